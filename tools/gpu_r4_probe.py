"""GPU box, round 4 probe: (1) end-to-end Monte-Carlo loop rate by host thread count, (2) replicas-per-call sweep of the headline shape
with and without the strip kernels for n <= 200, (3) kernel times of every prebuilt variant in build/var/."""
import glob, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from desirna_amd import engine as E
import bench

tg = bench.load_target("eteV1_69.txt"); L = len(tg)
out = {}

def seqs_for(R, seed=20260101):
    rng = np.random.default_rng(seed)
    return ["".join(rng.choice(list("ACGU"), L)) for _ in range(R)]

def time_batch(eng, seqs, flags, reps=12):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        eng.score_batch(seqs, flags)
        ts.append((time.perf_counter() - t0, eng.last_timing()))
    ts = ts[3:]
    return {"wall_ms": 1e3 * min(t[0] for t in ts), "mfe": min(t[1]["mfe"] for t in ts), "pf": min(t[1]["pf"] for t in ts),
            "total": min(t[1]["total"] for t in ts)}

if "mc" in sys.argv[1:] or len(sys.argv) == 1:
    eng = E.Engine(max_R=64, max_L=L)
    eng.set_targets([tg])
    base = time_batch(eng, seqs_for(64), E.NEED_PF | E.NEED_MFE | E.NEED_EVAL)
    out["kernel_only"] = base
    print("kernel only", base, flush=True)
    for T in (1, 2, 4, 8, 12):
        eng.set_option("mc_threads", T)
        r = bench.mc_loop_block(eng, tg, 64, 100, 64 / (base["wall_ms"] * 1e-3))
        out["mc_T%d" % T] = r
        print("mc_threads", T, {k: r[k] for k in ("scored_sequences_per_s", "ms_per_iteration", "kernel_ms_per_iteration", "host_us_per_iteration", "frac_of_kernel_only_rate", "host_threads")}, flush=True)
    eng.close()

if "sweep" in sys.argv[1:] or len(sys.argv) == 1:
    for R in (32, 64, 128, 256):
        sq = seqs_for(R)
        for strips in (1, 2):
            eng = E.Engine(max_R=R, max_L=L)
            eng.set_targets([tg])
            eng.set_option("strips", strips)
            both = time_batch(eng, sq, E.NEED_PF | E.NEED_MFE | E.NEED_EVAL)
            pf = time_batch(eng, sq, E.NEED_PF)
            mfe = time_batch(eng, sq, E.NEED_MFE)
            out["R%d_strips%d" % (R, strips)] = {"both": both, "pf_alone": pf, "mfe_alone": mfe, "folds_per_s": R / (both["wall_ms"] * 1e-3)}
            print("R", R, "strips", strips, "both", both, "pf alone %.4f mfe alone %.4f" % (pf["pf"], mfe["mfe"]), "folds/s %.0f" % (R / (both["wall_ms"] * 1e-3)),
                  "fallbacks", eng.get_option("sync_fallbacks"), flush=True)
            eng.close()

if "variants" in sys.argv[1:] or len(sys.argv) == 1:
    VR = int(os.environ.get("VAR_R", "64"))
    vflags = {"both": E.NEED_PF | E.NEED_MFE | E.NEED_EVAL, "pf": E.NEED_PF, "mfe": E.NEED_MFE}[os.environ.get("VAR_FLAGS", "both")]
    sq = seqs_for(VR)
    for lib in sorted(glob.glob(os.path.join(ROOT, "build", "var", "lib_*.so"))):
        name = os.path.basename(lib)[4:-3]
        for fused in ((1, 0) if VR <= 64 and os.environ.get("VAR_FUSED", "both") == "both" else (int(os.environ.get("VAR_FUSED", "1")),)):
            eng = E.Engine(max_R=VR, max_L=L, lib=lib)
            eng.set_targets([tg])
            try:
                eng.set_option("fused", fused)
            except Exception:
                pass
            try:
                r = time_batch(eng, sq, vflags, reps=int(os.environ.get("VAR_REPS", "40")))
            except Exception as ex:                       # timing builds leave phases out: results (and status words) may be off
                r = {"error": str(ex)[:80], **eng.last_timing()}
            out["var_%s_fused%d" % (name, fused)] = r
            print("variant %-20s fused %d" % (name, fused), r, "fallbacks", eng.get_option("sync_fallbacks"), flush=True)
            eng.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r4_probe.json"), "w"), indent=1)
