"""Diagnostic (GPU box): kernel times of the L=200, R=64 batch for builds with extra -D flags, one per argument
(e.g. `python tools/variants.py "" "-DDRNA_FIN_SYNC" "-DDRNA_SKIP=15"`); DUAL=0/1 prefix in an argument sets DRNA_DUAL."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from desirna_amd import engine as E
import bench
tg = bench.load_target("eteV1_69.txt"); L = len(tg); R = 64
rng = np.random.default_rng(20260101)
seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(R)]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
procs = []
for k, a in enumerate(sys.argv[1:]):
    flags = [f for f in a.split() if f.startswith("-D") or f.startswith("--offload-arch") or f.startswith("-m")]
    arch = [] if any(f.startswith("--offload-arch") for f in flags) else ["--offload-arch=gfx950"]
    out = os.path.join(ROOT, "gpurun_out", "libvar%d.so" % k)
    procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC"] + arch + flags +
                                  ["-shared", "-o", out, os.path.join(ROOT, "desirna_amd/csrc/engine.hip")], stderr=subprocess.DEVNULL))
for p in procs:
    p.wait()
for k, a in enumerate(sys.argv[1:]):
    out = os.path.join(ROOT, "gpurun_out", "libvar%d.so" % k)
    for dual in ("1", "0"):
        os.environ["DRNA_DUAL"] = dual
        eng = E.Engine(max_R=R, max_L=L, lib=out)
        eng.set_targets([tg])
        ts = []
        for _ in range(8):
            try:
                eng.score_batch(seqs, E.NEED_MFE | E.NEED_PF)
            except Exception as ex:
                pass
            ts.append(eng.last_timing())
        print("%-40s dual=%s: mfe %.3f ms  pf %.3f ms" % (a or "(default)", dual, min(x["mfe"] for x in ts[2:]), min(x["pf"] for x in ts[2:])), flush=True)
        eng.close()
