#!/usr/bin/env python3
"""Headline benchmark: replica-folds/s (MFE fill + traceback + partition function + eval_structure)
for R=64 replicas x L=200 per GPU (BASELINE.json configs[2]) on N MI355X of one node.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python bench.py --gpus N ...          (N > 1 without WORLD_SIZE in the environment: launches the N ranks itself, see launch_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One step = one pass of the hot path over one batch of R sequences resident in HBM: what the reference
does R times per Monte-Carlo iteration inside score_sequence() (utils/energy_scores.py:31-125).
Replicas fold independently, so N GPUs each take their own R replicas (weak scaling); the only
exchange is the all-gather of the R x N scoring-function values before a replica-exchange attempt
(reference utils/replica_exchange_monte_carlo.py:113-173), issued every --exchange-every steps as in
the reference's e=100 Monte-Carlo iterations per exchange step, and once after the last step.

Prints ONE JSON line (rank 0).  'roofline' prices the dominant kernel against the HARDWARE peaks only (HBM bytes, LDS-array
cycles, VALU issue at the SIMD-32 pipe rate) and reports the kernel's own finalize-only floor separately (see
roofline_block); 'cpu_baseline' times the CPU oracle (a port of the ViennaRNA recursions; ViennaRNA itself is not on this
box) on the same workload and its results are compared with the GPU's.
"""
import argparse
import csv
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic operand bytes per replica-fold (MB): n -> (MFE, PF with q5-only exterior)
B_ALG_MB = {100: (6.71, None), 200: (40.16, 80.5), 400: (221.4, None)}


def b_alg_bytes(n):
    """(MFE, PF) algorithmic bytes per fold; exact SURVEY figures where given, else the same formula:
    MFE = 4(2K + I + 2E + 2C), PF(q5) = 8(2K + I + 2E + 4C) with dense term counts."""
    C = (n - 4) * (n - 3) // 2
    K = sum((n - d) * (d - 1) for d in range(4, n)) // 1  # all split points u of every cell
    # the SURVEY counts splits with both parts longer than TURN: use its closed numbers when available
    I = 0
    for d in range(4, n):
        cnt = 0
        for s in range(0, 31):
            if d - 2 - s >= 4:
                cnt += s + 1
        I += (n - d) * cnt
    E = n * (n - 1) // 2
    Ks = sum((n - d) * max(0, d - 8) for d in range(4, n))
    mfe = 4 * (2 * Ks + I + 2 * E + 2 * C)
    pf = 8 * (2 * Ks + I + 2 * E + 4 * C)
    if n in B_ALG_MB:
        m, p = B_ALG_MB[n]
        mfe = m * 1e6
        if p is not None:
            pf = p * 1e6
    return float(mfe), float(pf)


def load_target(name):
    with open(os.path.join(ROOT, "tests", "golden", "eterna_v1_targets.csv")) as fh:
        for r in csv.DictReader(fh):
            if r["name"] == name:
                return r["structure"]
    raise KeyError(name)


def usable_cores():
    """CPU cores this process may really use: the affinity mask, capped by the cgroup CPU quota (cpu.max or the v1 pair).
    os.cpu_count() reports the host's 256 hardware threads on the GPU box although the lease is limited to a share."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def cpu_baseline(seqs, target, budget_s=25.0):
    """Time the CPU oracle (kind 'port': a plain-C restatement of the ViennaRNA recursions -- ViennaRNA itself is
    not on this box) on the GPU box's host cores, on a bounded sample of the benchmark batch.  Threads 1, 2, 4, ...
    up to the usable cores (affinity mask and cgroup quota, NOT os.cpu_count()), at least 8 folds per thread, threads
    bound to cores; the table of all thread counts is reported so that the scaling can be judged.  Also returns the
    oracle's results for the benchmark batch, which main() compares with what the timed GPU steps produced."""
    from desirna_amd import params
    from oracle import pyoracle
    pyoracle.build()
    native = pyoracle.build_native()                 # -march=native on this host; the shipped build is x86-64-v3
    try:
        orc = pyoracle.Oracle(params.load_blob(), lib=native)
    except OSError:
        native = None
        orc = pyoracle.Oracle(params.load_blob())
    cores, quota = usable_cores()
    flags = pyoracle.FLAG_PF | pyoracle.FLAG_MFE | pyoracle.FLAG_PIN
    table = []
    ref = None
    tstart = time.perf_counter()
    t = 1
    counts = []
    while t < cores:
        counts.append(t)
        t *= 2
    counts.append(cores)
    for th in counts:
        nseq = max(8 * th, 8)
        sample = (list(seqs) * (-(-nseq // len(seqs))))[:nseq]
        orc.score_batch(sample[:th], [target], flags, threads=th)           # sizes the threads' scratch arenas, untimed
        best = None
        for rep in range(3 if th > 1 else 1):
            t0 = time.perf_counter()
            res = orc.score_batch(sample, [target], flags, threads=th)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            if time.perf_counter() - tstart > budget_s and th != counts[-1]:
                break
        table.append({"threads": th, "folds": nseq, "folds_per_s": nseq / best})
    # the oracle's answers for the benchmark batch itself (parity check of the timed GPU path)
    ref = orc.score_batch(list(seqs), [target], flags, threads=cores)
    single, allc = table[0]["folds_per_s"], table[-1]["folds_per_s"]
    hw = os.cpu_count() or cores
    return {"value": allc, "unit": "replica-folds/s", "cores": cores, "kind": "port",
            "build": "gcc -O3 -march=native (built on this host)" if native else "gcc -O3 -march=x86-64-v3 (shipped build)",
            # the reference starts one process per replica (utils/replica_exchange_monte_carlo.py:233-271): on a host whose
            # hardware threads are all available to it, R replicas fold on min(R, hardware threads) cores at once.  NOT
            # measured (this lease's cgroup allows `cores` of them): single-core rate x that count, stated as an extrapolation
            "reference_style_R_processes": {"value": single * min(len(seqs), hw), "processes": min(len(seqs), hw),
                                            "host_hardware_threads": hw, "kind": "extrapolation: single_core_value x processes"},
            "sample": "%d x L=%d sequences (the benchmark batch tiled to 8 folds per thread), best of 3 after one warm-up, %d "
                      "threads = usable cores (affinity %d, cgroup quota %s, os.cpu_count %d), thread t bound to the t-th "
                      "CPU of the affinity mask; MFE fill+traceback + PF + eval each"
                      % (table[-1]["folds"], len(seqs[0]), cores, len(os.sched_getaffinity(0)), quota, os.cpu_count() or 0),
            "single_core_value": single, "effective_cores": allc / single, "scaling_table": table}, ref


def roofline_block(dom, dom_ms, L, R, tk):
    """The dominant fold kernel against HARDWARE peaks only -- `bound` is the largest of
         hbm : measured HBM bytes per launch (PMC: (2 FETCH_SIZE + WRITE_SIZE) KB) / kernel time / 8 TB/s
         lds : LDS-array busy cycles per CU (SQ_LDS_IDX_ACTIVE / workgroups) / kernel cycles at 2.4 GHz
         valu: VALU wave-instructions per SIMD (SQ_ACTIVE_INST_VALU / (4 workgroups)) x 2 cycles (SIMD-32 pipe rate,
               MI355X_MICROARCH.md 'Wave scheduling') / kernel cycles; the 4-cycle figure (what ONE wave alone sustains) is
               given beside it as frac_wave_issue_4cycle
    and `own_floor` = time of the same kernel with every sweep phase left out (finalize + barrier only) / kernel time, a
    property of this implementation, not of the hardware, reported separately.  Counter inputs come from profiles/
    (rocprofv3 PMC passes and the floor build of the SAME source, tools/gpu_round2.sh); kernel time is measured live here."""
    src = os.path.join(ROOT, "profiles", "roofline_inputs.json")
    mfe_b, pf_b = b_alg_bytes(L)
    stream = (pf_b if dom == "pf" else mfe_b) * R
    out = {"bound": None, "kernel": ("%s_lds_kernel<1024>" if L <= 200 else "%s_kernel<1024>") % dom, "achieved": None,
           "peak": None, "unit": None, "frac": None, "traffic": None,
           "operand_stream": {"bytes_per_launch": stream, "GB_per_s": stream / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0,
                              "note": "algorithmic operand bytes (SURVEY 8(d)); served from LDS/L2, not a bound"}}
    if not os.path.exists(src) or dom_ms <= 0:
        return out
    try:
        inp = json.load(open(src))
        k = inp["kernels"]["%s_L%d_R%d" % (dom, L, R)]
    except Exception:
        return out
    t = dom_ms * 1e-3
    if k.get("kernel"):
        out["kernel"] = k["kernel"].replace("void drna::", "").split("(")[0]
    clk = k.get("clock_hz", 2.4e9)
    bounds = {}
    if k.get("hbm_bytes_per_launch"):
        a = k["hbm_bytes_per_launch"] / t / 1e9
        bounds["hbm"] = {"achieved": a, "peak": 8000.0, "unit": "GB/s", "frac": a / 8000.0}
        out["traffic"] = k["hbm_bytes_per_launch"]
    if k.get("lds_busy_cycles_per_cu"):
        a = k["lds_busy_cycles_per_cu"] / t
        bounds["lds"] = {"achieved": a / 1e9, "peak": clk / 1e9, "unit": "G LDS-array cycles/s per CU", "frac": a / clk,
                         "bank_conflict_share": k.get("lds_bank_conflict_share")}
        if k.get("workgroups") and k["workgroups"] > R:
            # kernels with a helper workgroup per fold: the counters are sums over main and helper CUs; if ALL of the LDS
            # traffic were the main workgroups' (the helpers' is small), their CUs would be this busy
            bounds["lds"]["frac_main_cus_upper"] = a / clk * k["workgroups"] / R
    if k.get("valu_busy_cycles_per_simd"):
        # the inputs file prices an instruction at 4 cycles (valu_cycles_per_count); the pipe itself takes 2
        insts = k["valu_busy_cycles_per_simd"] / inp.get("valu_cycles_per_count", 4.0)
        a = insts * 2.0 / t
        bounds["valu"] = {"achieved": a / 1e9, "peak": clk / 1e9, "unit": "G VALU pipe cycles/s per SIMD (2 per wave64 instruction)",
                          "frac": a / clk, "frac_wave_issue_4cycle": insts * 4.0 / t / clk, "wave_instructions_per_simd": insts}
    if bounds:
        name = max(bounds, key=lambda b: bounds[b]["frac"])
        out.update({"bound": name, "achieved": bounds[name]["achieved"], "peak": bounds[name]["peak"],
                    "unit": bounds[name]["unit"], "frac": bounds[name]["frac"], "bounds": bounds,
                    "formulas": {"hbm": "(2*FETCH_SIZE + WRITE_SIZE)*1024 B / kernel_s / 8e12",
                                 "lds": "SQ_LDS_IDX_ACTIVE / workgroups / (kernel_s * 2.4e9)",
                                 "valu": "SQ_ACTIVE_INST_VALU / (4*workgroups) * 2 / (kernel_s * 2.4e9)"},
                    "inputs": "profiles/roofline_inputs.json (%s)" % inp.get("source", "?")})
    if k.get("floor_ms"):
        out["own_floor"] = {"floor_ms": k["floor_ms"], "frac": k["floor_ms"] / dom_ms,
                            "note": "one-workgroup-per-fold build of the same source with every sweep phase left out (finalize + "
                                    "barrier only): this implementation's dependency chain, not a hardware peak"}
    out["workgroups"] = k.get("workgroups")
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: start the N ranks from here, one process
    per GPU, as torch.distributed.run would (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), relay rank 0's JSON
    line and return non-zero if any rank failed.  This parent imports neither torch nor the engine and never touches the GPU;
    the ranks are plain child processes (nothing is exec'ed over an initialised process)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].stdout.read().decode()
    rcs = []
    deadline = time.time() + 1800
    for p_ in procs:
        try:
            rcs.append(p_.wait(timeout=max(1.0, deadline - time.time())))
        except subprocess.TimeoutExpired:
            p_.kill()
            rcs.append(-9)
    # ONE JSON line on stdout: whatever else rank 0 printed there (the gloo backend announces its connections on stdout) goes to stderr
    for line in out0.splitlines():
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(rcs) if c != 0]
    if bad:
        sys.stderr.write("bench.py: ranks failed (rank, exit code): %s\n" % bad)
        return 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--replicas", type=int, default=64, help="replicas per GPU")
    ap.add_argument("--target", default="eteV1_69.txt", help="Eterna100-V1 target giving L (69: L=200, 92: L=100, 53: L=400)")
    ap.add_argument("--exchange-every", type=int, default=100)
    ap.add_argument("--seqs", choices=["uniform", "design"], default="uniform")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))        # before torch / HIP are touched in this process
    if os.environ.get("DRNA_BENCH_ECHO_RANK"):           # tests/test_host_cpu.py: what a rank was started with, no GPU needed
        me = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        me["gpus"] = args.gpus
        with open(os.path.join(os.environ["DRNA_BENCH_ECHO_RANK"], "rank%s.json" % me["RANK"]), "w") as fh:
            json.dump(me, fh)
        if me["RANK"] in (None, "0"):
            print(json.dumps(me))
        raise SystemExit(3 if os.environ.get("DRNA_BENCH_ECHO_FAIL") == me["RANK"] else 0)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: running %d ranks\n" % (args.gpus, world, world))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # DRNA_BENCH_FORCE_DIST=1: run the exchange step's collectives (all-gather, barrier, max all-reduce) at N = 1 too -- a
    # one-rank communicator over RCCL on the one GPU a test box has (tests/test_gpu_parity.py); the N > 1 path is the same code
    multi = world > 1 or os.environ.get("DRNA_BENCH_FORCE_DIST", "0") == "1"
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    # DRNA_BENCH_BACKEND=gloo is a REHEARSAL mode for a one-GPU box: the ranks share the card (local_rank modulo the device
    # count) and the score gather goes through host tensors, so the launch contract (env, barriers, max over ranks, one JSON
    # line from rank 0) can be exercised without RCCL.  Real runs use the default, RCCL, one rank per GPU.
    backend = os.environ.get("DRNA_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank %= max(1, torch.cuda.device_count())
    elif torch.cuda.device_count() < world:
        raise SystemExit("bench.py: %d ranks over RCCL need %d GPUs, this node shows %d (DRNA_BENCH_BACKEND=gloo rehearses the "
                         "launch on fewer)" % (world, world, torch.cuda.device_count()))
    if multi:
        if backend == "gloo":
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if backend == "gloo" else dev      # where the collectives' tensors live

    from desirna_amd import engine as E
    target = load_target(args.target)
    L, R = len(target), args.replicas
    rng = np.random.default_rng(20260101 + rank)
    if args.seqs == "uniform":
        seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(R)]
    else:
        from desirna_amd.workloads import design_like_sequences
        seqs = design_like_sequences(target, R, rng)
    eng = E.Engine(max_R=R, max_L=L, device=local_rank)
    if backend == "gloo" and world > max(1, torch.cuda.device_count()):
        # rehearsal with several ranks on ONE card: the folds by several workgroups assume the whole chip (every workgroup of a
        # launch resident at once), which ranks sharing a card do not have
        eng.set_option("pf_helper", 0)
        eng.set_option("dual", 0)
    eng.set_targets([target])
    flags = E.NEED_PF | E.NEED_MFE | E.NEED_EVAL

    d_seqs = torch.from_numpy(np.frombuffer("".join(seqs).encode(), dtype=np.uint8).copy()).to(dev)
    d_Epf = torch.zeros(R, dtype=torch.float64, device=dev)
    d_Emfe = torch.zeros(R, dtype=torch.int32, device=dev)
    d_ss = torch.zeros(R * L, dtype=torch.uint8, device=dev)
    d_Ed = torch.zeros(R, dtype=torch.int32, device=dev)
    gathered = torch.zeros(R * world, dtype=torch.float64, device=cdev) if multi else None
    torch.cuda.synchronize()

    def step(k, last):
        eng.score_batch_device(d_seqs.data_ptr(), R, L, flags, d_Epf.data_ptr(), d_Emfe.data_ptr(),
                               d_ss.data_ptr(), d_Ed.data_ptr())
        if multi and (last or (k + 1) % args.exchange_every == 0):
            score = d_Ed.to(torch.float64) / 100.0 - d_Epf          # Ed - Epf, the default -sf term
            dist.all_gather_into_tensor(gathered, score.to(cdev))
            torch.cuda.current_stream().synchronize()               # d_Ed / d_Epf are read: the next step's kernels (other streams) overwrite them

    for k in range(args.warmup):
        step(k, False)
    tk = {"mfe": 0.0, "pf": 0.0, "eval": 0.0, "total": 0.0}
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    eng.timing_sums(reset=True)                     # the engine sums its HIP-event kernel times over the timed calls itself
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, k == args.steps - 1)
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    dt = time.perf_counter() - t0
    if multi:
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ts = eng.timing_sums()
    for key in tk:
        tk[key] = ts[key] / max(1, ts["calls"])

    if rank == 0:
        folds = R * world * args.steps
        dom = "pf" if tk["pf"] >= tk["mfe"] else "mfe"
        out = {
            "metric": "replica-folds/sec (MFE+PF, L=%d, R=%d)" % (L, R),
            "value": folds / dt, "unit": "replica-folds/s", "n_gpus": world, "collectives": (backend if multi else None), "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32 (MFE) + f64 (PF)", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: Eterna100-V1 #%s target (L=%d), R=%d %s-random sequences per GPU, "
                                   "-sf Ed-Epf (MFE fill+traceback, PF inside, eval_structure)"
                                   % (args.target[6:8], L, R, args.seqs),
                       "replicas_per_gpu": R, "L": L, "exchange_every": args.exchange_every,
                       "threads_per_workgroup": eng.info()["threads_per_wg"]},
            "kernel_ms": {k: round(v, 4) for k, v in tk.items()},
            "roofline": roofline_block(dom, tk[dom], L, R, tk),
        }
        hb = (out["roofline"].get("bounds") or {}).get("hbm")
        out["achieved_hbm_GB_s"] = hb["achieved"] if hb else None          # BASELINE metric: "HBM GB/s vs peak" (8000)
        out["achieved_hbm_frac_of_peak"] = hb["frac"] if hb else None
        out["sync_fallbacks"] = eng.get_option("sync_fallbacks")          # calls redone with one workgroup per fold (lost partner)
        out["cus_occupied"] = {"workgroups": eng.get_option("last_workgroups"), "compute_units": eng.info()["compute_units"]}
        if not args.no_cpu_baseline and world == 1:      # rank 0 at N=1 only: at N>1 the other ranks would idle at the barrier
            out["cpu_baseline"], ref = cpu_baseline(seqs, target)
            out["speedup_vs_cpu_usable_cores"] = out["value"] / world / out["cpu_baseline"]["value"]
            out["speedup_vs_cpu_reference_style_R_processes"] = out["value"] / world / out["cpu_baseline"]["reference_style_R_processes"]["value"]
            # what the timed steps left in the device buffers against the oracle's answers for the same batch:
            # MFE energy, MFE structure and E(target) bit for bit, Epf to 1e-9 kcal/mol
            r_Epf, r_Emfe, r_ss, r_Ed = ref
            g_ss = bytes(d_ss.cpu().numpy().tobytes()).decode()
            ok = (np.array_equal(d_Emfe.cpu().numpy(), r_Emfe) and np.array_equal(d_Ed.cpu().numpy(), r_Ed[:, 0])
                  and [g_ss[k * L:(k + 1) * L] for k in range(R)] == r_ss
                  and float(np.abs(d_Epf.cpu().numpy() - r_Epf).max()) < 1e-9)
            out["parity_checked"] = bool(ok)
            out["parity"] = {"sequences": R, "max_abs_dEpf": float(np.abs(d_Epf.cpu().numpy() - r_Epf).max()),
                             "Emfe_Ed_structures": "bit-exact" if ok else "MISMATCH"}
            if not ok:
                print(json.dumps(out))
                raise SystemExit("bench: GPU results differ from the oracle")
        print(json.dumps(out))
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
