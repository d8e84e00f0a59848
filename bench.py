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


def roofline_block(dom, dom_ms, L, R, tk, fused=False):
    """The dominant fold kernel against HARDWARE peaks only -- `bound` is the largest of
         hbm : measured HBM bytes per launch (PMC: (2 FETCH_SIZE + WRITE_SIZE) KB) / kernel time / 8 TB/s
         lds : LDS-array busy cycles per CU (SQ_LDS_IDX_ACTIVE / workgroups) / kernel cycles at 2.4 GHz
         valu: VALU wave-instructions per SIMD (SQ_ACTIVE_INST_VALU / (4 workgroups)) x 4 cycles / kernel cycles: MI355X's vector
               fp64 / int32 rate is 16 lanes per SIMD and clock (78.6 TFLOP/s fp64 over 256 CUs at 2.4 GHz), i.e. 4 cycles per
               wave64 instruction; only packed or dual-issue fp32 reaches the 2-cycle pipe rate, which is given beside it as
               frac_pipe_2cycle
    and `own_floor` = time of the SAME launch configuration built with every sweep phase left out (finalize, hand-shake and
    barriers only) / kernel time: a property of this implementation, not of the hardware, reported separately.  Counter inputs
    come from profiles/roofline_inputs.json (rocprofv3 PMC passes and the floor build of the source at the commit named there,
    tools/gpu_round4.sh); kernel time is measured live here, and a live time more than 5 % away from the profiled run's is flagged."""
    src = os.path.join(ROOT, "profiles", "roofline_inputs.json")
    mfe_b, pf_b = b_alg_bytes(L)
    stream = (pf_b + mfe_b if fused else pf_b if dom == "pf" else mfe_b) * R        # (one launch: both folds' operands)
    out = {"bound": None, "kernel": ("%s_lds_kernel<1024>" if L <= 200 else "%s_kernel<1024>") % dom, "achieved": None,
           "peak": None, "unit": None, "frac": None, "traffic": None,
           "operand_stream": {"bytes_per_launch": stream, "GB_per_s": stream / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0,
                              "note": "algorithmic operand bytes (SURVEY 8(d)); served from LDS/L2, not a bound"}}
    if not os.path.exists(src) or dom_ms <= 0:
        return out
    try:
        inp = json.load(open(src))
        key = "fused_L%d_R%d" % (L, R) if fused else "%s_L%d_R%d" % (dom, L, R)
        k = inp["kernels"][key]
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
        # the inputs file prices an instruction at valu_cycles_per_count (4) cycles
        insts = k["valu_busy_cycles_per_simd"] / inp.get("valu_cycles_per_count", 4.0)
        a = insts * 4.0 / t
        bounds["valu"] = {"achieved": a / 1e9, "peak": clk / 1e9, "unit": "G VALU issue cycles/s per SIMD (4 per wave64 fp64 / int32 instruction)",
                          "frac": a / clk, "frac_pipe_2cycle": insts * 2.0 / t / clk, "wave_instructions_per_simd": insts}
    if bounds:
        name = max(bounds, key=lambda b: bounds[b]["frac"])
        out.update({"bound": name, "achieved": bounds[name]["achieved"], "peak": bounds[name]["peak"],
                    "unit": bounds[name]["unit"], "frac": bounds[name]["frac"], "bounds": bounds,
                    "formulas": {"hbm": "(2*FETCH_SIZE + WRITE_SIZE)*1024 B / kernel_s / 8e12",
                                 "lds": "SQ_LDS_IDX_ACTIVE / workgroups / (kernel_s * 2.4e9)",
                                 "valu": "SQ_ACTIVE_INST_VALU / (4*workgroups) * 4 / (kernel_s * 2.4e9)"},
                    "inputs": "profiles/roofline_inputs.json (%s)" % inp.get("source", "?")})
    if k.get("floor_ms"):
        out["own_floor"] = {"floor_ms": k["floor_ms"], "frac": k["floor_ms"] / dom_ms, "build": k.get("floor_build"),
                            "note": "the same launch configuration with every sweep phase left out (finalize, hand-shake and barriers "
                                    "only): this implementation's dependency chain, not a hardware peak"}
    if k.get("rocprof_avg_ms"):
        drift = dom_ms / k["rocprof_avg_ms"] - 1.0
        out["profiled_kernel_ms"] = k["rocprof_avg_ms"]
        out["live_vs_profiled"] = drift
        if abs(drift) > 0.05:
            out["warning"] = ("live kernel time %.4f ms is %+.1f %% from the profiled run's %.4f ms (%s): counters per launch are "
                              "from that run" % (dom_ms, 100 * drift, k["rocprof_avg_ms"], inp.get("source", "?")))
            sys.stderr.write("bench.py: roofline: " + out["warning"] + "\n")
    out["workgroups"] = k.get("workgroups")
    # both folds of a step: measured HBM bytes of the two launches over the step's device time (BASELINE metric: HBM GB/s vs peak)
    both = [inp["kernels"].get("%s_L%d_R%d" % (f, L, R), {}).get("hbm_bytes_per_launch") for f in ("mfe", "pf")]
    if fused:
        both = [k.get("hbm_bytes_per_launch")]
    if all(both) and tk.get("total", 0) > 0:
        out["step_hbm"] = {"bytes_per_step": float(sum(both)), "GB_per_s": sum(both) / (tk["total"] * 1e-3) / 1e9,
                           "frac_of_peak": sum(both) / (tk["total"] * 1e-3) / 8e12}
    return out


def mc_loop_block(eng, target, R, iters, kernel_only_rate, headline_kernel_ms=None, seed=1):
    """The loop the reference actually runs (utils/replica_exchange_monte_carlo.py:176-210: propose -> score -> accept, e times per
    exchange step) through the native driver drna_mc_run, on the benchmark's target: every iteration proposes one mutation per
    replica (utils/sequence_utils.py:1008-1136), folds the R proposals on the GPU, computes SimScore and the -sf sum, and takes
    the Metropolis decision.  All replicas start from the reference's initial sequence; one untimed exchange step (warm-up, and
    the replicas diverge), then `iters` timed iterations.  Outside the headline's timed region."""
    import random
    from desirna_amd import design, engine as E, replica_exchange as rx
    L = len(target)
    prob = design.DesignProblem(target, "N" * L, None)
    hk = E.HostKernels()
    flags = E.NEED_PF | E.NEED_MFE | E.NEED_EVAL
    sf = [("Ed-Epf", 1.0)]
    init = prob.initial_sequence(random.Random(2137 + seed))
    cur = np.ascontiguousarray(np.tile(np.frombuffer(init.encode(), dtype=np.uint8), (R, 1)))
    Epf, Emfe, ss, Ed = eng.score_batch_arrays(cur, flags)
    mcc, _, _ = hk.simscore(target, ss)
    state = dict(seqs=cur, mfe_ss=np.ascontiguousarray(ss), score=np.ascontiguousarray(Ed[:, 0] / 100.0 - Epf),
                 mcc1=np.ascontiguousarray(1.0 - mcc), Epf=np.ascontiguousarray(Epf), Ed=np.ascontiguousarray(Ed[:, 0] / 100.0))
    temps = np.array(rx.get_rep_temps(R, 10.0, 150.0), dtype=np.float64)
    shelf = np.searchsorted(temps, temps).astype(np.int32)
    rng_state = np.empty((R, E.RNG_WORDS), dtype=np.uint32)
    counters = np.zeros(3, dtype=np.int64)
    k0 = int(np.lexsort((state["score"], state["mcc1"]))[0])
    best = dict(seq=cur[k0].copy(), ss=state["mfe_ss"][k0].copy(),
                vals=np.array([state["mcc1"][k0], state["score"][k0], state["Epf"][k0], state["Ed"][k0]], dtype=np.float64))

    def exchange_step(n):
        hk.rng_seed(np.arange(R), out=rng_state)               # random.seed(replica index) at every exchange step
        eng.mc_run(prob, n, shelf, R, 0.7, 0.0, True, temps, sf, flags, rng_state, state, counters, best)

    exchange_step(iters)                                       # untimed
    counters[:] = 0
    eng.timing_sums(reset=True)
    t0 = time.perf_counter()
    exchange_step(iters)
    dt = time.perf_counter() - t0
    ts = eng.timing_sums()
    ms_it = dt / iters * 1e3
    k_ms = ts["total"] / max(1, ts["calls"])
    rate = R * iters / dt
    return {"scored_sequences_per_s": rate, "ms_per_iteration": ms_it, "kernel_ms_per_iteration": k_ms,
            "host_us_per_iteration": (ms_it - k_ms) * 1e3, "iterations": iters, "replicas": R, "L": L,
            # the headline's rate (uniform-random sequences; a converging design's sequences fold a few per cent faster) and the rate
            # the same iterations would have with no host work between the launches beyond the headline's own (device time of these
            # iterations + the headline's host time per step)
            "frac_of_kernel_only_rate": rate / kernel_only_rate if kernel_only_rate else None,
            "frac_of_own_device_rate": (k_ms + (R / kernel_only_rate * 1e3 - headline_kernel_ms)) / ms_it
                                       if kernel_only_rate and headline_kernel_ms else None,
            "host_threads": eng.get_option("mc_threads_used"),
            "accepted": int(counters[0]), "rejected": int(counters[2]), "best_1_minus_mcc": float(best["vals"][0]),
            "what": "drna_mc_run: proposals (reference MT19937 streams) + GPU folds + SimScore + -sf Ed-Epf + Metropolis, e=%d "
                    "iterations of one exchange step after one untimed step; host_us = wall - device time per iteration" % iters}


def r_sweep_block(target, device, Rs=(32, 64, 128, 256), reps=12):
    """Replicas per call at the headline's shape (one GPU): device-resident batches of R uniform-random sequences, the time of a
    whole scoring call (both folds + E(target)), best of `reps` after three warm-up calls.  R = 64 is the headline's operating
    point (four workgroups per sequence); above it every fold has one workgroup and the chip is shared by more folds."""
    import torch
    from desirna_amd import engine as E
    L = len(target)
    rows = []
    dev = torch.device("cuda", device)
    for R in Rs:
        rng = np.random.default_rng(20260101)
        seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(R)]
        eng = E.Engine(max_R=R, max_L=L, device=device)
        eng.set_targets([target])
        d_seqs = torch.from_numpy(np.frombuffer("".join(seqs).encode(), dtype=np.uint8).copy()).to(dev)
        d_Epf = torch.zeros(R, dtype=torch.float64, device=dev); d_Emfe = torch.zeros(R, dtype=torch.int32, device=dev)
        d_ss = torch.zeros(R * L, dtype=torch.uint8, device=dev); d_Ed = torch.zeros(R, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        best, kt = None, None
        for k in range(reps + 3):
            t0 = time.perf_counter()
            eng.score_batch_device(d_seqs.data_ptr(), R, L, E.NEED_PF | E.NEED_MFE | E.NEED_EVAL, d_Epf.data_ptr(), d_Emfe.data_ptr(),
                                   d_ss.data_ptr(), d_Ed.data_ptr())
            dt = time.perf_counter() - t0
            if k >= 3 and (best is None or dt < best):
                best, kt = dt, eng.last_timing()
        rows.append({"R": R, "ms_per_call": best * 1e3, "replica_folds_per_s": R / best, "mfe_ms": kt["mfe"], "pf_ms": kt["pf"],
                     "workgroups": eng.get_option("last_workgroups"), "sync_fallbacks": eng.get_option("sync_fallbacks")})
        eng.close()
    return rows


def launch_ranks(n):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: start the N ranks from here, one process
    per GPU, as torch.distributed.run would (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), relay rank 0's JSON
    line and return non-zero if any rank failed.  This parent imports neither torch nor the engine and never touches the GPU;
    the ranks are plain child processes (nothing is exec'ed over an initialised process).  All ranks are supervised: the first
    rank that exits non-zero, or the deadline, ends the others (a rank that dies at start-up would otherwise leave the rest in
    the rendezvous until the backend's own timeout); rank 0's stdout is drained by a thread meanwhile."""
    import socket
    import subprocess
    import threading
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("DRNA_BENCH_LAUNCH_TIMEOUT", "1800"))
    rcs = [None] * n
    why = None
    while any(c is None for c in rcs):
        for k, p_ in enumerate(procs):
            if rcs[k] is None:
                rcs[k] = p_.poll()
        failed = [k for k, c in enumerate(rcs) if c not in (None, 0)]
        if failed or time.time() > deadline:
            why = "rank %d exited with code %s" % (failed[0], rcs[failed[0]]) if failed else "deadline reached"
            for k, p_ in enumerate(procs):
                if rcs[k] is None:
                    p_.kill()                                  # the exact child processes started above
                    p_.wait()
                    rcs[k] = -9
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    out0 = b"".join(chunks).decode()
    # ONE JSON line on stdout: whatever else rank 0 printed there (the gloo backend announces its connections on stdout) goes to stderr
    for line in out0.splitlines():
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(rcs) if c != 0]
    if bad:
        sys.stderr.write("bench.py: ranks failed (rank, exit code): %s%s\n" % (bad, " -- the others were stopped: " + why if why else ""))
        return 1
    return 0


def strong_shard(R_total, rank, world):
    """--scaling strong: the reference's R is the TOTAL number of replicas (utils/replica_exchange_monte_carlo.py:233-271);
    replica r lives on rank r mod world (replica_exchange.ReplicaShards)."""
    return list(range(rank, R_total, world))


def oracle_reference(seqs, target):
    """The CPU oracle's answers (Epf, Emfe, structures, E(target)) for a batch: the checker of the timed GPU path."""
    from desirna_amd import params
    from oracle import pyoracle
    pyoracle.build()
    orc = pyoracle.Oracle(params.load_blob())
    cores, _ = usable_cores()
    return orc.score_batch(list(seqs), [target], pyoracle.FLAG_PF | pyoracle.FLAG_MFE, threads=cores)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--replicas", type=int, default=64, help="replicas per GPU (--scaling weak) or in total (--scaling strong)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --replicas per GPU (the headline's R = 64 on every GPU); strong: --replicas in TOTAL, replica r on rank r mod N")
    ap.add_argument("--target", default="eteV1_69.txt", help="Eterna100-V1 target giving L (69: L=200, 92: L=100, 53: L=400)")
    ap.add_argument("--exchange-every", type=int, default=100)
    ap.add_argument("--seqs", choices=["uniform", "design"], default="uniform")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mc-loop", action="store_true", help="skip the end-to-end Monte-Carlo loop figure (mc_loop)")
    ap.add_argument("--r-sweep", action="store_true", help="(default at N = 1 since round 4; kept for old command lines)")
    ap.add_argument("--no-r-sweep", action="store_true", help="skip the replicas-per-call sweep (32 / 64 / 128 / 256) of the N = 1 line")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))        # before torch / HIP are touched in this process
    if os.environ.get("DRNA_BENCH_ECHO_RANK"):           # tests/test_host_cpu.py: what a rank was started with, no GPU needed
        me = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        me["gpus"] = args.gpus
        me["scaling"] = args.scaling
        me["local_replicas"] = (len(strong_shard(args.replicas, int(me["RANK"] or 0), int(me["WORLD_SIZE"] or 1)))
                                if args.scaling == "strong" else args.replicas)
        with open(os.path.join(os.environ["DRNA_BENCH_ECHO_RANK"], "rank%s.json" % me["RANK"]), "w") as fh:
            json.dump(me, fh)
        if me["RANK"] in (None, "0"):
            print(json.dumps(me))
        if os.environ.get("DRNA_BENCH_ECHO_HANG") == me["RANK"]:
            time.sleep(600)                              # a rank stuck in the rendezvous (test of the launcher's supervision)
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
    L = len(target)
    strong = args.scaling == "strong"
    if strong:
        # ONE set of --replicas sequences for the whole job, replica r on rank r mod N
        rng = np.random.default_rng(20260101)
        n_all = args.replicas
        mine = strong_shard(n_all, rank, world)
    else:
        rng = np.random.default_rng(20260101 + rank)
        n_all = args.replicas
        mine = list(range(n_all))
    if args.seqs == "uniform":
        all_seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(n_all)]
    else:
        from desirna_amd.workloads import design_like_sequences
        all_seqs = design_like_sequences(target, n_all, rng)
    seqs = [all_seqs[k] for k in mine]
    R = len(seqs)
    R_total = args.replicas if strong else args.replicas * world
    R_max = -(-args.replicas // world) if strong else args.replicas          # widest shard: the all-gather's row width
    if R < 1:
        raise SystemExit("bench.py: --scaling strong with fewer replicas than ranks leaves rank %d empty" % rank)
    eng = E.Engine(max_R=max(R, 1), max_L=L, device=local_rank)
    shared_card = backend == "gloo" and world > max(1, torch.cuda.device_count())
    if shared_card:
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
    gathered = torch.zeros(R_max * world, dtype=torch.float64, device=cdev) if multi else None
    mine_row = torch.full((R_max,), float("nan"), dtype=torch.float64, device=cdev) if multi else None
    torch.cuda.synchronize()
    gather_s = [0.0, 0]                                            # wall seconds inside the exchange step's all-gather, calls

    def step(k, last):
        eng.score_batch_device(d_seqs.data_ptr(), R, L, flags, d_Epf.data_ptr(), d_Emfe.data_ptr(),
                               d_ss.data_ptr(), d_Ed.data_ptr())
        if multi and (last or (k + 1) % args.exchange_every == 0):
            tg0 = time.perf_counter()
            score = d_Ed.to(torch.float64) / 100.0 - d_Epf          # Ed - Epf, the default -sf term
            mine_row[:R] = score.to(cdev)
            dist.all_gather_into_tensor(gathered, mine_row)         # ONE collective per exchange step
            torch.cuda.current_stream().synchronize()               # d_Ed / d_Epf are read: the next step's kernels (other streams) overwrite them
            gather_s[0] += time.perf_counter() - tg0
            gather_s[1] += 1

    for k in range(args.warmup):
        step(k, False)
    if multi and args.warmup > 0:
        step(0, True)                               # one untimed exchange step: the communicator's first collective sets it up
    tk = {"mfe": 0.0, "pf": 0.0, "eval": 0.0, "total": 0.0}
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    eng.timing_sums(reset=True)                     # the engine sums its HIP-event kernel times over the timed calls itself
    gather_s[0], gather_s[1] = 0.0, 0
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, k == args.steps - 1)
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0               # this rank's own clock, before it waits for the others
    if multi:
        dist.barrier()
    dt = time.perf_counter() - t0
    per_rank = None
    if multi:
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        # every rank's own time per step and time in the all-gather (outside the timed region: reporting only)
        mine_t = torch.tensor([dt_own / args.steps * 1e3, gather_s[0] / max(1, gather_s[1]) * 1e6], dtype=torch.float64, device=cdev)
        all_t = torch.zeros(2 * world, dtype=torch.float64, device=cdev)
        dist.all_gather_into_tensor(all_t, mine_t)
        per_rank = all_t.cpu().numpy().reshape(world, 2)
    ts = eng.timing_sums()
    for key in tk:
        tk[key] = ts[key] / max(1, ts["calls"])

    if rank == 0:
        folds = R_total * args.steps
        dom = "pf" if tk["pf"] >= tk["mfe"] else "mfe"
        fused = bool(eng.get_option("last_fused"))
        out = {
            "metric": "replica-folds/sec (MFE+PF, L=%d, R=%d)" % (L, args.replicas),
            "value": folds / dt, "unit": "replica-folds/s", "n_gpus": world, "collectives": (backend if multi else None), "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "int32 (MFE) + f64 (PF)", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: Eterna100-V1 #%s target (L=%d), R=%d %s-random sequences %s, "
                                   "-sf Ed-Epf (MFE fill+traceback, PF inside, eval_structure)"
                                   % (args.target[6:8], L, args.replicas, args.seqs,
                                      "in total, replica r on GPU r mod N" if strong else "per GPU"),
                       "replicas_per_gpu": R if not strong else None, "replicas_total": R_total,
                       "replicas_on_rank0": R, "L": L, "exchange_every": args.exchange_every,
                       "threads_per_workgroup": eng.info()["threads_per_wg"],
                       "launches_per_step": 1 if fused else 2},
            "kernel_ms": {k: round(v, 4) for k, v in tk.items()},
            # (one launch for both folds: the kernel whose counters the inputs hold is that launch, its duration the step's device time)
            "roofline": roofline_block(dom, tk["total"] if fused else tk[dom], L, R, tk, fused),
        }
        rb = out["roofline"]
        hb = rb.get("step_hbm")
        out["achieved_hbm_GB_s"] = hb["GB_per_s"] if hb else None          # BASELINE metric: "HBM GB/s vs peak" (8000): both folds of a step
        out["achieved_hbm_frac_of_peak"] = hb["frac_of_peak"] if hb else None
        out["sync_fallbacks"] = eng.get_option("sync_fallbacks")          # calls redone with one workgroup per fold (lost partner)
        out["cus_occupied"] = {"workgroups": eng.get_option("last_workgroups"), "compute_units": eng.info()["compute_units"],
                               # folds by several workgroups: every launch is sized against the occupancy query (engine.hip, pair_blocks_per_cu)
                               "resident_check": "%d workgroups <= %d blocks/CU x %d CUs (hipOccupancyMaxActiveBlocksPerMultiprocessor)"
                                                 % (eng.get_option("last_workgroups"), eng.get_option("fused_blocks_per_cu" if fused else "pair_blocks_per_cu"),
                                                    eng.info()["compute_units"])}
        if per_rank is not None:
            out["ranks"] = {"ms_per_step_min": float(per_rank[:, 0].min()), "ms_per_step_max": float(per_rank[:, 0].max()),
                            "allgather_us_mean": float(per_rank[:, 1].mean()), "allgather_us_max": float(per_rank[:, 1].max()),
                            "allgather_calls": gather_s[1], "allgather_bytes_per_rank": 8 * R_max,
                            "note": "ms_per_step: every rank's own clock over the timed steps, before the closing barrier; "
                                    "allgather_us: wall time of the exchange step's collective incl. staging the scores"}
        if strong:
            out["strong_scaling_note"] = ("a step is one chain of %d diagonals per fold whatever the batch (%.3f ms of kernel time with %d "
                                          "replicas on this GPU, chain-bound: SURVEY 8(d), DESIGN 3.9), so with R = %d in TOTAL ms_per_step "
                                          "stays flat as N grows and value does not scale; the weak mode (R per GPU) is the throughput mode"
                                          % (L - 4, tk["total"], R, args.replicas))
        ref = None
        if not args.no_cpu_baseline and world == 1:      # the timed CPU baseline on rank 0 at N=1 only: at N>1 the other ranks would idle at the barrier
            out["cpu_baseline"], ref = cpu_baseline(seqs, target)
            out["speedup_vs_cpu_usable_cores"] = out["value"] / world / out["cpu_baseline"]["value"]
            out["speedup_vs_cpu_reference_style_R_processes"] = out["value"] / world / out["cpu_baseline"]["reference_style_R_processes"]["value"]
        elif not args.no_cpu_baseline:
            ref = oracle_reference(seqs, target)          # N > 1: rank 0's shard against the oracle (tens of ms of CPU), no timing
        if ref is not None:
            # what the timed steps left in the device buffers against the oracle's answers for the same batch:
            # MFE energy, MFE structure and E(target) bit for bit, Epf to 1e-9 kcal/mol
            r_Epf, r_Emfe, r_ss, r_Ed = ref
            g_ss = bytes(d_ss.cpu().numpy().tobytes()).decode()
            ok = (np.array_equal(d_Emfe.cpu().numpy(), r_Emfe) and np.array_equal(d_Ed.cpu().numpy(), r_Ed[:, 0])
                  and [g_ss[k * L:(k + 1) * L] for k in range(R)] == r_ss
                  and float(np.abs(d_Epf.cpu().numpy() - r_Epf).max()) < 1e-9)
            out["parity_checked"] = bool(ok)
            out["parity"] = {"sequences": R, "max_abs_dEpf": float(np.abs(d_Epf.cpu().numpy() - r_Epf).max()),
                             "Emfe_Ed_structures": "bit-exact" if ok else "MISMATCH",
                             "of": "rank 0's shard" if world > 1 else "the whole batch"}
            if not ok:
                print(json.dumps(out))
                raise SystemExit("bench: GPU results differ from the oracle")
        if not args.no_mc_loop and not shared_card:
            out["mc_loop"] = mc_loop_block(eng, target, R, args.exchange_every, out["value"] / world, tk["total"])
        if not args.no_r_sweep and world == 1 and not shared_card:
            out["r_sweep"] = r_sweep_block(target, local_rank)
        print(json.dumps(out))
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
