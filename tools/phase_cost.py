"""Diagnostic (GPU box): marginal cost of each sweep phase of the LDS kernels.  Builds the engine with -DDRNA_SKIP=mask
(a phase left out; results are wrong by construction) into gpurun_out/ and reports kernel times for the L=200, R=64 batch."""
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
masks = [int(x) for x in sys.argv[1:]] or [0, 1, 2, 4, 8, 3, 15]
for m in masks:
    out = os.path.join(ROOT, "gpurun_out", "libskip%d.so" % m)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DDRNA_SKIP=%d" % m,
                           "-shared", "-o", out, os.path.join(ROOT, "desirna_amd/csrc/engine.hip")], stderr=subprocess.DEVNULL)
    eng = E.Engine(max_R=R, max_L=L, lib=out)
    eng.set_targets([tg])
    ts = []
    for _ in range(6):
        try:
            eng.score_batch(seqs, E.NEED_MFE | E.NEED_PF)
        except Exception:
            pass
        ts.append(eng.last_timing())
    t = ts[-1]
    print("skip mask %2d: mfe %.3f ms  pf %.3f ms" % (m, min(x["mfe"] for x in ts[2:]), min(x["pf"] for x in ts[2:])), flush=True)
    eng.close()
