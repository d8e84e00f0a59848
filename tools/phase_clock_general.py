"""Diagnostic (GPU box): per-phase time of the general PF kernel (-DDRNA_PHASECLK build into gpurun_out/), L=400, R=128."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "libphaseclk.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DDRNA_PHASECLK",
                       "-shared", "-o", out, os.path.join(ROOT, "desirna_amd/csrc/engine.hip")])
from desirna_amd import engine as E
import bench
tg = bench.load_target("eteV1_53.txt"); L = len(tg); R = 128
rng = np.random.default_rng(20260101)
seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(R)]
eng = E.Engine(max_R=R, max_L=L, lib=out)
eng.set_targets([tg])
for _ in range(2):
    eng.score_batch(seqs, E.NEED_PF)
print(eng.last_timing())

# LDS-resident MFE kernel (L=200): fill / traceback split of block 0
tg = bench.load_target("eteV1_69.txt"); L = len(tg); R = 64
seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(R)]
eng2 = E.Engine(max_R=R, max_L=L, lib=out)
eng2.set_targets([tg])
eng2.score_batch(seqs, E.NEED_MFE)
print(eng2.last_timing())
