"""Diagnostic (GPU box): run the L=200, R=64 benchmark batch a few times through a given build of the library
(e.g. a -DDRNA_SKIP variant); meant to sit behind `rocprofv3 --pmc ... -- python3 tools/run_lib.py <lib.so>`."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from desirna_amd import engine as E
import bench
lib = sys.argv[1]
R = int(sys.argv[2]) if len(sys.argv) > 2 else 64
tg = bench.load_target("eteV1_69.txt"); L = len(tg)
rng = np.random.default_rng(20260101)
seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(R)]
eng = E.Engine(max_R=R, max_L=L, lib=lib)
eng.set_targets([tg])
for _ in range(5):
    try:
        eng.score_batch(seqs, E.NEED_MFE | E.NEED_PF)
    except Exception:
        pass
print(lib, eng.last_timing())
