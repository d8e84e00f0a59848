import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from desirna_amd import engine as E
tg = bench.load_target("eteV1_53.txt"); L = len(tg)
rng = np.random.default_rng(5)
def _rand(rng, L, a="ACGU"): return "".join(rng.choice(list(a), L))
seqs = [_rand(rng, L) for _ in range(6)] + [_rand(rng, L, "GGCCAU") for _ in range(2)]
eng = E.Engine(max_R=128, max_L=400)
eng.set_targets([tg, tg, tg])
for name, flags in (("pf", E.NEED_PF), ("mfe", E.NEED_MFE), ("mfe+pk", E.NEED_MFE | E.NEED_PK), ("mfe+pk+pf", E.NEED_MFE | E.NEED_PK | E.NEED_PF),
                    ("all", E.NEED_MFE | E.NEED_PK | E.NEED_PF | E.NEED_EVAL)):
    print("case", name, flush=True)
    for it in range(3):
        eng.score_batch(seqs, flags)
    print("  ok", eng.last_timing(), flush=True)
