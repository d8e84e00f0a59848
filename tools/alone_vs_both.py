"""Diagnostic (GPU box): the two folds of the headline batch (R = 64 x L = 200) alone and side by side."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from desirna_amd import engine as E
import bench
tg = bench.load_target("eteV1_69.txt"); L = len(tg); R = int(os.environ.get("VAR_R", "64"))
rng = np.random.default_rng(20260101)
seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(R)]
eng = E.Engine(max_R=R, max_L=L)
eng.set_targets([tg])
for what, flags in (("mfe alone", E.NEED_MFE), ("pf alone", E.NEED_PF), ("both", E.NEED_MFE | E.NEED_PF), ("both + eval", E.NEED_MFE | E.NEED_PF | E.NEED_EVAL)):
    ts = []
    for _ in range(12):
        eng.score_batch(seqs, flags)
        ts.append(eng.last_timing())
    print("%-12s mfe %.4f  pf %.4f  total %.4f ms   workgroups %d" % (what, min(t["mfe"] for t in ts[2:]), min(t["pf"] for t in ts[2:]),
                                                                     min(t["total"] for t in ts[2:]), eng.get_option("last_workgroups")))
eng.close()
