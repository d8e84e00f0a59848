"""Diagnostic (GPU box): MFE with pseudoknot rounds + PF on strips, by DRNA_PF_GATE (the pk round after which the PF launch starts)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from desirna_amd import engine as E
cases = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(400, 128)]
for L, R in cases:
    rs = np.random.default_rng(1000 * L + R)
    seqs = ["".join(rs.choice(list("ACGU"), L)) for _ in range(R)]
    eng = E.Engine(max_R=R, max_L=L)
    ts = []
    for _ in range(4):
        eng.score_batch(seqs, E.NEED_MFE | E.NEED_PK | E.NEED_PF)
        ts.append(eng.last_timing())
    b = min(ts[1:], key=lambda t: t["total"])
    print("gate %s  L=%d R=%d  total %.3f (mfe %.2f pf %.2f)" % (os.environ.get("DRNA_PF_GATE", "-"), L, R, b["total"], b["mfe"], b["pf"]), flush=True)
    eng.close()
