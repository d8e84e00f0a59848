"""Diagnostic (GPU box): MFE strips plain vs blocked (option mfe_fark_min_strips), alone and beside the PF strips."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from desirna_amd import engine as E
cases = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(370, 256), (385, 256), (400, 256)]
for L, R in cases:
    rs = np.random.default_rng(1000 * L + R)
    seqs = ["".join(rs.choice(list("ACGU"), L)) for _ in range(R)]
    eng = E.Engine(max_R=R, max_L=400 if L <= 400 else L)
    row = []
    for thr in (99, int(os.environ.get("THR", "3"))):
        eng.set_option("mfe_fark_min_strips", thr)
        for what, flags in (("mfe", E.NEED_MFE), ("both", E.NEED_MFE | E.NEED_PF)):
            ts = []
            for _ in range(4):
                eng.score_batch(seqs, flags)
                ts.append(eng.last_timing())
            b = min(ts[1:], key=lambda t: t["total"])
            row.append("%s[%d] %.3f (mfe %.2f pf %.2f)" % (what, thr, b["total"], b["mfe"], b["pf"]))
    print("L=%d R=%d  %s" % (L, R, "  ".join(row)), flush=True)
    eng.close()
