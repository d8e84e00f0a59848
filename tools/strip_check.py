"""Diagnostic (GPU box): the strip kernels (several workgroups per sequence) against the one-workgroup kernels -- same
structures and energies (MFE, with the pseudoknot rounds), same Epf to 1e-9 kcal/mol -- and kernel times by length and batch."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from desirna_amd import engine as E

cases = [(240, 64), (300, 64), (400, 64), (400, 128), (400, 256), (600, 32)]
if len(sys.argv) > 1:
    cases = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
rng = np.random.default_rng(11)
for L, R in cases:
    seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(R)]
    eng = E.Engine(max_R=R, max_L=L)
    for what, flags in (("pf", E.NEED_PF), ("mfe", E.NEED_MFE), ("mfe+pk", E.NEED_MFE | E.NEED_PK), ("mfe+pk+pf", E.NEED_MFE | E.NEED_PK | E.NEED_PF)):
        out = {}
        modes = (0, 1) if L > 200 else (1, 2)
        for mode in modes:
            eng.set_option("strips", mode)
            ts = []
            for _ in range(4):
                r = eng.score_batch(seqs, flags)
                ts.append(eng.last_timing()["total"])
            out[mode] = (r, min(ts[1:]))
        a, b = out[modes[0]][0], out[modes[1]][0]
        ok = True
        if flags & E.NEED_PF:
            ok = ok and np.abs(a["Epf"] - b["Epf"]).max() < 1e-9
        if flags & E.NEED_MFE:
            ok = ok and a["mfe_ss"] == b["mfe_ss"] and (a["Emfe"] == b["Emfe"]).all()
        print("L=%d R=%d %-10s one workgroup %.3f ms, strips %.3f ms (x%.2f) %s" %
              (L, R, what, out[modes[0]][1], out[modes[1]][1], out[modes[0]][1] / out[modes[1]][1], "same" if ok else "DIFFERENT"), flush=True)
    eng.close()
