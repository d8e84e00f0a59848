"""Diagnostic (GPU box): the strip kernels of the partition function against the one-workgroup kernels -- same Epf
(bitwise where the summation order is the same, else to 1e-9 kcal/mol) and kernel times by length and batch size."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from desirna_amd import engine as E

cases = [(200, 64), (240, 64), (300, 64), (400, 64), (400, 128), (400, 256), (600, 32)]
if len(sys.argv) > 1:
    cases = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
rng = np.random.default_rng(11)
for L, R in cases:
    seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(R)]
    eng = E.Engine(max_R=R, max_L=L)
    out = {}
    for mode in (0, 2):
        eng.set_option("strips", mode)
        ts = []
        for _ in range(5):
            r = eng.score_batch(seqs, E.NEED_PF)
            ts.append(eng.last_timing()["pf"])
        out[mode] = (np.array(r["Epf"]), min(ts[1:]))
    d = np.abs(out[0][0] - out[2][0]).max()
    print("L=%d R=%d: one workgroup %.3f ms, strips %.3f ms (x%.2f), max |dEpf| %.3g" %
          (L, R, out[0][1], out[2][1], out[0][1] / out[2][1], d), flush=True)
    eng.close()
