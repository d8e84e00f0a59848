"""GPU box: randomized parity sweep of the production kernels against the CPU oracle (lengths 1 ... 200 for the LDS-resident
path, a few longer ones for the general path; uniform, GC-rich, AU-rich and low-complexity sequences)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from desirna_amd import engine as E, params  # noqa: E402
from oracle.pyoracle import Oracle, FLAG_PF, FLAG_MFE  # noqa: E402

orc = Oracle(params.load_blob())
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
eng = E.Engine(max_R=64, max_L=420, device=0)
alph = ["ACGU", "GC", "GCGCAU", "AU", "GGGGGGCCCCCCAAAU", "ACGUACGUGGCC"]
bad = 0
t0 = time.time()
lengths = list(range(1, 201)) + [201, 222, 256, 257, 300, 333, 400, 420]
for L in lengths:
    R = 12 if L <= 200 else 4
    seqs = ["".join(rng.choice(list(alph[k % len(alph)]), L)) for k in range(R)]
    tg = orc.mfe(seqs[0])[0]
    eng.set_targets([tg])
    out = eng.score_batch(seqs)
    rEpf, rEmfe, rss, rEd = orc.score_batch(seqs, [tg], FLAG_PF | FLAG_MFE, threads=0)
    for k, s in enumerate(seqs):
        ok = (out["mfe_ss"][k] == rss[k] and int(out["Emfe"][k]) == int(rEmfe[k]) and
              abs(float(out["Epf"][k]) - float(rEpf[k])) < 1e-9 and int(out["Ed"][k, 0]) == int(rEd[k, 0]))
        if not ok:
            bad += 1
            print("MISMATCH L=%d %s\n  gpu %s %d %.9f %d\n  cpu %s %d %.9f %d" % (
                L, s, out["mfe_ss"][k], out["Emfe"][k], out["Epf"][k], out["Ed"][k, 0], rss[k], rEmfe[k], rEpf[k], rEd[k, 0]))
print("lengths %d, sequences checked, mismatches %d, %.1f s" % (len(lengths), bad, time.time() - t0))
sys.exit(1 if bad else 0)
