"""GPU box: long sequences through the general kernels against the oracle (maximum-size check, not a bench line)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from desirna_amd import engine as E  # noqa: E402
from oracle import pyoracle  # noqa: E402

from desirna_amd import params  # noqa: E402
pyoracle.build()
orc = pyoracle.Oracle(params.load_blob())
rng = np.random.default_rng(4242)
for L in (600, 2046):
    seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(2)]
    tg = "." * L
    eng = E.Engine(max_R=2, max_L=L, device=0)
    eng.set_targets([tg])
    t0 = time.time()
    out = eng.score_batch(seqs, E.NEED_MFE | E.NEED_PF | E.NEED_EVAL)
    t1 = time.time()
    tim = eng.last_timing()
    for k, s in enumerate(seqs):
        t2 = time.time()
        ss, e = orc.mfe(s)
        f = orc.pf(s)
        t3 = time.time()
        ok = (ss == out["mfe_ss"][k], e == int(out["Emfe"][k]), abs(f - float(out["Epf"][k])))
        print("L=%d seq %d: structure %s, Emfe %s (%d), Epf %.9f vs %.9f, |dEpf| %.2e; gpu %.1f ms (mfe %.1f pf %.1f), oracle %.1f s" %
              (L, k, ok[0], ok[1], e, float(out["Epf"][k]), f, ok[2], (t1 - t0) * 1e3, tim["mfe"], tim["pf"], t3 - t2), flush=True)
    eng.close()
