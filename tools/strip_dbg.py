import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from desirna_amd import engine as E
rng = np.random.default_rng(11)
L, R = 400, 64
seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(R)]
eng = E.Engine(max_R=R, max_L=L)
for it in range(30):
    t = time.time()
    try:
        eng.score_batch(seqs, E.NEED_PF)
    except Exception as ex:
        print("iter", it, "FAILED after %.2f s" % (time.time() - t), ex, flush=True)
        break
else:
    print("30 calls ok", eng.last_timing())
