"""GPU box: kernel time of the one-workgroup MFE fold (dual off) at L = 200 for the libraries given (diagnostic builds)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from desirna_amd import engine as E
import bench
tg = bench.load_target("eteV1_69.txt"); L = len(tg)
rng = np.random.default_rng(20260101)
for R in (64, 128):
    seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(R)]
    for lib in sys.argv[1:]:
        eng = E.Engine(max_R=R, max_L=L, lib=os.path.join(ROOT, lib))
        eng.set_option("dual", 0)
        eng.set_targets([tg])
        ts = []
        for _ in range(30):
            try:
                eng.score_batch(seqs, E.NEED_MFE)
            except Exception:
                pass
            ts.append(eng.last_timing()["mfe"])
        print("R %3d %-32s mfe %.4f ms (median %.4f)" % (R, lib, min(ts[3:]), float(np.median(ts[3:]))), flush=True)
        eng.close()
