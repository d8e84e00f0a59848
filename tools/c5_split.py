"""GPU box: config 5's shape (400 nt x 128, pseudoknot rounds, two alternative targets) by the number of parts the MFE chain's
batch is split into (option mfe_split) and by the round the partition function's launch waits for (DRNA_PF_GATE)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from desirna_amd import engine as E
tg = bench.load_target("eteV1_53.txt"); L = len(tg); R = int(os.environ.get("C5_R", "128"))
rng = np.random.default_rng(20260101)
seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(R)]
flags = E.NEED_MFE | E.NEED_PF | E.NEED_EVAL | E.NEED_PK
for split in [int(x) for x in (sys.argv[1:] or ["1", "2", "3", "4"])]:
    eng = E.Engine(max_R=R, max_L=L)
    eng.set_targets([tg, tg, tg])
    eng.set_option("mfe_split", split)
    ts = []
    for _ in range(5):
        eng.score_batch(seqs, flags)
        ts.append(eng.last_timing())
    b = min(ts[1:], key=lambda t: t["total"])
    m = eng.score_batch(seqs, E.NEED_MFE | E.NEED_PK) and eng.last_timing()
    print("mfe_split %d gate %s: total %.3f ms (mfe %.3f pf %.3f)   mfe chain alone %.3f   fallbacks %d" %
          (split, os.environ.get("DRNA_PF_GATE", "rule"), b["total"], b["mfe"], b["pf"], m["mfe"], eng.get_option("sync_fallbacks")), flush=True)
    eng.close()
