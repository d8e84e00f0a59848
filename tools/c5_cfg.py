import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from desirna_amd import engine as E
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
tg = bench.load_target("eteV1_53.txt"); L = len(tg); R = 128
seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(R)]
eng = E.Engine(max_R=R, max_L=L, device=0)
eng.set_targets([tg, tg, tg])
for _ in range(6):
    eng.score_batch(seqs, E.NEED_MFE | E.NEED_PF | E.NEED_EVAL | E.NEED_PK)
    t = eng.last_timing(); print({k: round(v, 2) for k, v in t.items()})
