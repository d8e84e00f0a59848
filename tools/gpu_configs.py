"""GPU box: device time of the BASELINE parity configs other than the headline one (not bench lines)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from desirna_amd import engine as E  # noqa: E402

out = {}
rng = np.random.default_rng(20260101)


def run(name, tg, R, flags, alts=(), reps=4):
    L = len(tg)
    seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(R)]
    eng = E.Engine(max_R=R, max_L=L, device=0)
    eng.set_targets([tg] + list(alts))
    for _ in range(reps):
        eng.score_batch(seqs, flags)
    t = eng.last_timing()
    out[name] = {"R": R, "L": L, "mfe_ms": t["mfe"], "pf_ms": t["pf"], "eval_ms": t["eval"], "total_ms": t["total"],
                 "folds_per_s": R / (t["total"] * 1e-3)}
    eng.close()


run("config2_L100_R64_mfe_only", bench.load_target("eteV1_92.txt"), 64, E.NEED_MFE | E.NEED_EVAL)
run("config3_L200_R64", bench.load_target("eteV1_69.txt"), 64, E.NEED_MFE | E.NEED_PF | E.NEED_EVAL)
run("config3_shape_R128", bench.load_target("eteV1_69.txt"), 128, E.NEED_MFE | E.NEED_PF | E.NEED_EVAL)
run("config3_shape_R256", bench.load_target("eteV1_69.txt"), 256, E.NEED_MFE | E.NEED_PF | E.NEED_EVAL)
tg = bench.load_target("eteV1_53.txt")
pk = list(tg)
run("config5_L400_R128_pk_alt", tg, 128, E.NEED_MFE | E.NEED_PF | E.NEED_EVAL | E.NEED_PK, alts=[tg, tg], reps=3)
print(json.dumps(out, indent=1))
