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


ONLY = os.environ.get("CONFIGS_ONLY", "")          # "4": config 4 only; "5": config 5 only


def run_if(tag, *a, **k):
    if not ONLY or ONLY == tag:
        run(*a, **k)


run_if("2", "config2_L100_R64_mfe_only", bench.load_target("eteV1_92.txt"), 64, E.NEED_MFE | E.NEED_EVAL)
run_if("3", "config3_L200_R64", bench.load_target("eteV1_69.txt"), 64, E.NEED_MFE | E.NEED_PF | E.NEED_EVAL)
run_if("3", "config3_shape_R128", bench.load_target("eteV1_69.txt"), 128, E.NEED_MFE | E.NEED_PF | E.NEED_EVAL)
run_if("3", "config3_shape_R256", bench.load_target("eteV1_69.txt"), 256, E.NEED_MFE | E.NEED_PF | E.NEED_EVAL)
tg = bench.load_target("eteV1_53.txt")
pk = list(tg)
run_if("5", "config5_L400_R128_pk_alt", tg, 128, E.NEED_MFE | E.NEED_PF | E.NEED_EVAL | E.NEED_PK, alts=[tg, tg], reps=3)
if ONLY and ONLY != "4":
    print(json.dumps(out, indent=1))
    sys.exit(0)
# config 4: the whole Eterna100-V1 set, 32 replicas per puzzle, one ragged call
import csv
rows = list(csv.DictReader(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "eterna_v1_solutions.csv"))))
seqs, tof = [], []
for p, r in enumerate(rows):
    for _ in range(32):
        s = list(r["sequence"])
        for pos in rng.choice(len(s), size=min(3, len(s)), replace=False):
            s[pos] = "ACGU"[rng.integers(4)]
        seqs.append("".join(s))
        tof.append(p)
eng = E.Engine(max_R=len(seqs), max_L=400, device=0)
eng.set_targets_ragged([r["structure"] for r in rows])
for _ in range(2):
    eng.score_ragged(seqs, tof)
t = eng.last_timing()
out["config4_eterna100_R32_ragged"] = {"sequences": len(seqs), "nucleotides": sum(len(s) for s in seqs), "mfe_ms": t["mfe"],
                                       "pf_ms": t["pf"], "eval_ms": t["eval"], "total_ms": t["total"],
                                       "folds_per_s": len(seqs) / (t["total"] * 1e-3), "workspace_GB": eng.info()["workspace_bytes"] / 1e9}
eng.close()
print(json.dumps(out, indent=1))
