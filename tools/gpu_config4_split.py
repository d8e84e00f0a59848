"""GPU box: config 4 (Eterna100-V1, 32 replicas per puzzle, one ragged call) split by kernel family: n <= 200 / n > 200."""
import csv
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from desirna_amd import engine as E  # noqa: E402

rng = np.random.default_rng(20260101)
rows = list(csv.DictReader(open(os.path.join(ROOT, "tests", "golden", "eterna_v1_solutions.csv"))))
out = {}
for name, keep in (("all", lambda n: True), ("n<=200", lambda n: n <= 200), ("n>200", lambda n: n > 200)):
    sub = [r for r in rows if keep(len(r["sequence"]))]
    seqs, tof = [], []
    for p, r in enumerate(sub):
        for _ in range(32):
            s = list(r["sequence"])
            for pos in rng.choice(len(s), size=min(3, len(s)), replace=False):
                s[pos] = "ACGU"[rng.integers(4)]
            seqs.append("".join(s))
            tof.append(p)
    eng = E.Engine(max_R=len(seqs), max_L=400, device=0)
    eng.set_targets_ragged([r["structure"] for r in sub])
    for _ in range(3):
        eng.score_ragged(seqs, tof)
    t = eng.last_timing()
    out[name] = {"puzzles": len(sub), "sequences": len(seqs), "sum_n3_in_200cubed": sum(len(s) ** 3 for s in seqs) / 200.0 ** 3,
                 "mfe_ms": t["mfe"], "pf_ms": t["pf"], "total_ms": t["total"]}
    eng.close()
print(json.dumps(out, indent=1))
