"""GPU box: time the ensemble-defect path (general inside kernel + outside kernel) at configs 3 and 5 shapes."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from desirna_amd import engine, workloads  # noqa: E402

out = {}
for R, L in ((64, 200), (128, 400)):
    rng = np.random.default_rng(20260101)
    seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(R)]
    eng = engine.Engine(max_R=R, max_L=L, device=0)
    eng.set_targets(["." * L])
    for _ in range(3):
        eng.ensemble_defect(seqs)
    t = eng.last_edef_timing()
    out["R%d_L%d" % (R, L)] = {"inside_ms": t["inside"], "outside_ms": t["outside"],
                               "defects_per_s": R / ((t["inside"] + t["outside"]) * 1e-3)}
    eng.close()
print(json.dumps(out))
