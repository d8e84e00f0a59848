"""GPU box: device time of the two-strand path (co-fold MFE + PF + eval)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from desirna_amd import engine as E
out = {}
rng = np.random.default_rng(5)
for R, la, lb in ((64, 18, 18), (64, 50, 50), (64, 100, 100)):
    seqs = ["".join(rng.choice(list("ACGU"), la)) + "&" + "".join(rng.choice(list("ACGU"), lb)) for _ in range(R)]
    eng = E.Engine(max_R=R, max_L=la + lb, device=0)
    eng.set_targets(["." * (la + lb)])
    for _ in range(3):
        eng.cofold_batch(seqs)
    t = eng.last_timing()
    out["R%d_%d+%d" % (R, la, lb)] = {"mfe_ms": t["mfe"], "pf_ms": t["pf"], "pairs_per_s": R / (t["total"] * 1e-3)}
    eng.close()
print(json.dumps(out))
