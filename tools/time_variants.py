"""Diagnostic (GPU box): kernel times of the L=200, R=64 batch for every prebuilt engine variant in build/var/ (tools/build_variants.sh)."""
import glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from desirna_amd import engine as E
import bench
tg = bench.load_target("eteV1_69.txt"); L = len(tg); R = int(os.environ.get("VAR_R", "64"))
rng = np.random.default_rng(20260101)
seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(R)]
want = sys.argv[1:]
for lib in sorted(glob.glob(os.path.join(ROOT, "build", "var", "lib_*.so"))):
    name = os.path.basename(lib)[4:-3]
    if want and name not in want:
        continue
    eng = E.Engine(max_R=R, max_L=L, lib=lib)
    eng.set_targets([tg])
    ts = []
    for _ in range(10):
        try:
            eng.score_batch(seqs, E.NEED_MFE | E.NEED_PF)
        except Exception:
            pass
        ts.append(eng.last_timing())
    print("%-28s mfe %.4f ms  pf %.4f ms" % (name, min(x["mfe"] for x in ts[2:]), min(x["pf"] for x in ts[2:])), flush=True)
    eng.close()
