"""Diagnostic (GPU box): strip-kernel times (PF and MFE alone, then both at once) of LxR batches for every prebuilt engine
variant in build/var/ (tools/build_variants.sh); Epf of every variant is compared with the first one's.
   python tools/time_strip_variants.py [LxR ...] [--only name,name]"""
import glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from desirna_amd import engine as E

cases, want = [], None
for a in sys.argv[1:]:
    if a.startswith("--only="):
        want = a[7:].split(",")
    else:
        cases.append(tuple(int(x) for x in a.split("x")))
cases = cases or [(400, 64), (400, 256)]
rng = np.random.default_rng(11)
ref = {}
for lib in sorted(glob.glob(os.path.join(ROOT, "build", "var", "lib_*.so"))):
    name = os.path.basename(lib)[4:-3]
    if want and name not in want:
        continue
    for L, R in cases:
        rs = np.random.default_rng(1000 * L + R)
        seqs = ["".join(rs.choice(list("ACGU"), L)) for _ in range(R)]
        eng = E.Engine(max_R=R, max_L=L, lib=lib)
        row = []
        for what, flags in (("pf", E.NEED_PF), ("mfe", E.NEED_MFE), ("both", E.NEED_MFE | E.NEED_PF))[:int(os.environ.get("NWHAT", "3"))]:
            ts = []
            for _ in range(5):
                try:
                    r = eng.score_batch(seqs, flags)
                except E.EngineError as ex:                 # timing builds leave phases out: their status words may be off -- said, not swallowed
                    print("   (%s %s: %s)" % (name, what, str(ex)[:100]), flush=True)
                ts.append(eng.last_timing()["total"])
            row.append("%s %.3f" % (what, min(ts[1:])))
            if flags == E.NEED_PF:
                key = (L, R)
                if key not in ref:
                    ref[key] = r["Epf"].copy()
                dev = float(np.abs(r["Epf"] - ref[key]).max())
        fb = eng.get_option("sync_fallbacks")             # calls redone with one workgroup per fold: such a time is not a strip time
        print("%-22s L=%d R=%-4d %s ms   max|dEpf| vs first %.2e  fallbacks %d" % (name, L, R, "  ".join(row), dev, fb), flush=True)
        eng.close()
        if fb:
            sys.exit("time_strip_variants: %s lost a strip %d times (ST_SYNC): the times above include one-workgroup re-runs" % (name, fb))
