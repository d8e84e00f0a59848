"""Diagnostic (GPU box): partition-function kernel times with the strip kernels for builds with extra -D flags, one per
argument (e.g. `python tools/strip_variants.py "" "-DSTRIP_DIAG=1"`); results of diagnostic builds are wrong by construction."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from desirna_amd import engine as E
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
procs = []
for k, a in enumerate(sys.argv[1:]):
    out = os.path.join(ROOT, "gpurun_out", "libsv%d.so" % k)
    procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950"] + a.split() +
                                  ["-shared", "-o", out, os.path.join(ROOT, "desirna_amd/csrc/engine.hip")], stderr=subprocess.DEVNULL))
for p in procs:
    p.wait()
rng = np.random.default_rng(11)
cases = [(400, 64), (400, 128), (400, 256), (300, 128)]
WHAT = os.environ.get("WHAT", "pf")
seqs = {c: ["".join(rng.choice(list("ACGU"), c[0])) for _ in range(c[1])] for c in cases}
for k, a in enumerate(sys.argv[1:]):
    out = os.path.join(ROOT, "gpurun_out", "libsv%d.so" % k)
    res = []
    for (L, R) in cases:
        eng = E.Engine(max_R=R, max_L=L, lib=out)
        eng.set_option("strips", 2)
        ts = []
        for _ in range(5):
            try:
                eng.score_batch(seqs[(L, R)], E.NEED_PF if WHAT == "pf" else E.NEED_MFE)
            except Exception:
                pass
            ts.append(eng.last_timing()[WHAT])
        res.append("L=%d R=%d %.3f ms" % (L, R, min(ts[1:])))
        eng.close()
    print("%-28s %s" % (a or "(default)", " | ".join(res)), flush=True)
