"""Diagnostic (GPU box): MFE-only and MFE+PF kernel times (R=64, one workgroup per fold) by length, for builds with extra -D flags."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from desirna_amd import engine as E
procs = []
for k, a in enumerate(sys.argv[1:]):
    out = os.path.join(ROOT, "gpurun_out", "liblen%d.so" % k)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950"] + a.split() +
                                  ["-shared", "-o", out, os.path.join(ROOT, "desirna_amd/csrc/engine.hip")], stderr=subprocess.DEVNULL))
for p in procs:
    p.wait()
for k, a in enumerate(sys.argv[1:]):
    out = os.path.join(ROOT, "gpurun_out", "liblen%d.so" % k)
    rng = np.random.default_rng(7)
    res = []
    for L in (60, 100, 140, 200):
        seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(64)]
        eng = E.Engine(max_R=64, max_L=L, lib=out)
        eng.set_option("dual", 0)
        eng.set_targets(["." * L])
        ts = []
        for _ in range(8):
            eng.score_batch(seqs, E.NEED_MFE | E.NEED_PF)
            ts.append(eng.last_timing())
        res.append("L=%d mfe %.3f pf %.3f" % (L, min(t["mfe"] for t in ts[2:]), min(t["pf"] for t in ts[2:])))
        eng.close()
    print("%-24s %s" % (a or "(default)", " | ".join(res)), flush=True)
