"""Diagnostic (GPU box): PF kernel time against the set of finalize waves that join the work queue (-DDRNA_JOIN_MASK)."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from desirna_amd import engine as E
import bench
tg = bench.load_target("eteV1_69.txt"); L = len(tg); R = 64
rng = np.random.default_rng(20260101)
seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(R)]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for m in [int(x, 0) for x in sys.argv[1:]] or [0xF, 0x6, 0xE, 0x7, 0x2, 0x4]:
    out = os.path.join(ROOT, "gpurun_out", "libjoin%d.so" % m)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DDRNA_JOIN_MASK=%d" % m,
                           "-shared", "-o", out, os.path.join(ROOT, "desirna_amd/csrc/engine.hip")], stderr=subprocess.DEVNULL)
    eng = E.Engine(max_R=R, max_L=L, lib=out)
    eng.set_targets([tg])
    ts = []
    for _ in range(8):
        eng.score_batch(seqs, E.NEED_MFE | E.NEED_PF)
        ts.append(eng.last_timing())
    print("join mask 0x%x: pf %.4f ms (median %.4f)  mfe %.4f ms" % (m, min(x["pf"] for x in ts[2:]), float(np.median([x["pf"] for x in ts[2:]])), min(x["mfe"] for x in ts[2:])), flush=True)
    eng.close()
