"""Diagnostic (GPU box): per-diagonal step time of the PF LDS kernel with one clock read per step by one thread
(-DDRNA_STEPCLK=block+1), for the L=200, R=64 batch.  Prints cycles per step for d = 5 .. n-1."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from desirna_amd import engine as E
import bench
tg = bench.load_target("eteV1_69.txt"); L = len(tg); R = 64
rng = np.random.default_rng(20260101)
seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(R)]
extra = sys.argv[1:]
out = os.path.join(ROOT, "gpurun_out", "libstepclk.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DDRNA_STEPCLK=1", "-DDRNA_STAMPS_API"] + extra +
                      ["-shared", "-o", out, os.path.join(ROOT, "desirna_amd/csrc/engine.hip")], stderr=subprocess.DEVNULL)
eng = E.Engine(max_R=R, max_L=L, lib=out)
eng.set_targets([tg])
for _ in range(3):
    eng.score_batch(seqs, E.NEED_MFE | E.NEED_PF)
print(eng.last_timing())
ld = L + 2
eng._L.drna_debug_read_pf_ws.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]
buf = np.zeros(256, dtype=np.float64)
eng._L.drna_debug_read_pf_ws(eng._h, 5 * ld * ld, 256, buf.ctypes.data)
t = buf.view(np.int64)[4:L + 1]
dt = np.diff(t)
print("cycles per step, d = 4 ..", " ".join(str(int(x)) for x in dt))
print("sum", int(dt.sum()), "mean", float(dt.mean()))
