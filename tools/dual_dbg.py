"""Diagnostic (GPU box): cycle counters of the two-workgroup MFE kernel (-DDRNA_DUALDBG build): who waits for whom."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "libdualdbg.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DDRNA_DUALDBG", "-DDRNA_STAMPS_API"] + sys.argv[1:] +
                      ["-shared", "-o", out, os.path.join(ROOT, "desirna_amd/csrc/engine.hip")], stderr=subprocess.DEVNULL)
from desirna_amd import engine as E
import bench
tg = bench.load_target("eteV1_69.txt"); L = len(tg); R = 64
rng = np.random.default_rng(20260101)
seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(R)]
eng = E.Engine(max_R=R, max_L=L, lib=out)
eng.set_targets([tg])
for _ in range(3):
    eng.score_batch(seqs, E.NEED_MFE | E.NEED_PF)
print(eng.last_timing())
ld = L + 2
eng._L.drna_debug_read_mfe_ws.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]
buf = np.zeros(64 * 64 * 2, dtype=np.int32)
eng._L.drna_debug_read_mfe_ws(eng._h, 2 * ld * ld + 8192, buf.size, buf.ctypes.data)
d = buf.view(np.int64).reshape(64, 64)
names = {0: "A blocking waits (count)", 1: "A cycles in blocking waits", 2: "A total cycles", 3: "A drain cycles (wave 0)", 4: "A barrier cycles (wave 0)",
         8: "B inbound wait cycles", 9: "B inbound copy cycles", 10: "B tables cycles", 11: "B outbound cycles", 12: "B worker(3) item cycles",
         13: "B barrier cycles (wave 0)", 14: "B worker(3) items", 15: "B total cycles"}
for k, nm in names.items():
    print("%-32s median %10d   min %10d   max %10d" % (nm, np.median(d[:, k]), d[:, k].min(), d[:, k].max()))
