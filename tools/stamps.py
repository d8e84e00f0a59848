"""Diagnostic: build the engine with -DDRNA_STAMPS into gpurun_out/, run one L=200 batch, dump per-phase cycles."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "libstamps.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DDRNA_STAMPS",
                       "-shared", "-o", out, os.path.join(ROOT, "desirna_amd/csrc/engine.hip")])
import torch
from desirna_amd import engine as E
import bench
tg = bench.load_target("eteV1_69.txt"); L = len(tg); R = 64
rng = np.random.default_rng(20260101)
seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(R)]
eng = E.Engine(max_R=R, max_L=L, lib=out)
eng.set_targets([tg])
for _ in range(3):
    o = eng.score_batch(seqs, E.NEED_MFE | E.NEED_PF)
print(eng.last_timing())

ld = L + 2
buf = np.zeros(16 * 8 * 2, dtype=np.int32)
eng._L.drna_debug_read_mfe_ws.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]
rc = eng._L.drna_debug_read_mfe_ws(eng._h, 2 * ld * ld, buf.size, buf.ctypes.data)
st = buf.view(np.int64).reshape(16, 8)
names = ["T", "E", "#E-items", "barrier", "finalize", "K", "pop", "#K-items"]
print("per-wave cycles (block 0), total over %d diagonals" % (L - 4))
for w in range(16):
    print("wave %2d " % w + "  ".join("%s=%7d" % (names[k], st[w, k]) for k in range(8)), " sum=%d" % (st[w, :7].sum() - st[w, 2]))


# ---- PF kernel: stamps live in the (unused) U table of block 0
tab = ld * ld
eng._L.drna_debug_read_pf_ws.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]
pbuf = np.zeros(1024 + 12 * 256, dtype=np.float64)
rc = eng._L.drna_debug_read_pf_ws(eng._h, 5 * tab, pbuf.size, pbuf.ctypes.data)
pst = pbuf.view(np.int64)
names = ["T", "E", "X", "barrier", "finalize", "K", "queue-empty pop"]
print("PF per-wave cycles (block 0)")
for w in range(16):
    print("wave %2d " % w + "  ".join("%s=%7d" % (names[k], pst[w * 8 + k]) for k in range(7)), " sum=%d" % pst[w * 8:w * 8 + 7].sum())
t = pst[512:512 + L + 1]
dt = np.diff(t[4:L + 1])
print("PF per-diagonal cycles (sweep wave 0, barrier to barrier), d = 5 ..:")
print(" ".join("%d" % x for x in dt))
fin = np.diff(pst[256 + 4:256 + L + 1])
print("PF finalize busy cycles per step (thread 0):")
print(" ".join("%d" % x for x in fin))
bw = np.array([np.diff(pst[1024 + a * 256 + 4:1024 + a * 256 + L + 1]) for a in range(12)])
print("PF sweep waves: barrier wait per diagonal, min / max over the 12 waves")
print(" ".join("%d/%d" % (bw[:, k].min(), bw[:, k].max()) for k in range(bw.shape[1])))
