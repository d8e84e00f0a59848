"""Diagnostic (GPU box): where the MFE LDS kernel's time goes outside the diagonal loop (-DDRNA_PROCLK build)."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "libproclk.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DDRNA_PROCLK", "-DDRNA_STAMPS_API"] + sys.argv[1:] +
                      ["-shared", "-o", out, os.path.join(ROOT, "desirna_amd/csrc/engine.hip")], stderr=subprocess.DEVNULL)
from desirna_amd import engine as E
import bench
tg = bench.load_target("eteV1_69.txt"); L = len(tg); R = 64
rng = np.random.default_rng(20260101)
seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(R)]
for dual in ("0", "1"):
    os.environ["DRNA_DUAL"] = dual
    eng = E.Engine(max_R=R, max_L=L, lib=out)
    eng.set_targets([tg])
    for _ in range(3):
        eng.score_batch(seqs, E.NEED_MFE | E.NEED_PF)
    ld = L + 2
    eng._L.drna_debug_read_mfe_ws.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]
    buf = np.zeros(16, dtype=np.int32)
    eng._L.drna_debug_read_mfe_ws(eng._h, 2 * ld * ld + 1024, buf.size, buf.ctypes.data)
    t = buf.view(np.int64)
    names = ["table loads + sequence", "prologue of the fill (tables, pairable lists)", "first-diagonal tables", "diagonal loop", "exterior tail",
             "traceback", "round bookkeeping + output"]
    print("dual=%s  %s" % (dual, eng.last_timing()))
    for k, nm in enumerate(names):
        print("   %-48s %8d cycles" % (nm, t[k + 1] - t[k]))
    eng.close()
