"""Diagnostic (GPU box): per-wave cycle totals (own job | items | barrier wait) of the last MFE strip of sequence 0, for a
build with -DMSTRIP_STAMPS (plus optional extra flags in argv)."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DRNA_STRIP_DEBUG"] = "1"
from desirna_amd import engine as E
for k, extra in enumerate(sys.argv[1:] or [""]):
    out = os.path.join(ROOT, "gpurun_out", "libst%d.so" % k)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DMSTRIP_STAMPS"] + extra.split() +
                          ["-shared", "-o", out, os.path.join(ROOT, "desirna_amd/csrc/engine.hip")], stderr=subprocess.DEVNULL)
    rng = np.random.default_rng(11)
    L, R = 400, 64
    seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(R)]
    eng = E.Engine(max_R=R, max_L=L, lib=out)
    for _ in range(3):
        try:
            eng.score_batch(seqs, E.NEED_MFE)
        except Exception:
            pass
    t = eng.last_timing()["mfe"]
    buf = np.zeros((R, 18, 2), dtype=np.int64)
    eng._L.drna_debug_strip_clocks.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    eng._L.drna_debug_strip_clocks(eng._h, buf.ctypes.data, R)
    st = buf.reshape(-1)[16:16 + 64].reshape(16, 4)
    print("build '%s': kernel %.3f ms; per wave k-cycles (own job | items | barrier):" % (extra, t))
    names = ["fin0", "fin1", "svcA", "svcB"] + ["tow%d" % i for i in range(6)] + ["flt%d" % i for i in range(6)]
    for w in range(16):
        print("   %-5s %7.0f %7.0f %7.0f" % (names[w], st[w, 0] / 1e3, st[w, 1] / 1e3, st[w, 2] / 1e3))
    eng.close()
