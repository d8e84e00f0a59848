"""Diagnostic (GPU box, DRNA_STRIP_DEBUG=1): how long each strip workgroup of an MFE launch lives, by strip index."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DRNA_STRIP_DEBUG"] = "1"
from desirna_amd import engine as E
rng = np.random.default_rng(11)
L = 400
for R in (64, 128, 256):
    seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(R)]
    eng = E.Engine(max_R=R, max_L=L)
    for _ in range(3):
        eng.score_batch(seqs, E.NEED_MFE)
    t = eng.last_timing()["mfe"]
    out = np.zeros((R, 18, 2), dtype=np.int64)
    eng._L.drna_debug_strip_clocks.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    n = eng._L.drna_debug_strip_clocks(eng._h, out.ctypes.data, R)
    t0 = out[:, :4, 0].min()
    dur = (out[:, :4, 1] - out[:, :4, 0]) / 100.0      # us
    start = (out[:, :4, 0] - t0) / 100.0
    end = (out[:, :4, 1] - t0) / 100.0
    print("R=%d kernel %.3f ms; strip s=0..3 (top .. last): mean life %s us, mean start %s us, last end %.0f us" %
          (R, t, np.round(dur.mean(0)), np.round(start.mean(0)), end.max()), flush=True)
    print("   first 3 sequences start:", np.round(start[:3]).tolist(), "end:", np.round(end[:3]).tolist())
    print("   last sequence start:", np.round(start[-1]).tolist(), "end:", np.round(end[-1]).tolist())
    eng.close()
