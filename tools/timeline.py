"""Diagnostic (GPU box): per-wave timeline of the PF LDS kernel's main workgroup of sequence 0, from a -DDRNA_TL build in
build/var/lib_tl.so (tools/build_variants.sh tl "-DDRNA_TL").  Per step and wave: 100 MHz clock at barrier exit, after the
finalize / tower step, after the last item.  Prints averages over step ranges in microseconds."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from desirna_amd import engine as E
import bench
tg = bench.load_target("eteV1_69.txt"); L = len(tg); R = int(os.environ.get("TL_R", "64"))      # TL_R=128: no helper workgroups
rng = np.random.default_rng(20260101)
seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(R)]
lib = os.path.join(ROOT, "build", "var", "lib_%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "tl"))
TURN1 = 4
MFE = len(sys.argv) > 2 and sys.argv[2] == "mfe"          # the MFE kernel's marks (table 2 of its workspace) instead of the PF kernel's
eng = E.Engine(max_R=R, max_L=L, lib=lib)
eng.set_targets([tg])
for kv in os.environ.get("TL_OPTS", "").split():          # e.g. TL_OPTS="pf_helper=0 dual=0"
    k, v = kv.split("="); eng.set_option(k, int(v))
for _ in range(3):
    eng.score_batch(seqs, E.NEED_MFE | E.NEED_PF)
print(eng.last_timing())
ld = L + 2; tab = ld * ld
eng._L.drna_debug_read_pf_ws.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]
buf = np.zeros(16 * 3 * 256 + 4 * 3 * 256 + 16 * 256, dtype=np.float64)
if MFE:
    eng._L.drna_debug_read_mfe_ws.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]
    eng._L.drna_debug_read_mfe_ws(eng._h, 2 * tab, 16 * 3 * 256 * 2, buf.ctypes.data)
else:
    eng._L.drna_debug_read_pf_ws(eng._h, 4 * tab + tab // 2, buf.size, buf.ctypes.data)
t = buf.view(np.int64)[:16 * 3 * 256].reshape(16, 3, 256).astype(np.float64) / 100.0      # microseconds
t2 = buf.view(np.int64)[16 * 3 * 256:16 * 3 * 256 + 4 * 3 * 256].reshape(4, 3, 256).astype(np.float64) / 100.0
t3 = buf.view(np.int64)[60 * 256:76 * 256].reshape(16, 256).astype(np.float64) / 100.0       # PF sweep waves: tile products done (zero without tiles)
if not MFE and t[0, 0, 0] > 0:
    print("PF main workgroup: kernel entry -> first step %.2f us, steps %.2f us, last barrier -> Z stored %.2f us" %
          (t[0, 0, TURN1] - t[0, 0, 0], t[0, 0, 2] - t[0, 0, TURN1], t[0, 0, 3] - t[0, 0, 2]))
if MFE and t[0, 0, 0] > 0:
    print("MFE main workgroup: kernel entry -> first step %.2f us, steps %.2f us, traceback %.2f us" %
          (t[0, 0, TURN1] - t[0, 0, 0], t[0, 0, 1] - t[0, 0, TURN1], t[0, 0, 2] - t[0, 0, 1]))
    e = t[0, 0, 0]
    print("   prologue, from kernel entry: tables + sequence in LDS %.2f, codes published %.2f, LDS tables of the fill %.2f, list rows %.2f, first step %.2f us" %
          (t[0, 1, 0] - e, t[0, 1, 1] - e, t[0, 1, 2] - e, t[0, 1, 3] - e, t[0, 0, TURN1] - e))
if MFE:
    late = int(buf.view(np.int64)[12287])
    print("steps in which the main role had to wait for its helper (its results were not there a step ahead), by 25 steps:",
          [(late >> (8 * b)) & 255 for b in range(8)])
if MFE:
    raw64 = np.zeros(22 * 256, dtype=np.int64)
    eng._L.drna_debug_read_mfe_ws(eng._h, 2 * tab + 2 * (49 << 8), 2 * raw64.size, raw64.ctypes.data)
    h = raw64.reshape(22, 256).astype(np.float64) / 100.0
    print("helper of sequence 0, per step (us from its top): inbound copy done, outbound stores done, next diagonal's tables done, first / last worker wave done, step length")
    for lo, hi in ((20, 72), (72, 125), (125, 150), (150, 175), (175, 197)):
        ks = np.arange(lo, hi)
        print("   steps %3d..%3d: inbound +%.2f  outbound +%.2f  tables +%.2f  workers +%.2f / +%.2f  step %.2f   (main role's step %.2f)" %
              (lo, hi, (h[1, ks] - h[0, ks]).mean(), (h[2, ks] - h[0, ks]).mean(), (h[5, ks] - h[0, ks]).mean(), (h[3, ks] - h[0, ks]).mean(), (h[4, ks] - h[0, ks]).mean(),
               np.diff(h[0, lo:hi + 1]).mean(), np.diff(t[0, 0, lo:hi + 1]).mean()))
        wk = h[6 + 3:6 + 16][:, ks] - h[0, ks]
        print("        worker waves 3..15 done at: " + " ".join("%.2f" % x for x in wk.mean(axis=1)) + "   last of them +%.2f" % wk.max(axis=0).mean())
if MFE and os.environ.get("TL_TB"):
    raw = np.zeros(64, dtype=np.int64)
    eng._L.drna_debug_read_mfe_ws(eng._h, 2 * tab + 2 * (72 << 8), 2 * raw.size, raw.ctypes.data)
    print("traceback of sequence 0 (-DDRNA_TL -DDRNA_TL_TB): per wave start, end (us from the fill's end), busy us, sectors, pair events")
    for w in range(8):
        o = raw[8 * w: 8 * w + 5]
        print("   wave %d: %.2f .. %.2f  busy %.2f  sectors %d  events %d" % (w, o[0] / 100.0 - t[0, 0, 1], o[1] / 100.0 - t[0, 0, 1], o[2] / 100.0, o[3], o[4]))
for lo, hi in ((10, 40), (40, 72), (72, 110), (110, 150), (150, 196)):
    ks = np.arange(lo, hi)
    t0 = t[:, 0, ks].min(axis=0)                      # first wave out of the barrier
    step = np.diff(t[0, 0, lo:hi + 1]).mean()
    print("steps %3d..%3d: step %.2f us" % (lo, hi, step))
    for w in range(16):
        a = (t[w, 0, ks] - t0).mean(); b = (t[w, 1, ks] - t0).mean(); c = (t[w, 2, ks] - t0).mean()
        extra = ""
        if w < 4 and not MFE:
            extra = "   [requests issued +%.2f  cells finalized +%.2f]" % ((t2[w, 0, ks] - t0).mean(), (t2[w, 1, ks] - t0).mean())
        if w >= 4 and not MFE and t3[w, ks].max() > 0:
            extra = "   [tile products: %.2f us]" % (t3[w, ks] - t[w, 1, ks]).mean()
        print("   wave %2d: out of barrier +%.2f  own job done +%.2f  items done +%.2f%s" % (w, a, b, c, extra))
