"""Stress (GPU box): random lengths / batch sizes / flags through the strip kernels against the one-workgroup kernels."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from desirna_amd import engine as E
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 123)
eng = E.Engine(max_R=160, max_L=700)
bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    L = int(rng.integers(201, 700)); R = int(rng.choice([1, 2, 7, 8, 9, 31, 33, 64, 100, 160]))
    if L > 450: R = min(R, 33)
    alpha = rng.choice(["ACGU", "GC", "GGCCAU", "AU"])
    seqs = ["".join(rng.choice(list(alpha), L)) for _ in range(R)]
    flags = int(rng.choice([E.NEED_PF, E.NEED_MFE, E.NEED_MFE | E.NEED_PK, E.NEED_MFE | E.NEED_PK | E.NEED_PF, E.NEED_MFE | E.NEED_PF]))
    res = []
    for mode in (1, 0):
        eng.set_option("strips", mode)
        try:
            res.append(eng.score_batch(seqs, flags))
        except E.EngineError as ex:
            res.append(ex.code)
    a, b = res
    if isinstance(a, int) or isinstance(b, int):
        if a != b and not (isinstance(a, int) and isinstance(b, int)):
            bad += 1
            print("ERROR MISMATCH L=%d R=%d flags=%d: %s vs %s" % (L, R, flags, a if isinstance(a, int) else "ok", b if isinstance(b, int) else "ok"), flush=True)
        continue
    ok = True
    if flags & E.NEED_PF: ok &= bool(np.abs(a["Epf"] - b["Epf"]).max() < 1e-9)
    if flags & (E.NEED_MFE | E.NEED_PK): ok &= a["mfe_ss"] == b["mfe_ss"] and bool((a["Emfe"] == b["Emfe"]).all())
    if not ok:
        bad += 1
        print("MISMATCH L=%d R=%d flags=%d alphabet=%s" % (L, R, flags, alpha), flush=True)
print("cases done, mismatches:", bad, "fallbacks:", eng.get_option("sync_fallbacks"))
# ragged batches with mixed lengths
for case in range(6):
    lens = [int(x) for x in rng.integers(12, 700, size=int(rng.integers(3, 40)))]
    seqs = ["".join(rng.choice(list("ACGU"), n)) for n in lens]
    fl = E.NEED_PF | E.NEED_MFE | (E.NEED_PK if case & 1 else 0)
    eng.set_option("strips", 1); a = eng.score_ragged(seqs, flags=fl)
    eng.set_option("strips", 0); b = eng.score_ragged(seqs, flags=fl)
    ok = a["mfe_ss"] == b["mfe_ss"] and bool(np.abs(np.array(a["Epf"]) - np.array(b["Epf"])).max() < 1e-9)
    print("ragged", len(lens), "ok" if ok else "MISMATCH", flush=True)
