"""Diagnostic (GPU box): MFE-only batches (R=64), one- vs two-workgroup kernel, by sequence length and pk rounds."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from desirna_amd import engine as E
rng = np.random.default_rng(7)
for L in ([int(x) for x in sys.argv[1:]] or [60, 100, 130, 160, 200]):
    seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(64)]
    eng = E.Engine(max_R=64, max_L=L)
    eng.set_targets(["." * L])
    for pk in (0, E.NEED_PK):
        row = []
        for dual in (0, 2):
            eng.set_option("dual", dual)
            ts = []
            for _ in range(8):
                eng.score_batch(seqs, E.NEED_MFE | pk)
                ts.append(eng.last_timing()["mfe"])
            row.append(min(ts[2:]))
        print("L=%3d pk=%d  one workgroup %.3f ms   two workgroups %.3f ms" % (L, bool(pk), row[0], row[1]), flush=True)
    # both folds of the batch (the engine's own choice for the partition function), MFE by one / two workgroups: the call's device time
    row = []
    for dual in (0, 2):
        eng.set_option("dual", dual)
        ts = []
        for _ in range(8):
            eng.score_batch(seqs, E.NEED_MFE | E.NEED_PF | E.NEED_EVAL)
            ts.append(eng.last_timing()["total"])
        row.append(min(ts[2:]))
    print("L=%3d MFE + PF + eval: one workgroup per MFE fold %.3f ms   two %.3f ms (one launch: %d)" % (L, row[0], row[1], eng.get_option("last_fused")), flush=True)
    eng.close()
