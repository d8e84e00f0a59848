"""Profiling target (GPU box, under rocprofv3): the strip kernels on a batch that fills the chip (L = 400, R = 256) and one
that does not (R = 64); MFE (no pseudoknot rounds) and partition function in separate calls so that the kernels do not overlap."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from desirna_amd import engine as E
rng = np.random.default_rng(11)
L = 400
for R in (256, 64):
    seqs = ["".join(rng.choice(list("ACGU"), L)) for _ in range(R)]
    eng = E.Engine(max_R=R, max_L=L)
    for mode in (1, 0):
        eng.set_option("strips", mode)
        for _ in range(3):
            eng.score_batch(seqs, E.NEED_PF)
            eng.score_batch(seqs, E.NEED_MFE)
    eng.close()
