"""Synthetic inputs for benchmarks and tests: uniform-random and design-like sequences."""
import numpy as np

from .sim_score import pair_table

_WC = {"A": "U", "U": "A", "G": "C", "C": "G"}
_PAIRS = [("G", "C"), ("C", "G"), ("A", "U"), ("U", "A"), ("G", "U"), ("U", "G")]


def uniform_sequences(L, R, rng):
    return ["".join(rng.choice(list("ACGU"), L)) for _ in range(R)]


def design_like_sequences(target, R, rng, max_mutations=50):
    """What the engine sees during a design run: the reference's initial-sequence rule
    (utils/sequence_utils.py:686-724 with all-N restraints: unpaired -> A, first base of an unpaired
    stretch of length >= 2 that follows a paired base -> G, paired -> random G/C Watson-Crick pair)
    followed by k in [0, max_mutations] random structure-compatible point / pair mutations."""
    pt = pair_table(target.replace("[", ".").replace("]", ".").replace("<", ".").replace(">", ".")
                    .replace("{", ".").replace("}", "."))
    n = len(target)
    out = []
    for _ in range(R):
        s = ["A"] * n
        for i in range(1, n - 1):
            if pt[i] < 0 and pt[i - 1] >= 0 and pt[i + 1] < 0:
                s[i] = "G"
        for i in range(n):
            if pt[i] > i:
                s[i] = "CG"[int(rng.integers(2))]
                s[pt[i]] = _WC[s[i]]
        for _ in range(int(rng.integers(0, max_mutations + 1))):
            i = int(rng.integers(n))
            if pt[i] >= 0:
                a, b = _PAIRS[int(rng.integers(len(_PAIRS)))]
                lo, hi = min(i, pt[i]), max(i, pt[i])
                s[lo], s[hi] = a, b
            else:
                s[i] = "ACGU"[int(rng.integers(4))]
        out.append("".join(s))
    return out
