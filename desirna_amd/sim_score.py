"""Structure-similarity metrics of the scoring path (host side, exact integer/fp arithmetic).

Counterpart of the reference's ``utils/sim_score.py`` (``pairing_positions`` :28-59, ``SimScore``
:62-147).  Same interface and bit-identical results, different algorithm: the
reference matches brackets with an O(#open x #close) double loop in pure Python (0.8 ms at L=200);
here every bracket family is matched with a stack in one pass and the confusion matrix is counted
with numpy, for whole batches at a time.

Semantics kept from the reference:
  * nine bracket families ``() [] <> {} Aa Bb Cc Dd Ee``; ``.`` and ``-`` are unpaired;
  * the confusion matrix is per POSITION: a correctly paired base is one TP (so a correct pair
    counts twice), a base unpaired in both is a TN, a base paired in the query but unpaired in the
    reference is a FP, every other disagreement is a FN;
  * ``mcc = round(num / (den + 1e-5), 3)`` with the all-unpaired special case -> 1,
    ``recall = round(tp / (tp + fn + 0.001), 3)``, ``precision = round(tp / (tp + fp + 0.001), 3)``.
"""
import math

import numpy as np

_OPEN = {"(": 0, "[": 1, "<": 2, "{": 3, "A": 4, "B": 5, "C": 6, "D": 7, "E": 8}
_CLOSE = {")": 0, "]": 1, ">": 2, "}": 3, "a": 4, "b": 5, "c": 6, "d": 7, "e": 8}


def pair_table(ss):
    """numpy int32 array: partner index (0-based) or -1 for unpaired positions.

    For balanced input this equals the reference's ``pairing_positions`` (an open bracket pairs
    with the nearest still-free close bracket of its family to its right, opens taken right to
    left), which is ordinary stack matching per family."""
    n = len(ss)
    pt = np.full(n, -1, dtype=np.int32)
    stacks = [[] for _ in range(9)]
    for i, ch in enumerate(ss):
        f = _OPEN.get(ch)
        if f is not None:
            stacks[f].append(i)
            continue
        f = _CLOSE.get(ch)
        if f is not None:
            if not stacks[f]:
                raise ValueError("unbalanced structure: %r" % ss)
            o = stacks[f].pop()
            pt[o] = i
            pt[i] = o
        elif ch not in ".-&":          # '&': strand separator of the design drivers (an unpaired position)
            raise ValueError("unexpected character %r in structure" % ch)
    if any(stacks):
        raise ValueError("unbalanced structure: %r" % ss)
    return pt


def pairing_positions(s1):
    """dict position -> partner (or -1), the reference's return type."""
    return {i: int(p) for i, p in enumerate(pair_table(s1))}


def confusion(pt_ref, pt_query):
    """(tp, fp, fn, tn) for one reference table against one or many query tables (last axis = position)."""
    r = np.asarray(pt_ref)
    q = np.asarray(pt_query)
    same = q == r
    tp = np.count_nonzero(same & (r != -1), axis=-1)
    tn = np.count_nonzero(same & (r == -1), axis=-1)
    fp = np.count_nonzero(~same & (r == -1), axis=-1)
    fn = np.count_nonzero(~same & (r != -1), axis=-1)
    return tp, fp, fn, tn


def mcc_from(tp, fp, fn, tn):
    tp, fp, fn, tn = int(tp), int(fp), int(fn), int(tn)
    if tp == 0 and fp == 0 and fn == 0 and tn != 0:
        num, den = 1, 1
    else:
        num = tp * tn - fp * fn
        den = math.sqrt((tp + fp) * (tp + fn) * (tn + fn) * (tn + fp))
    return round(num / (den + 0.00001), 3)


def recall_from(tp, fp, fn, tn):
    return round(int(tp) / (int(tp) + int(fn) + 0.001), 3)


def precision_from(tp, fp, fn, tn):
    return round(int(tp) / (int(tp) + int(fp) + 0.001), 3)


class SimScore:
    """Same call sequence as the reference: SimScore(ref, query); find_basepairs(); cofusion_matrix(); mcc()..."""

    def __init__(self, ref_ss, query_ss):
        self.ref_ss = ref_ss
        self.query_ss = query_ss

    def find_basepairs(self):
        self._pt_r = pair_table(self.ref_ss)
        self._pt_q = pair_table(self.query_ss)
        self.bp_dict_r = {i: int(p) for i, p in enumerate(self._pt_r)}
        self.bp_dict_q = {i: int(p) for i, p in enumerate(self._pt_q)}

    def cofusion_matrix(self):
        tp, fp, fn, tn = confusion(self._pt_r, self._pt_q)
        self.conf_mat = (int(tp), int(fp), int(fn), int(tn))

    def mcc(self):
        return mcc_from(*self.conf_mat)

    def recall(self):
        return recall_from(*self.conf_mat)

    def precision(self):
        return precision_from(*self.conf_mat)

    def fscore(self):
        den = self.precision() + self.recall()
        if den < 0.001:
            den = 0.001
        return round(2 * (self.precision() * self.recall() / den), 4)

    def mcc_reverse(self):
        return -self.mcc()

    def recall_reverse(self):
        return -self.recall()

    def precision_reverse(self):
        return -self.precision()

    def fscore_reverse(self):
        return -self.fscore()


def batch_metrics(ref_ss, query_list):
    """(mcc, recall, precision) rounded like the reference, for many query structures against one reference."""
    r = pair_table(ref_ss)
    q = np.stack([pair_table(s) for s in query_list]) if query_list else np.zeros((0, len(r)), dtype=np.int32)
    tp, fp, fn, tn = confusion(r, q)
    return [(mcc_from(a, b, c, d), recall_from(a, b, c, d), precision_from(a, b, c, d))
            for a, b, c, d in zip(tp, fp, fn, tn)]
