"""The candidate-list form of the MFE multiloop splits (tests/study_candidate_lists.py) is exact: on the
oracle's tables the recurrence DML(i,j) = min(DML(i,j-1) + MLbase, candidate split points) gives every cell's split minimum."""
import numpy as np
import pytest

from tests.study_candidate_lists import check


@pytest.mark.parametrize("n,alphabet", [(40, "ACGU"), (64, "ACGU"), (64, "GC"), (90, "ACGU")])
def test_candidate_recurrence_gives_every_split_minimum(oracle, n, alphabet):
    rng = np.random.default_rng(900 + n)
    seq = "".join(rng.choice(list(alphabet), n))
    assert check(oracle, seq) == 0


def test_candidate_recurrence_with_masked_positions(oracle):
    rng = np.random.default_rng(77)
    n = 72
    seq = "".join(rng.choice(list("ACGU"), n))
    nopair = np.zeros(n + 2, dtype=np.uint8)
    nopair[rng.choice(np.arange(1, n + 1), size=20, replace=False)] = 1       # a pseudoknot round's mask
    assert check(oracle, seq, nopair=nopair) == 0


def test_the_unpaired_end_term_is_needed(oracle):
    """without DML(i,j-1) + MLbase the candidates alone miss split minima (the closing of a multiloop reads them bare)"""
    rng = np.random.default_rng(5)
    missed = sum(check(oracle, "".join(rng.choice(list("ACGU"), 70)), with_prev=False) for _ in range(3))
    assert missed > 0
