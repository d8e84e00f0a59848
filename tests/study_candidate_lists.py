"""Study (CPU, uses the oracle's table dump; not a test): how many multiloop split points of the MFE fold are NOT dominated.

fML(i,j) = min( fML(i+1,j) + MLbase, fML(i,j-1) + MLbase, c(i,j) + stem, min_m fML(i,m-1) + fML(m,j) )   (oracle.c:554-569).
A split point m of cell (i,j) whose right part fML(m,j) is itself reached by one of the other three cases is dominated:
  fML(m,j) = fML(m,u-1) + fML(u,j)  =>  fML(i,m-1) + fML(m,j) >= fML(i,u-1) + fML(u,j)        (split point u)
  fML(m,j) = fML(m+1,j) + MLbase    =>  ... >= fML(i,m) + fML(m+1,j)                            (split point m+1)
  fML(m,j) = fML(m,j-1) + MLbase    =>  ... >= fML(i,j-1) + MLbase                              (another case of the cell)
so only the m with fML(m,j) strictly below those three (a single stem (m,j) is the only way to reach it) are needed: the
'candidates' of column j.  Values of the table do not change, so the traceback (which reads the tables) does not either.
Careful with the third case: the split minimum DML(i,j) on its own -- the multiloop CLOSING uses it without fML's other cases
(oracle.c:487, 674) -- needs the term the third case is dominated by,
  DML(i,j) = min( DML(i,j-1) + MLbase, min over the candidates m of column j: fML(i,m-1) + fML(m,j) ),
which check() below verifies cell by cell (tests/test_candidate_lists.py runs it).  Built and measured in the one-workgroup
LDS kernel in round 4, not kept there (profiles/r4/candidate_lists_cost.txt).
Prints the mean number of split points per cell and the mean number of candidates among them.

  python tests/study_candidate_lists.py [n ...]
"""
import csv
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from desirna_amd import params  # noqa: E402
from oracle import pyoracle  # noqa: E402

INF = 10000000


def candidates(f, n, mlbase=0):
    """cand[m, j]: fML(m, j) is strictly below its cell's three other cases"""
    cand = np.zeros((n + 2, n + 2), dtype=bool)
    for j in range(1, n + 1):
        for m in range(1, j):
            v = f[m, j]
            if v >= INF:
                continue
            other = min(f[m + 1, j] + mlbase, f[m, j - 1] + mlbase)
            if j - m >= 2:
                u = np.arange(m + 1, j + 1)
                other = min(other, int((f[m, m:j] + f[u, j]).min()))
            cand[m, j] = v < other
    return cand


def check(orc, seq, mlbase=0, nopair=None, with_prev=True):
    """cells whose split minimum differs between the full range and the candidate recurrence (0 = exact)"""
    n = len(seq)
    _, f, _ = orc.mfe_tables(seq, nopair)
    f = np.minimum(f.astype(np.int64), INF)
    cand = candidates(f, n, mlbase)
    dml = np.full((n + 2, n + 2), INF, dtype=np.int64)       # by the candidate recurrence
    bad = 0
    for d in range(9, n):
        for i in range(1, n - d + 1):
            j = i + d
            ms = np.arange(i + 5, j - 3)                      # fML(i, m-1) + fML(m, j), both parts at least TURN + 2 long
            full = int(min(INF, (f[i, ms - 1] + f[ms, j]).min())) if ms.size else INF
            mc = ms[cand[ms, j]] if ms.size else ms
            v = int(min(INF, (f[i, mc - 1] + f[mc, j]).min())) if mc.size else INF
            if with_prev and dml[i, j - 1] < INF:
                v = min(v, int(dml[i, j - 1]) + mlbase)
            dml[i, j] = v
            bad += v != full
    return bad


def study(orc, seq, mlbase=0):
    n = len(seq)
    c, f, _ = orc.mfe_tables(seq)
    f = f.astype(np.int64)
    cand = np.zeros((n + 2, n + 2), dtype=bool)
    for j in range(1, n + 1):
        for m in range(1, j):
            v = f[m, j]
            if v >= INF:
                continue
            other = min(f[m + 1, j] + mlbase, f[m, j - 1] + mlbase)
            if j - m >= 2:
                u = np.arange(m + 1, j + 1)                       # split: fML(m,u-1) + fML(u,j)
                other = min(other, int((f[m, m:j] + f[u, j]).min()))
            cand[m, j] = v < other
    splits = cands = cells = 0
    for i in range(1, n + 1):
        for j in range(i + 1, n + 1):
            cells += 1
            splits += j - i
            cands += int(cand[i + 1:j + 1, j].sum())
    per_col = cand.sum(axis=0)[1:n + 1]
    return splits / cells, cands / cells, per_col.mean(), per_col.max()


def main():
    pyoracle.build()
    orc = pyoracle.Oracle(params.load_blob())
    rng = np.random.default_rng(20260101)
    ns = [int(x) for x in sys.argv[1:]] or [200, 400]
    for n in ns:
        seq = "".join(rng.choice(list("ACGU"), n))
        print("uniform %4d nt: split points per cell %.1f, candidates among them %.2f; candidates per column mean %.1f max %d"
              % ((n,) + study(orc, seq)), flush=True)
    rows = list(csv.DictReader(open(os.path.join(ROOT, "tests", "golden", "eterna_v1_solutions.csv"))))
    for r in sorted(rows, key=lambda r: -len(r["sequence"]))[:2] + [r for r in rows if 190 <= len(r["sequence"]) <= 200][:1]:
        s = r["sequence"]
        print("designed %4d nt: split points per cell %.1f, candidates among them %.2f; candidates per column mean %.1f max %d"
              % ((len(s),) + study(orc, s)), flush=True)


if __name__ == "__main__":
    main()
