"""Geometry of the partition function's 4 x 4 tile products (desirna_amd/csrc/fold_pf_lds.hpp), restated in Python so that
a CPU test can check what the kernel relies on: every split point of every cell is counted exactly once (tile product or near slot),
a tile's operands are final when its first step comes, the near slots fit their rows, and the helper's flag means what the main
workgroup reads into it.  check(n, pke) returns the number of violations."""
TURN = 3


def consts(pke):
    nl = pke + 3
    return dict(KT_NL=nl, KT_U=(2 * nl + 3) // 4, KT_BMIN=(12 + 2 * pke + 3) // 4, KT_D0=4 * ((12 + 2 * pke + 3) // 4) - 3)


def near_counts(i, d, pke):
    """kt_near_counts: split points of cell (i, i+d) taken per cell, from below and from above"""
    c = consts(pke)
    j = i + d
    a, cc, tot = (i - 1) >> 2, (j - 1) >> 2, max(d - 2 * TURN - 2, 0)
    if cc - a >= c["KT_BMIN"]:
        return 4 * a + 4 + pke - i, j - 4 * cc - 1 + pke
    lo = min(tot, c["KT_NL"])
    return lo, tot - lo


def far_range(a, B, pke):
    """k_tile_issue: split points m of tile (a, a + B) that go through the matrix instruction"""
    return 4 * a + 9 + pke, 4 * (a + B) - 3 - pke


def check(n, pke):
    c = consts(pke)
    bad = 0
    bmax = (n - 1) >> 2
    for d in range(TURN + 1, n):
        for i in range(1, n - d + 1):
            j = i + d
            want = set(range(i + TURN + 2, j - TURN))            # m = i+TURN+2 .. j-TURN-1
            a, cc = (i - 1) >> 2, (j - 1) >> 2
            nl, nh = near_counts(i, d, pke)
            low = [i + TURN + 2 + x for x in range(nl)]
            high = [j - TURN - 1 - y for y in range(nh)]
            far = []
            if cc - a >= c["KT_BMIN"]:
                lo, hi = far_range(a, cc - a, pke)
                far = list(range(lo, hi + 1))
                # both operands of every far term final before the tile's first step (4B - 3 - 4), i.e. on diagonals <= 4B - 5 - pke
                for m in far:
                    if m - 1 - i > 4 * (cc - a) - 5 - pke or j - m > 4 * (cc - a) - 5 - pke:
                        bad += 1
            got = low + high + far
            if len(got) != len(set(got)) or set(got) != want:
                bad += 1
            if nl < 0 or nh < 0 or nl > c["KT_NL"] or nh > c["KT_NL"] or nl + nh > 4 * c["KT_U"]:
                bad += 1
            # the finalize waves ask for a far sum exactly where a tile stores one, never before diagonal KT_D0
            if (cc - a >= c["KT_BMIN"]) != bool(far) or (far and d < c["KT_D0"]):
                bad += 1
    # the helper's flag after the round of block distance B says "far sums complete below diagonal 4B + 1": every cell of a
    # diagonal <= 4B belongs to a block distance <= B; the main workgroup's own tiles of block distance B are multiplied in steps
    # 4B-7 .. 4B-4, reading rows <= step - 2
    for B in range(c["KT_BMIN"], bmax + 1):
        for d in range(TURN + 1, min(4 * B, n - 1) + 1):
            for i in range(1, n - d + 1):
                if ((i + d - 1) >> 2) - ((i - 1) >> 2) > B:
                    bad += 1
        if 4 * B - 5 - pke > (4 * B - 7) - 2:
            bad += 1
    return bad


if __name__ == "__main__":
    import sys
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    for pke in (5, 6, 7):
        print("n", n, "PKE", pke, consts(pke), "violations", check(n, pke))
