"""The schedule of the tile products in fold_pf_strip.hpp (blocked multiloop sums), restated: for every block distance B the
chunks of the far range are dealt to the PKT_W steps before the tile's first cell is due, from the middle outward.  check()
verifies that every chunk is taken exactly once and only at a step where both its operands are final AND visible (diagonals
<= step - 2).  Used by tests/test_host_cpu.py; run it to print the table."""


def plan(B, W=16, L=4, t=0):
    """[(step, [chunk ids])] of tile (t, t + B); None if the tile has no far range"""
    bj = t + B
    m_lo, m_hi = 16 * t + 31 + L, 16 * bj - 13 - L
    if m_hi < m_lo:
        return None
    nch = (m_hi - m_lo + 4) >> 2
    nl, nh = (nch + 1) >> 1, nch >> 1
    cl, ch = (nl + W - 1) // W, (nh + W - 1) // W
    d_min = 16 * B - 15
    out = []
    for g in range(W):
        e = W - 1 - g
        cs = list(range(e * cl, min(nl, (e + 1) * cl))) + [nch - 1 - q for q in range(e * ch, min(nh, (e + 1) * ch))]
        out.append((d_min - W + g, cs))
    return out


def check(W=16, L=4, bmax=140):
    """number of (tile, chunk) pairs taken too early; asserts that every chunk is taken exactly once"""
    bad = 0
    for B in range(1, bmax):
        for t in (0, 3):
            p = plan(B, W, L, t)
            if p is None:
                continue
            bj = t + B
            m_lo, m_hi = 16 * t + 31 + L, 16 * bj - 13 - L
            nch = (m_hi - m_lo + 4) >> 2
            seen = []
            for k, cs in p:
                for c in cs:
                    m_first = m_lo + 4 * c
                    m_last = min(m_first + 3, m_hi)
                    a_diag = (m_last - 1) - (16 * t + 1)          # QM(i_min, m_last - 1)
                    b_diag = (16 * bj + 16) - m_first             # QM1(m_first, j_max)
                    bad += max(a_diag, b_diag) + 2 > k
                    seen.append(c)
            assert sorted(seen) == list(range(nch)), (B, t)
    return bad


if __name__ == "__main__":
    for W in (16, 8):
        for L in (2, 3, 4):
            print("PKT_W %2d PKT_L %d: chunks taken too early: %d" % (W, L, check(W, L)))
