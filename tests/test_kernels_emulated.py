"""CPU-only logic check of the UNMODIFIED HIP kernel sources: tests/emu/ compiles desirna_amd/csrc/*.hpp
against a stand-in for the few HIP constructs they use (threads = OS threads, real barriers) and the
results are compared with the oracle.  This is not the product path and proves nothing about the GPU
build -- the -m gpu tests do that through the C ABI -- it only lets kernel logic be debugged here."""
import numpy as np
import pytest

from tests.emu.emu import Emu


@pytest.fixture(scope="module")
def emu(blob):
    return Emu(blob)


def _rand(rng, L, alphabet="ACGU"):
    return "".join(rng.choice(list(alphabet), L))


@pytest.mark.parametrize("L,nt", [(5, 64), (9, 64), (36, 128), (70, 128), (97, 256)])
def test_mfe_pf_random(emu, oracle, L, nt):
    rng = np.random.default_rng(100 + L)
    seqs = [_rand(rng, L) for _ in range(3)] + [_rand(rng, L, "GC"), _rand(rng, L, "GGCCAU")]
    E, ss, st = emu.mfe(seqs, nt=nt)
    Ep, stp = emu.pf(seqs, nt=nt)
    assert not st.any() and not stp.any()
    for k, s in enumerate(seqs):
        oss, oe = oracle.mfe(s)
        assert (ss[k], int(E[k])) == (oss, oe), s
        assert abs(Ep[k] - oracle.pf(s)) < 1e-9, s


def test_mfe_tables_match_oracle(emu, oracle):
    rng = np.random.default_rng(7)
    s = _rand(rng, 60)
    E, ss, st, Wc, F = emu.mfe([s], nt=128, dump=True)
    c, f, f5 = oracle.mfe_tables(s)
    n, INF_DEV = len(s), 1 << 22
    for d in range(4, n):
        for i in range(1, n - d + 1):
            cc = Wc[d, i] >> 8
            assert (c[i, i + d] if c[i, i + d] < 10000000 else INF_DEV) == cc
            assert (f[i, i + d] if f[i, i + d] < 10000000 else INF_DEV) == F[d, i]


def test_pk_rounds(emu, oracle, traj_golden):
    rows = [r for r in traj_golden if r["run"] == "Pseudoknot_design_input" and "[" in r["mfe_ss"]][:6]
    seqs = [r["sequence"] for r in rows]
    E, ss, st = emu.mfe(seqs, pk_rounds=3, nt=128)
    assert not st.any()
    for r, got in zip(rows, ss):
        assert got == r["mfe_ss"]


def test_eval_kernel(emu, oracle, traj_golden, example_inputs):
    rows = [r for r in traj_golden if r["run"] == "Alternative_structures_design_input"][:12]
    inp = example_inputs["Alternative_structures_design_input"]
    targets = [inp["sec_struct"][0]] + inp["alt_sec_struct"]
    seqs = [r["sequence"] for r in rows]
    Ed = emu.eval(seqs, targets)
    for k, s in enumerate(seqs):
        for t, tg in enumerate(targets):
            assert Ed[k, t] == oracle.eval_structure(s, tg)
    # long loops (> MAXLOOP) and a multiloop
    s = "G" + "A" * 40 + "GGGAAACCC" + "A" * 35 + "GGGAAACCC" + "A" * 3 + "C"
    tg = "(" + "." * 40 + "(((...)))" + "." * 35 + "(((...)))" + "." * 3 + ")"
    s2 = "GGG" + "A" * 45 + "CCC" + "AAAA"
    tg2 = "(((" + "." * 45 + ")))" + "...."
    assert emu.eval([s], [tg])[0, 0] == oracle.eval_structure(s, tg)
    assert emu.eval([s2], [tg2])[0, 0] == oracle.eval_structure(s2, tg2)


def test_bad_character_flag(emu):
    E, ss, st = emu.mfe(["GGGAAANCCC"], nt=64)
    assert st[0] == 1


# ---- LDS-resident production kernels (fold_mfe_lds.hpp / fold_pf_lds.hpp); nt < 0 selects them in the emulator.
# 256 threads (4 waves: 1 finalize + 3 sweep) cover n <= 64; 1024 threads is the shipped geometry.

@pytest.mark.parametrize("L,nt", [(5, -256), (9, -256), (36, -256), (70, -1024)])
def test_lds_kernels_random(emu, oracle, L, nt):
    rng = np.random.default_rng(300 + L)
    seqs = [_rand(rng, L), _rand(rng, L, "GC")]
    if nt == -1024:
        seqs = seqs[:1]                  # 1024 OS threads per workgroup: one sequence keeps the CPU suite short
    E, ss, st = emu.mfe(seqs, nt=nt)
    Ep, stp = emu.pf(seqs, nt=nt)
    assert not st.any() and not stp.any()
    for k, s in enumerate(seqs):
        assert (ss[k], int(E[k])) == oracle.mfe(s), s
        assert abs(Ep[k] - oracle.pf(s)) < 1e-9, s


def test_lds_kernel_pk_rounds(emu, oracle, traj_golden):
    rows = [r for r in traj_golden if r["run"] == "Pseudoknot_design_input" and "[" in r["mfe_ss"]][:3]
    E, ss, st = emu.mfe([r["sequence"] for r in rows], pk_rounds=3, nt=-256)
    assert not st.any()
    for r, got in zip(rows, ss):
        assert got == r["mfe_ss"]


# ---- outside recursion / ensemble defect (fold_outside.hpp after the general pf_kernel)

@pytest.mark.parametrize("L,nt", [(5, 64), (12, 64), (36, 128), (70, 128), (97, 256)])
def test_outside_kernel_bpp_and_defect(emu, oracle, L, nt):
    rng = np.random.default_rng(500 + L)
    seqs = [_rand(rng, L), _rand(rng, L), _rand(rng, L, "GC")]
    target = oracle.mfe(seqs[0])[0]
    ed, st, B = emu.edef(seqs, target, nt=nt, bpp=True)
    assert not st.any()
    for k, s in enumerate(seqs):
        oe, ob = oracle.ensemble_defect(s, target, want_bpp=True)
        assert abs(ed[k] - oe) < 1e-12, s
        assert np.abs(B[k] - ob).max() < 1e-12, s


# ---- ragged batches: per-workgroup length / offsets (drna_score_ragged)

@pytest.mark.parametrize("lds", [False, True])
def test_ragged_lengths_in_one_launch(emu, oracle, lds):
    rng = np.random.default_rng(77)
    seqs = [_rand(rng, L) for L in ((40, 1, 7, 23, 64, 5) if not lds else (33, 1, 7, 21))]
    E, ss, Ep, st = emu.ragged(seqs, lds=lds)
    assert not st.any()
    for k, s in enumerate(seqs):
        assert (ss[k], int(E[k])) == oracle.mfe(s), s
        assert abs(Ep[k] - oracle.pf(s)) < 1e-9, s


# ---- two strands: co-fold MFE / PF kernels and the eval kernel with the nick (fold_cofold.hpp)

def test_cofold_kernels_goldens_and_random(emu, oracle, traj_golden, example_inputs):
    for run in ("RNA_RNA_complex_design_input", "Homodimer_design_input"):
        rows = [r for r in traj_golden if r["run"] == run][:2]
        tg = example_inputs[run]["sec_struct"][0]
        E, ss, F4, st, Ed = emu.cofold([r["sequence"] for r in rows], tg, nt=128)
        assert not st.any()
        for k, r in enumerate(rows):
            assert ss[k] == r["mfe_ss"]
            assert abs(F4[k, 3] - float(r["Epf"])) < 2e-6
            assert Ed[k] == round(float(r["edesired"]) * 100)
    rng = np.random.default_rng(11)
    pairs = [_rand(rng, a) + "&" + _rand(rng, b) for a, b in ((1, 1), (9, 6))]
    for s in pairs:
        E, ss, F4, st, _ = emu.cofold([s], None, nt=64)
        assert not st.any()
        oss, oe = oracle.cofold_mfe(s)
        assert (ss[0], int(E[0])) == (oss, oe), s
        assert np.abs(F4[0] - np.array(oracle.cofold_pf(s))).max() < 1e-9, s


# ---- second-best structure energy (fold_subopt.hpp), SURVEY 8(f)-4

def test_subopt_kernel_two_best(emu, oracle):
    rng = np.random.default_rng(21)
    for L, nt in ((6, 64), (14, 64), (27, 128)):
        seqs = [_rand(rng, L), _rand(rng, L, "GC")] + (["A" * L] if L == 6 else [])
        E2, E12, st = emu.subopt(seqs, nt=nt)
        assert not st.any()
        for k, s in enumerate(seqs):
            assert tuple(int(x) for x in E12[k]) == oracle.two_best(s), s
            assert int(E2[k]) == oracle.subopt_energy(s), s
            assert int(E12[k, 0]) == oracle.mfe(s)[1]


# ---- K lowest-energy structures with their strings (kbest_kernel), the call behind get_alt_mcc: checked against
# explicit enumeration of every structure of short sequences (the reference pins nothing here)

def _all_structures(seq):
    n = len(seq)
    pairs = {("A", "U"), ("U", "A"), ("G", "C"), ("C", "G"), ("G", "U"), ("U", "G")}
    out = []

    def rec(i, cur, stack):
        if i == n:
            if not stack:
                out.append("".join(cur))
            return
        cur.append("."); rec(i + 1, cur, stack); cur.pop()
        if n - i - 1 >= len(stack) + 1:
            cur.append("("); stack.append(i); rec(i + 1, cur, stack); stack.pop(); cur.pop()
        if stack and i - stack[-1] > 3 and (seq[stack[-1]], seq[i]) in pairs:
            o = stack.pop(); cur.append(")"); rec(i + 1, cur, stack); cur.pop(); stack.append(o)

    rec(0, [], [])
    return out


def test_kbest_structures_against_enumeration(emu, oracle):
    rng = np.random.default_rng(77)
    cases = [(_rand(rng, 13), 4, 64), (_rand(rng, 15, "GC"), 4, 128), (_rand(rng, 14, "GCAU"), 8, 64), ("GGGAAACCCA", 4, 64),
             ("AAAAAAAA", 4, 64)]
    for s, K, nt in cases:
        E, ss, st = emu.kbest([s], K, nt=nt)
        assert not st.any()
        en = sorted(oracle.eval_structure(s, x) for x in _all_structures(s))
        want = (en + [10000000] * K)[:K]
        assert [int(x) for x in E[0]] == want, s
        got = [x for x, e in zip(ss[0], E[0]) if e < 10000000]
        assert len(set(got)) == len(got), (s, got)                      # K different structures
        for x, e in zip(ss[0], E[0]):
            if e < 10000000:
                assert oracle.eval_structure(s, x) == int(e), (s, x)    # each string has the energy reported for it
            else:
                assert x == "." * len(s)


@pytest.mark.parametrize("L,nt,pk", [(24, 256, 0), (41, 256, 3), (44, 1024, 0)])       # 1024 threads: the production roles (tower waves by size and diagonal parity, list staging by sweep waves, helper-built list rows)
def test_two_workgroup_mfe_kernel(emu, oracle, L, nt, pk):
    """fold_mfe_dual.hpp on the CPU: the main and the helper workgroup of every sequence run side by side (OS threads), rows
    and flags go through ordinary memory; two calls in a row exercise the epoch arithmetic of the never-reset flags.
    Energies and (pk-annotated) structures must equal the oracle's, a bad character must end both roles."""
    rng = np.random.default_rng(900 + L)
    seqs = ["".join(rng.choice(list("ACGU"), L))] + ["".join(rng.choice(list("GC"), L))]
    E, ss, st = emu.mfe_dual(seqs, pk_rounds=pk, nt=nt, calls=2 if L == 24 else 1)
    assert (st == 0).all()
    for k, s in enumerate(seqs):
        ref, e = oracle.mfe(s)
        if pk:
            ref = oracle.pk_struct(s, ref)
        assert (ss[k], int(E[k])) == (ref, e), s
    if L == 24:
        _, _, st = emu.mfe_dual(["ACGUNACGUACGUACGUACGUACG"], nt=nt)
        assert st[0] == 1


def test_pf_strip_kernel(emu, oracle):
    """fold_pf_strip.hpp on the CPU: the three strips of a sequence run side by side (OS threads), records and flags go
    through ordinary memory; the second call exercises the epoch arithmetic of the never-reset flags.  100 nt at 256 threads:
    strips of 33 / 34 columns (a halo reaches 31), towers that walk through all three strips, multiloop sums dealt 1 / 2 / 4 ways."""
    rng = np.random.default_rng(4110)
    seqs = [_rand(rng, 100, "GGCCAU")]
    Ep, st = emu.pf_strip(seqs, 3, nt=256, calls=2)
    assert (st == 0).all()
    for k, s in enumerate(seqs):
        assert abs(Ep[k] - oracle.pf(s)) < 1e-9, s


@pytest.mark.parametrize("flags,tag", [((), ""), (("-DDRNA_PKT_W=6", "-DDRNA_PKT_L=3"), "_pktw6")])
def test_pf_strip_kernel_blocked_sums(blob, oracle, flags, tag):
    """The blocked multiloop sums of fold_pf_strip.hpp (16 x 16 tiles of cells; the far split points as v_mfma_f64_16x16x4_f64
    products spread over the PKT_W steps before the tile is due, here in the emulation's restatement of the instruction's
    operand layout; near split points masked per cell in the per-diagonal items).  150 nt in three strips of 50 columns: tiles
    of block distance 5 .. 9 have far ranges of up to 90 split points; a second window length moves every boundary."""
    e = Emu(blob, flags=flags, tag=tag) if flags else Emu(blob)
    rng = np.random.default_rng(4113)
    seqs = [_rand(rng, 150, "GGCCAU" if flags else "ACGU")]
    Ep, st = e.pf_strip(seqs, 3, nt=256, calls=1)
    assert (st == 0).all()
    for k, s in enumerate(seqs):
        assert abs(Ep[k] - oracle.pf(s)) < 1e-9, s


def test_mfe_strip_kernel(emu, oracle):
    """fold_mfe_strip.hpp on the CPU: per pseudoknot round the two strips of a sequence side by side, then the traceback
    "launch" on the tables they left behind; energies and pk-annotated structures must equal the oracle's; a bad character is
    reported by the traceback kernel."""
    rng = np.random.default_rng(4111)
    seqs = [_rand(rng, 84, "GGCCAU")]
    E, ss, st = emu.mfe_strip(seqs, 2, pk_rounds=3, nt=256, calls=1)
    assert (st == 0).all()
    for k, s in enumerate(seqs):
        ref, e = oracle.mfe(s)
        assert (ss[k], int(E[k])) == (oracle.pk_struct(s, ref), e), s
    _, ss, st = emu.mfe_strip(["ACGUN" * 20], 2, nt=256)
    assert st[0] == 1 and ss[0] == "." * 100


def test_mfe_strip_kernel_blocked_splits(emu, oracle):
    """The blocked form of the multiloop splits in fold_mfe_strip.hpp (StripLink::fark; the engine switches it on for long folds):
    16 x 16 tiles of cells, the far split points as (min,+) tile products by DPP row broadcasts and lane fetches spread over the
    16 steps before the tile is due, near split points masked per cell.  150 nt in three strips of 50 columns, with a pseudoknot
    round: energies and structures must equal the oracle's."""
    rng = np.random.default_rng(4112)
    seqs = [_rand(rng, 150, "GGCCAU")]
    E, ss, st = emu.mfe_strip(seqs, 3, pk_rounds=1, nt=256, calls=1, fark=1)
    assert (st == 0).all()
    for k, s in enumerate(seqs):
        ref, e = oracle.mfe(s)
        assert (ss[k], int(E[k])) == (oracle.pk_struct(s, ref), e), s


@pytest.mark.parametrize("L,nt", [(30, -257), (64, -257), (96, -1025)])
def test_pf_helper_workgroup_is_bit_identical(emu, oracle, L, nt):
    """fold_pf_lds.hpp with a helper workgroup per sequence (far multiloop split points, nt = -257 / -1025: main and helper side
    by side as OS threads, each with an LDS image of its own, rows and flags through ordinary memory): Epf must equal the
    one-workgroup kernel's BIT FOR BIT (one canonical summation order whoever computes the far part) and the oracle's to 1e-9."""
    rng = np.random.default_rng(700 + L)
    seqs = [_rand(rng, L)] + ([_rand(rng, L, "GC")] if nt == -257 else [])
    Eh, sth = emu.pf(seqs, nt=nt)
    E1, st1 = emu.pf(seqs, nt=nt + 1)
    assert not sth.any() and not st1.any()
    assert (Eh.view(np.int64) == E1.view(np.int64)).all()
    for k, s in enumerate(seqs):
        assert abs(Eh[k] - oracle.pf(s)) < 1e-9, s
