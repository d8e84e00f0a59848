"""GPU parity tests proper: HIP path through the C ABI (ctypes -> libdesirna_amd.so) against the CPU
oracle and the committed golden vectors.  Integer quantities (MFE energy, E(target)) and bracket
strings must be bit-exact; Epf must agree with the oracle to EPF_TOL_ORACLE and with the reference's
float32 goldens to EPF_TOL_GOLDEN (north_star: 'stated fp tolerance')."""
import numpy as np
import pytest

INJECTED_FALLBACKS = []     # fault-injected sync fallbacks (test_lost_strip_falls_back_to_one_workgroup_per_fold)

pytestmark = pytest.mark.gpu

EPF_TOL_ORACLE = 1e-9   # kcal/mol, fp64 summation-order differences only
EPF_TOL_GOLDEN = 2e-6   # kcal/mol, goldens are float32
TIE_OUTLIER = "GGUACAGCCGUGCCCCUUAGGGCACCGUGGUGUACC"
SINGLE = ("Standard_design_input", "Seed_sequence_design_input",
          "Alternative_structures_design_input", "Pseudoknot_design_input")


@pytest.fixture(scope="module")
def eng400():
    from desirna_amd import engine
    e = engine.Engine(max_R=128, max_L=400, device=0)
    yield e
    e.close()


def _rand(rng, L, alphabet="ACGU"):
    return "".join(rng.choice(list(alphabet), L))


def _check_against_oracle(eng, oracle, seqs, targets, pk=False):
    from desirna_amd import engine as E
    eng.set_targets(targets)
    flags = E.NEED_PF | E.NEED_MFE | E.NEED_EVAL | (E.NEED_PK if pk else 0)
    out = eng.score_batch(seqs, flags)
    for k, s in enumerate(seqs):
        ss, e = oracle.mfe(s)
        if pk:
            ss = oracle.pk_struct(s, ss)
        assert out["mfe_ss"][k] == ss, (s, out["mfe_ss"][k], ss)
        assert int(out["Emfe"][k]) == e, s
        assert abs(float(out["Epf"][k]) - oracle.pf(s)) < EPF_TOL_ORACLE, s
        for t, tg in enumerate(targets):
            assert int(out["Ed"][k, t]) == oracle.eval_structure(s, tg), (s, tg)
    return out


def test_library_loaded_is_in_tree():
    from desirna_amd import engine
    import os
    assert os.path.exists(engine.LIB_PATH)
    engine.load_library()


def test_golden_trajectories(eng400, traj_golden, example_inputs):
    from desirna_amd import engine as E
    for run in SINGLE:
        rows = [r for r in traj_golden if r["run"] == run]
        inp = example_inputs[run]
        targets = [inp["sec_struct"][0]] + inp.get("alt_sec_struct", [])
        eng400.set_targets(targets)
        pk = bool(int(rows[0]["pk_on"]))
        flags = E.NEED_PF | E.NEED_MFE | E.NEED_EVAL | (E.NEED_PK if pk else 0)
        for b in range(0, len(rows), 128):
            chunk = rows[b:b + 128]
            out = eng400.score_batch([r["sequence"] for r in chunk], flags)
            for k, r in enumerate(chunk):
                assert abs(float(out["Epf"][k]) - float(r["Epf"])) < EPF_TOL_GOLDEN, r["sequence"]
                assert int(out["Ed"][k, 0]) == round(float(r["edesired"]) * 100), r["sequence"]
                if r["sequence"] != TIE_OUTLIER:
                    assert out["mfe_ss"][k] == r["mfe_ss"], r["sequence"]
                if len(targets) > 1:
                    alt = [int(x) / 100.0 for x in out["Ed"][k, 1:]]
                    assert abs(sum(alt) / len(alt) - float(r["edesired2"])) < 1e-5


def test_eterna_v1_solutions(eng400, oracle, eterna_solutions):
    from desirna_amd import engine as E
    by_len = {}
    for r in eterna_solutions:
        by_len.setdefault(len(r["sequence"]), []).append(r)
    for L, rows in by_len.items():
        for r in rows:
            eng400.set_targets([r["structure"]])
            out = eng400.score_batch([r["sequence"]], E.NEED_PF | E.NEED_MFE | E.NEED_EVAL)
            assert out["mfe_ss"][0] == r["structure"], r["name"]
            assert int(out["Emfe"][0]) == int(out["Ed"][0, 0]), r["name"]
            assert abs(float(out["Epf"][0]) - oracle.pf(r["sequence"])) < EPF_TOL_ORACLE, r["name"]     # all 100, up to 400 nt


@pytest.mark.parametrize("L,R", [(1, 3), (4, 2), (5, 4), (8, 4), (9, 4), (10, 4), (11, 4), (12, 4), (13, 4), (63, 8), (64, 8), (65, 8), (74, 8), (75, 8),
                                 (100, 16), (129, 8), (138, 8), (139, 8)])      # (10 / 11, 74 / 75, 138 / 139: one tower block more)
def test_random_vs_oracle(eng400, oracle, L, R):
    rng = np.random.default_rng(1000 + L)
    seqs = [_rand(rng, L) for _ in range(R - 1)] + [_rand(rng, L, "GC")]
    _check_against_oracle(eng400, oracle, seqs, ["." * L])


def test_config3_full_batch_vs_oracle(eng400, oracle, eterna_targets):
    """BASELINE config 3: L=200 (Eterna V1 #69 target), R=64, uniform-random sequences, MFE+PF+eval."""
    tg = eterna_targets["eteV1_69.txt"]
    rng = np.random.default_rng(20260101)
    seqs = [_rand(rng, len(tg)) for _ in range(64)]
    _check_against_oracle(eng400, oracle, seqs, [tg])


def test_config2_L100(eng400, oracle, eterna_targets):
    tg = eterna_targets["eteV1_92.txt"]
    rng = np.random.default_rng(2)
    seqs = [_rand(rng, len(tg)) for _ in range(64)]
    _check_against_oracle(eng400, oracle, seqs, [tg])


def test_config5_L400_pk_alt(eng400, oracle, eterna_targets):
    """L=400 with pk heuristic and alternative targets (two extra eval structures)."""
    tg = eterna_targets["eteV1_53.txt"]
    L = len(tg)
    rng = np.random.default_rng(5)
    seqs = [_rand(rng, L) for _ in range(6)] + [_rand(rng, L, "GGCCAU") for _ in range(2)]
    alts = [eterna_targets["eteV1_22.txt"], eterna_targets["eteV1_63.txt"]]
    assert all(len(a) == L for a in alts)
    _check_against_oracle(eng400, oracle, seqs, [tg] + alts, pk=True)


def test_properties_at_full_size(eng400, eterna_targets):
    """Size-independent properties at R=128, L=200: E(MFE structure) == MFE energy, Epf <= MFE,
    bit-identical results across two calls, and batch-composition independence."""
    from desirna_amd import engine as E
    L = 200
    rng = np.random.default_rng(77)
    seqs = [_rand(rng, L) for _ in range(128)]
    eng400.set_targets(["." * L])
    a = eng400.score_batch(seqs, E.NEED_PF | E.NEED_MFE)
    b = eng400.score_batch(seqs, E.NEED_PF | E.NEED_MFE)
    assert a["mfe_ss"] == b["mfe_ss"] and (a["Emfe"] == b["Emfe"]).all()
    assert (a["Epf"].view(np.int64) == b["Epf"].view(np.int64)).all()
    c = eng400.score_batch(seqs[::-1][:17], E.NEED_PF | E.NEED_MFE)
    assert c["mfe_ss"] == a["mfe_ss"][::-1][:17]
    assert (c["Epf"].view(np.int64) == a["Epf"][::-1][:17].view(np.int64)).all()
    for k, s in enumerate(seqs):
        eng400.set_targets([a["mfe_ss"][k]])
        ed = eng400.score_batch([s], E.NEED_EVAL)["Ed"][0, 0]
        assert int(ed) == int(a["Emfe"][k])
        assert float(a["Epf"][k]) <= a["Emfe"][k] / 100.0 + 1e-9


def test_errors(eng400):
    from desirna_amd import engine as E
    eng400.set_targets(["." * 10])
    with pytest.raises(E.EngineError) as ei:
        eng400.score_batch(["GGGAAANCCC"])
    assert ei.value.code == -4
    with pytest.raises(E.EngineError) as ei:
        eng400.set_targets(["((..."])
    assert ei.value.code == -5
    with pytest.raises(E.EngineError) as ei:
        eng400.score_batch(["A" * 401], E.NEED_MFE)
    assert ei.value.code == -1


def test_replica_scorer_matches_reference_fields(traj_golden, example_inputs):
    """Host mirror of score_sequence(): ScoreSeq fields against the committed trajectory columns."""
    from types import SimpleNamespace
    from desirna_amd.energy_scores import ReplicaScorer, parse_scoring_functions
    inp = example_inputs["Standard_design_input"]
    input_file = SimpleNamespace(sec_struct=inp["sec_struct"][0], alt_sec_struct=None, alt_sec_structs=None)
    opts = SimpleNamespace(oligo_state="none", pks="off", subopt="off", motifs=None, param="1999",
                           scoring_f=parse_scoring_functions("Ed-Epf:1.0"))
    rows = [r for r in traj_golden if r["run"] == "Standard_design_input"][:64]
    sc = ReplicaScorer(input_file, opts, max_replicas=64)
    res = sc.score([r["sequence"] for r in rows])
    for r, s in zip(rows, res):
        assert s.mfe_ss == r["mfe_ss"]
        assert s.mcc == float(r["one_minus_mcc"]) and s.recall == float(r["one_minus_recall"])
        assert s.precision == float(r["one_minus_precision"])
        assert abs(s.Epf - float(r["Epf"])) < EPF_TOL_GOLDEN
        assert abs(s.edesired - float(r["edesired"])) < 1e-6
        assert abs(s.scoring_function - (s.edesired - s.Epf)) < 1e-12


# ---- ensemble defect: inside + outside recursion on the GPU against the oracle.  The reference
# holds no golden for this quantity ("parity unpinned" against the reference for Edef); the oracle's Z, P(i,j) and defect
# are verified to 1e-12 against the explicit Boltzmann-weighted sum over ALL structures of 37 short sequences
# (tests/test_oracle_golden.py::test_pf_bpp_defect_against_enumeration), and the kernel against that oracle here.
EDEF_TOL = 1e-10


@pytest.mark.parametrize("L", [1, 4, 5, 9, 36, 100, 200])
def test_ensemble_defect_vs_oracle(eng400, oracle, L):
    rng = np.random.default_rng(900 + L)
    seqs = [_rand(rng, L) for _ in range(5)] + [_rand(rng, L, "GC")]
    target = oracle.mfe(seqs[0])[0]
    eng400.set_targets([target])
    ed, bpp = eng400.ensemble_defect(seqs, want_bpp=True)
    for k, s in enumerate(seqs):
        oe, ob = oracle.ensemble_defect(s, target, want_bpp=True)
        assert abs(ed[k] - oe) < EDEF_TOL, (s, ed[k], oe)
        assert np.abs(bpp[k] - ob).max() < EDEF_TOL, s
    # bitwise repeatability and independence of batch composition
    ed2 = eng400.ensemble_defect(seqs[::-1])
    assert np.array_equal(ed2[::-1], ed)


def test_ensemble_defect_config5_shape(eng400, oracle, eterna_targets):
    """BASELINE config 5 shape: L=400 target, full-width batch; probabilities are a distribution per base."""
    target = eterna_targets["eteV1_53.txt"]
    assert len(target) == 400
    rng = np.random.default_rng(4005)
    seqs = [_rand(rng, 400) for _ in range(8)]
    eng400.set_targets([target])
    ed, bpp = eng400.ensemble_defect(seqs, want_bpp=True)
    for k in range(2):
        oe = oracle.ensemble_defect(seqs[k], target)
        assert abs(ed[k] - oe) < EDEF_TOL
    P = bpp + bpp.transpose(0, 2, 1)
    assert (P.sum(axis=2) <= 1.0 + 1e-9).all() and (bpp >= 0).all()
    assert ((0 <= ed) & (ed <= 1)).all()


def test_edef_scoring_function(eng400, oracle):
    """-sf Edef:1.0 through ReplicaScorer (reference energy_scores.py:93-94, :397-398)."""
    from types import SimpleNamespace
    from desirna_amd.energy_scores import ReplicaScorer
    target = "((((((.((((((((....))))).)).).))))))"
    inp = SimpleNamespace(sec_struct=target, alt_sec_struct=None, alt_sec_structs=None)
    opts = SimpleNamespace(oligo_state="none", subopt="off", pks="off", scoring_f=[("Edef", 1.0)], motifs={}, param="1999")
    sc = ReplicaScorer(inp, opts, max_replicas=4, engine=eng400)
    seqs = ["GGUGACACCGACGGCUACUGCCGUACGUGCGUCACC", "CGCGGGAGGGGGCCGGAAACGGCCACCACACCCGCG"]
    res = sc.score(seqs)
    for s, r in zip(seqs, res):
        assert abs(r.ensemble_defect - oracle.ensemble_defect(s, target)) < EDEF_TOL
        assert r.scoring_function == r.ensemble_defect


# ---- ragged batches (BASELINE config 4: the whole Eterna100-V1 set, 12 ... 400 nt, in ONE call)

def test_ragged_eterna100_one_call(eng400, oracle, eterna_solutions):
    """All 100 Eterna100-V1 solutions (reference eterna_benchmark results: MFE(sequence) == structure) as one ragged
    batch: strings exact, Emfe == E(structure), Epf against the oracle for the short ones and against the uniform-call
    path for a long one."""
    from desirna_amd import engine as E
    rows = eterna_solutions
    seqs = [r["sequence"] for r in rows]
    structs = [r["structure"] for r in rows]
    eng400.set_targets_ragged(structs)
    out = eng400.score_ragged(seqs, list(range(len(rows))))
    for k, r in enumerate(rows):
        assert out["mfe_ss"][k] == r["structure"], r["name"]
        assert int(out["Emfe"][k]) == int(out["Ed"][k]), r["name"]
    for k in sorted(range(len(rows)), key=lambda k: len(seqs[k]))[:25]:
        assert abs(float(out["Epf"][k]) - oracle.pf(seqs[k])) < EPF_TOL_ORACLE, rows[k]["name"]
        assert int(out["Emfe"][k]) == oracle.mfe(seqs[k])[1]
    k = max(range(len(rows)), key=lambda k: len(seqs[k]))
    eng400.set_targets([structs[k]])
    one = eng400.score_batch([seqs[k]])
    assert float(one["Epf"][0]) == float(out["Epf"][k]) and int(one["Ed"][0, 0]) == int(out["Ed"][k])


def test_ragged_config4_shape_matches_uniform_calls(eng400, eterna_solutions):
    """Config 4 shape at reduced replica count (100 puzzles x 4 mutated replicas = 400 sequences, lengths 12 ... 400, sorted
    longest-first inside the engine, LDS-resident and general kernels in the same call): every value equals what the
    per-puzzle uniform call returns (bitwise: the kernels are the same, only the batching differs)."""
    from desirna_amd import engine as E
    big = E.Engine(max_R=400, max_L=400, device=0)
    try:
        rng = np.random.default_rng(404)
        rows = eterna_solutions
        seqs, tof = [], []
        for p, r in enumerate(rows):
            for _ in range(4):
                s = list(r["sequence"])
                for pos in rng.choice(len(s), size=min(3, len(s)), replace=False):
                    s[pos] = "ACGU"[rng.integers(4)]
                seqs.append("".join(s))
                tof.append(p)
        big.set_targets_ragged([r["structure"] for r in rows])
        out = big.score_ragged(seqs, tof)
        for p in (0, 17, 52, 68, 99):
            big.set_targets([rows[p]["structure"]])
            mine = [k for k in range(len(seqs)) if tof[k] == p]
            ref = big.score_batch([seqs[k] for k in mine])
            for a, k in enumerate(mine):
                assert ref["mfe_ss"][a] == out["mfe_ss"][k]
                assert float(ref["Epf"][a]) == float(out["Epf"][k])
                assert int(ref["Emfe"][a]) == int(out["Emfe"][k]) and int(ref["Ed"][a, 0]) == int(out["Ed"][k])
    finally:
        big.close()


def test_ragged_edges_and_errors(eng400, oracle):
    from desirna_amd import engine as E
    seqs = ["G", "GGGAAACCC", "ACGU", "GGGGGAAAAACCCCC", "A" * 201]
    eng400.set_targets_ragged(["." * len(s) for s in seqs])
    out = eng400.score_ragged(seqs, list(range(len(seqs))))
    for k, s in enumerate(seqs):
        assert (out["mfe_ss"][k], int(out["Emfe"][k])) == oracle.mfe(s)
        assert int(out["Ed"][k]) == 0
    with pytest.raises(E.EngineError) as ei:
        eng400.score_ragged(["GGGAAACCC"], [0])            # structure 0 has length 1
    assert ei.value.code == -1
    with pytest.raises(E.EngineError) as ei:
        eng400.score_ragged(["GGGANACCC"], None)
    assert ei.value.code == -4


# ---- two strands: fc.mfe_dimer / fc.pf_dimer / two-strand eval_structure on the GPU

def test_cofold_golden_trajectories(eng400, traj_golden, example_inputs):
    """All 708 two-strand golden rows (hetero-dimer and homodimer example runs) through drna_cofold_batch: mfe_dimer strings
    exact, FAB = pf_dimer()[-1] within the float32 storage of the goldens, E(target) exact."""
    for run in ("RNA_RNA_complex_design_input", "Homodimer_design_input"):
        rows = [r for r in traj_golden if r["run"] == run]
        tg = example_inputs[run]["sec_struct"][0]
        eng400.set_targets([tg.replace("&", "")])
        for b in range(0, len(rows), 128):
            chunk = rows[b:b + 128]
            out = eng400.cofold_batch([r["sequence"] for r in chunk])
            for k, r in enumerate(chunk):
                assert out["mfe_ss"][k] == r["mfe_ss"], r["sequence"]
                assert abs(float(out["FAB"][k]) - float(r["Epf"])) < EPF_TOL_GOLDEN, r["sequence"]
                assert int(out["Ed"][k, 0]) == round(float(r["edesired"]) * 100), r["sequence"]


def test_cofold_vs_oracle_random(eng400, oracle):
    rng = np.random.default_rng(2024)
    for la, lb in ((1, 1), (2, 5), (17, 18), (40, 40), (33, 90), (100, 100)):
        seqs = [_rand(rng, la) + "&" + _rand(rng, lb) for _ in range(4)] + [_rand(rng, la, "GC") + "&" + _rand(rng, lb, "GC")]
        if la == lb:
            a = _rand(rng, la)
            seqs.append(a + "&" + a)                                 # homodimer: symmetry correction
        eng400.set_targets(["." * (la + lb)])
        out = eng400.cofold_batch(seqs)
        for k, s in enumerate(seqs):
            oss, oe = oracle.cofold_mfe(s)
            assert (out["mfe_ss"][k], int(out["Emfe"][k])) == (oss, oe), s
            fa, fb, fcab, fab = oracle.cofold_pf(s)
            got = [float(out[x][k]) for x in ("FA", "FB", "FcAB", "FAB")]
            assert max(abs(g - o) for g, o in zip(got, (fa, fb, fcab, fab))) < EPF_TOL_ORACLE, s
            assert int(out["Ed"][k, 0]) == 0


def test_two_strand_scorer_and_design_run(eng400, oracle, traj_golden, example_inputs):
    """oligo_state heterodimer / homodimer through ReplicaScorer (Epf = FAB, dimer MFE structure, two-strand E(target),
    SimScore with the '&' -> 'Ee' trick, oligomer bonus from FA / FB / FcAB), -o on (monomer bonus), and a short design run."""
    from types import SimpleNamespace
    from desirna_amd import design
    from desirna_amd.energy_scores import ReplicaScorer, oligo_fraction, kTlog_oligo_fraction, kTlog_monomer_fraction
    for run, state in (("RNA_RNA_complex_design_input", "heterodimer"), ("Homodimer_design_input", "homodimer")):
        tg = example_inputs[run]["sec_struct"][0]
        rows = [r for r in traj_golden if r["run"] == run][:40]
        inp = SimpleNamespace(sec_struct=tg, alt_sec_struct=None, alt_sec_structs=None)
        opts = SimpleNamespace(oligo_state=state, subopt="off", pks="off", scoring_f=[("Ed-Epf", 1.0)], motifs={}, param="1999")
        sc = ReplicaScorer(inp, opts, max_replicas=64, engine=eng400)
        res = sc.score([r["sequence"] for r in rows])
        for r, s in zip(rows, res):
            assert s.mfe_ss == r["mfe_ss"] and abs(s.Epf - float(r["Epf"])) < EPF_TOL_GOLDEN
            assert abs(s.edesired - float(r["edesired"])) < 1e-6 and s.mcc == float(r["one_minus_mcc"])
            fa, fb, fcab, fab = oracle.cofold_pf(r["sequence"])
            frac = oligo_fraction(fa, fb, fcab)
            ss1, ss2 = tg.split("&")
            bonus = kTlog_oligo_fraction(frac) if (state == "heterodimer" or ss1 != ss2) else kTlog_monomer_fraction(frac)
            assert abs(s.scoring_function - (s.edesired - s.Epf + bonus)) < 1e-7
    # -o on: one strand, monomer-fraction bonus from the homodimer of the sequence with itself
    tg = example_inputs["Standard_design_input"]["sec_struct"][0]
    inp = SimpleNamespace(sec_struct=tg, alt_sec_struct=None, alt_sec_structs=None)
    opts = SimpleNamespace(oligo_state="avoid", subopt="off", pks="off", scoring_f=[("Ed-Epf", 1.0)], motifs={}, param="1999")
    seq = "GGUGACACCGACGGCUACUGCCGUACGUGCGUCACC"
    s = ReplicaScorer(inp, opts, max_replicas=4, engine=eng400).score([seq])[0]
    fa, fb, fcab, fab = oracle.cofold_pf(seq + "&" + seq)
    assert abs(s.monomer_bonus - kTlog_monomer_fraction(oligo_fraction(fa, fb, fcab))) < 1e-7
    assert abs(s.scoring_function - (s.edesired - s.Epf + s.monomer_bonus)) < 1e-9
    # a short hetero-dimer design run through the Python driver
    d = example_inputs["RNA_RNA_complex_design_input"]
    inp = SimpleNamespace(name="cx", sec_struct=d["sec_struct"][0], seq_restr=d["seq_restr"][0], seed_seq=None, alt_sec_struct=None,
                          alt_sec_structs=None)
    res = design.run_design(inp, replicas=8, exchange=20, steps=3, seed=4)
    b = res["best"]
    assert b.sequence.count("&") == 1 and len(b.sequence) == len(inp.sec_struct) and b.mfe_ss.index("&") == inp.sec_struct.index("&")


# ---- second-best structure energy (-nd on): no golden in the reference ("parity unpinned"), the oracle's
# two-best dynamic programme is checked against exhaustive enumeration in tests/test_oracle_golden.py

def test_subopt_energy_vs_oracle(eng400, oracle):
    rng = np.random.default_rng(77)
    for L in (5, 13, 36, 100, 200, 260):
        seqs = [_rand(rng, L) for _ in range(4)] + [_rand(rng, L, "GC"), "A" * L]
        E2, E12 = eng400.subopt_energy(seqs, want_both=True)
        for k, s in enumerate(seqs):
            assert tuple(int(x) for x in E12[k]) == oracle.two_best(s), s
            assert int(E2[k]) == oracle.subopt_energy(s), s
            assert int(E12[k, 0]) == oracle.mfe(s)[1]


def test_small_engines(oracle):
    """engines sized exactly for short sequences (BASELINE config 1: L=16, R=4): the loop-size tables must not depend on max_L"""
    from desirna_amd import engine as E
    rng = np.random.default_rng(16)
    for L, R in ((16, 4), (9, 2), (5, 1)):
        eng = E.Engine(max_R=R, max_L=L, device=0)
        tg = "(((((......)))))" if L == 16 else "." * L
        eng.set_targets([tg])
        for rep in range(3):
            seqs = [_rand(rng, L, "GC" if k % 2 else "ACGU") for k in range(R)]
            out = eng.score_batch(seqs, E.NEED_MFE | E.NEED_PF | E.NEED_EVAL)
            for k, s in enumerate(seqs):
                ss, e = oracle.mfe(s)
                assert out["mfe_ss"][k] == ss and int(out["Emfe"][k]) == e, s
                assert abs(float(out["Epf"][k]) - oracle.pf(s)) < 1e-9, s
                assert int(out["Ed"][k, 0]) == oracle.eval_structure(s, tg), s
        eng.close()


def test_maximum_length(oracle):
    """the engine's length limit (MAXN - 2 = 2046 nt; 18 strips per fold, and the general one-workgroup kernels with "strips"
    off): same bits as the oracle"""
    from desirna_amd import engine as E
    rng = np.random.default_rng(4242)
    for L in (601, 2046):
        seqs = [_rand(rng, L)]
        eng = E.Engine(max_R=1, max_L=L, device=0)
        eng.set_targets(["." * L])
        out = eng.score_batch(seqs, E.NEED_MFE | E.NEED_PF | E.NEED_EVAL)
        ss, e = oracle.mfe(seqs[0])
        assert out["mfe_ss"][0] == ss and int(out["Emfe"][0]) == e and int(out["Ed"][0, 0]) == 0
        assert abs(float(out["Epf"][0]) - oracle.pf(seqs[0])) < 1e-9
        eng.set_option("strips", 0)
        gen = eng.score_batch(seqs, E.NEED_MFE | E.NEED_PF | E.NEED_EVAL)
        assert gen["mfe_ss"] == out["mfe_ss"] and int(gen["Emfe"][0]) == int(out["Emfe"][0]) and abs(gen["Epf"][0] - out["Epf"][0]) < 1e-9
        eng.close()
    with pytest.raises(Exception):
        E.Engine(max_R=1, max_L=2047, device=0)


def test_ranked_structures(eng400, oracle):
    """K lowest-energy structures (the call behind get_alt_mcc): energies ascending, rank 0 = MFE, rank 1 =
    the oracle's second-best energy, every string distinct and worth exactly its energy (oracle.eval_structure); short
    sequences against explicit enumeration in tests/test_kernels_emulated.py."""
    rng = np.random.default_rng(78)
    for L, K in ((5, 4), (13, 4), (36, 8), (100, 4), (200, 3), (260, 8)):
        seqs = [_rand(rng, L) for _ in range(3)] + [_rand(rng, L, "GC"), "A" * L]
        E, ss = eng400.subopt_structs(seqs, K)
        for k, s in enumerate(seqs):
            e = [int(x) for x in E[k]]
            assert e == sorted(e), s
            two = oracle.two_best(s)
            assert e[0] == oracle.mfe(s)[1] == two[0] and (K < 2 or e[1] == two[1]), s
            real = [x for x, v in zip(ss[k], e) if v < 10000000]
            assert len(set(real)) == len(real), s
            for x, v in zip(ss[k], e):
                if v < 10000000:
                    assert oracle.eval_structure(s, x) == v, (s, x)
                else:
                    assert x == "." * L


def test_ranked_structures_chunks_and_errors(eng400, oracle):
    """more sequences than one workspace chunk (16): results do not depend on the chunk a sequence lands in; error codes"""
    from desirna_amd.engine import EngineError
    rng = np.random.default_rng(80)
    seqs = [_rand(rng, 30) for _ in range(37)]
    E, ss = eng400.subopt_structs(seqs, 2)
    E1, ss1 = eng400.subopt_structs(seqs[20:23], 2)
    assert (E[20:23] == E1).all() and ss[20:23] == ss1
    for k, s in enumerate(seqs):
        assert tuple(int(x) for x in E[k]) == oracle.two_best(s), s
    with pytest.raises(EngineError) as ei:
        eng400.subopt_structs(["ACGUXACGUA"], 2)
    assert ei.value.code == -4
    with pytest.raises(EngineError) as ei:
        eng400.subopt_structs(["ACGUACGUA"], 9)
    assert ei.value.code == -1


def test_get_alt_mcc_records(eng400, oracle):
    """outputs.get_alt_mcc / sort_and_filter_alternative (reference sequence_utils.py:766-793, stats_inputs_outputs.py:422-460)."""
    from desirna_amd import outputs
    from desirna_amd.sim_score import SimScore
    rng = np.random.default_rng(79)
    L = 40
    alts = ["((((....))))" + "." * (L - 12), "." * (L - 12) + "((((....))))"]
    seqs = [_rand(rng, L) for _ in range(6)]
    recs = [{"sequence": s, "mcc": 0.1 * (k % 3), "edesired_minus_Epf": 1.0 + k, "Epf": -3.0, "scoring_function": 0.5 * k}
            for k, s in enumerate(seqs)] + [{"sequence": seqs[0], "mcc": 0.0, "edesired_minus_Epf": 9.0, "Epf": -1.0, "scoring_function": 7.0}]
    top, aug = outputs.sort_and_filter_alternative(recs, alts, eng400, num_results=4)
    assert len(aug) == 6 and len(top) == 4
    E, ss = eng400.subopt_structs(seqs, 3)
    for d in aug:
        k = seqs.index(d["sequence"])
        for a in (1, 2):
            assert d["alt_struct_%d" % a] == ss[k][a]
            sc = SimScore(alts[a - 1], ss[k][a]); sc.find_basepairs(); sc.cofusion_matrix()
            assert d["mcc_%d" % a] == 1 - sc.mcc()
    keys = [(d["mcc"], d["mcc_1"], d["mcc_2"]) for d in top]
    assert keys == sorted(keys)


def test_negative_design_scoring(eng400, oracle, traj_golden, example_inputs):
    """-nd on through ReplicaScorer (reference energy_scores.py:105-108): for candidates whose MFE structure is the target
    the scoring function loses (E_subopt - Epf)."""
    from types import SimpleNamespace
    from desirna_amd.energy_scores import ReplicaScorer
    tg = example_inputs["Standard_design_input"]["sec_struct"][0]
    rows = [r for r in traj_golden if r["run"] == "Standard_design_input"]
    solved = [r["sequence"] for r in rows if float(r["one_minus_mcc"]) == 0.0][:6]
    other = [r["sequence"] for r in rows if float(r["one_minus_mcc"]) > 0.0][:3]
    inp = SimpleNamespace(sec_struct=tg, alt_sec_struct=None, alt_sec_structs=None)
    opts = SimpleNamespace(oligo_state="none", subopt="on", pks="off", scoring_f=[("Ed-Epf", 1.0)], motifs={}, param="1999")
    res = ReplicaScorer(inp, opts, max_replicas=16, engine=eng400).score(solved + other)
    for s, sc in zip(solved + other, res):
        if sc.mcc == 0:
            assert sc.subopt_e == oracle.subopt_energy(s) / 100.0
            assert abs(sc.scoring_function - (sc.edesired_minus_Epf - (sc.subopt_e - sc.Epf))) < 1e-9
        else:
            assert sc.subopt_e == 0 and sc.scoring_function == sc.edesired_minus_Epf
    assert sum(sc.mcc == 0 for sc in res) == len(solved)


# ---- BASELINE configs 4 and 5 at their FULL sizes (round-1 verdict, item 8): the sizes where workspace, queueing and the
# mix of LDS-resident and general kernels in one call matter

def test_config4_full_size_3200_sequences(oracle, eterna_solutions):
    """Config 4 as stated: the 100 Eterna100-V1 puzzles x 32 mutated replicas = 3,200 sequences (12 ... 400 nt) in ONE
    ragged call.  Size-independent properties on every entry (E(MFE structure) == MFE energy through the oracle's
    evaluation for a sample, Epf <= Emfe, repeatability bit for bit) and 256 random entries against the oracle."""
    from desirna_amd import engine as E
    rng = np.random.default_rng(4004)
    rows = eterna_solutions
    seqs, tof = [], []
    for p, r in enumerate(rows):
        for _ in range(32):
            s = list(r["sequence"])
            for pos in rng.choice(len(s), size=min(3, len(s)), replace=False):
                s[pos] = "ACGU"[rng.integers(4)]
            seqs.append("".join(s))
            tof.append(p)
    assert len(seqs) == 3200
    big = E.Engine(max_R=3200, max_L=400, device=0)
    try:
        big.set_targets_ragged([r["structure"] for r in rows])
        out = big.score_ragged(seqs, tof)
        again = big.score_ragged(seqs, tof)
        assert out["mfe_ss"] == again["mfe_ss"] and (out["Emfe"] == again["Emfe"]).all() and (out["Ed"] == again["Ed"]).all()
        assert (out["Epf"].view(np.int64) == again["Epf"].view(np.int64)).all()
        assert (out["Epf"] <= out["Emfe"] / 100.0 + 1e-9).all()
        for k in range(len(seqs)):
            assert len(out["mfe_ss"][k]) == len(seqs[k]) and out["mfe_ss"][k].count("(") == out["mfe_ss"][k].count(")")
        for k in rng.choice(len(seqs), size=256, replace=False):
            k = int(k)
            ss, e = oracle.mfe(seqs[k])
            assert out["mfe_ss"][k] == ss and int(out["Emfe"][k]) == e, (k, rows[tof[k]]["name"])
            assert abs(float(out["Epf"][k]) - oracle.pf(seqs[k])) < EPF_TOL_ORACLE
            assert int(out["Ed"][k]) == oracle.eval_structure(seqs[k], rows[tof[k]]["structure"])
            assert oracle.eval_structure(seqs[k], ss) == e
    finally:
        big.close()


def test_config5_full_size_R128_L400_pk_alts_edef(eng400, oracle, eterna_targets):
    """Config 5 as stated: L=400 target, R=128, pk heuristic, two alternative targets and the ensemble defect, all in one
    test on one engine: properties on all 128 entries, 32 entries (every fourth, and both ends of the two sequence classes)
    against the oracle -- MFE structure with its pseudoknot layers, energies, Epf, E(target and alternatives), ensemble defect;
    the oracle's 32 x (up to four fills + inside + outside) run on its worker threads."""
    from desirna_amd import engine as E
    tg = eterna_targets["eteV1_53.txt"]
    alts = [eterna_targets["eteV1_22.txt"], eterna_targets["eteV1_63.txt"]]
    L = len(tg)
    rng = np.random.default_rng(5128)
    seqs = [_rand(rng, L) for _ in range(120)] + [_rand(rng, L, "GGCCAU") for _ in range(8)]
    eng400.set_targets([tg] + alts)
    flags = E.NEED_PF | E.NEED_MFE | E.NEED_EVAL | E.NEED_PK
    out = eng400.score_batch(seqs, flags)
    ed = eng400.ensemble_defect(seqs)
    again = eng400.score_batch(seqs, flags)
    assert out["mfe_ss"] == again["mfe_ss"] and (out["Ed"] == again["Ed"]).all()
    assert (out["Epf"].view(np.int64) == again["Epf"].view(np.int64)).all()
    assert ((0.0 <= ed) & (ed <= 1.0)).all() and (out["Epf"] <= out["Emfe"] / 100.0 + 1e-9).all()
    for k in range(128):
        s = out["mfe_ss"][k]
        for o, c in ("()", "[]", "<>", "{}"):
            assert s.count(o) == s.count(c)
        plain = "".join(ch if ch in "()" else "." for ch in s)
        # the pk-annotated string's '(' ')' layer is the unconstrained MFE structure: its energy is the MFE energy
        if k < 12:
            assert oracle.eval_structure(seqs[k], plain) == int(out["Emfe"][k])
    picks = sorted(set(list(range(0, 116, 4)) + [119, 120, 127]))        # 29 + 3
    assert len(picks) == 32

    def check(k):
        ss, e = oracle.mfe(seqs[k])
        assert out["mfe_ss"][k] == oracle.pk_struct(seqs[k], ss) and int(out["Emfe"][k]) == e
        assert abs(float(out["Epf"][k]) - oracle.pf(seqs[k])) < EPF_TOL_ORACLE
        for t, st in enumerate([tg] + alts):
            assert int(out["Ed"][k, t]) == oracle.eval_structure(seqs[k], st)
        assert abs(ed[k] - oracle.ensemble_defect(seqs[k], tg)) < 1e-10
        return k

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=8) as pool:          # the oracle's C calls release the GIL (ctypes)
        assert sorted(pool.map(check, picks)) == picks


def test_pf_range_error_code():
    """DRNA_ERR_PF_RANGE (-6): a parameter blob whose stacking energies are -40 kcal/mol drives the partition function of a
    36-nt hairpin beyond fp64 (Z ~ e^840); the engine must report it instead of returning inf / nan (LDS-resident and
    general kernel)."""
    from desirna_amd import engine as E, params
    blob = np.array(params.load_blob(), dtype=np.int32, copy=True)
    blob[3:3 + 64] = np.where(blob[3:3 + 64] < 0, -4000, blob[3:3 + 64])       # stack[8][8] follows the 3 header words
    seq = "GGGGGGGGGGGGGGGGAAAACCCCCCCCCCCCCCCC"
    for max_L, s in ((36, seq), (260, seq + "A" * 224)):
        eng = E.Engine(max_R=2, max_L=max_L, device=0, params=blob)
        try:
            eng.set_targets(["." * len(s)])
            with pytest.raises(E.EngineError) as ei:
                eng.score_batch([s, s], E.NEED_PF)
            assert ei.value.code == -6
            ok = eng.score_batch(["A" * len(s)] * 2, E.NEED_PF)               # the engine stays usable
            assert np.isfinite(ok["Epf"]).all()
        finally:
            eng.close()


def test_two_workgroup_mfe_equals_one_workgroup(eng400, oracle, eterna_targets):
    """Small batches fold the MFE with two workgroups per sequence (main + helper on another CU, fold_mfe_dual.hpp): same
    energies, same structures (pk rounds included), bit for bit, as the one-workgroup kernel ("dual" off) and as the oracle;
    lengths around the kernel's limits, a batch that does not fill the chip, repeated calls (the hand-shake flags are never
    reset: they grow with every call)."""
    from desirna_amd import engine as E
    rng = np.random.default_rng(2200)
    for L, R, pk in ((200, 64, False), (200, 3, True), (137, 40, True), (64, 17, False), (23, 64, True), (12, 5, False)):
        seqs = [_rand(rng, L) for _ in range(R - 1)] + [_rand(rng, L, "GC")]
        eng400.set_targets(["." * L])
        flags = E.NEED_MFE | E.NEED_PF | (E.NEED_PK if pk else 0)
        eng400.set_option("dual", 2)                     # two workgroups even beside the partition function
        try:
            a = eng400.score_batch(seqs, flags)
            b = eng400.score_batch(seqs, flags)
            eng400.set_option("dual", 0)
            c = eng400.score_batch(seqs, flags)
        finally:
            eng400.set_option("dual", 1)
        d = eng400.score_batch(seqs, E.NEED_MFE | (E.NEED_PK if pk else 0))      # the default policy's two-workgroup case
        assert d["mfe_ss"] == a["mfe_ss"] and (d["Emfe"] == a["Emfe"]).all()
        assert a["mfe_ss"] == b["mfe_ss"] == c["mfe_ss"] and (a["Emfe"] == c["Emfe"]).all() and (a["Emfe"] == b["Emfe"]).all()
        assert (a["Epf"].view(np.int64) == c["Epf"].view(np.int64)).all()
        for k in (0, R - 1):
            ss, e = oracle.mfe(seqs[k])
            if pk:
                ss = oracle.pk_struct(seqs[k], ss)
            assert a["mfe_ss"][k] == ss and int(a["Emfe"][k]) == e
    with pytest.raises(E.EngineError):
        eng400.set_option("no-such-option", 1)


def test_strip_kernels_equal_one_workgroup_kernels(eng400, oracle):
    """Sequences longer than 200 nt are folded by strips of columns, one workgroup each (fold_pf_strip.hpp, fold_mfe_strip.hpp):
    MFE energies and structures (pseudoknot rounds included) equal the general one-workgroup kernels' ("strips" off) and the
    oracle's bit for bit, the ensemble free energy to 1e-9 kcal/mol; two to four strips, lengths at the strip-count boundaries,
    batches below and far above one workgroup per CU (workgroups then queue behind each other: the strips of a sequence are
    dispatched upstream first), repeated calls (the flags are never reset), and the two-strip form of a 200-nt batch against
    the LDS-resident kernels."""
    from desirna_amd import engine as E
    rng = np.random.default_rng(2400)
    big = E.Engine(max_R=300, max_L=400, device=0)
    for L, R, pk in ((201, 5, True), (240, 64, False), (241, 9, True), (360, 30, False), (361, 8, True), (400, 128, True), (400, 300, False)):
        seqs = [_rand(rng, L) for _ in range(R - 1)] + [_rand(rng, L, "GC")]
        eng = big if R > 128 else eng400
        flags = E.NEED_PF | E.NEED_MFE | (E.NEED_PK if pk else 0)
        try:
            a = eng.score_batch(seqs, flags)
            b = eng.score_batch(seqs, flags)
            eng.set_option("strips", 0)
            c = eng.score_batch(seqs, flags)
        finally:
            eng.set_option("strips", 1)
        assert (a["Epf"].view(np.int64) == b["Epf"].view(np.int64)).all()
        assert np.abs(a["Epf"] - c["Epf"]).max() < 1e-9
        assert a["mfe_ss"] == b["mfe_ss"] == c["mfe_ss"] and (a["Emfe"] == c["Emfe"]).all() and (a["Emfe"] == b["Emfe"]).all()
        if pk and R >= 32:                 # the pseudoknot rounds of such a batch run in two halves on two streams: same answers as in one
            try:
                eng.set_option("mfe_split", 1)
                d1 = eng.score_batch(seqs, flags)
            finally:
                eng.set_option("mfe_split", 2)
            assert d1["mfe_ss"] == a["mfe_ss"] and (d1["Emfe"] == a["Emfe"]).all()
        for k in (0, R - 1):
            assert abs(a["Epf"][k] - oracle.pf(seqs[k])) < 1e-9
            ss, e = oracle.mfe(seqs[k])
            if pk:
                ss = oracle.pk_struct(seqs[k], ss)
            assert a["mfe_ss"][k] == ss and int(a["Emfe"][k]) == e
    big.close()
    seqs = [_rand(rng, 200) for _ in range(64)]
    flags = E.NEED_PF | E.NEED_MFE | E.NEED_PK
    try:
        eng400.set_option("strips", 2)
        a = eng400.score_batch(seqs, flags)
    finally:
        eng400.set_option("strips", 1)
    c = eng400.score_batch(seqs, flags)
    assert np.abs(a["Epf"] - c["Epf"]).max() < 1e-9      # (the multiloop sums are dealt to the waves differently: last bits differ)
    assert a["mfe_ss"] == c["mfe_ss"] and (a["Emfe"] == c["Emfe"]).all()
    with pytest.raises(E.EngineError):                   # a bad character is reported by the strip path too
        eng400.score_batch(["ACGU" * 60 + "N" + "ACGU" * 2], E.NEED_MFE | E.NEED_PF)


def test_blocked_mfe_strips_equal_plain_strips(eng400, oracle):
    """From four strips on (n > 360) the MFE strips fold their multiloop splits in blocked form (tile products of the far split
    points, fold_mfe_strip.hpp MKT_L).  Forced here for two to four strips ("mfe_fark_min_strips" = 2), with pseudoknot rounds and
    a batch above one workgroup per CU: structures and energies must equal the plain strips' and the oracle's; and 700 nt through
    the default switch against the general kernel."""
    from desirna_amd import engine as E
    rng = np.random.default_rng(2410)
    for L, R, pk in ((205, 6, True), (333, 40, True), (400, 96, False)):
        seqs = [_rand(rng, L) for _ in range(R - 1)] + [_rand(rng, L, "GC")]
        flags = E.NEED_MFE | (E.NEED_PK if pk else 0)
        a = eng400.score_batch(seqs, flags)
        try:
            eng400.set_option("mfe_fark_min_strips", 2)
            b = eng400.score_batch(seqs, flags)
            c = eng400.score_batch(seqs, flags)
        finally:
            eng400.set_option("mfe_fark_min_strips", 4)
        assert a["mfe_ss"] == b["mfe_ss"] == c["mfe_ss"] and (a["Emfe"] == b["Emfe"]).all() and (b["Emfe"] == c["Emfe"]).all()
        for k in (0, R - 1):
            ss, e = oracle.mfe(seqs[k])
            if pk:
                ss = oracle.pk_struct(seqs[k], ss)
            assert b["mfe_ss"][k] == ss and int(b["Emfe"][k]) == e
    eng = E.Engine(max_R=8, max_L=700, device=0)
    seqs = [_rand(rng, 700) for _ in range(3)]
    a = eng.score_batch(seqs, E.NEED_MFE | E.NEED_PK)
    # a ragged call: 6, 4, 4, 3 and 1 strips side by side (blocked from four on), with and without the partition function beside them
    rs = [_rand(rng, n) for n in (700, 450, 371, 260, 130)]
    r1 = eng.score_ragged(rs, flags=E.NEED_MFE)
    r2 = eng.score_ragged(rs, flags=E.NEED_MFE | E.NEED_PF)
    eng.set_option("strips", 0)
    b = eng.score_batch(seqs, E.NEED_MFE | E.NEED_PK)
    r0 = eng.score_ragged(rs, flags=E.NEED_MFE | E.NEED_PF)
    eng.close()
    assert a["mfe_ss"] == b["mfe_ss"] and (a["Emfe"] == b["Emfe"]).all()
    assert r1["mfe_ss"] == r0["mfe_ss"] == r2["mfe_ss"] and list(r1["Emfe"]) == list(r0["Emfe"]) == list(r2["Emfe"])
    assert np.abs(np.array(r2["Epf"]) - np.array(r0["Epf"])).max() < 1e-9


def test_lost_strip_falls_back_to_one_workgroup_per_fold(eng400, oracle):
    """HIP promises no dispatch order, so a fold by several workgroups may lose a partner (bounded wait -> ST_SYNC).  That must
    not fail the call: the engine redoes it with one workgroup per fold.  Injected here ("strip_fault": the top strip of every
    sequence gives up at once): results stay exact, the fallback is counted, and the next call uses the strips again."""
    from desirna_amd import engine as E
    rng = np.random.default_rng(2500)
    seqs = [_rand(rng, 260) for _ in range(5)]
    flags = E.NEED_PF | E.NEED_MFE | E.NEED_PK
    before = eng400.get_option("sync_fallbacks")
    try:
        eng400.set_option("strip_fault", 1)
        a = eng400.score_batch(seqs, flags)
    finally:
        eng400.set_option("strip_fault", 0)
    assert eng400.get_option("sync_fallbacks") == before + 1 and eng400.get_option("strips") == 1
    b = eng400.score_batch(seqs, flags)
    assert eng400.get_option("sync_fallbacks") == before + 1
    assert a["mfe_ss"] == b["mfe_ss"] and (a["Emfe"] == b["Emfe"]).all() and np.abs(a["Epf"] - b["Epf"]).max() < 1e-9
    ss, e = oracle.mfe(seqs[0])
    assert a["mfe_ss"][0] == oracle.pk_struct(seqs[0], ss) and int(a["Emfe"][0]) == e
    lens = [260, 100, 301]
    rs = [_rand(rng, n) for n in lens]
    try:
        eng400.set_option("strip_fault", 1)
        r1 = eng400.score_ragged(rs, flags=E.NEED_PF | E.NEED_MFE)
    finally:
        eng400.set_option("strip_fault", 0)
    r2 = eng400.score_ragged(rs, flags=E.NEED_PF | E.NEED_MFE)
    assert eng400.get_option("sync_fallbacks") == before + 2
    INJECTED_FALLBACKS.append(2)          # tests/test_zz_gpu_last.py: every other fallback of the suite is a real lost partner
    assert r1["mfe_ss"] == r2["mfe_ss"] and np.abs(np.array(r1["Epf"]) - np.array(r2["Epf"])).max() < 1e-9


def test_flag_epochs_are_reset_before_the_compare_range_runs_out(eng400, oracle):
    """The hand-over flags of the multi-workgroup folds hold monotone epoch values compared wrap-safe, which is only valid over
    half the 32-bit range: a slot never written (value 0) would read as published after 2^19 launches (round-2 advisor finding).
    The engine zeroes the flags and restarts the epochs long before that; jump to the reset point here and check that the
    calls on both sides of it give the same, correct, results for the strips (260 nt) and the two-workgroup MFE fold (200 nt)."""
    from desirna_amd import engine as E
    rng = np.random.default_rng(77001)
    long_seqs = [_rand(rng, 260) for _ in range(3)]
    short = [_rand(rng, 200) for _ in range(4)]
    flags = E.NEED_PF | E.NEED_MFE
    ref_l, ref_s = eng400.score_batch(long_seqs, flags), eng400.score_batch(short, flags)
    r0, f0 = eng400.get_option("flag_resets"), eng400.get_option("sync_fallbacks")
    eng400.set_option("debug_epoch", (1 << 17) - 1)
    for _ in range(3):
        a, b = eng400.score_batch(long_seqs, flags), eng400.score_batch(short, flags)
        assert a["mfe_ss"] == ref_l["mfe_ss"] and (a["Epf"].view(np.int64) == ref_l["Epf"].view(np.int64)).all()
        assert b["mfe_ss"] == ref_s["mfe_ss"] and (b["Epf"].view(np.int64) == ref_s["Epf"].view(np.int64)).all()
    assert eng400.get_option("flag_resets") == r0 + 3          # strips, two-workgroup MFE and partition-function helper flags, once each
    assert eng400.get_option("debug_epoch") < 64 and eng400.get_option("sync_fallbacks") == f0
    ss, e = oracle.mfe(short[0])
    assert ref_s["mfe_ss"][0] == ss and int(ref_s["Emfe"][0]) == e


def test_lost_pf_helper_costs_milliseconds(eng400, oracle):
    """The helper workgroup of the partition function may never show up (HIP promises no dispatch order).  Its waits are bounded
    by TIME (10 ms of the wall clock, not a poll count worth seconds): injected here -- the helpers leave at once -- the call
    comes back within tens of milliseconds, redone without helpers, with the same bits, and is counted."""
    import time
    from desirna_amd import engine as E
    rng = np.random.default_rng(31337)
    L = 200
    seqs = [_rand(rng, L) for _ in range(8)]
    flags = E.NEED_PF | E.NEED_MFE | E.NEED_EVAL
    eng400.set_targets(["." * L])
    ref = eng400.score_batch(seqs, flags)
    assert eng400.get_option("last_workgroups") == 4 * len(seqs)          # two workgroups per fold for both folds
    before = eng400.get_option("sync_fallbacks")
    try:
        eng400.set_option("helper_fault", 1)
        t0 = time.perf_counter()
        a = eng400.score_batch(seqs, flags)
        dt = time.perf_counter() - t0
    finally:
        eng400.set_option("helper_fault", 0)
    assert eng400.get_option("sync_fallbacks") == before + 1
    INJECTED_FALLBACKS.append(1)
    assert dt < 0.05, dt
    assert a["mfe_ss"] == ref["mfe_ss"] and (a["Epf"].view(np.int64) == ref["Epf"].view(np.int64)).all() and (a["Ed"] == ref["Ed"]).all()
    b = eng400.score_batch(seqs, flags)                                      # and the next call uses the helpers again
    assert eng400.get_option("sync_fallbacks") == before + 1 and (b["Epf"].view(np.int64) == ref["Epf"].view(np.int64)).all()
    assert abs(float(ref["Epf"][0]) - oracle.pf(seqs[0])) < EPF_TOL_ORACLE


def test_tile_products_give_the_same_bits_whoever_computes_them(eng400, oracle):
    """The far multiloop split points of the partition function are 4 x 4 tile products (fold_pf_lds.hpp): computed by
    the helper workgroup in small batches, by the main workgroup's sweep waves otherwise -- same device functions, so Epf must
    agree BITWISE between the two (the engine gives a fold a helper from 95 nt on; the emulated kernels compare the two at 30, 64
    and 96 nt), at lengths that leave ragged tiles; and with the oracle to 1e-9, also where a single block distance has a far
    range (21 .. 24 nt: the first tiles) or none at all."""
    from desirna_amd import engine as E
    rng = np.random.default_rng(4242)
    try:
        for L in (20, 21, 22, 23, 24, 37, 64, 94, 95, 97, 120, 121, 122, 123, 131, 150, 177, 199, 200):
            seqs = [_rand(rng, L) for _ in range(4)] + ["GC" * (L // 2) + "A" * (L % 2)]
            eng400.set_targets(["." * L])
            eng400.set_option("pf_helper", 1)
            a = eng400.score_batch(seqs, E.NEED_PF)
            assert eng400.get_option("last_workgroups") == (2 if L >= 95 else 1) * len(seqs), L       # main + helper per fold from 95 nt on
            eng400.set_option("pf_helper", 0)
            b = eng400.score_batch(seqs, E.NEED_PF)
            assert eng400.get_option("last_workgroups") == len(seqs), L
            assert (a["Epf"].view(np.int64) == b["Epf"].view(np.int64)).all(), L
            for k in (0, len(seqs) - 1):
                assert abs(float(a["Epf"][k]) - oracle.pf(seqs[k])) < EPF_TOL_ORACLE, (L, k)
    finally:
        eng400.set_option("pf_helper", 1)


def test_repeated_lost_partners_switch_the_multi_workgroup_paths_off_for_a_while(eng400):
    """A GPU shared with another process loses partners call after call, and every lost call costs its wait budget before it is
    redone (round-2 advisor finding).  Three fallbacks in a row: the engine folds with one workgroup per fold for the next 1000
    calls (no wait, no fallback), results unchanged; setting one of the path options ends that at once."""
    from desirna_amd import engine as E
    rng = np.random.default_rng(77003)
    seqs = [_rand(rng, 200) for _ in range(6)]
    flags = E.NEED_PF | E.NEED_MFE
    ref = eng400.score_batch(seqs, flags)
    wgs = eng400.get_option("last_workgroups")
    before = eng400.get_option("sync_fallbacks")
    try:
        eng400.set_option("helper_fault", 1)
        for k in range(3):
            a = eng400.score_batch(seqs, flags)
            assert eng400.get_option("sync_fallbacks") == before + k + 1
        INJECTED_FALLBACKS.append(3)
        assert eng400.get_option("solo_calls_left") == 1000
        b = eng400.score_batch(seqs, flags)                      # the fault is still on: nobody waits for a helper now
        assert eng400.get_option("sync_fallbacks") == before + 3 and eng400.get_option("solo_calls_left") == 999
        assert eng400.get_option("last_workgroups") < wgs
    finally:
        eng400.set_option("helper_fault", 0)
    eng400.set_option("pf_helper", 1)                            # any path option: probe again at once
    assert eng400.get_option("solo_calls_left") == 0
    c = eng400.score_batch(seqs, flags)
    assert eng400.get_option("last_workgroups") == wgs and eng400.get_option("sync_fallbacks") == before + 3
    for r in (a, b, c):
        assert r["mfe_ss"] == ref["mfe_ss"] and (r["Emfe"] == ref["Emfe"]).all()
        assert (r["Epf"].view(np.int64) == ref["Epf"].view(np.int64)).all()


def test_batches_larger_than_the_workspace_go_in_chunks(eng400, oracle, monkeypatch):
    """The O(L^2) workspaces hold DRNA_WS_GB (default 8) at most; a batch with more sequences than fit goes through them in
    chunks (ragged call: consecutive runs of the length-sorted order, back to back on the same streams; uniform call: sub-batches).
    Forced here with a tiny budget: results must equal the unchunked engine's bit for bit, in the caller's order."""
    from desirna_amd import engine as E
    rng = np.random.default_rng(8080)
    lens = [260, 31, 199, 7, 301, 120, 64, 230, 200, 1, 88, 270, 150, 45, 201, 12] * 3
    seqs = [_rand(rng, n) for n in lens]
    flags = E.NEED_PF | E.NEED_MFE
    ref = eng400.score_ragged(seqs, flags=flags)
    monkeypatch.setenv("DRNA_WS_GB", "0.1")
    small = E.Engine(max_R=len(seqs), max_L=301, device=0)
    try:
        slots = small.get_option("workspace_slots")
        assert 8 <= slots < len(seqs) // 2, slots                       # three chunks at least
        out = small.score_ragged(seqs, flags=flags)
        assert out["mfe_ss"] == ref["mfe_ss"] and (np.array(out["Emfe"]) == np.array(ref["Emfe"])).all()
        assert (np.array(out["Epf"]).view(np.int64) == np.array(ref["Epf"]).view(np.int64)).all()
        uni = [_rand(rng, 130) for _ in range(slots + 9)]
        a = small.score_batch(uni, flags)
        b = eng400.score_batch(uni, flags)
        assert a["mfe_ss"] == b["mfe_ss"] and (a["Epf"].view(np.int64) == b["Epf"].view(np.int64)).all()
        # the ensemble defect goes through the same workspaces: more sequences than slots in sub-batches (advisor r3: it used to
        # return an argument error, which aborted a native Monte-Carlo run with an Edef term at its first defect call)
        tg130 = "((((....))))" + "." * 118
        small.set_targets([tg130]); eng400.set_targets([tg130])
        ea, eb = small.ensemble_defect(uni), eng400.ensemble_defect(uni)
        assert (ea.view(np.int64) == eb.view(np.int64)).all() and abs(ea[-1] - oracle.ensemble_defect(uni[-1], tg130)) < 1e-10
    finally:
        small.close()
    k = 4
    assert (ref["mfe_ss"][k], int(ref["Emfe"][k])) == oracle.mfe(seqs[k]) and abs(float(ref["Epf"][k]) - oracle.pf(seqs[k])) < EPF_TOL_ORACLE


def test_bench_exchange_step_over_rccl_single_rank():
    """bench.py's exchange step (all-gather of the replica scores, barriers, max all-reduce of the time: reference
    utils/replica_exchange_monte_carlo.py:113-173) through RCCL itself: DRNA_BENCH_FORCE_DIST=1 builds a one-rank communicator
    on the one GPU this box has and runs the same collectives on device tensors that N ranks run; the line must say so and
    carry the same metric."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DRNA_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "DRNA_BENCH_BACKEND"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "1", "--no-cpu-baseline",
                        "--exchange-every", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [x for x in p.stdout.splitlines() if x.startswith("{")][-1]
    d = json.loads(line)
    assert d["collectives"] == "nccl" and d["n_gpus"] == 1 and d["metric"] and d["value"] > 0


def test_bench_line_carries_parity_roofline_baseline_and_the_monte_carlo_loop():
    """The driver's own run in small: `bench.py --steps 3 --warmup 1` WITH the CPU baseline must end in one JSON line whose
    timed device buffers were checked against the oracle (parity_checked), with the roofline of the dominant kernel, the CPU
    baseline beside it, and the end-to-end Monte-Carlo loop figure (reference utils/replica_exchange_monte_carlo.py:176-210)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "DRNA_BENCH_BACKEND", "DRNA_BENCH_FORCE_DIST"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--exchange-every", "20"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([x for x in p.stdout.splitlines() if x.startswith("{")][-1])
    assert d["parity_checked"] is True and d["parity"]["Emfe_Ed_structures"] == "bit-exact" and d["parity"]["max_abs_dEpf"] < 1e-9
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["value"] > 0 and d["sync_fallbacks"] == 0
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "lds", "valu") and 0.0 < rf["frac"] <= 1.0 and rf["traffic"] > 0 and rf["own_floor"]["floor_ms"] > 0
    assert d["achieved_hbm_GB_s"] > 0
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and cb["kind"] == "port" and cb["cores"] >= 1
    mc = d["mc_loop"]
    assert mc["scored_sequences_per_s"] > 0 and mc["iterations"] == 20 and mc["replicas"] == 64 and mc["L"] == 200
    assert mc["ms_per_iteration"] >= mc["kernel_ms_per_iteration"] > 0 and mc["accepted"] + mc["rejected"] == 20 * 64
    assert "workgroups <= " in d["cus_occupied"]["resident_check"]
    rs = {x["R"]: x for x in d["r_sweep"]}                     # replicas per call: more folds per call must not cost throughput
    assert sorted(rs) == [32, 64, 128, 256] and all(x["sync_fallbacks"] == 0 for x in rs.values())
    assert rs[128]["replica_folds_per_s"] > rs[64]["replica_folds_per_s"] and rs[256]["replica_folds_per_s"] > 0.95 * rs[128]["replica_folds_per_s"]


def test_one_launch_form_equals_the_two_launches(eng400, oracle, eterna_targets):
    """Option "fused" (the default): both folds of a small batch in ONE launch of 4 R workgroups (fold_fused.hpp).  Same device
    functions, so every output is bit for bit that of the two launches; odd batch sizes leave idle blocks at the end of the grid;
    pseudoknot rounds run inside the MFE roles."""
    from desirna_amd import engine as E
    rng = np.random.default_rng(4141)
    tg = eterna_targets["eteV1_69.txt"]
    for L, R, pk in ((200, 64, False), (200, 63, True), (180, 5, False), (130, 9, False)):      # (two workgroups per MFE fold from 125 nt on, 170 with pk rounds)
        t = tg[:L] if L == 200 else "." * L
        seqs = [_rand(rng, L) for _ in range(R)]
        eng400.set_targets([t])
        flags = E.NEED_PF | E.NEED_MFE | E.NEED_EVAL | (E.NEED_PK if pk else 0)
        default = eng400.get_option("fused")
        assert default == 1
        try:
            eng400.set_option("fused", 0)
            a = eng400.score_batch(seqs, flags)
            assert eng400.get_option("last_fused") == 0
            eng400.set_option("fused", 1)
            b = eng400.score_batch(seqs, flags)
            assert eng400.get_option("last_fused") == 1 and eng400.get_option("last_workgroups") == 4 * R
            c = eng400.score_batch(seqs, flags)
        finally:
            eng400.set_option("fused", default)
        for x in (b, c):
            assert x["mfe_ss"] == a["mfe_ss"] and (x["Emfe"] == a["Emfe"]).all() and (x["Ed"] == a["Ed"]).all()
            assert (x["Epf"].view(np.int64) == a["Epf"].view(np.int64)).all()
        tm = eng400.last_timing()
        assert 0 < tm["mfe"] <= tm["total"] * 1.05 and 0 < tm["pf"] <= tm["total"] * 1.05
        ss, e = oracle.mfe(seqs[0])
        assert (a["mfe_ss"][0] == (oracle.pk_struct(seqs[0], ss) if pk else ss)) and int(a["Emfe"][0]) == e
