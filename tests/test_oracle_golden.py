"""Pin the CPU oracle against every golden vector the reference tree holds for the scoring path
(SURVEY.md App. D): committed example trajectories and Eterna100-V1 solutions."""
import numpy as np
import pytest

RUN_INPUT = {
    "Standard_design_input": "Standard_design_input",
    "Seed_sequence_design_input": "Seed_sequence_design_input",
    "Alternative_structures_design_input": "Alternative_structures_design_input",
    "Pseudoknot_design_input": "Pseudoknot_design_input",
    "RNA_RNA_complex_design_input": "RNA_RNA_complex_design_input",
    "Homodimer_design_input": "Homodimer_design_input",
}
SINGLE = ("Standard_design_input", "Seed_sequence_design_input",
          "Alternative_structures_design_input", "Pseudoknot_design_input")
# one exact-energy tie in the legacy Seed run resolves differently
TIE_OUTLIER = "GGUACAGCCGUGCCCCUUAGGGCACCGUGGUGUACC"
EPF_TOL = 2e-6  # kcal/mol: goldens are float32


def _rows(traj_golden, runs):
    return [r for r in traj_golden if r["run"] in runs]


def test_known_answers(oracle):
    tgt = "((((((.((((((((....))))).)).).))))))"
    for s, epf, ed in [("GGUGACACCGACGGCUACUGCCGUACGUGCGUCACC", -20.843525, -2080),
                       ("CGCGGGAGGGGGCCGGAAACGGCCACCACACCCGCG", -29.483187, -2890),
                       ("GCCCCGGCCCCCGGCGAAAGCCGGUGGAGGCGGGGC", -32.114521, -3210)]:
        assert abs(oracle.pf(s) - epf) < EPF_TOL
        assert oracle.eval_structure(s, tgt) == ed
    assert oracle.mfe("GCCUGGAUUAACAGGC") == ("(((((......)))))", -790)


def test_eval_structure_exact(oracle, traj_golden, example_inputs):
    n = 0
    for r in _rows(traj_golden, SINGLE):
        tgt = example_inputs[r["run"]]["sec_struct"][0]
        ed = oracle.eval_structure(r["sequence"], tgt)
        assert ed == round(float(r["edesired"]) * 100), r["sequence"]
        n += 1
    assert n == 621 + 1330 + 67 + 581


def test_eval_alt_structures_exact(oracle, traj_golden, example_inputs):
    alts = example_inputs["Alternative_structures_design_input"]["alt_sec_struct"]
    for r in _rows(traj_golden, ("Alternative_structures_design_input",)):
        es = [oracle.eval_structure(r["sequence"], a) for a in alts]
        assert abs(sum(es) / len(es) / 100.0 - float(r["edesired2"])) < 1e-5


def test_eval_two_strand_exact(oracle, traj_golden, example_inputs):
    n = 0
    for r in _rows(traj_golden, ("RNA_RNA_complex_design_input", "Homodimer_design_input")):
        tgt = example_inputs[r["run"]]["sec_struct"][0]
        cut = r["sequence"].index("&")
        ed = oracle.eval_structure(r["sequence"], tgt, cut=cut)
        assert ed == round(float(r["edesired"]) * 100), r["sequence"]
        n += 1
    assert n == 538 + 170


def test_epf_within_float32(oracle, traj_golden):
    worst = 0.0
    for r in _rows(traj_golden, SINGLE):
        worst = max(worst, abs(oracle.pf(r["sequence"]) - float(r["Epf"])))
    assert worst < EPF_TOL, worst


def test_mfe_structures_exact(oracle, traj_golden):
    miss = []
    for r in _rows(traj_golden, SINGLE):
        ss, _ = oracle.mfe(r["sequence"])
        if int(r["pk_on"]):
            ss = oracle.pk_struct(r["sequence"], ss)
        if ss != r["mfe_ss"]:
            miss.append(r["sequence"])
    assert set(miss) <= {TIE_OUTLIER}, miss


def test_mfe_energy_equals_eval_of_mfe_structure(oracle, traj_golden):
    for r in _rows(traj_golden, ("Standard_design_input",))[:200]:
        ss, e = oracle.mfe(r["sequence"])
        assert oracle.eval_structure(r["sequence"], ss) == e


def test_simscore_matches_committed_metrics(oracle, traj_golden, example_inputs):
    n = 0
    for r in traj_golden:
        tgt = example_inputs[r["run"]]["sec_struct"][0].replace("&", "Ee")
        (mcc, rec, prec), _ = oracle.simscore(tgt, r["mfe_ss"].replace("&", "Ee"))
        assert 1 - mcc == float(r["one_minus_mcc"]), r
        assert 1 - rec == float(r["one_minus_recall"]), r
        assert 1 - prec == float(r["one_minus_precision"]), r
        n += 1
    assert n > 3000


def test_eterna_v1_solutions_fold_to_target(oracle, eterna_solutions):
    assert len(eterna_solutions) == 100
    for r in eterna_solutions:
        ss, e = oracle.mfe(r["sequence"])
        assert ss == r["structure"], r["name"]
        assert oracle.eval_structure(r["sequence"], r["structure"]) == e


def test_pf_is_below_mfe_and_scale_free(oracle, eterna_solutions):
    for r in eterna_solutions[:40]:
        _, e = oracle.mfe(r["sequence"])
        assert oracle.pf(r["sequence"]) <= e / 100.0 + 1e-9


def _enumerate(seq, oracle):
    """all secondary structures of a short sequence with their energies (brute force)"""
    n = len(seq)
    pairs = {("A", "U"), ("U", "A"), ("G", "C"), ("C", "G"), ("G", "U"), ("U", "G")}
    out = []

    def rec(i, cur, stack):
        if i == n:
            if not stack:
                out.append("".join(cur))
            return
        cur.append(".")
        rec(i + 1, cur, stack)
        cur.pop()
        cur.append("(")
        stack.append(i)
        rec(i + 1, cur, stack)
        stack.pop()
        cur.pop()
        if stack and i - stack[-1] > 3 and (seq[stack[-1]], seq[i]) in pairs:
            o = stack.pop()
            cur.append(")")
            rec(i + 1, cur, stack)
            cur.pop()
            stack.append(o)

    rec(0, [], [])
    return out


def test_pf_and_bpp_against_enumeration(oracle):
    """Not pinned by the reference (no golden for bpp / ensemble defect): check the recursions
    against explicit enumeration of all structures, with Boltzmann weights from the INTEGER loop
    model (pf_smooth off for that comparison is not available, so compare bpp-derived quantities
    only through self-consistency: sum_j P_ij <= 1, defect of MFE structure in [0,1])."""
    rng = np.random.default_rng(5)
    for _ in range(5):
        seq = "".join(rng.choice(list("ACGU"), 14))
        ss, _ = oracle.mfe(seq)
        ed, bpp = oracle.ensemble_defect(seq, ss, want_bpp=True)
        assert 0.0 <= ed <= 1.0
        p = bpp + bpp.T
        assert (p.sum(axis=1) <= 1 + 1e-9).all()


# ---- two strands: co-fold MFE and partition function, pinned on the reference's two-strand trajectories

def test_cofold_goldens(oracle, traj_golden):
    """fc.mfe_dimer() strings and fc.pf_dimer()[-1] free energies of every sequence of the hetero-dimer (538) and homodimer
    (170, rotational-symmetry correction) example runs; E(structure) of the two-strand evaluation agrees with the DP."""
    rows = [r for r in traj_golden if "&" in r["sequence"]]
    assert len(rows) == 708
    worst = 0.0
    for r in rows:
        ss, e = oracle.cofold_mfe(r["sequence"])
        assert ss == r["mfe_ss"], r["sequence"]
        a, b = r["sequence"].split("&")
        assert e == oracle.eval_structure(a + b, ss.replace("&", ""), cut=len(a)), r["sequence"]
        fab = oracle.cofold_pf(r["sequence"])[3]
        worst = max(worst, abs(fab - float(r["Epf"])))
    assert worst < 2e-6          # goldens are float32


def test_cofold_reduces_to_monomers_when_strands_cannot_interact(oracle):
    """Two strands with no possible inter-strand pair: the dimer ensemble is the product of the monomer ensembles."""
    a, b = "GGGAAACCC", "GGGAAAACCC"          # G-C only inside each strand... but G/C across strands can pair: use poly-A spacer
    a, b = "AAAAAAAAA", "AAAAAAAAAA"
    fa, fb, fcab, fab = oracle.cofold_pf(a + "&" + b)
    assert abs(fa - oracle.pf(a)) < 1e-9 and abs(fb - oracle.pf(b)) < 1e-9
    assert abs(fab - (fa + fb)) < 1e-9 and fcab == 999.0
    ss, e = oracle.cofold_mfe(a + "&" + b)
    assert ss == "." * len(a) + "&" + "." * len(b) and e == 0


def test_two_best_against_enumeration(oracle):
    """-nd on takes the energy of the second entry of ViennaRNA's sorted subopt list; the reference holds no golden for it
    ("parity unpinned"): the oracle's two-best dynamic programme is checked against explicit enumeration of all structures."""
    rng = np.random.default_rng(3)
    for trial in range(40):
        L = int(rng.integers(8, 16))
        s = "".join(rng.choice(list("ACGU" if trial % 3 else "GC"), L))
        en = sorted(oracle.eval_structure(s, st) for st in _enumerate(s, oracle))
        assert oracle.two_best(s) == (en[0], en[1] if len(en) > 1 else 10000000), s
    assert oracle.subopt_energy("AAAAAAAA") == 0                       # one structure only: the reference's fallback
