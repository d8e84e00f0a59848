"""Batched design driver (host side): constraints, move set, and a full REMC run.  The CPU tests drive the
loop with an ORACLE-backed scorer (test infrastructure); the GPU test runs BASELINE config 1 through the engine."""
import random
from types import SimpleNamespace

import pytest

from desirna_amd import design
from desirna_amd.energy_scores import ScoreSeq, parse_scoring_functions
from desirna_amd.sim_score import batch_metrics

ETE1 = "(((((......)))))"


class OracleScorer:
    """Same contract as energy_scores.ReplicaScorer.score, numbers from the CPU oracle (tests only)."""

    def __init__(self, oracle, target, scoring_f):
        self.o, self.target, self.sf = oracle, target, scoring_f
        self.pk = bool(set(target) - set(".()"))

    def score(self, seqs):
        out = []
        for s in seqs:
            ss, _ = self.o.mfe(s)
            if self.pk:
                ss = self.o.pk_struct(s, ss)
            sc = ScoreSeq(s)
            sc.get_Epf(self.o.pf(s))
            sc.get_mfe_ss(ss)
            sc.get_edesired(self.o.eval_structure(s, self.target) / 100.0)
            sc.get_edesired_minus_Epf(sc.Epf, sc.edesired)
            mcc, rec, prec = batch_metrics(self.target, [ss])[0]
            sc.get_precision(prec); sc.get_recall(rec); sc.get_mcc(mcc)
            sc.get_scoring_function(self.sf)
            out.append(sc)
        return out


class OracleEngine:
    """Duck-typed stand-in for engine.Engine in run_design_fast's per-iteration loop (native_loop=False), numbers from the
    CPU oracle: lets the sharded fast driver (native proposer + Metropolis + MT19937 streams) run without a GPU (tests only)."""
    TERM_IDS = {}                                  # no drna_mc_run here: forces the per-iteration loop

    def __init__(self, oracle):
        self.o, self.targets = oracle, []

    def set_targets(self, targets):
        self.targets = list(targets)

    def score_batch_arrays(self, seqs_u8, flags=0):
        import numpy as np
        seqs = [bytes(r).decode() for r in seqs_u8]
        Epf, Emfe, ss, Ed = self.o.score_batch(seqs, self.targets, threads=4)
        return Epf, Emfe, np.frombuffer("".join(ss).encode(), dtype=np.uint8).reshape(len(seqs), -1).copy(), Ed


def test_fast_driver_equals_python_driver_with_oracle(oracle):
    """For a fixed seed the batched driver (native proposer with the reference's MT19937 streams, native Metropolis) and the
    per-replica Python driver walk the same trajectory: same records after every exchange step, same counters."""
    inp = SimpleNamespace(name="ete1", sec_struct=ETE1, seq_restr="N" * 16, seed_seq=None, alt_sec_struct=None,
                          alt_sec_structs=None)
    sc = OracleScorer(oracle, ETE1, parse_scoring_functions("Ed-Epf:1.0"))
    a = design.run_design(inp, replicas=6, exchange=15, steps=4, seed=9, scorer=sc)
    b = design.run_design_fast(inp, replicas=6, exchange=15, steps=4, seed=9, engine=OracleEngine(oracle), native_loop=False)
    assert [r["sequence"] for r in a["simulation_data"]] == [r["sequence"] for r in b["simulation_data"]]
    assert [r["temp_shelf"] for r in a["simulation_data"]] == [r["temp_shelf"] for r in b["simulation_data"]]
    assert [r["sim_step"] for r in b["simulation_data"][-6:]] == [4 * 15] * 6          # stats.step = global_step * RE_attempt
    for k in ("acc_mc", "acc_mc_better", "rej_mc", "acc_re", "rej_re", "scored"):
        assert a["stats"][k] == b["stats"][k], k
    assert a["best"].sequence == b["best"].sequence and abs(a["best"].scoring_function - b["best"].scoring_function) < 1e-9


def test_problem_constraints_and_moves():
    p = design.DesignProblem("((((....))))..", "NNNSNNNNWNNNNA")
    assert p.allowed[13] == ["A"] and 13 not in p.mutable
    assert p.allowed[3] == ["G"] and p.allowed[8] == ["U"]        # S pairs with W -> only G-U survives
    rng = random.Random(1)
    s = p.initial_sequence(rng)
    assert len(s) == 14 and set(s) <= set("ACGU") and s[13] == "A"
    for i, j in p.pairs:
        assert s[j] in design.CAN_PAIR[s[i]]
    for _ in range(200):
        pos = rng.choice(p.mutable)
        m = p.mutate(s, pos, rng)
        for i, j in p.pairs:
            assert m[j] in design.CAN_PAIR[m[i]]
        assert m[13] == "A"
        s = m
    with pytest.raises(ValueError):
        design.DesignProblem("(....)", "ANNNNC")


def test_targeted_position_window():
    p = design.DesignProblem(ETE1)
    rng = random.Random(0)
    # current structure misses everything: all target pairs are false negatives -> window covers 1..15
    hits = {p.mutation_position("." * 16, 0, 4, 1.0, 1.0, True, rng) for _ in range(300)}
    assert hits <= set(range(1, 16)) and len(hits) > 8
    assert p.mutation_position(ETE1, 0, 4, 0.7, 0.0, True, rng) in p.mutable    # solved: uniform choice


def test_read_input_format(tmp_path):
    f = tmp_path / "in.txt"
    f.write_text(">name\nX\n>seq_restr\nNNNNNNNNNNNNNNNN\n>sec_struct\n%s\n" % ETE1)
    inp = design.read_input(str(f))
    assert inp.name == "X" and inp.sec_struct == ETE1 and inp.alt_sec_struct is None


def test_design_run_solves_eterna1_with_oracle_scorer(oracle):
    inp = SimpleNamespace(name="ete1", sec_struct=ETE1, seq_restr="N" * 16, seed_seq=None, alt_sec_struct=None,
                          alt_sec_structs=None)
    sc = OracleScorer(oracle, ETE1, parse_scoring_functions("Ed-Epf:1.0"))
    res = design.run_design(inp, replicas=4, exchange=20, steps=6, seed=5, scorer=sc, stop_when_solved=True)
    assert res["solved"] and res["best"].mfe_ss == ETE1 and res["best"].mcc == 0.0
    assert res["stats"]["scored"] >= 4


@pytest.mark.gpu
def test_config1_design_run_on_gpu(eterna_targets):
    """BASELINE configs[0] plumbing on the GPU: Eterna100 V1 #1, R=4, -sf Ed-Epf."""
    tg = eterna_targets["eteV1_01.txt"]
    inp = SimpleNamespace(name="ete1", sec_struct=tg, seq_restr="N" * len(tg), seed_seq=None, alt_sec_struct=None,
                          alt_sec_structs=None)
    res = design.run_design(inp, replicas=4, exchange=50, steps=10, seed=3, stop_when_solved=True)
    assert res["solved"] and res["best"].mfe_ss == tg


@pytest.mark.gpu
def test_design_run_L100_on_gpu(eterna_targets):
    tg = eterna_targets["eteV1_92.txt"]
    inp = SimpleNamespace(name="ete92", sec_struct=tg, seq_restr="N" * len(tg), seed_seq=None, alt_sec_struct=None,
                          alt_sec_structs=None)
    res = design.run_design(inp, replicas=16, exchange=100, steps=4, seed=1)
    assert res["best"].mcc < 0.6 and res["stats"]["scored"] == 16 + 4 * 100 * 16


# ---- native batched host helpers (no GPU needed: plain CPU code in the C-ABI library)

@pytest.fixture(scope="module")
def hk():
    import __graft_entry__ as g
    g.build()
    from desirna_amd.engine import HostKernels
    return HostKernels()


def test_native_simscore_matches_reference_columns(hk, traj_golden, example_inputs):
    import numpy as np
    for run in ("Standard_design_input", "Pseudoknot_design_input", "RNA_RNA_complex_design_input"):
        rows = [r for r in traj_golden if r["run"] == run]
        ref = example_inputs[run]["sec_struct"][0].replace("&", "Ee")
        q = np.array([np.frombuffer(r["mfe_ss"].replace("&", "Ee").encode(), dtype=np.uint8) for r in rows])
        mcc, rec, prec = hk.simscore(ref, q)
        for k, r in enumerate(rows):
            assert 1 - mcc[k] == float(r["one_minus_mcc"]) and 1 - rec[k] == float(r["one_minus_recall"])
            assert 1 - prec[k] == float(r["one_minus_precision"])


def test_native_proposals_respect_constraints(hk):
    import numpy as np
    target = "((((....))))..[[..]]"
    restr = "NNNSNNNNWNNNNANNNNNN"
    p = design.DesignProblem(target, restr)
    amask = np.array([sum(1 << "ACGU".index(c) for c in a) for a in p.allowed], dtype=np.uint8)
    R = 32
    seq = p.initial_sequence(random.Random(2))
    cur = np.tile(np.frombuffer(seq.encode(), dtype=np.uint8), (R, 1)).copy()
    ss = np.tile(np.frombuffer(("." * len(target)).encode(), dtype=np.uint8), (R, 1)).copy()
    rng = hk.rng_seed(np.arange(R))
    changed = 0
    for it in range(60):
        out = hk.propose(target, amask, cur, ss, np.zeros(R, dtype=np.int32), R, 0.7, 0.0, True, rng)
        for r in range(R):
            s = out[r].tobytes().decode()
            assert all(s[i] in p.allowed[i] for i in range(len(s)))
            for i, j in p.pairs:
                assert s[j] in design.CAN_PAIR[s[i]]
            d = sum(a != b for a, b in zip(s, cur[r].tobytes().decode()))
            assert d <= 2
            changed += d > 0
        cur = out
    assert changed > 0.8 * 60 * R
    # two replicas with the same stream state and the same inputs make the same move
    rng2 = hk.rng_seed([7, 7])
    o2 = hk.propose(target, amask, cur[:2] * 0 + cur[0], ss[:2], np.zeros(2, dtype=np.int32), R, 0.7, 0.0, True, rng2)
    assert (o2[0] == o2[1]).all()


def test_native_rng_is_cpythons_mersenne_twister(hk):
    """drna_rng_seed / drna_rng_random against CPython's random.seed(int) / random.random() (the reference's worker streams,
    utils/replica_exchange_monte_carlo.py:227-228), including a seed above 2^32 and the refill of the state after 624 words."""
    import numpy as np
    seeds = [0, 1, 2, 63, 2137, 2 ** 31, 2 ** 32 + 5, 2 ** 63 + 11]
    st = hk.rng_seed(seeds)
    pys = [random.Random(s) for s in seeds]
    for _ in range(700):
        got = hk.rng_random(st)
        assert list(got) == [p.random() for p in pys]


def test_native_proposals_match_recorded_reference_proposals(hk):
    """The 600 single-strand proposals recorded from the reference's mutate_sequence (tests/golden/host_golden.json, same
    vectors as tests/test_host_golden.py::test_proposals_same_position_and_draws): the native proposer seeded like the
    reference's worker must consume the same draws (next random() equal) and produce the same sequence, with the same
    exemptions as the Python mirror (second letter of a pair move / snake state order depend on str-set order)."""
    import json, os
    import numpy as np
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "host_golden.json")))
    n = n_exact = 0
    probs = {}
    for c in gold["proposals"]:
        if c.get("oligo_state", "none") != "none":
            continue                                  # two strands run through the Python driver
        d = gold["inputs"][c["input"]]
        key = c["input"]
        if key not in probs:
            probs[key] = design.DesignProblem(d["sec_struct"], d["seq_restr"], d["alt_sec_structs"])
        p = probs[key]
        st = hk.rng_seed([c["seed"]])
        cur = np.frombuffer(c["sequence"].encode(), dtype=np.uint8).reshape(1, -1).copy()
        ss = np.frombuffer(c["mfe_ss"].encode(), dtype=np.uint8).reshape(1, -1).copy()
        out = hk.propose_alt(p, cur, ss, np.array([c["shelf"]], dtype=np.int32), c["n_shelves"], 0.7, 0.0, True, st)
        mine = out[0].tobytes().decode()
        assert hk.rng_random(st)[0] == c["next_random"], c          # same number of draws consumed
        # and the same proposal as the Python mirror, letter for letter
        rr = random.Random(c["seed"])
        pos = p.mutation_position(c["mfe_ss"], c["shelf"], c["n_shelves"], 0.7, 0.0, True, rr)
        assert mine == p.mutate(c["sequence"], pos, rr), c
        n += 1
        n_exact += mine == c["proposed"]
    assert n == 600 and n_exact == n          # every recorded reference proposal is reproduced letter for letter


def test_native_metropolis_semantics(hk):
    import numpy as np
    rng = hk.rng_seed(np.arange(4))
    before = rng.copy()
    acc, better = hk.metropolis([1.0, 1.0, 1.0, 1.0], [0.5, 1.0, 1.0 + 1e-9, 50.0], [10.0, 10.0, 1e9, 10.0], rng)
    assert list(acc[:2]) == [True, True] and list(better) == [True, True, False, False]
    assert acc[2] and not acc[3]                     # tiny uphill at huge T is accepted, huge uphill is not
    assert (rng[:2] == before[:2]).all() and (rng[2:] != before[2:]).any(axis=1).all()   # a draw only when the mutant is worse


@pytest.mark.gpu
def test_fast_driver_solves_L200_on_gpu(eterna_targets):
    tg = eterna_targets["eteV1_69.txt"]
    inp = SimpleNamespace(name="ete69", sec_struct=tg, seq_restr="N" * len(tg), seed_seq=None, alt_sec_struct=None,
                          alt_sec_structs=None)
    res = design.run_design_fast(inp, replicas=64, exchange=100, steps=6, seed=1, stop_when_solved=True)
    assert res["solved"] and res["stats"]["scored"] >= 64


@pytest.mark.gpu
def test_fast_driver_long_target_strip_path_equals_general_path(eterna_targets):
    """A 400-nt puzzle (BASELINE config 5's target) through the native driver: the folds take the strip kernels (several
    workgroups per sequence); with "strips" off the general one-workgroup kernels.  Same seed -> the same records, because the
    kernels' answers are the same (MFE strings and energies bit for bit, Epf to 1e-13)."""
    from desirna_amd import engine as E
    tg = eterna_targets["eteV1_53.txt"]
    inp = SimpleNamespace(name="ete53", sec_struct=tg, seq_restr="N" * len(tg), seed_seq=None, alt_sec_struct=None,
                          alt_sec_structs=None)
    out = []
    for strips in (1, 0):
        eng = E.Engine(max_R=16, max_L=len(tg), device=0)
        eng.set_option("strips", strips)
        res = design.run_design_fast(inp, replicas=16, exchange=10, steps=3, seed=7, engine=eng)
        out.append(res)
        eng.close()
    a, b = out
    assert a["best"].sequence == b["best"].sequence and a["best"].mfe_ss == b["best"].mfe_ss
    assert a["stats"]["scored"] == b["stats"]["scored"] and a["stats"]["acc_mc"] == b["stats"]["acc_mc"]
    assert [(r["sequence"], r["mfe_ss"]) for r in a["simulation_data"]] == [(r["sequence"], r["mfe_ss"]) for r in b["simulation_data"]]


# ---- alternative structures: snakes (reference utils/sequence_utils.py:143-388, :1081-1095)

ALT_TARGET = "((((((.((((((((....))))).)).).))))))"
ALT_STRUCTS = ["(((((((((((((....)))..)).)).).))))).",
               "(((((((((((((....)))))...)).).))))).",
               ]


def _snake_invariants(p, s):
    for nodes, states in p.snakes:
        assert "".join(s[v] for v in nodes) in states
    for i, j in p.pairs:
        if p.snake_of[i] < 0:
            assert s[j] in design.CAN_PAIR[s[i]]
    # every pair of the target AND of every alternative structure can form (Watson-Crick inside snakes)
    for st in [ALT_TARGET] + ALT_STRUCTS:
        pt = design.pair_table(st)
        for i, j in enumerate(pt):
            if j > i:
                assert s[int(j)] in design.CAN_PAIR[s[i]], (st, i, j)


def test_native_snake_proposals(hk):
    import numpy as np
    p = design.DesignProblem(ALT_TARGET, "N" * len(ALT_TARGET), ALT_STRUCTS)
    assert len(p.snakes) == 2
    R = 16
    seq = p.initial_sequence(random.Random(3))
    _snake_invariants(p, seq)
    cur = np.tile(np.frombuffer(seq.encode(), dtype=np.uint8), (R, 1)).copy()
    ss = np.tile(np.frombuffer(("." * p.n).encode(), dtype=np.uint8), (R, 1)).copy()
    rng = hk.rng_seed(np.arange(R))
    moved_snake = 0
    for it in range(40):
        out = hk.propose_alt(p, cur, ss, np.zeros(R, dtype=np.int32), R, 0.7, 0.0, True, rng)
        for r in range(R):
            s = out[r].tobytes().decode()
            _snake_invariants(p, s)
            moved_snake += sum(a != b for a, b in zip(s, cur[r].tobytes().decode())) > 2
        cur = out
    assert moved_snake > 0
    # Python mirror keeps the same invariants
    rr = random.Random(5)
    s = seq
    for _ in range(200):
        pos = p.mutation_position("." * p.n, 0, 10, 0.7, 0.0, True, rr)
        s = p.mutate(s, pos, rr)
        _snake_invariants(p, s)


@pytest.mark.gpu
def test_alt_structure_design_on_gpu():
    """Alternative-structures example input through both drivers: score = (Ed - Epf) + (mean Ed_alt - Epf)."""
    inp = SimpleNamespace(name="alt", sec_struct=ALT_TARGET, seq_restr="N" * len(ALT_TARGET), seed_seq=None,
                          alt_sec_struct=ALT_STRUCTS[0], alt_sec_structs=ALT_STRUCTS)
    res = design.run_design_fast(inp, replicas=16, exchange=50, steps=4, seed=2)
    b = res["best"]
    p = design.DesignProblem(ALT_TARGET, inp.seq_restr, ALT_STRUCTS)
    _snake_invariants(p, b.sequence)
    res2 = design.run_design(inp, replicas=8, exchange=10, steps=2, seed=2)
    _snake_invariants(p, res2["best"].sequence)
    sc = res2["best"]
    assert abs(sc.scoring_function - (sc.edesired_minus_Epf + sc.edesired2_minus_Epf)) < 1e-9


def test_design_run_writes_reference_result_files(oracle, tmp_path):
    """A short run through the Python driver, then the reference's result files (formats pinned byte-for-byte in
    tests/test_host_golden.py): one record per replica per exchange step, header = vars(ScoreSeq) order."""
    from desirna_amd import outputs
    inp = SimpleNamespace(name="ete1", sec_struct=ETE1, seq_restr="N" * len(ETE1), seed_seq=None, alt_sec_struct=None,
                          alt_sec_structs=None)
    res = design.run_design(inp, replicas=4, exchange=10, steps=3, seed=3,
                            scorer=OracleScorer(oracle, ETE1, parse_scoring_functions("Ed-Epf:1.0")))
    assert len(res["simulation_data"]) == 4 * (3 + 1)
    st = res["stats"]
    stats = SimpleNamespace(step=st["acc_mc"] + st["rej_mc"], global_step=res["steps"], acc_mc_step=st["acc_mc"],
                            acc_mc_better_e=st["acc_mc_better"], rej_mc_step=st["rej_mc"], acc_re_step=st["acc_re"],
                            rej_re_step=st["rej_re"])
    top, ok = outputs.write_all(res["simulation_data"], "ete1", "ete1.txt", "out", stats, 1.0, 60, "NOW", directory=str(tmp_path))
    hdr = open(tmp_path / "out_traj.csv").readline().strip().split(",")
    assert hdr[:5] == ["sequence", "scoring_function", "replica_num", "temp_shelf", "sim_step"]
    assert ok == res["solved"] and top[0]["mcc"] == min(r["mcc"] for r in res["simulation_data"])
    assert open(tmp_path / "out_best_str").read().startswith(">ete1,%s," % ok)


@pytest.mark.gpu
def test_native_mc_loop_equals_python_loop_on_gpu(eterna_targets):
    """drna_mc_run (the whole inner loop of an exchange step in native code) replays exactly what the per-iteration Python loop
    over the same native helpers does: same random streams, same accept decisions, same best sequence."""
    tg = eterna_targets["eteV1_92.txt"]
    inp = SimpleNamespace(name="ete92", sec_struct=tg, seq_restr="N" * len(tg), seed_seq=None, alt_sec_struct=None,
                          alt_sec_structs=None)
    a = design.run_design_fast(inp, replicas=16, exchange=25, steps=3, seed=7, native_loop=True, scoring_f="Ed-Epf:0.9")
    b = design.run_design_fast(inp, replicas=16, exchange=25, steps=3, seed=7, native_loop=False, scoring_f="Ed-Epf:0.9")
    assert a["best"].sequence == b["best"].sequence and a["best"].scoring_function == b["best"].scoring_function
    for k in ("acc_mc", "acc_mc_better", "rej_mc", "acc_re", "rej_re", "scored"):
        assert a["stats"][k] == b["stats"][k], k
    assert [r["sequence"] for r in a["simulation_data"]] == [r["sequence"] for r in b["simulation_data"]]
    # ... and what the per-replica Python driver (random.Random(replica) streams, ReplicaScorer) does for the same seed
    c = design.run_design(inp, replicas=16, exchange=25, steps=3, seed=7, scoring_f="Ed-Epf:0.9")
    assert [r["sequence"] for r in a["simulation_data"]] == [r["sequence"] for r in c["simulation_data"]]
    assert [r["temp_shelf"] for r in a["simulation_data"]] == [r["temp_shelf"] for r in c["simulation_data"]]
    for k in ("acc_mc", "acc_mc_better", "rej_mc", "acc_re", "rej_re", "scored"):
        assert a["stats"][k] == c["stats"][k], k


@pytest.mark.gpu
def test_native_mc_loop_with_ensemble_defect_term(eterna_targets):
    """-sf with an Edef term (reference utils/energy_scores.py:362-374,397-398) stays on the native loop (term 6 of drna_mc_run:
    inside + outside recursion of every proposal) and walks the same trajectory as the per-iteration Python loop."""
    tg = "((((((.((((((((....))))).)).).))))))"
    inp = SimpleNamespace(name="edef", sec_struct=tg, seq_restr="N" * len(tg), seed_seq=None, alt_sec_struct=None,
                          alt_sec_structs=None)
    sf = "Ed-Epf:0.5,Edef:1.0"
    a = design.run_design_fast(inp, replicas=8, exchange=10, steps=3, seed=5, native_loop=True, scoring_f=sf)
    b = design.run_design_fast(inp, replicas=8, exchange=10, steps=3, seed=5, native_loop=False, scoring_f=sf)
    assert a["used_native_loop"] and not b["used_native_loop"]
    assert [r["sequence"] for r in a["simulation_data"]] == [r["sequence"] for r in b["simulation_data"]]
    assert [r["scoring_function"] for r in a["simulation_data"]] == [r["scoring_function"] for r in b["simulation_data"]]
    for k in ("acc_mc", "acc_mc_better", "rej_mc", "acc_re", "rej_re", "scored"):
        assert a["stats"][k] == b["stats"][k], k


@pytest.mark.gpu
def test_avoid_oligomerization_design_run_builds_its_own_engines():
    """-oa on through run_design WITHOUT an injected engine: every candidate is also folded against itself (s & s, 2 L
    nucleotides), which needs an engine sized for 2 L (reference utils/energy_scores.py:411-418)."""
    tg = "((((((.((((((((....))))).)).).))))))"
    inp = SimpleNamespace(name="oa", sec_struct=tg, seq_restr="N" * len(tg), seed_seq=None, alt_sec_struct=None,
                          alt_sec_structs=None)
    res = design.run_design(inp, replicas=4, exchange=5, steps=2, seed=4, oligo="on")
    assert res["stats"]["scored"] == 4 + 2 * 5 * 4
    for r in res["simulation_data"]:
        assert 0.0 <= r["oligo_fraction"] <= 1.0
