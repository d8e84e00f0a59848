#!/usr/bin/env python3
"""Generate golden vectors for the HOST rows of the hot path (a3, a4, a5, a12-a14 helpers) by
importing the reference's pure-Python modules and recording inputs -> outputs.  Runs only in the build
container (the reference does not travel); the JSON it writes is data, not code.

    PYTHONHASHSEED=0 python tools/make_host_golden.py /root/reference tests/golden/host_golden.json

ViennaRNA is absent here, so a stand-in ``RNA`` module (no arithmetic: every call raises) is put on the path
only to let ``import RNA`` at the top of the reference modules succeed; ``score_sequence`` is replaced by a
recorder so that ``mutate_sequence`` returns the proposed string.  Nothing numeric comes from the stand-in.

Recorded:
  rep_temps        get_rep_temps for several (R, Tmin, Tmax)                     utils/sequence_utils.py:811-860
  mc_delta         (score_o, score_m, T, seed) -> accept, better, draws consumed   utils/replica_exchange_monte_carlo.py:26-57
  exchange         (temps, scores, global_step, seed) -> new temps, acc, rej       :113-173
  inputs           read_input on the six example inputs                            utils/stats_inputs_outputs.py:183-214
  problems         per example input: pairs, letters_allowed (sorted), snake graphs/states, excluded alt pairs
                                                                                    utils/sequence_utils.py:120-525
  initial          initial_sequence_generator with random.seed(k)                  :686-763
  outputs          the reference's writers (_traj.csv, _multifasta.fas, _best_fasta.fas, _results.csv, _best_str, _stats text,
                   get_outname) on a synthetic simulation_data list               utils/stats_inputs_outputs.py:304-669
  proposals        mutate_sequence with random.seed(k) on fixed (sequence, mfe_ss, temp_shelf): proposed sequence
                   and the value of random.random() right after (pins the number of draws)   :926-1136
"""
import json
import os
import random
import sys
import tempfile
import types


def main():
    ref, out = sys.argv[1], sys.argv[2]
    stub_dir = tempfile.mkdtemp()
    with open(os.path.join(stub_dir, "RNA.py"), "w") as fh:
        fh.write("class md:\n    def __init__(self):\n        self.compute_bpp = 0\n"
                 "def params_load(*a, **k):\n    pass\n"
                 "def fold_compound(*a, **k):\n    raise RuntimeError('stand-in')\n"
                 "def fold(*a, **k):\n    raise RuntimeError('stand-in')\n")
    sys.path.insert(0, stub_dir)
    sys.path.insert(0, ref)
    from utils import replica_exchange_monte_carlo as remc
    from utils import sequence_utils as su
    from utils import stats_inputs_outputs as sio
    from utils import energy_scores as es

    G = {"hashseed": os.environ.get("PYTHONHASHSEED", "unset")}

    # ---- temperature ladder
    G["rep_temps"] = []
    for R, tmin, tmax in ((1, 10, 150), (2, 10, 150), (10, 10, 150), (64, 10, 150), (32, 5.5, 90.25), (128, 10, 150)):
        o = types.SimpleNamespace(replicas=R, T_min=tmin, T_max=tmax)
        G["rep_temps"].append({"R": R, "T_min": tmin, "T_max": tmax, "temps": su.get_rep_temps(o)})

    # ---- Metropolis
    opt = types.SimpleNamespace(L=504.12)
    G["mc_delta"] = []
    rng = random.Random(11)
    for k in range(60):
        so = round(rng.uniform(-5, 20), 3)
        sm = so + rng.choice([-1.5, -0.01, 0.0, 0.005, 0.05, 0.3, 1.0, 4.0])
        T = rng.choice([10.0, 12.222, 47.778, 150.0])
        random.seed(k)
        acc, better = remc.mc_delta(so, sm, T, opt)
        G["mc_delta"].append({"score_o": so, "score_m": sm, "T": T, "seed": k, "accept": bool(acc), "better": bool(better),
                              "next_random": random.random()})

    # ---- replica exchange
    class Stats:
        def __init__(self, step):
            self.global_step = step
            self.acc = self.accb = self.rej = 0

        def update_acc_re_step(self):
            self.acc += 1

        def update_acc_re_better_e(self):
            self.accb += 1

        def update_rej_re_step(self):
            self.rej += 1

    G["exchange"] = []
    for case, (R, step, seed) in enumerate(((4, 1, 0), (4, 2, 1), (10, 3, 2), (10, 4, 3), (7, 5, 4), (7, 6, 5), (64, 7, 6), (64, 8, 7))):
        temps = su.get_rep_temps(types.SimpleNamespace(replicas=R, T_min=10, T_max=150))
        rng = random.Random(100 + case)
        shelf = temps[:]
        rng.shuffle(shelf)
        scores = [round(rng.uniform(0, 12), 3) for _ in range(R)]
        objs = []
        for r in range(R):
            s = es.ScoreSeq(sequence="A")
            s.get_replica_num(r + 1)
            s.get_temp_shelf(shelf[r])
            s.scoring_function = scores[r]
            objs.append(s)
        st = Stats(step)
        random.seed(seed)
        res, st = remc.replica_exchange(objs, st, opt)
        G["exchange"].append({"temps": shelf, "scores": scores, "global_step": step, "seed": seed,
                              "new_temps": [o.temp_shelf for o in res], "acc": st.acc, "acc_better": st.accb, "rej": st.rej,
                              "next_random": random.random()})

    # ---- example inputs: parsing, design problem, initial sequence, proposals
    recorded = {}

    def fake_score(seq, input_file, sim_options):
        recorded["seq"] = seq
        return es.ScoreSeq(sequence=seq)

    es.score_sequence = fake_score
    G["inputs"], G["problems"], G["initial"], G["proposals"] = {}, {}, [], []
    in_dir = os.path.join(ref, "example_files", "inputs")
    for fn in sorted(os.listdir(in_dir)):
        name = fn[:-4]
        inp = sio.read_input(os.path.join(in_dir, fn))
        G["inputs"][name] = {"name": inp.name, "sec_struct": inp.sec_struct, "seq_restr": inp.seq_restr,
                             "seed_seq": inp.seed_seq, "alt_sec_structs": inp.alt_sec_structs}
        two = "&" in inp.sec_struct
        # two-strand inputs: the RNA-RNA complex example is a hetero-dimer, the homodimer example runs with -d on
        oligo_state = "none" if not two else ("homodimer" if "omodimer" in name else "heterodimer")
        inp.pairs = su.check_dot_bracket(inp.sec_struct)
        inp.set_target_pairs_tupl()
        if inp.alt_sec_structs is not None:
            alt_pairs = su.get_pairs_for_graphs(inp)
            inp.graphs = su.generate_graphs(alt_pairs)
            su.update_graphs(inp)
        nt_list = su.get_nt_list(inp)
        su.check_input_logic(nt_list)
        G["problems"][name] = {
            "pairs": sorted([sorted(p) for p in inp.pairs]),
            "target_pairs": sorted([sorted(p) for p in inp.target_pairs_tupl]),
            "letters_allowed": ["".join(sorted(nt.letters_allowed)) for nt in nt_list],
            "pairs_with": [(-1 if nt.pairs_with is None else nt.pairs_with) for nt in nt_list],
            "snake": [bool(nt.snake) for nt in nt_list],
            "graphs": ([{"numbers": g["numbers"], "states": ["".join(s) for s in g["states"]]} for g in inp.graphs]
                       if inp.graphs is not None else None),
            "excluded_alt_pairs": (sorted([sorted(p) for p in inp.excluded_alt_pairs])
                                   if inp.excluded_alt_pairs is not None else None),
        }
        R = 10
        temps = su.get_rep_temps(types.SimpleNamespace(replicas=R, T_min=10, T_max=150))
        opts = types.SimpleNamespace(acgu_percentages="off", point_mutations="on", tm_max=0.7, tm_min=0.0,
                                     rep_temps_shelfs=temps, oligo_state=oligo_state, pks="off")
        for k in range(6):
            random.seed(k)
            init = su.initial_sequence_generator(nt_list, inp, opts)
            G["initial"].append({"input": name, "seed": k, "sequence": init, "next_random": random.random()})
        # proposals: current MFE structure = the target with its first helix opened (false negatives) and, in a second
        # variant, the exact target (no false cases)
        tgt = inp.sec_struct
        only = "".join(c if c in "().&" else "." for c in tgt)
        broken = list(only)
        first_open = only.find("(")
        if first_open >= 0:
            partner = dict((a, b) for a, b in (sorted(p) for p in su.check_dot_bracket(only)))
            broken[first_open] = "."
            broken[partner[first_open]] = "."
        variants = {"broken": "".join(broken), "exact": tgt}
        random.seed(12345)
        base_seq = su.initial_sequence_generator(nt_list, inp, opts)
        for vname, ss in variants.items():
            for shelf in (0, 4, 9):
                for k in range(25):
                    so = es.ScoreSeq(sequence=base_seq)
                    so.get_mfe_ss(ss)
                    so.get_temp_shelf(temps[shelf])
                    so.get_replica_num(1)
                    random.seed(1000 * shelf + k)
                    su.mutate_sequence(so, nt_list, opts, inp)
                    G["proposals"].append({"input": name, "variant": vname, "oligo_state": oligo_state, "sequence": base_seq, "mfe_ss": ss,
                                           "shelf": shelf, "n_shelves": R, "seed": 1000 * shelf + k,
                                           "proposed": recorded["seq"], "next_random": random.random()})
    # ---- output files: reference writers run on a synthetic simulation_data list (records = vars(ScoreSeq))
    rng = random.Random(77)
    sim = []
    structs = ["((((....))))", "(((......)))", "............", "((((....))))"]
    for k in range(14):
        sc = es.ScoreSeq(sequence="".join(rng.choice("ACGU") for _ in range(12)) if k % 5 else "GGGGAAAACCCC")
        sc.get_replica_num(k % 4 + 1)
        sc.get_temp_shelf(round(10 + 46.667 * (k % 4), 3))
        sc.get_sim_step(k // 4)
        sc.get_Epf(-rng.uniform(1, 9))
        sc.get_mfe_ss(structs[k % 4])
        sc.get_edesired(-rng.uniform(0, 8))
        sc.get_edesired_minus_Epf(sc.Epf, sc.edesired)
        sc.get_mcc(1.0 if k % 3 == 0 else round(rng.uniform(0, 1), 3))
        sc.get_precision(round(rng.uniform(0, 1), 3))
        sc.get_recall(round(rng.uniform(0, 1), 3))
        sc.scoring_function = sc.edesired_minus_Epf
        sim.append(vars(sc))
    opts = types.SimpleNamespace(oligo="off", dimer="off", subopt="off", num_results=10, infile="toy.txt", outname="toyout", timlim=60)
    inp0 = types.SimpleNamespace(name="Toy", sec_struct="((((....))))", graphs=None)
    st = sio.Stats()
    st.global_step, st.step, st.acc_mc_step, st.acc_mc_better_e, st.rej_mc_step, st.acc_re_step, st.rej_re_step = 3, 300, 211, 150, 89, 4, 5
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)
    try:
        import copy
        data = copy.deepcopy(sim)
        traj = sio.sort_trajectory(data)
        sio.generate_trajectory_csv(traj, opts.outname)
        sio.generate_multifasta(traj, opts, "NOW")
        sio.generate_best_fasta(data, opts, "NOW")
        res = sio.sort_and_filter_simulation_data(data, opts, inp0)
        sio.generate_csv_from_data(res, opts.outname)
        txt, ok = sio.check_if_design_solved(res[:10], inp0, opts)
        stats_txt = sio.generate_simulation_stats_text(st, res, ok, 83.4, opts)
        files = {}
        for suffix in ("_traj.csv", "_multifasta.fas", "_best_fasta.fas", "_results.csv"):
            with open(opts.outname + suffix, newline="") as fh:
                files[suffix] = fh.read()
    finally:
        os.chdir(cwd)
    o2 = types.SimpleNamespace(infile="Standard_design_input.txt", replicas=10, RE_attempt=100, timlim=60, pks="off",
                               acgu_percentages="off", T_min=10, T_max=150, param="1999", scoring_f=[("Ed-Epf", 1.0)],
                               oligo="off", dimer="off", point_mutations="on")
    G["outputs"] = {"simulation_data": [[[k, v] for k, v in r.items()] for r in sim],   # key order = vars(ScoreSeq) = CSV header
                     "files": files, "best_str": txt, "solved": bool(ok), "stats_txt": stats_txt,
                    "stats": {"global_step": 3, "step": 300, "acc_mc_step": 211, "acc_mc_better_e": 150, "rej_mc_step": 89,
                              "acc_re_step": 4, "rej_re_step": 5}, "finish_time": 83.4,
                    "outname_case": {k: getattr(o2, k) for k in vars(o2)}, "outname": sio.get_outname(o2)}
    # ---- -acgu on (weighted letter choices, default content A15 C30 G30 U15): initial sequences and proposals on the
    # standard example input
    G["acgu"] = {"percentages": {"A": 15, "C": 30, "G": 30, "U": 15}, "initial": [], "proposals": []}
    inp = sio.read_input(os.path.join(in_dir, "Standard_design_input.txt"))
    inp.pairs = su.check_dot_bracket(inp.sec_struct)
    inp.set_target_pairs_tupl()
    nt_list = su.get_nt_list(inp)
    temps = su.get_rep_temps(types.SimpleNamespace(replicas=10, T_min=10, T_max=150))
    opts = types.SimpleNamespace(acgu_percentages="on", nt_percentages=G["acgu"]["percentages"], point_mutations="on", tm_max=0.7,
                                 tm_min=0.0, rep_temps_shelfs=temps, oligo_state="none", pks="off")
    for k in range(8):
        random.seed(k)
        init = su.initial_sequence_generator(nt_list, inp, opts)
        G["acgu"]["initial"].append({"seed": k, "sequence": init, "next_random": random.random()})
    random.seed(4242)
    base_seq = su.initial_sequence_generator(nt_list, inp, opts)
    for k in range(60):
        so = es.ScoreSeq(sequence=base_seq)
        so.get_mfe_ss("." * len(base_seq))
        so.get_temp_shelf(temps[k % 10])
        so.get_replica_num(1)
        random.seed(7000 + k)
        su.mutate_sequence(so, nt_list, opts, inp)
        G["acgu"]["proposals"].append({"sequence": base_seq, "mfe_ss": "." * len(base_seq), "shelf": k % 10, "n_shelves": 10,
                                       "seed": 7000 + k, "proposed": recorded["seq"], "next_random": random.random()})
    with open(out, "w") as fh:
        json.dump(G, fh, indent=0, sort_keys=True)
    print({k: (len(v) if hasattr(v, "__len__") else v) for k, v in G.items()})


if __name__ == "__main__":
    main()
