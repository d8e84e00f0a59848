"""Batched replica-exchange Monte-Carlo design driver on top of the GPU scoring engine.

Counterpart of the reference's host loop: ``run_functions`` (``DesiRNA.py:333-386``),
``mutate_sequence_re`` / ``single_replica_design`` (``utils/replica_exchange_monte_carlo.py:176-271``),
``mutate_sequence`` / ``get_mutation_position`` / ``expand_cases`` (``utils/sequence_utils.py:926-1136``),
``get_nt_list`` (:454-525), ``initial_sequence_generator`` (:686-763) and ``read_input``
(``utils/stats_inputs_outputs.py:183-214``).  SURVEY rows a1, a2, a5 (8(f)-1).

What is kept from the reference: the input-file format, IUPAC sequence restraints, the move set (single
unpaired position / Watson-Crick-or-GU pair move on target pairs), targeted mutations around false
negatives / false positives of the current MFE structure (+-3 window, per-temperature probability
``linspace(tm_max, tm_min, R)`` rounded to 2 dp), the initial-sequence rule, Metropolis acceptance and the
even/odd neighbour exchange, one ``random.Random(replica_index)`` stream per replica re-seeded at every
exchange step.

What is different by design: the R chains advance in LOCK-STEP -- one proposal per replica, one batched
``score`` call for all of them, R Metropolis decisions -- instead of R forked Python workers.  Trajectories
are statistically equivalent, not draw-for-draw identical: the reference itself is not reproducible across
processes (it iterates over ``set`` objects of strings, whose order depends on PYTHONHASHSEED).

Alternative structures: "snake" moves (connected components of the pair graph of target + alternative structures
switch between their Watson-Crick colourings) as in the reference (:143-388, :1081-1095).

``-acgu on`` (weighted letter choices) is available in the Python driver (``DesignProblem(acgu=...)``).

Not supported here (raises): ``-nd``.
"""
import argparse
import random
import time
from types import SimpleNamespace

import numpy as np

from . import energy_scores as es
from . import replica_exchange as rx
from .sim_score import pair_table

IUPAC = {
    'N': ['A', 'C', 'G', 'U'], 'W': ['A', 'U'], 'S': ['C', 'G'], 'M': ['A', 'C'], 'K': ['G', 'U'],
    'R': ['A', 'G'], 'Y': ['C', 'U'], 'B': ['C', 'G', 'U'], 'D': ['A', 'G', 'U'], 'H': ['A', 'C', 'U'],
    'V': ['A', 'C', 'G'], 'C': ['C'], 'A': ['A'], 'G': ['G'], 'U': ['U'], '&': ['&'],
}
CAN_PAIR = {'A': ['U'], 'U': ['G', 'A'], 'G': ['U', 'C'], 'C': ['G']}
WC = {'A': 'U', 'U': 'A', 'G': 'C', 'C': 'G'}


def read_input(path):
    """``>key`` block format of the reference (name, seq_restr, sec_struct, optional seed_seq / alt_sec_struct)."""
    with open(path, encoding='utf-8') as fh:
        text = fh.read()
    data = {}
    for block in text.lstrip(">").rstrip("\n").split("\n>"):
        lines = block.split("\n")
        data[lines[0]] = lines[1:]
    inp = SimpleNamespace(name=data['name'][0], sec_struct=data['sec_struct'][0].strip(),
                          seq_restr=data['seq_restr'][0].strip(), seed_seq=None, alt_sec_struct=None, alt_sec_structs=None)
    if 'seed_seq' in data:
        inp.seed_seq = data['seed_seq'][0].strip()
    if 'alt_sec_struct' in data:
        inp.alt_sec_structs = [x.strip() for x in data['alt_sec_struct']]
        inp.alt_sec_struct = inp.alt_sec_structs[0]
    return inp


class DesignProblem:
    """Target structure + restraints -> per-position letter sets and pairing partners (reference get_nt_list)."""

    def __init__(self, sec_struct, seq_restr=None, alt_sec_structs=None, acgu=None):
        # acgu: None (= -acgu off) or the nucleotide percentages of -acgu on, e.g. {'A': 15, 'C': 30, 'G': 30, 'U': 15}:
        # weighted letter choices in the initial sequence and in pair moves (reference :725-741, :1052-1064)
        self.acgu = acgu
        # two strands: the '&' stays in every string as a fixed, unpaired "letter" exactly as in the reference (its positions
        # count; seq_restr carries it too), so indices, windows and draw counts are the reference's
        self.two_strands = "&" in sec_struct
        self.sec_struct = sec_struct
        n = len(sec_struct)
        self.seq_restr = seq_restr or "N" * n
        if len(self.seq_restr) != n:
            raise ValueError("Secondary structure and sequence restraints are of different length. Check input file.")
        self.partner = pair_table(sec_struct).copy()          # all bracket families count as design pairs
        self.target_pairs = {(i, int(p)) for i, p in enumerate(self.partner) if p > i}
        self.pairs = set(self.target_pairs)
        self.snakes = []                                      # [(positions, [state strings])]
        self.snake_of = [-1] * n
        if alt_sec_structs:
            self._build_snakes(alt_sec_structs)
        letters = []
        for ch in self.seq_restr:
            if ch not in IUPAC:
                raise ValueError("Not allowed characters in sequence restraints. Check input file.")
            letters.append(list(IUPAC[ch]))
        allowed = [list(l) for l in letters]
        for i, j in self.pairs:
            pi = sorted({b for a in letters[j] for b in CAN_PAIR[a]} & set(letters[i]))
            pj = sorted({b for a in letters[i] for b in CAN_PAIR[a]} & set(letters[j]))
            if not pi or not pj:
                raise ValueError("Wrong restraints in the input file. Nucleotide %d %s, cannot pair with nucleotide %d %s"
                                 % (i + 1, letters[i], j + 1, letters[j]))
            allowed[i], allowed[j] = pi, pj
        self.letters, self.allowed = letters, allowed
        self.mutable = [i for i in range(n) if len(allowed[i]) != 1]
        self.n = n
        for k, (nodes, states) in enumerate(self.snakes):
            ok = [st for st in states if all(st[x] in letters[v] for x, v in enumerate(nodes))]
            if not ok:
                raise ValueError("The structural restraints are contradictive to sequence restraints. Check your input.")
            self.snakes[k] = (nodes, ok)

    def _build_snakes(self, alt_sec_structs):
        """Alternative structures (reference get_pairs_for_graphs / generate_graphs / update_graphs,
        utils/sequence_utils.py:143-388): the pairs of all structures form a graph; a pair with an end that
        also sits in another pair belongs to a "snake" (a connected component that must be two-coloured with one
        Watson-Crick letter pair: A/U, U/A, C/G or G/C); the remaining alternative pairs become ordinary design
        pairs.  State order follows the reference: colour 0 goes to the component's first vertex in the
        reference's traversal; both colourings are in the list, so only the order depends on it."""
        n = len(self.sec_struct)
        alt = set()
        for a in alt_sec_structs:
            if len(a) != n:
                raise ValueError("alternative structure and target are of different length")
            pt = pair_table(a)
            alt |= {(i, int(p)) for i, p in enumerate(pt) if p > i}
        merged = self.target_pairs | alt
        count = [0] * n
        for a, b in merged:
            count[a] += 1
            count[b] += 1
        graph_pairs = sorted(p for p in merged if count[p[0]] > 1 or count[p[1]] > 1)
        excluded = sorted(p for p in merged if not (count[p[0]] > 1 or count[p[1]] > 1))
        for a, b in excluded:                                 # ordinary pairs (the target's own ones are already there)
            self.pairs.add((a, b))
            self.partner[a], self.partner[b] = b, a
        adj = {}
        for a, b in graph_pairs:
            adj.setdefault(a, []).append(b)
            adj.setdefault(b, []).append(a)
        seen = set()
        comps = []
        for v in sorted(adj):
            if v in seen:
                continue
            colour = {v: 0}
            queue = [v]
            while queue:
                x = queue.pop(0)
                for y in adj[x]:
                    if y not in colour:
                        colour[y] = 1 - colour[x]
                        queue.append(y)
                    elif colour[y] == colour[x]:
                        raise ValueError("The structural restraints in the alternative structures cannot be solved. "
                                         "Check your input.")
            seen |= set(colour)
            nodes = sorted(colour)
            cols = [colour[x] for x in nodes]
            states = ["".join(m[c] for c in cols) for m in ("AU", "UA", "CG", "GC")]
            comps.append((nodes, states))
        comps.sort(key=lambda c: c[0][0])
        self.snakes = comps
        for k, (nodes, _) in enumerate(comps):
            for v in nodes:
                self.snake_of[v] = k

    def initial_sequence(self, rng):
        """reference initial_sequence_generator with -acgu off: unpaired -> A, first base of an unpaired stretch (length
        >= 2) after a paired base -> G (else U), pairs -> random G/C (else A/U) Watson-Crick pair, rest random."""
        n, part = self.n, self.partner
        s = list(self.seq_restr)
        for i in range(n):
            if part[i] < 0 and "A" in self.letters[i]:
                s[i] = "A"
        for i in range(1, n - 1):
            if part[i] < 0 and part[i - 1] >= 0 and part[i + 1] < 0:
                if "G" in self.letters[i]:
                    s[i] = "G"
                elif "U" in self.letters[i]:
                    s[i] = "U"
        for i, j in sorted(self.pairs):
            a, b = self.allowed[i], self.allowed[j]
            if self.acgu is not None:
                opts_i = sorted(a)
                s[i] = rng.choices(opts_i, weights=[self.acgu[x] for x in opts_i])[0]
                s[j] = WC[s[i]]
                continue
            if "C" in a and "G" in a and "C" in b and "G" in b:
                s[i] = rng.choice(["C", "G"]); s[j] = WC[s[i]]
            elif "C" in a and "G" in b:
                s[i], s[j] = "C", "G"
            elif "G" in a and "C" in b:
                s[i], s[j] = "G", "C"
            elif "A" in a and "U" in a and "A" in b and "U" in b:
                s[i] = rng.choice(["A", "U"]); s[j] = WC[s[i]]
            elif "A" in a and "U" in b:
                s[i], s[j] = "A", "U"
            elif "U" in a and "A" in b:
                s[i], s[j] = "U", "A"
            else:
                # restraints that only leave G-U wobble pairs (the reference falls through to unconstrained
                # random letters here): take any compatible pair
                s[i] = rng.choice(a)
                s[j] = rng.choice(sorted(set(b) & set(CAN_PAIR[s[i]])) or b)
        for i in range(n):
            if s[i] not in "ACGU":
                s[i] = rng.choice(IUPAC[s[i]])
        for nodes, states in self.snakes:                     # reference :750-760: every snake starts in its first state
            for x, v in enumerate(nodes):
                s[v] = states[0][x]
        return "".join(s)

    # ---- move set
    def mutation_position(self, mfe_ss, shelf_index, n_shelves, tm_max, tm_min, point_mutations, rng):
        if not point_mutations:
            return rng.choice(self.mutable)
        q = pair_table(mfe_ss)
        query = {(i, int(p)) for i, p in enumerate(q) if p > i}
        mutable = set(self.mutable)
        false_cases = {x for pr in (self.target_pairs - query) | (query - self.target_pairs) for x in pr if x in mutable}
        if not false_cases:
            return rng.choice(self.mutable)
        prob = round(float(np.linspace(tm_max, tm_min, num=n_shelves)[shelf_index]), 2)
        expanded = sorted({c + k for c in false_cases for k in range(-3, 4) if 0 < c + k <= self.n - 1})
        pool = rng.choices([expanded, self.mutable], weights=[prob, 1 - prob])[0]
        return rng.choice(pool)

    def mutate(self, seq, pos, rng, oligo_state="none"):
        """One move at `pos`; for homodimers the reference's strand-copy rules are applied afterwards (:1104-1126)."""
        out = self._mutate(seq, pos, rng)
        if oligo_state == "homodimer":
            j = int(self.partner[pos])
            ss1, ss2 = self.sec_struct.split("&")
            s1o, s2o = seq.split("&")
            s1m, s2m = out.split("&")
            if j >= 0 and self.snake_of[pos] < 0 and ss1 != ss2:
                d = sorted([pos, j])
                a, b = d[0], d[1] - len(s1o) - 1
                s1c = s1m[:b] + s2m[b] + s1m[b + 1:]
                s2c = s2m[:a] + s1m[a] + s2m[a + 1:]
                out = s1c + "&" + s2c
                s1m, s2m = s1c, s2c
            if ss1 == ss2:
                if s1m != s1o:
                    out = s1m + "&" + s1m
                elif s2m != s2o:
                    out = s2m + "&" + s2m
        return out

    def _mutate(self, seq, pos, rng):
        s = list(seq)
        j = int(self.partner[pos])
        if self.snake_of[pos] >= 0:                           # reference :1081-1095: move the whole snake to another state
            nodes, states = self.snakes[self.snake_of[pos]]
            x = nodes.index(pos)
            cur = next((st for st in states if st[x] == s[pos]), None)
            new_states = [st for st in states if st != cur]
            if new_states:
                st = rng.choice(new_states)
                for y, v in enumerate(nodes):
                    s[v] = st[y]
        elif j < 0:
            opts = [x for x in self.allowed[pos] if x != s[pos]] if len(self.allowed[pos]) > 1 else []
            if opts:
                s[pos] = rng.choice(opts)
        else:
            opts1 = sorted(x for x in self.allowed[pos] if x != s[pos]) if len(self.allowed[pos]) != 1 else list(self.allowed[pos])
            if self.acgu is not None:
                n1 = rng.choices(opts1, weights=[self.acgu[x] for x in opts1])[0]
            else:
                n1 = rng.choice(opts1)
            opts2 = sorted(set(self.allowed[j]) & set(CAN_PAIR[n1]))
            if opts2:
                if self.acgu is not None:
                    s[pos], s[j] = n1, rng.choices(opts2, weights=[self.acgu[x] for x in opts2])[0]
                else:
                    s[pos], s[j] = n1, rng.choice(opts2)
        return "".join(s)


def _sws_reached(simulation_data, num_results, oligo_state):
    """The reference's -sws test (utils/stats_inputs_outputs.py:240-264, run every 10th exchange step): records
    de-duplicated by sequence, sorted like the result file, cut to num_results -- stop when all of them are solved."""
    from . import outputs
    top = outputs.sort_and_filter(simulation_data, num_results, oligo_state=oligo_state)
    return len(top) == num_results and all(r["mcc"] == 0.0 for r in top)


def run_design(input_file, replicas=10, exchange=100, steps=None, timelimit=60, t_min=10.0, t_max=150.0,
               scoring_f="Ed-Epf:1.0", tm_max=0.7, tm_min=0.0, point_mutations="on", seed=0, stop_when_solved=False,
               device=0, shards=None, scorer=None, progress=None, dimer="off", oligo="off", acgu=None, subopt="off",
               num_results=None):
    """Replica-exchange Monte-Carlo design of one target.  Returns dict(best=ScoreSeq, solved=bool, history=..., stats=...).

    ``shards`` (a ``replica_exchange.ReplicaShards``) splits the replicas over ranks; every rank proposes and scores its
    own replicas; ONE all-gather per exchange step carries the scores (and the ranks' solved / time-is-up flags), then every
    rank replays the same swaps on the whole temperature ladder, which it tracks itself.  Stop decisions are collective:
    they are taken from the gathered flags (rank 0's clock decides the time limit), so no rank leaves the loop alone.

    Loop semantics of the reference (``DesiRNA.py:361-383``): the time limit holds also when ``steps`` is given; records
    carry ``sim_step = global_step * exchange``; ``stop_when_solved`` with ``num_results`` is the reference's ``-sws on``
    (every 10th step: stop when the best ``num_results`` distinct sequences are all solved), without it the run stops at the
    first solved replica (an extension used by the tests and by the puzzle-set driver)."""
    prob = DesignProblem(input_file.sec_struct, input_file.seq_restr, input_file.alt_sec_structs, acgu=acgu)
    pks = "on" if set(input_file.sec_struct) - set(".()&") else "off"
    if prob.two_strands:
        oligo_state = "homodimer" if dimer == "on" else "heterodimer"       # reference DesiRNA.py:474-485
    else:
        oligo_state = "avoid" if oligo == "on" else "none"
    opts = SimpleNamespace(oligo_state=oligo_state, pks=pks, subopt=subopt, motifs=None, param="1999",
                           scoring_f=es.parse_scoring_functions(scoring_f))
    shards = shards or rx.ReplicaShards(replicas, 0, 1)
    local = shards.local
    scorer = scorer or es.ReplicaScorer(input_file, opts, max_replicas=max(1, len(local)), device=device)
    temps = rx.get_rep_temps(replicas, t_min, t_max)           # temperature shelf of EVERY replica, replayed on every rank
    shelves = list(temps)
    # main stream (initial sequence, swap acceptance; DesiRNA.py:659-664): the same on every rank
    main_rng = random.Random(shards.broadcast_seed(2137 + seed if seed else random.random()))
    init = prob.initial_sequence(main_rng)
    # input_file.seed_seq is read but ignored, as in the reference
    cur = scorer.score([init] * len(local)) if local else []
    for k, r in enumerate(local):
        cur[k].get_replica_num(r + 1)
        cur[k].get_temp_shelf(temps[r])
    best = min(cur, key=lambda s: (s.mcc, s.scoring_function)) if cur else None
    simulation_data = [dict(vars(s)) for s in cur]            # reference DesiRNA.py:353: one record per replica and exchange step
    stats = dict(acc_mc=0, acc_mc_better=0, rej_mc=0, acc_re=0, rej_re=0, scored=len(local))
    t_start = time.time()
    global_step = 0
    solved = best is not None and best.mcc == 0.0
    if shards.world > 1:                                      # a solved start must be known to every rank before the loop
        _, ex = shards.allgather_scores([s.scoring_function for s in cur], extras=[float(solved)])
        solved = bool(ex[:, 0].any())
    stop = stop_when_solved and solved and num_results is None
    while not stop:
        if steps is not None and global_step >= steps:
            break
        global_step += 1
        rngs = [random.Random(r) for r in local]              # re-seeded with the replica index every exchange step
        for _ in range(exchange):
            props = []
            for k, r in enumerate(local):
                shelf = shelves.index(cur[k].temp_shelf)
                pos = prob.mutation_position(cur[k].mfe_ss, shelf, replicas, tm_max, tm_min, point_mutations == "on", rngs[k])
                props.append(prob.mutate(cur[k].sequence, pos, rngs[k], oligo_state))
            cand = scorer.score(props)
            stats["scored"] += len(props)
            for k in range(len(local)):
                acc, better = rx.mc_delta(cur[k].scoring_function, cand[k].scoring_function, cur[k].temp_shelf, rngs[k])
                if acc:
                    cand[k].get_replica_num(cur[k].replica_num)
                    cand[k].get_temp_shelf(cur[k].temp_shelf)
                    cand[k].get_sim_step(global_step * exchange)
                    cur[k] = cand[k]
                    stats["acc_mc"] += 1
                    stats["acc_mc_better"] += int(better)
                    if (cur[k].mcc, cur[k].scoring_function) < (best.mcc, best.scoring_function):
                        best = cur[k]
                else:
                    stats["rej_mc"] += 1
        # replica exchange: ONE all-gather (scores + control flags); all ranks replay the same swaps on the whole ladder
        flags = [float(any(s.mcc == 0.0 for s in cur)), float(time.time() - t_start >= timelimit), 0.0]
        for k, r in enumerate(local):
            cur[k].get_sim_step(global_step * exchange)       # reference DesiRNA.py:371-373: stats.step = global_step * RE_attempt
        if stop_when_solved and num_results is not None and global_step % 10 == 0:
            flags[2] = float(_sws_reached(simulation_data + [dict(vars(s)) for s in cur], num_results, oligo_state))
        all_scores, ex = shards.allgather_scores([s.scoring_function for s in cur], extras=flags)
        solved = solved or bool(ex[:, 0].any())
        temps, acc, _, rej = rx.replica_exchange(list(temps), list(all_scores), global_step, main_rng)
        stats["acc_re"] += acc
        stats["rej_re"] += rej
        for k, r in enumerate(local):
            cur[k].get_temp_shelf(temps[r])
        simulation_data += [dict(vars(s)) for s in cur]       # reference DesiRNA.py:375
        if progress:
            progress(global_step, best, stats)
        stop = bool(ex[0, 1]) or (stop_when_solved and (solved if num_results is None else bool(ex[:, 2].all())))
    if shards.world > 1:                                      # the best replica of the whole job, on every rank
        cands = rx.gather_results({shards.rank: best}, shards.world)
        best = min((b for b in cands.values() if b is not None), key=lambda s: (s.mcc, s.scoring_function))
    stats["elapsed_s"] = time.time() - t_start
    return {"best": best, "solved": solved, "replicas": cur, "stats": stats, "steps": global_step, "temps": list(temps),
            "simulation_data": simulation_data, "engine": getattr(scorer, "engine", None)}


def run_design_fast(input_file, replicas=10, exchange=100, steps=None, timelimit=60, t_min=10.0, t_max=150.0,
                    scoring_f="Ed-Epf:1.0", tm_max=0.7, tm_min=0.0, point_mutations="on", seed=0, stop_when_solved=False,
                    device=0, engine=None, keep_records=True, native_loop=None, shards=None, num_results=None):
    """Same loop as :func:`run_design` with the per-replica host work in native code and no per-step Python objects:
    proposals, SimScore and Metropolis run batched in the C library, the replica state lives in numpy arrays.  The
    per-replica random streams are the reference's (MT19937 seeded with the replica index at every exchange step, CPython's
    draw mapping: ``host_driver.hpp``), so for a fixed seed this driver and :func:`run_design` walk the same trajectory.

    ``shards`` (``replica_exchange.ReplicaShards``): this rank holds replicas ``shards.local`` only (its engine needs
    ``max_R >= len(shards.local)``); per exchange step ONE all-gather carries the scores and the stop flags, every rank
    replays the swaps on the whole ladder.  Results do not depend on the sharding (streams are seeded by the GLOBAL
    replica index and the kernels' results do not depend on the batch composition)."""
    from . import engine as _engine
    if stop_when_solved and num_results is not None and not keep_records:
        raise ValueError("the -sws rule with num_results ranks the recorded sequences: it needs keep_records=True "
                         "(without records the stop test could never fire and the run would only end at its step / time limit)")
    prob = DesignProblem(input_file.sec_struct, input_file.seq_restr, input_file.alt_sec_structs)
    if prob.two_strands:
        raise NotImplementedError("two-strand inputs run through run_design (the native batched proposer is one-strand)")
    n_alt = len(input_file.alt_sec_structs) if input_file.alt_sec_structs else 0
    sf = es.parse_scoring_functions(scoring_f)
    for name, _ in sf:
        if name not in es.AVAILABLE_SCORING_FUNCTIONS:
            raise ValueError("%s is not an available option for scoring function. Check your command." % name)
    shards = shards or rx.ReplicaShards(replicas, 0, 1)
    local = np.array(shards.local, dtype=np.int64)
    R, L, Rl = replicas, prob.n, len(shards.local)
    eng = engine or _engine.Engine(max_R=max(1, Rl), max_L=L, device=device)
    hk = _engine.HostKernels()
    eng.set_targets([input_file.sec_struct] + list(input_file.alt_sec_structs or []))
    flags = _engine.NEED_PF | _engine.NEED_MFE | _engine.NEED_EVAL
    if set(input_file.sec_struct) - set(".()&"):
        flags |= _engine.NEED_PK
    amask = np.array([sum(1 << "ACGU".index(c) for c in a) for a in prob.allowed], dtype=np.uint8)
    temps = np.array(rx.get_rep_temps(R, t_min, t_max), dtype=np.float64)      # the whole ladder, replayed on every rank
    shelves = temps.copy()
    main_rng = random.Random(shards.broadcast_seed(2137 + seed if seed else random.random()))
    ref_ss = input_file.sec_struct

    def score(seqs_u8):
        Epf, Emfe, ss, Ed = eng.score_batch_arrays(seqs_u8, flags)
        mcc, rec, prec = hk.simscore(ref_ss, ss)
        ed = Ed[:, 0] / 100.0
        total = np.zeros(len(Epf))
        for name, w in sf:
            if name == "Ed-Epf":
                total += (ed - Epf) * w
            elif name == "1-MCC":
                total += (1 - mcc) * 10 * w
            elif name == "sln_Epf":
                total += (Epf + 0.3759 * L + 5.7534) / 10 * w
            elif name == "Ed-MFE":
                total += (ed - Emfe / 100.0) * w
            elif name == "1-precision":
                total += (1 - prec) * 10 * w
            elif name == "1-recall":
                total += (1 - rec) * 10 * w
            elif name == "Edef":
                total += eng.ensemble_defect_arrays(seqs_u8) * w
        if n_alt:                                                 # reference energy_scores.py:98-102
            total += Ed[:, 1:].sum(axis=1) / 100.0 / n_alt - Epf
        return total, 1 - mcc, ss, Epf, ed

    init = prob.initial_sequence(main_rng)
    cur = np.tile(np.frombuffer(init.encode(), dtype=np.uint8), (max(1, Rl), 1)).copy()
    cur_score, cur_mcc, cur_ss, cur_epf, cur_ed = score(cur)
    k0 = int(np.lexsort((cur_score, cur_mcc))[0])
    best = dict(sequence=cur[k0].tobytes().decode(), mfe_ss=cur_ss[k0].tobytes().decode(), mcc=float(cur_mcc[k0]),
                scoring_function=float(cur_score[k0]), Epf=float(cur_epf[k0]), edesired=float(cur_ed[k0]))
    stats = dict(acc_mc=0, acc_mc_better=0, rej_mc=0, acc_re=0, rej_re=0, scored=Rl)

    def records(step_no):
        out = []
        for k in range(Rl):
            sc = es.ScoreSeq(cur[k].tobytes().decode())
            sc.scoring_function = float(cur_score[k])
            sc.get_replica_num(int(local[k]) + 1)
            sc.get_temp_shelf(float(temps[local[k]]))
            sc.get_sim_step(step_no)
            sc.get_Epf(float(cur_epf[k]))
            sc.get_edesired(float(cur_ed[k]))
            sc.get_edesired_minus_Epf(sc.Epf, sc.edesired)
            sc.mcc = float(cur_mcc[k])
            sc.get_mfe_ss(cur_ss[k].tobytes().decode())
            out.append(dict(vars(sc)))
        return out

    simulation_data = records(0) if keep_records else []
    if native_loop is None:
        native_loop = True
    native_loop = native_loop and all(name in eng.TERM_IDS for name, _ in sf)   # (every -sf term, Edef included, has a native id)
    cur = np.ascontiguousarray(cur); cur_ss = np.ascontiguousarray(cur_ss)
    cur_score = np.ascontiguousarray(cur_score, dtype=np.float64); cur_mcc = np.ascontiguousarray(cur_mcc, dtype=np.float64)
    cur_epf = np.ascontiguousarray(cur_epf, dtype=np.float64); cur_ed = np.ascontiguousarray(cur_ed, dtype=np.float64)
    rng_state = np.empty((max(1, Rl), _engine.RNG_WORDS), dtype=np.uint32)
    t_start = time.time()
    step = 0
    solved = best["mcc"] == 0.0
    if shards.world > 1:
        _, ex = shards.allgather_scores(cur_score[:Rl], extras=[float(solved)])
        solved = bool(ex[:, 0].any())
    targeted = point_mutations == "on"
    stop = stop_when_solved and solved and num_results is None
    while not stop:
        if steps is not None and step >= steps:
            break
        step += 1
        hk.rng_seed(local if Rl else [0], out=rng_state)        # random.seed(replica index) at every exchange step (App. C2)
        tl = temps[local] if Rl else temps[:1]
        shelf_idx = np.searchsorted(shelves, tl).astype(np.int32)
        if native_loop and Rl:
            # the whole inner loop of the exchange step in native code (drna_mc_run): one call, no per-iteration Python
            state = dict(seqs=cur, mfe_ss=cur_ss, score=cur_score, mcc1=cur_mcc, Epf=cur_epf, Ed=cur_ed)
            counters = np.zeros(3, dtype=np.int64)
            bst = dict(seq=np.frombuffer(best["sequence"].encode(), dtype=np.uint8).copy(),
                       ss=np.frombuffer(best["mfe_ss"].encode(), dtype=np.uint8).copy(),
                       vals=np.array([best["mcc"], best["scoring_function"], best["Epf"], best["edesired"]], dtype=np.float64))
            eng.mc_run(prob, exchange, shelf_idx, R, tm_max, tm_min, targeted, np.ascontiguousarray(tl), sf, flags, rng_state, state,
                       counters, bst)
            best = dict(sequence=bst["seq"].tobytes().decode(), mfe_ss=bst["ss"].tobytes().decode(), mcc=float(bst["vals"][0]),
                        scoring_function=float(bst["vals"][1]), Epf=float(bst["vals"][2]), edesired=float(bst["vals"][3]))
            stats["acc_mc"] += int(counters[0]); stats["acc_mc_better"] += int(counters[1]); stats["rej_mc"] += int(counters[2])
            stats["scored"] += Rl * exchange
        for _ in range(0 if (native_loop or not Rl) else exchange):
            prop = hk.propose_alt(prob, cur, cur_ss, shelf_idx, R, tm_max, tm_min, targeted, rng_state)
            p_score, p_mcc, p_ss, p_epf, p_ed = score(prop)
            acc, better = hk.metropolis(cur_score, p_score, tl, rng_state)
            cur[acc] = prop[acc]; cur_ss[acc] = p_ss[acc]
            cur_score[acc] = p_score[acc]; cur_mcc[acc] = p_mcc[acc]; cur_epf[acc] = p_epf[acc]; cur_ed[acc] = p_ed[acc]
            na = int(acc.sum())
            stats["acc_mc"] += na
            stats["acc_mc_better"] += int((acc & better).sum())
            stats["rej_mc"] += Rl - na
            stats["scored"] += Rl
            # the best state is tracked replica by replica in replica order, like the native loop (first strictly better wins)
            for kb in np.nonzero(acc)[0]:
                if (cur_mcc[kb], cur_score[kb]) < (best["mcc"], best["scoring_function"]):
                    best = dict(sequence=cur[kb].tobytes().decode(), mfe_ss=cur_ss[kb].tobytes().decode(), mcc=float(cur_mcc[kb]),
                                scoring_function=float(cur_score[kb]), Epf=float(cur_epf[kb]), edesired=float(cur_ed[kb]))
        # ONE collective per exchange step: scores + (solved, time is up); rank 0's clock decides the time limit
        ctl = [float(bool(Rl) and bool((cur_mcc[:Rl] == 0.0).any())), float(time.time() - t_start >= timelimit), 0.0]
        if stop_when_solved and num_results is not None and keep_records and step % 10 == 0:      # the reference's -sws rule
            ctl[2] = float(_sws_reached(simulation_data + records(step * exchange), num_results, "none"))
        all_scores, ex = shards.allgather_scores(cur_score[:Rl], extras=ctl)
        solved = solved or bool(ex[:, 0].any())
        new_temps, a, _, rj = rx.replica_exchange(list(temps), list(all_scores), step, main_rng)
        temps = np.array(new_temps, dtype=np.float64)
        stats["acc_re"] += a
        stats["rej_re"] += rj
        if keep_records:
            simulation_data += records(step * exchange)          # reference: stats.step = global_step * RE_attempt
        stop = bool(ex[0, 1]) or (stop_when_solved and (solved if num_results is None else bool(ex[:, 2].all())))
    if shards.world > 1:
        cands = rx.gather_results({shards.rank: best if Rl else None}, shards.world)
        best = min((b for b in cands.values() if b is not None), key=lambda b: (b["mcc"], b["scoring_function"]))
    stats["elapsed_s"] = time.time() - t_start
    return {"best": SimpleNamespace(**best), "solved": solved, "stats": stats, "steps": step, "simulation_data": simulation_data,
            "engine": eng, "temps": [float(t) for t in temps], "local": [int(r) for r in local],
            "used_native_loop": bool(native_loop)}


def run_puzzle_set(inputs, rank=0, world=1, driver=None, **kw):
    """Design every problem of `inputs` (BASELINE config 4), the set sharded over `world` ranks by sum n^3
    (``replica_exchange.shard_puzzles``): each rank runs its own puzzles with all their replicas on its GPU and the
    results are gathered once at the end.  Returns {index: dict(name, solved, sequence, mfe_ss, score)} on every rank."""
    driver = driver or run_design_fast
    owner = rx.shard_puzzles([len(i.sec_struct) for i in inputs], world)
    mine = {}
    for k, inp in enumerate(inputs):
        if owner[k] != rank:
            continue
        res = driver(inp, **kw)
        b = res["best"]
        mine[k] = dict(name=inp.name, solved=bool(res["solved"]), sequence=b.sequence, mfe_ss=b.mfe_ss,
                       score=float(b.scoring_function), rank=rank)
    return rx.gather_results(mine, world)


def main(argv=None):
    ap = argparse.ArgumentParser(description="GPU replica-exchange RNA design (flag names follow DesiRNA.py)")
    ap.add_argument("-f", "--filename", required=True, dest="name")
    ap.add_argument("-R", "--replicas", type=int, default=10)
    ap.add_argument("-e", "--exchange", type=int, default=100)
    ap.add_argument("-t", "--timelimit", type=int, default=60, dest="timlim")
    ap.add_argument("-s", "--steps", type=int, default=None)
    ap.add_argument("-tmin", "--tmin", type=float, default=10, dest="t_min")
    ap.add_argument("-tmax", "--tmax", type=float, default=150, dest="t_max")
    ap.add_argument("-sf", "--scoring_function", default="Ed-Epf:1.0", dest="scoring_f")
    ap.add_argument("-tm", "--target_mutations", default="on", choices=["off", "on"], dest="pm")
    ap.add_argument("-tm_perc_max", type=float, default=0.7, dest="tm_max")
    ap.add_argument("-tm_perc_min", type=float, default=0.0, dest="tm_min")
    ap.add_argument("-seed", "--seed_number", type=int, default=0, dest="in_seed")
    ap.add_argument("-sws", "--stop_when_solved", default="off", choices=["off", "on"], dest="sws")
    ap.add_argument("-r", "--results_number", type=int, default=10, dest="num_results")
    ap.add_argument("-d", "--dimer", default="off", choices=["off", "on"], dest="dimer", help="homodimer design (two-strand input)")
    ap.add_argument("-oa", "--avoid_oligomerization", default="off", choices=["off", "on"], dest="oligo")
    ap.add_argument("-nd", "--negative_design", default="off", choices=["off", "on"], dest="subopt")
    ap.add_argument("-acgu", "--ACGU", default="off", choices=["off", "on"], dest="percs", help="weighted nucleotide choices")
    ap.add_argument("-acgu_content", "--ACGU_content", default="", dest="acgu_content", help="A,C,G,U percentages (sum 100)")
    ap.add_argument("--python-host", action="store_true", help="per-replica Python host loop instead of the native batched one")
    ap.add_argument("-o", "--outdir", default=None, help="write the reference's result files (_results.csv, _traj.csv, "
                    "_stats, _best_str, fasta files) into this directory")
    a = ap.parse_args(argv)
    inp = read_input(a.name)
    two = "&" in inp.sec_struct or a.oligo == "on" or a.percs == "on" or a.subopt == "on"
    extra = dict(dimer=a.dimer, oligo=a.oligo, subopt=a.subopt) if two else {}
    if a.percs == "on":
        vals = [int(x) for x in a.acgu_content.split(",")] if a.acgu_content else [15, 30, 30, 15]
        if sum(vals) != 100:
            raise SystemExit("The ACGU content should sum up to 100, check your command.")
        extra["acgu"] = dict(zip("ACGU", vals))
    res = (run_design if (a.python_host or two) else run_design_fast)(inp, **extra, replicas=a.replicas, exchange=a.exchange, steps=a.steps, timelimit=a.timlim, t_min=a.t_min,
                     t_max=a.t_max, scoring_f=a.scoring_f, tm_max=a.tm_max, tm_min=a.tm_min, point_mutations=a.pm,
                     seed=a.in_seed, stop_when_solved=a.sws == "on", num_results=a.num_results if a.sws == "on" else None)
    if a.outdir:
        import os
        from . import outputs
        os.makedirs(a.outdir, exist_ok=True)
        sf = es.parse_scoring_functions(a.scoring_f)
        pks = "on" if set(inp.sec_struct) - set(".()&") else "off"
        outname = outputs.get_outname(os.path.basename(a.name), a.replicas, a.exchange, a.timlim, pks, "off", a.t_min, a.t_max,
                                      "1999", sf, "off", "off", a.pm)
        st = res["stats"]
        stats = SimpleNamespace(step=st["acc_mc"] + st["rej_mc"], global_step=res["steps"], acc_mc_step=st["acc_mc"],
                                acc_mc_better_e=st["acc_mc_better"], rej_mc_step=st["rej_mc"], acc_re_step=st["acc_re"],
                                rej_re_step=st["rej_re"])
        oligo_state = ("homodimer" if a.dimer == "on" else "heterodimer") if "&" in inp.sec_struct else ("avoid" if a.oligo == "on" else "none")
        outputs.write_all(res["simulation_data"], inp.name, os.path.basename(a.name), outname, stats, st["elapsed_s"], a.timlim,
                          time.strftime("%Y%m%d.%H%M%S"), num_results=a.num_results, directory=a.outdir,
                          alt_sec_structs=list(inp.alt_sec_structs or []) or None, engine=res.get("engine"),
                          oligo_state=oligo_state, subopt=a.subopt, sec_struct=inp.sec_struct)
    b = res["best"]
    print("Design solved succesfully!" if res["solved"] else "Design not solved.")
    print(b.sequence)
    print(b.mfe_ss)
    print("Epf=%.3f Ed=%.3f 1-MCC=%.3f score=%.3f  steps=%d scored=%d in %.1fs" %
          (b.Epf, b.edesired, b.mcc, b.scoring_function, res["steps"], res["stats"]["scored"], res["stats"]["elapsed_s"]))


if __name__ == "__main__":
    main()
