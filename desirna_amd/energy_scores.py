"""Scoring function of the design loop, batched over replicas and backed by the gfx950 engine.

Counterpart of the reference's ``utils/energy_scores.py``: ``score_sequence`` (:31-125),
``get_mfe_e_ss`` (:128-159) and ``ScoreSeq`` (:162-450).  The ``-sf`` plug-in surface is unchanged:
the same term names (``Ed-Epf``, ``1-MCC``, ``sln_Epf``, ``Ed-MFE``, ``1-precision``, ``1-recall``,
``Edef``) with the same weights and x10 factors (:376-398), the alt-structure term (:98-102) and the
motif bonus (:121-123, :443-450).  What changes is where the numbers come from: one
``Engine.score_batch`` call returns Epf, the MFE (or pk-annotated) structure, the MFE energy and
E(target), E(alt targets) for every replica of the batch, instead of 3-8 ViennaRNA calls per
sequence inside a forked worker.

``Edef`` is served by ``Engine.ensemble_defect`` (inside + outside recursion on the GPU).

Two strands (``oligo_state`` heterodimer / homodimer, and ``avoid`` = ``-o on``) go through ``Engine.cofold_batch``
(co-fold MFE + partition function + two-strand evaluation on the GPU); the oligomer / monomer bonus terms of
``utils/dimer_multichain_energy.py`` are computed here from its free energies.

``-nd on`` (negative design): the energy of the second-best structure comes from ``Engine.subopt_energy`` (two-best
dynamic programme on the GPU; no golden in the reference: checked against exhaustive enumeration).
"""
from . import engine as _engine
from .sim_score import batch_metrics

# reference utils/dimer_multichain_energy.py:24-28
KB = 0.001987204259
RHO = 55.14
TEMP = 273.15 + 37
CONC = 1e-3


def oligo_fraction(FA, FB, FcAB):
    """reference dimer_multichain_energy.oligo_fraction (:30-45): fraction of strands bound in the dimer at 1 mM."""
    import numpy as np
    dF = FcAB - FA - FB
    rhs = CONC / RHO * np.exp(-dF / (KB * TEMP))
    return 1 - (np.sqrt(1 + 4 * rhs) - 1) / (2 * rhs)


def kTlog_oligo_fraction(frac):
    import numpy as np
    return -KB * TEMP * np.log(frac)


def kTlog_monomer_fraction(frac):
    import numpy as np
    return -KB * TEMP * np.log(1 - frac)


AVAILABLE_SCORING_FUNCTIONS = ['Ed-Epf', '1-MCC', 'sln_Epf', 'Ed-MFE', '1-precision', '1-recall', 'Edef']


def parse_scoring_functions(scoring_f_str, first_term_only=True):
    """``-sf`` string -> [(name, weight), ...].

    ``first_term_only=True`` reproduces the reference, whose ``return`` sits inside the loop
    (``utils/stats_inputs_outputs.py:232-237``): only the first term survives.
    """
    out = []
    for item in scoring_f_str.split(','):
        if ':' not in item:
            raise ValueError(f"Invalid scoring function format: {item}. Expected format: 'function:weight'")
        function, weight = item.split(':')
        out.append((function, float(weight)))
        if first_term_only:
            return out
    return out


class ScoreSeq:
    """Per-replica state record.  The attribute NAMES and their ORDER are the reference's (``vars(obj)`` is its CSV schema,
    ``utils/energy_scores.py:176-195``) and so are the method names the host code calls; the setters themselves are generated
    from three small tables below the class instead of being written out one by one."""

    # (attribute, initial value) in the order of the reference's __init__ = the column order of _traj.csv / _results.csv
    _SCHEMA = (("sequence", None), ("scoring_function", 0), ("replica_num", None), ("temp_shelf", None), ("sim_step", 0),
               ("edesired_minus_Epf", 0), ("Epf", 0), ("edesired", 0), ("mcc", 0), ("mcc_alt", 0), ("mfe_ss", None),
               ("subopt_e", 0), ("esubopt_minus_Epf", 0), ("sln_Epf", 0), ("MFE", 0), ("edesired_minus_MFE", 0), ("recall", 0),
               ("precision", 0), ("edesired2", 0), ("edesired2_minus_Epf", 0))
    # -sf term -> value taken from the record (reference :376-398); a name that is not listed contributes nothing, as there
    _TERMS = {"Ed-Epf": lambda r: r.edesired_minus_Epf, "1-MCC": lambda r: r.mcc * 10, "sln_Epf": lambda r: r.sln_Epf,
              "Ed-MFE": lambda r: r.edesired_minus_MFE, "1-precision": lambda r: r.precision * 10,
              "1-recall": lambda r: r.recall * 10, "Edef": lambda r: r.ensemble_defect}

    def __init__(self, sequence):
        for name, value in self._SCHEMA:
            setattr(self, name, value)
        self.sequence = sequence

    def get_sln_Epf(self):
        self.sln_Epf = (self.Epf + 0.3759 * len(self.sequence) + 5.7534) / 10

    def get_edesired_minus_MFE(self):
        self.edesired_minus_MFE = self.edesired - self.MFE

    def get_scoring_function_w_subopt(self):
        self.scoring_function = self.scoring_function - self.esubopt_minus_Epf

    def get_scoring_function(self, scoring_f):
        total = 0
        for function, weight in scoring_f:
            term = self._TERMS.get(function)
            if term is not None:
                total += term(self) * weight
        self.scoring_function = total

    def get_scoring_function_w_alt_ss(self):
        self.scoring_function = self.scoring_function + self.edesired2_minus_Epf

    def update_scoring_function_w_motifs(self, motif_bonus):
        self.scoring_function += motif_bonus


def _install_setters(cls):
    """``get_<attr>(value)`` stores a value as it comes (MFE: the reference re-folds with RNA.fold(), :350-354, the engine's fill
    already produced f5[n]; ensemble_defect: the reference runs mfe / rescale / pf / ensemble_defect on a new fold compound,
    :362-374, here the value comes from the engine's inside + outside kernels -- and the attribute is created on first use, as
    there); ``get_<attr>(x)`` for precision / recall / mcc stores 1 - x; ``get_<attr>_minus_Epf(Epf, e)`` stores e - Epf."""
    def plain(attr):
        def setter(self, value):
            setattr(self, attr, value)
        return setter

    def complement(attr):
        def setter(self, value):
            setattr(self, attr, 1 - value)
        return setter

    def minus_epf(attr):
        def setter(self, Epf, energy):
            setattr(self, attr, energy - Epf)
        return setter

    for attr in ("replica_num", "temp_shelf", "sim_step", "Epf", "mfe_ss", "edesired", "edesired2", "MFE", "subopt_e", "ensemble_defect"):
        setattr(cls, "get_" + attr, plain(attr))
    for attr in ("precision", "recall", "mcc"):
        setattr(cls, "get_" + attr, complement(attr))
    for attr in ("edesired_minus_Epf", "edesired2_minus_Epf", "esubopt_minus_Epf"):
        setattr(cls, "get_" + attr, minus_epf(attr))


_install_setters(ScoreSeq)


def score_motifs(seq, sim_options):
    """reference utils/sequence_utils.py:1231-1251"""
    motif_score = 0
    for motif in sim_options.motifs:
        if sim_options.motifs[motif][0].search(seq):
            motif_score += sim_options.motifs[motif][1]
    return motif_score


class ReplicaScorer:
    """Binds an engine to one design problem (target + alt structures + options) and scores batches."""

    def __init__(self, input_file, sim_options, max_replicas, device=0, engine=None):
        self.oligo_state = getattr(sim_options, "oligo_state", "none")
        if self.oligo_state not in ("none", "avoid", "heterodimer", "homodimer"):
            raise ValueError("unknown oligo_state %r" % self.oligo_state)
        self.subopt = getattr(sim_options, "subopt", "off") == "on"
        if self.subopt and self.oligo_state in ("heterodimer", "homodimer"):
            raise NotImplementedError("-nd on with two-strand inputs")
        self.input_file = input_file
        self.sim_options = sim_options
        self.target = input_file.sec_struct.replace("&", "")
        L = len(self.target)
        self.engine = engine or _engine.Engine(max_R=max_replicas, max_L=L, device=device,
                                               params=str(getattr(sim_options, "param", "1999")))
        # -oa on folds every candidate against itself (s & s: 2 L nucleotides, reference :411-418): that needs an engine
        # sized for 2 L; the scoring engine stays sized for L (its workspace pitch follows max_L)
        self.dimer_engine = self.engine
        if self.oligo_state == "avoid" and self.engine.max_L < 2 * L:
            self.dimer_engine = _engine.Engine(max_R=max(max_replicas, 1), max_L=2 * L, device=device,
                                               params=str(getattr(sim_options, "param", "1999")))
        targets = [self.target]
        if getattr(input_file, "alt_sec_struct", None) is not None:
            targets += [a.replace("&", "") for a in input_file.alt_sec_structs]
        self.engine.set_targets(targets)
        self.flags = _engine.NEED_PF | _engine.NEED_MFE | _engine.NEED_EVAL
        if getattr(sim_options, "pks", "off") == "on":
            self.flags |= _engine.NEED_PK
        for function, _ in sim_options.scoring_f:
            if function not in AVAILABLE_SCORING_FUNCTIONS:
                raise ValueError("%s is not an available option for scoring function. Check your command." % function)
        # reference :93-94: get_ensemble_defect(input_file.sec_struct) whenever 'Edef' is among the -sf terms
        self.want_edef = any(function == 'Edef' for function, _ in sim_options.scoring_f)

    def score(self, seqs):
        """list of sequences -> list of ScoreSeq (reference score_sequence(), once per replica)."""
        if self.oligo_state in ("heterodimer", "homodimer"):
            return self._score_two_strands(list(seqs))
        out = self.engine.score_batch(list(seqs), self.flags)
        metrics = batch_metrics(self.input_file.sec_struct.replace("&", "Ee"),
                                [s.replace("&", "Ee") for s in out["mfe_ss"]])
        edef = self.engine.ensemble_defect(list(seqs)) if self.want_edef else None
        res = []
        for k, seq in enumerate(seqs):
            sc = ScoreSeq(sequence=seq)
            sc.get_Epf(float(out["Epf"][k]))
            sc.get_mfe_ss(out["mfe_ss"][k])
            sc.get_edesired(int(out["Ed"][k, 0]) / 100.0)
            sc.get_edesired_minus_Epf(sc.Epf, sc.edesired)
            mcc, recall, precision = metrics[k]
            sc.get_precision(precision)
            sc.get_recall(recall)
            sc.get_mcc(mcc)
            for function, _ in self.sim_options.scoring_f:
                if function == 'sln_Epf':
                    sc.get_sln_Epf()
                if function == 'Ed-MFE':
                    sc.get_MFE(int(out["Emfe"][k]) / 100.0)
                    sc.get_edesired_minus_MFE()
                if function == 'Edef':
                    sc.get_ensemble_defect(float(edef[k]))
            sc.get_scoring_function(self.sim_options.scoring_f)
            if getattr(self.input_file, "alt_sec_struct", None) is not None:
                energies = [int(e) / 100.0 for e in out["Ed"][k, 1:]]
                sc.get_edesired2(sum(energies) / len(energies))
                sc.get_edesired2_minus_Epf(sc.Epf, sc.edesired2)
                sc.get_scoring_function_w_alt_ss()
            res.append(sc)
        if self.subopt:
            # reference :105-108 (-nd on): for solved candidates (1-MCC == 0) the energy of the first sub-optimal structure
            hit = [k for k, sc in enumerate(res) if sc.mcc == 0]
            if hit:
                e2 = self.engine.subopt_energy([seqs[k] for k in hit])
                for k, v in zip(hit, e2):
                    res[k].get_subopt_e(int(v) / 100.0)
                    res[k].get_esubopt_minus_Epf(res[k].Epf, res[k].subopt_e)
                    res[k].get_scoring_function_w_subopt()
        if self.oligo_state == "avoid":
            # reference get_scoring_function_monomer (:411-418): homodimer of the sequence with itself, monomer fraction bonus
            # (applied before the motif bonus in the reference; both are additive)
            co = self.dimer_engine.cofold_batch([s + "&" + s for s in seqs], _engine.NEED_PF)
            for k, sc in enumerate(res):
                sc.oligo_fraction = float(oligo_fraction(co["FA"][k], co["FB"][k], co["FcAB"][k]))
                sc.monomer_bonus = float(kTlog_monomer_fraction(sc.oligo_fraction))
                sc.scoring_function = sc.scoring_function + sc.monomer_bonus
        if getattr(self.sim_options, "motifs", None):
            for seq, sc in zip(seqs, res):
                sc.update_scoring_function_w_motifs(score_motifs(seq, self.sim_options))
        return res

    def _score_two_strands(self, seqs):
        """reference score_sequence() for oligo_state heterodimer / homodimer (:70-118): Epf = pf_dimer()[-1] (FAB), MFE
        structure of mfe_dimer() with the '&' re-inserted, E(target) of the two-strand evaluation, then the oligomer bonus
        (hetero-dimer, or homodimer with two different structures) or the monomer-fraction term (two equal structures)."""
        out = self.engine.cofold_batch(seqs, _engine.NEED_PF | _engine.NEED_MFE | _engine.NEED_EVAL)
        metrics = batch_metrics(self.input_file.sec_struct.replace("&", "Ee"),
                                [s.replace("&", "Ee") for s in out["mfe_ss"]])
        ss1, ss2 = self.input_file.sec_struct.split("&")
        res = []
        for k, seq in enumerate(seqs):
            sc = ScoreSeq(sequence=seq)
            sc.get_Epf(float(out["FAB"][k]))
            sc.get_mfe_ss(out["mfe_ss"][k])
            sc.get_edesired(int(out["Ed"][k, 0]) / 100.0)
            sc.get_edesired_minus_Epf(sc.Epf, sc.edesired)
            mcc, recall, precision = metrics[k]
            sc.get_precision(precision)
            sc.get_recall(recall)
            sc.get_mcc(mcc)
            for function, _ in self.sim_options.scoring_f:
                if function == 'sln_Epf':
                    sc.get_sln_Epf()
                if function in ('Ed-MFE', 'Edef'):
                    raise NotImplementedError("-sf %s is a one-strand quantity in the reference (RNA.fold / md defaults)" % function)
            sc.get_scoring_function(self.sim_options.scoring_f)
            sc.oligo_fraction = float(oligo_fraction(out["FA"][k], out["FB"][k], out["FcAB"][k]))
            if self.oligo_state == "heterodimer" or ss1 != ss2:
                sc.oligomer_bonus = float(kTlog_oligo_fraction(sc.oligo_fraction))
            else:
                sc.oligomer_bonus = float(kTlog_monomer_fraction(sc.oligo_fraction))
            sc.scoring_function = sc.scoring_function + sc.oligomer_bonus
            if getattr(self.sim_options, "motifs", None):
                sc.update_scoring_function_w_motifs(score_motifs(seq, self.sim_options))
            res.append(sc)
        return res


def score_sequence(seq, input_file, sim_options, scorer=None):
    """Single-sequence form with the reference's signature (a batch of one)."""
    scorer = scorer or ReplicaScorer(input_file, sim_options, max_replicas=1)
    return scorer.score([seq])[0]
