"""ctypes binding of the gfx950 scoring engine (include/desirna_amd.h -> libdesirna_amd.so).

This is the Python side of the drop-in boundary: what the reference gets from the ViennaRNA SWIG
module per sequence (``RNA.fold_compound(seq, md).pf() / .mfe() / .eval_structure()``, reference
``utils/energy_scores.py:147-151,75``) is obtained here for a whole batch of replicas per call.
There is deliberately NO CPU fallback: if the HIP library is missing or no GPU is present the
constructor raises.
"""
import ctypes as C
import os

import numpy as np

from . import params as _params

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdesirna_amd.so")

NEED_PF, NEED_MFE, NEED_PK, NEED_EVAL = 1, 2, 4, 8

_ERRORS = {-1: "bad argument", -2: "bad parameter blob", -3: "HIP/device error", -4: "bad sequence character",
           -5: "unbalanced structure", -6: "partition function out of fp64 range", -7: "internal (traceback)"}

EXPORTS = ("drna_create", "drna_destroy", "drna_last_error", "drna_set_targets", "drna_score_batch",
           "drna_score_batch_device", "drna_last_timing", "drna_info", "drna_simscore_batch", "drna_propose_batch",
           "drna_metropolis_batch", "drna_ensemble_defect_batch", "drna_ensemble_defect_batch_device",
           "drna_last_edef_timing", "drna_propose_batch_alt", "drna_set_targets_ragged", "drna_score_ragged", "drna_cofold_batch", "drna_mc_run", "drna_subopt_energy_batch",
           "drna_subopt_structs_batch", "drna_rng_seed", "drna_rng_random", "drna_set_option", "drna_timing_sums", "drna_debug_strip_clocks", "drna_get_option", "drna_abi_version")

ABI_VERSION = 3        # DRNA_ABI_VERSION this binding was written against (include/desirna_amd.h)

RNG_WORDS = 625        # DRNA_RNG_WORDS: uint32 words of one replica's MT19937 stream


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("desirna_amd engine error %d (%s): %s" % (code, _ERRORS.get(code, "?"), msg))
        self.code = code


def load_library(path=None):
    """dlopen the C-ABI library and declare its signatures (raises if it has not been built)."""
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise FileNotFoundError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C desirna_amd/csrc`); there is no CPU fallback" % path)
    L = C.CDLL(path)
    vp, ci, u32 = C.c_void_p, C.c_int, C.c_uint32
    L.drna_create.restype = ci
    L.drna_create.argtypes = [vp, ci, ci, ci, ci, C.POINTER(vp)]
    L.drna_destroy.restype = None
    L.drna_destroy.argtypes = [vp]
    L.drna_last_error.restype = C.c_char_p
    L.drna_last_error.argtypes = [vp]
    L.drna_set_targets.restype = ci
    L.drna_set_targets.argtypes = [vp, ci, ci, C.c_char_p]
    L.drna_score_batch.restype = ci
    L.drna_score_batch.argtypes = [vp, ci, ci, C.c_char_p, u32, vp, vp, vp, vp]
    L.drna_score_batch_device.restype = ci
    L.drna_score_batch_device.argtypes = [vp, ci, ci, vp, u32, vp, vp, vp, vp]
    L.drna_abi_version.restype = ci
    L.drna_abi_version.argtypes = []
    if L.drna_abi_version() != ABI_VERSION:
        raise EngineError(-1, "library ABI version %d, this binding expects %d: rebuild desirna_amd/csrc" % (L.drna_abi_version(), ABI_VERSION))
    L.drna_last_timing.restype = ci
    L.drna_last_timing.argtypes = [vp, vp]
    L.drna_info.restype = ci
    L.drna_info.argtypes = [vp, vp]
    L.drna_timing_sums.restype = ci
    L.drna_timing_sums.argtypes = [vp, vp, ci]
    L.drna_set_option.restype = ci
    L.drna_set_option.argtypes = [vp, C.c_char_p, ci]
    L.drna_get_option.restype = ci
    L.drna_get_option.argtypes = [vp, C.c_char_p, vp]
    L.drna_ensemble_defect_batch.restype = ci
    L.drna_ensemble_defect_batch.argtypes = [vp, ci, ci, C.c_char_p, vp, vp]
    L.drna_ensemble_defect_batch_device.restype = ci
    L.drna_ensemble_defect_batch_device.argtypes = [vp, ci, ci, vp, vp, vp]
    L.drna_last_edef_timing.restype = ci
    L.drna_last_edef_timing.argtypes = [vp, vp]
    L.drna_set_targets_ragged.restype = ci
    L.drna_set_targets_ragged.argtypes = [vp, ci, vp, C.c_char_p]
    L.drna_score_ragged.restype = ci
    L.drna_score_ragged.argtypes = [vp, ci, vp, C.c_char_p, vp, u32, vp, vp, vp, vp]
    L.drna_cofold_batch.restype = ci
    L.drna_cofold_batch.argtypes = [vp, ci, ci, ci, C.c_char_p, u32, vp, vp, vp, vp]
    L.drna_mc_run.restype = ci
    L.drna_mc_run.argtypes = [vp, ci, ci, ci, C.c_char_p, vp, vp, vp, ci, vp, vp, vp, vp, vp, ci, C.c_double, C.c_double, ci, vp,
                              C.c_double, ci, vp, vp, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.drna_subopt_energy_batch.restype = ci
    L.drna_subopt_energy_batch.argtypes = [vp, ci, ci, C.c_char_p, vp, vp]
    L.drna_subopt_structs_batch.restype = ci
    L.drna_subopt_structs_batch.argtypes = [vp, ci, ci, C.c_char_p, ci, vp, vp]
    L.drna_simscore_batch.restype = ci
    L.drna_simscore_batch.argtypes = [ci, ci, C.c_char_p, vp, vp, vp, vp]
    L.drna_propose_batch.restype = ci
    L.drna_propose_batch.argtypes = [ci, ci, C.c_char_p, vp, vp, vp, vp, ci, C.c_double, C.c_double, ci, vp, vp]
    L.drna_propose_batch_alt.restype = ci
    L.drna_propose_batch_alt.argtypes = [ci, ci, C.c_char_p, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, ci, C.c_double,
                                         C.c_double, ci, vp, vp]
    L.drna_metropolis_batch.restype = ci
    L.drna_metropolis_batch.argtypes = [ci, vp, vp, vp, C.c_double, vp, vp, vp]
    L.drna_rng_seed.restype = ci
    L.drna_rng_seed.argtypes = [ci, vp, vp]
    L.drna_rng_random.restype = ci
    L.drna_rng_random.argtypes = [ci, vp, vp]
    return L


_LIVE = None          # weak set of open engines, _CLOSED_FALLBACKS: the counters of the closed ones (sync_fallbacks_total)
_CLOSED_FALLBACKS = 0


def sync_fallbacks_total():
    """Calls of this process, over all engines, in which a fold by several workgroups lost a partner and was redone with one
    workgroup per fold (``drna_get_option("sync_fallbacks")``): 0 on a healthy box; the GPU test suite asserts it."""
    return _CLOSED_FALLBACKS + sum(e.get_option("sync_fallbacks") for e in list(_LIVE or ()) if e._h and e._h.value)


class Engine:
    """One engine per GPU (``RNA.params_load`` + all ``RNA.fold_compound`` allocations, done once)."""

    def __init__(self, max_R, max_L, device=0, params="1999", lib=None):
        self._L = load_library(lib)
        self._h = C.c_void_p()
        blob = params if isinstance(params, np.ndarray) else _params.load_params(params)
        blob = np.ascontiguousarray(blob, dtype=np.int32)
        rc = self._L.drna_create(blob.ctypes.data, blob.size, int(device), int(max_R), int(max_L), C.byref(self._h))
        if rc != 0:
            raise EngineError(rc, self._L.drna_last_error(None).decode())
        self.max_R, self.max_L, self.device = int(max_R), int(max_L), int(device)
        self.n_targets = 0
        self.L = None
        global _LIVE
        if _LIVE is None:
            import weakref
            _LIVE = weakref.WeakSet()
        _LIVE.add(self)

    def set_option(self, name, value):
        """Engine option (``drna_set_option``): ``"dual"`` = fold small batches with two workgroups per sequence (default on)."""
        self._check(self._L.drna_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name):
        """Reads an option back, or the counter ``"sync_fallbacks"`` (``drna_get_option``)."""
        v = C.c_int(0)
        self._check(self._L.drna_get_option(self._h, name.encode(), C.byref(v)))
        return v.value

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            global _CLOSED_FALLBACKS
            try:
                _CLOSED_FALLBACKS += self.get_option("sync_fallbacks")
            except Exception:
                pass
            self._L.drna_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise EngineError(rc, self._L.drna_last_error(self._h).decode())

    def set_targets(self, targets):
        """targets[0] = design target, targets[1:] = alternative structures ('&' already removed)."""
        targets = list(targets)
        L = len(targets[0]) if targets else 1
        if any(len(t) != L for t in targets):
            raise ValueError("all target structures must have the same length")
        self._check(self._L.drna_set_targets(self._h, len(targets), L, "".join(targets).encode("ascii")))
        self.n_targets, self.L = len(targets), L

    def score_batch(self, seqs, flags=NEED_PF | NEED_MFE | NEED_EVAL):
        """seqs: list of equal-length strings.  Returns dict(Epf, Emfe, mfe_ss, Ed) (None if not requested);
        energies in ViennaRNA's units: Epf kcal/mol (float), Emfe / Ed int dcal/mol."""
        R = len(seqs)
        L = len(seqs[0])
        if any(len(s) != L for s in seqs):
            raise ValueError("all sequences of a batch must have the same length")
        sb = "".join(seqs).encode("ascii")
        Epf = np.zeros(R, dtype=np.float64) if flags & NEED_PF else None
        want_mfe = flags & (NEED_MFE | NEED_PK)
        Emfe = np.zeros(R, dtype=np.int32) if want_mfe else None
        ss = np.zeros((R, L), dtype=np.uint8) if want_mfe else None
        Ed = np.zeros((R, max(1, self.n_targets)), dtype=np.int32) if flags & NEED_EVAL else None
        ptr = lambda a: a.ctypes.data if a is not None else None
        self._check(self._L.drna_score_batch(self._h, R, L, sb, flags, ptr(Epf), ptr(Emfe), ptr(ss), ptr(Ed)))
        return {"Epf": Epf, "Emfe": Emfe,
                "mfe_ss": [bytes(r).decode("ascii") for r in ss] if ss is not None else None, "Ed": Ed}

    def score_batch_arrays(self, seqs_u8, flags=NEED_PF | NEED_MFE | NEED_EVAL):
        """Array form for hot host loops: seqs_u8 is an (R, L) uint8 array of ASCII letters; returns
        (Epf float64[R], Emfe int32[R], mfe_ss uint8[R, L], Ed int32[R, n_targets]) without building Python strings."""
        seqs_u8 = np.ascontiguousarray(seqs_u8, dtype=np.uint8)
        R, L = seqs_u8.shape
        Epf = np.zeros(R, dtype=np.float64)
        Emfe = np.zeros(R, dtype=np.int32)
        ss = np.zeros((R, L), dtype=np.uint8)
        Ed = np.zeros((R, max(1, self.n_targets)), dtype=np.int32)
        self._check(self._L.drna_score_batch(self._h, R, L, seqs_u8.ctypes.data_as(C.c_char_p), flags,
                                             Epf.ctypes.data, Emfe.ctypes.data, ss.ctypes.data, Ed.ctypes.data))
        return Epf, Emfe, ss, Ed

    def score_batch_device(self, d_seqs, R, L, flags, d_Epf=None, d_Emfe=None, d_ss=None, d_Ed=None):
        """Device-resident variant: arguments are raw device pointers (e.g. ``tensor.data_ptr()``).
        The caller must have made the inputs visible (``torch.cuda.synchronize()``) before the call."""
        self._check(self._L.drna_score_batch_device(self._h, R, L, d_seqs, flags, d_Epf, d_Emfe, d_ss, d_Ed))

    def set_targets_ragged(self, structures):
        """Structures of different lengths for :meth:`score_ragged` ('&' already removed)."""
        lens = np.array([len(t) for t in structures], dtype=np.int32)
        self._check(self._L.drna_set_targets_ragged(self._h, len(structures), lens.ctypes.data, "".join(structures).encode("ascii")))
        self.n_ragged_targets = len(structures)

    def score_ragged(self, seqs, target_of=None, flags=NEED_PF | NEED_MFE | NEED_EVAL):
        """Sequences of DIFFERENT lengths in one call (BASELINE config 4: many puzzles x replicas).  target_of[r] is the index
        (into :meth:`set_targets_ragged`) of the structure sequence r is evaluated on.  Returns dict(Epf, Emfe, mfe_ss, Ed[R])."""
        R = len(seqs)
        lens = np.array([len(s) for s in seqs], dtype=np.int32)
        total = int(lens.sum())
        if target_of is None:
            flags &= ~NEED_EVAL
        tof = np.ascontiguousarray(target_of, dtype=np.int32) if target_of is not None else None
        Epf = np.zeros(R, dtype=np.float64) if flags & NEED_PF else None
        want_mfe = flags & (NEED_MFE | NEED_PK)
        Emfe = np.zeros(R, dtype=np.int32) if want_mfe else None
        ss = np.zeros(total, dtype=np.uint8) if want_mfe else None
        Ed = np.zeros(R, dtype=np.int32) if flags & NEED_EVAL else None
        ptr = lambda a: a.ctypes.data if a is not None else None
        self._check(self._L.drna_score_ragged(self._h, R, lens.ctypes.data, "".join(seqs).encode("ascii"), ptr(tof), flags,
                                              ptr(Epf), ptr(Emfe), ptr(ss), ptr(Ed)))
        out_ss = None
        if ss is not None:
            b = ss.tobytes().decode("ascii")
            offs = np.concatenate(([0], np.cumsum(lens)))
            out_ss = [b[offs[k]:offs[k + 1]] for k in range(R)]
        return {"Epf": Epf, "Emfe": Emfe, "mfe_ss": out_ss, "Ed": Ed}

    def cofold_batch(self, seqs, flags=NEED_PF | NEED_MFE | NEED_EVAL):
        """Two-strand scoring: seqs are 'AAAA&BBBB' strings with the same strand lengths.  Returns dict(FA, FB, FcAB, FAB
        (kcal/mol; the reference's Epf is FAB), Emfe (dcal/mol), mfe_ss (with the '&' re-inserted), Ed (vs. set_targets))."""
        a0, b0 = seqs[0].split("&")
        cut, L = len(a0), len(a0) + len(b0)
        flat = []
        for s in seqs:
            a, b = s.split("&")
            if len(a) != cut or len(a) + len(b) != L:
                raise ValueError("all pairs of a batch must have the same strand lengths")
            flat.append(a + b)
        R = len(flat)
        if not (flags & NEED_EVAL and self.n_targets):
            flags &= ~NEED_EVAL
        F4 = np.zeros((R, 4), dtype=np.float64) if flags & NEED_PF else None
        Emfe = np.zeros(R, dtype=np.int32) if flags & NEED_MFE else None
        ss = np.zeros((R, L), dtype=np.uint8) if flags & NEED_MFE else None
        Ed = np.zeros((R, max(1, self.n_targets)), dtype=np.int32) if flags & NEED_EVAL else None
        ptr = lambda a: a.ctypes.data if a is not None else None
        self._check(self._L.drna_cofold_batch(self._h, R, L, cut, "".join(flat).encode("ascii"), flags, ptr(F4), ptr(Emfe),
                                              ptr(ss), ptr(Ed)))
        out = {"Emfe": Emfe, "Ed": Ed, "mfe_ss": None, "FA": None, "FB": None, "FcAB": None, "FAB": None}
        if ss is not None:
            out["mfe_ss"] = [bytes(r[:cut]).decode() + "&" + bytes(r[cut:]).decode() for r in ss]
        if F4 is not None:
            out.update(FA=F4[:, 0], FB=F4[:, 1], FcAB=F4[:, 2], FAB=F4[:, 3])
        return out

    TERM_IDS = {"Ed-Epf": 0, "1-MCC": 1, "sln_Epf": 2, "Ed-MFE": 3, "1-precision": 4, "1-recall": 5, "Edef": 6}

    def mc_run(self, prob, n_iter, shelf_index, n_shelves, tm_max, tm_min, targeted, temps, scoring_f, flags, rng_state, state,
               counters, best, L_const=504.12):
        """n_iter Monte-Carlo iterations of all replicas in native code (drna_mc_run).  `state` holds the arrays seqs, mfe_ss
        (uint8 R x L), score, mcc1, Epf, Ed (float64 R); `best` holds seq, ss (uint8 L) and vals (float64 4); all updated in place."""
        pk = getattr(prob, "_native_pack", None)
        if pk is None:
            HostKernels._pack(prob)
            pk = prob._native_pack
        am, partner, snake_of, off, nodes, nst, chars = pk
        R, L = state["seqs"].shape
        assert rng_state.dtype == np.uint32 and rng_state.shape == (R, RNG_WORDS) and rng_state.flags.c_contiguous
        ids = np.array([self.TERM_IDS[n] for n, _ in scoring_f], dtype=np.int32)
        ws = np.array([w for _, w in scoring_f], dtype=np.float64)
        sh = np.ascontiguousarray(shelf_index, dtype=np.int32)
        tt = np.ascontiguousarray(temps, dtype=np.float64)
        p = lambda a: a.ctypes.data
        self._check(self._L.drna_mc_run(self._h, R, L, int(n_iter), prob.sec_struct.encode("ascii"), p(partner), p(am), p(snake_of),
                                        len(prob.snakes), p(off), p(nodes), p(nst), p(chars), p(sh), int(n_shelves), float(tm_max),
                                        float(tm_min), int(bool(targeted)), p(tt), float(L_const), len(ids), p(ids), p(ws), int(flags),
                                        p(rng_state), p(state["seqs"]), p(state["mfe_ss"]), p(state["score"]), p(state["mcc1"]),
                                        p(state["Epf"]), p(state["Ed"]), p(counters), p(best["seq"]), p(best["ss"]), p(best["vals"])))

    def subopt_energy(self, seqs, want_both=False):
        """Energy (dcal/mol) of the second-best structure of each sequence as the reference's -nd on path takes it from
        ViennaRNA's subopt (0 if none within 49 kcal/mol); with want_both also the (R, 2) array of the two lowest energies."""
        R, L = len(seqs), len(seqs[0])
        if any(len(s) != L for s in seqs):
            raise ValueError("all sequences of a batch must have the same length")
        E2 = np.zeros(R, dtype=np.int32)
        E12 = np.zeros((R, 2), dtype=np.int32) if want_both else None
        self._check(self._L.drna_subopt_energy_batch(self._h, R, L, "".join(seqs).encode("ascii"), E2.ctypes.data,
                                                     E12.ctypes.data if want_both else None))
        return (E2, E12) if want_both else E2

    def subopt_structs(self, seqs, K):
        """The K (<= 8) lowest-energy structures of each sequence: (R, K) int32 energies in dcal/mol (ascending; 10000000 where
        a sequence has fewer structures) and a list of R lists of K dot-bracket strings.  Rank k is entry k of ViennaRNA's
        energy-sorted subopt list as get_first_suboptimal_structure_and_energy(seq, fc, k) indexes it (reference
        utils/energy_scores.py:453-488); the order among structures of equal energy is the engine's own."""
        R, L = len(seqs), len(seqs[0])
        if any(len(s) != L for s in seqs):
            raise ValueError("all sequences of a batch must have the same length")
        E = np.zeros((R, K), dtype=np.int32)
        ss = np.zeros((R, K, L), dtype=np.uint8)
        self._check(self._L.drna_subopt_structs_batch(self._h, R, L, "".join(seqs).encode("ascii"), int(K), E.ctypes.data,
                                                      ss.ctypes.data))
        raw = ss.tobytes().decode("ascii")
        return E, [[raw[(r * K + k) * L:(r * K + k + 1) * L] for k in range(K)] for r in range(R)]

    def ensemble_defect(self, seqs, want_bpp=False):
        """Ensemble defect of each sequence against targets[0] (reference ScoreSeq.get_ensemble_defect,
        utils/energy_scores.py:362-374).  Returns float64[R]; with want_bpp also the (R, L+1, L+1) base-pair
        probability matrices (1-based, upper triangle)."""
        R = len(seqs)
        L = len(seqs[0])
        if any(len(s) != L for s in seqs):
            raise ValueError("all sequences of a batch must have the same length")
        ed = np.zeros(R, dtype=np.float64)
        bpp = np.zeros((R, L + 1, L + 1), dtype=np.float64) if want_bpp else None
        self._check(self._L.drna_ensemble_defect_batch(self._h, R, L, "".join(seqs).encode("ascii"), ed.ctypes.data,
                                                       bpp.ctypes.data if want_bpp else None))
        return (ed, bpp) if want_bpp else ed

    def ensemble_defect_arrays(self, seqs_u8):
        """Array form of :meth:`ensemble_defect`: (R, L) uint8 ASCII letters -> float64[R]."""
        seqs_u8 = np.ascontiguousarray(seqs_u8, dtype=np.uint8)
        R, L = seqs_u8.shape
        ed = np.zeros(R, dtype=np.float64)
        self._check(self._L.drna_ensemble_defect_batch(self._h, R, L, seqs_u8.ctypes.data_as(C.c_char_p),
                                                       ed.ctypes.data, None))
        return ed

    def last_edef_timing(self):
        out = (C.c_float * 2)()
        self._check(self._L.drna_last_edef_timing(self._h, out))
        return {"inside": out[0], "outside": out[1]}

    def last_timing(self):
        """ms of device time of the last call: dict(mfe, pf, eval, total) from HIP events on the engine's streams."""
        out = (C.c_float * 4)()
        self._check(self._L.drna_last_timing(self._h, out))
        return {"mfe": out[0], "pf": out[1], "eval": out[2], "total": out[3]}

    def timing_sums(self, reset=False):
        """device ms summed over the score_batch[_device] calls since the last reset: dict(mfe, pf, eval, total, calls)"""
        out = (C.c_double * 5)()
        self._check(self._L.drna_timing_sums(self._h, out, int(bool(reset))))
        return {"mfe": out[0], "pf": out[1], "eval": out[2], "total": out[3], "calls": int(out[4])}

    def info(self):
        out = (C.c_int64 * 6)()
        self._check(self._L.drna_info(self._h, out))
        return {"device": out[0], "max_R": out[1], "max_L": out[2], "threads_per_wg": out[3],
                "compute_units": out[4], "workspace_bytes": out[5]}


class HostKernels:
    """Native batched host helpers of the MC inner loop (no GPU needed): SimScore, proposals, Metropolis."""

    def __init__(self, lib=None):
        self._L = load_library(lib)

    @staticmethod
    def _pack(prob):
        """arrays of a design.DesignProblem in the layout of drna_propose_batch_alt / drna_mc_run (cached on the problem)"""
        am = np.array([sum(1 << "ACGU".index(c) for c in a) for a in prob.allowed], dtype=np.uint8)
        partner = np.ascontiguousarray(prob.partner, dtype=np.int32)
        snake_of = np.ascontiguousarray(prob.snake_of, dtype=np.int32)
        off, nodes, nst, chars = [0], [], [], b""
        for nd, states in prob.snakes:
            nodes += list(nd)
            off.append(len(nodes))
            nst.append(len(states))
            chars += "".join(states).encode() + b"." * (len(nd) * (4 - len(states)))
        prob._native_pack = (am, partner, snake_of, np.array(off, dtype=np.int32), np.array(nodes or [0], dtype=np.int32),
                             np.array(nst or [0], dtype=np.int32), np.frombuffer(chars or b".", dtype=np.uint8).copy())

    def rng_seed(self, seeds, out=None):
        """One MT19937 stream per entry of `seeds`, seeded like ``random.seed(int)`` (the reference re-seeds every worker
        with its replica index at each exchange step, utils/replica_exchange_monte_carlo.py:227-228,250)."""
        sd = np.ascontiguousarray(seeds, dtype=np.uint64)
        st = out if out is not None else np.empty((sd.shape[0], RNG_WORDS), dtype=np.uint32)
        assert st.dtype == np.uint32 and st.shape == (sd.shape[0], RNG_WORDS) and st.flags.c_contiguous
        rc = self._L.drna_rng_seed(sd.shape[0], sd.ctypes.data, st.ctypes.data)
        if rc != 0:
            raise EngineError(rc, "drna_rng_seed")
        return st

    def rng_random(self, rng_state):
        """one ``random.random()`` from every stream"""
        out = np.empty(rng_state.shape[0])
        rc = self._L.drna_rng_random(rng_state.shape[0], rng_state.ctypes.data, out.ctypes.data)
        if rc != 0:
            raise EngineError(rc, "drna_rng_random")
        return out

    def simscore(self, ref, queries_u8):
        """ref: reference structure string ('&' -> 'Ee' already applied); queries_u8: (R, L) uint8.
        Returns rounded (mcc, recall, precision) arrays exactly as the reference's SimScore computes them."""
        q = np.ascontiguousarray(queries_u8, dtype=np.uint8)
        R, L = q.shape
        mcc, rec, prec = np.zeros(R), np.zeros(R), np.zeros(R)
        rc = self._L.drna_simscore_batch(R, L, ref.encode("ascii"), q.ctypes.data, mcc.ctypes.data, rec.ctypes.data,
                                         prec.ctypes.data)
        if rc != 0:
            raise EngineError(rc, "drna_simscore_batch")
        return mcc, rec, prec

    def propose(self, target, allowed_mask, seqs_u8, ss_u8, shelf_index, n_shelves, tm_max, tm_min, targeted, rng_state):
        s = np.ascontiguousarray(seqs_u8, dtype=np.uint8)
        R, L = s.shape
        out = np.empty_like(s)
        ss = np.ascontiguousarray(ss_u8, dtype=np.uint8)
        am = np.ascontiguousarray(allowed_mask, dtype=np.uint8)
        sh = np.ascontiguousarray(shelf_index, dtype=np.int32)
        assert rng_state.dtype == np.uint32 and rng_state.shape == (R, RNG_WORDS) and rng_state.flags.c_contiguous
        rc = self._L.drna_propose_batch(R, L, target.encode("ascii"), am.ctypes.data, s.ctypes.data, ss.ctypes.data,
                                        sh.ctypes.data, int(n_shelves), float(tm_max), float(tm_min), int(bool(targeted)),
                                        rng_state.ctypes.data, out.ctypes.data)
        if rc != 0:
            raise EngineError(rc, "drna_propose_batch")
        return out

    def propose_alt(self, prob, seqs_u8, ss_u8, shelf_index, n_shelves, tm_max, tm_min, targeted, rng_state):
        """Proposals for a ``design.DesignProblem`` that may hold alternative-structure snakes."""
        s = np.ascontiguousarray(seqs_u8, dtype=np.uint8)
        R, L = s.shape
        out = np.empty_like(s)
        ss = np.ascontiguousarray(ss_u8, dtype=np.uint8)
        pk = getattr(prob, "_native_pack", None)
        if pk is None:
            HostKernels._pack(prob)
            pk = prob._native_pack
        am, partner, snake_of, off, nodes, nst, chars = pk
        sh = np.ascontiguousarray(shelf_index, dtype=np.int32)
        assert rng_state.dtype == np.uint32 and rng_state.shape == (R, RNG_WORDS) and rng_state.flags.c_contiguous
        rc = self._L.drna_propose_batch_alt(R, L, prob.sec_struct.encode("ascii"), partner.ctypes.data, am.ctypes.data,
                                            snake_of.ctypes.data, len(prob.snakes), off.ctypes.data, nodes.ctypes.data,
                                            nst.ctypes.data, chars.ctypes.data, s.ctypes.data, ss.ctypes.data, sh.ctypes.data,
                                            int(n_shelves), float(tm_max), float(tm_min), int(bool(targeted)),
                                            rng_state.ctypes.data, out.ctypes.data)
        if rc != 0:
            raise EngineError(rc, "drna_propose_batch_alt")
        return out

    def metropolis(self, score_o, score_m, temps, rng_state, L_const=504.12):
        so = np.ascontiguousarray(score_o, dtype=np.float64)
        sm = np.ascontiguousarray(score_m, dtype=np.float64)
        tt = np.ascontiguousarray(temps, dtype=np.float64)
        R = so.shape[0]
        assert rng_state.dtype == np.uint32 and rng_state.shape == (R, RNG_WORDS) and rng_state.flags.c_contiguous
        acc = np.zeros(R, dtype=np.uint8)
        bet = np.zeros(R, dtype=np.uint8)
        rc = self._L.drna_metropolis_batch(R, so.ctypes.data, sm.ctypes.data, tt.ctypes.data, float(L_const),
                                           rng_state.ctypes.data, acc.ctypes.data, bet.ctypes.data)
        if rc != 0:
            raise EngineError(rc, "drna_metropolis_batch")
        return acc.astype(bool), bet.astype(bool)
