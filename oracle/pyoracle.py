"""ctypes loader for the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product package desirna_amd/ never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

FLAG_PF, FLAG_MFE, FLAG_PK, FLAG_PIN = 1, 2, 4, 8      # FLAG_PIN: bind worker threads to cores for this call (bench.py only)


def build(force=False):
    if force or not os.path.exists(_LIB) or (
            os.path.getmtime(_LIB) < max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("oracle.c", "oracle.h"))):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return _LIB


def build_native():
    """bench.py's cpu_baseline leg: the same source built on THIS host with -march=native (liboracle_native.so, never shipped:
    the dev container's CPU may differ from the GPU box's).  Returns the path, or None when the compiler refuses."""
    out = os.path.join(_HERE, "liboracle_native.so")
    try:
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-std=gnu11", "-shared", "-o", out,
                               os.path.join(_HERE, "oracle.c"), "-lm"], stderr=subprocess.DEVNULL)
        return out
    except Exception:
        return None


class Oracle:
    def __init__(self, blob, lib=None):
        if lib is None and not os.path.exists(_LIB):
            build()
        L = C.CDLL(lib or _LIB)
        self._L = L
        L.orc_params_create.restype = C.c_void_p
        L.orc_params_create.argtypes = [C.c_void_p, C.c_int]
        L.orc_params_destroy.argtypes = [C.c_void_p]
        L.orc_eval_structure.restype = C.c_int
        L.orc_eval_structure.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int]
        L.orc_eval_structure_cut.restype = C.c_int
        L.orc_eval_structure_cut.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int]
        L.orc_mfe.restype = C.c_int
        L.orc_mfe.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_char_p]
        L.orc_pf.restype = C.c_double
        L.orc_pf.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.orc_pk_struct.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p]
        L.orc_ensemble_defect.restype = C.c_double
        L.orc_ensemble_defect.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_void_p]
        L.orc_simscore.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_score_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_char_p,
                                      C.c_uint, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_mfe_tables.restype = C.c_int
        L.orc_mfe_tables.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        blob = np.ascontiguousarray(blob, dtype=np.int32)
        self._P = L.orc_params_create(blob.ctypes.data, blob.size)
        if not self._P:
            raise ValueError("oracle rejected the parameter blob")

    def __del__(self):
        try:
            if getattr(self, "_P", None):
                self._L.orc_params_destroy(self._P)
                self._P = None
        except Exception:
            pass

    def eval_structure(self, seq, db, cut=0):
        db = db.replace("&", "")
        seq = seq.replace("&", "")
        return self._L.orc_eval_structure_cut(self._P, seq.encode(), db.encode(), len(seq), cut)

    def mfe(self, seq, nopair=None):
        n = len(seq)
        buf = C.create_string_buffer(n + 1)
        mp = None
        if nopair is not None:
            m = np.ascontiguousarray(nopair, dtype=np.uint8)
            mp = m.ctypes.data
        e = self._L.orc_mfe(self._P, seq.encode(), n, mp, buf)
        return buf.value.decode(), e

    def mfe_tables(self, seq, nopair=None):
        n = len(seq)
        c = np.empty((n + 2, n + 2), dtype=np.int32)
        f = np.empty((n + 2, n + 2), dtype=np.int32)
        f5 = np.empty(n + 1, dtype=np.int32)
        mp = None
        if nopair is not None:
            m = np.ascontiguousarray(nopair, dtype=np.uint8)
            mp = m.ctypes.data
        self._L.orc_mfe_tables(self._P, seq.encode(), n, mp, c.ctypes.data, f.ctypes.data, f5.ctypes.data)
        return c, f, f5

    def pf(self, seq):
        return self._L.orc_pf(self._P, seq.encode(), len(seq))

    def pk_struct(self, seq, ss_nopk):
        n = len(seq)
        buf = C.create_string_buffer(n + 1)
        self._L.orc_pk_struct(self._P, seq.encode(), n, ss_nopk.encode(), buf)
        return buf.value.decode()

    def ensemble_defect(self, seq, target, want_bpp=False):
        n = len(seq)
        bpp = np.zeros((n + 1, n + 1)) if want_bpp else None
        ed = self._L.orc_ensemble_defect(self._P, seq.encode(), n, target.encode(),
                                         bpp.ctypes.data if want_bpp else None)
        return (ed, bpp) if want_bpp else ed

    def boltzmann_weight(self, seq, db):
        """exp(-E/kT) of one structure under the partition function's loop model (no pf_scale)"""
        self._L.orc_boltzmann_weight.restype = C.c_double
        self._L.orc_boltzmann_weight.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int]
        return self._L.orc_boltzmann_weight(self._P, seq.encode(), db.encode(), len(seq))

    def cofold_mfe(self, seq_with_amp):
        """fc.mfe_dimer(): (structure with '&' re-inserted, energy in dcal/mol)"""
        a, b = seq_with_amp.split("&")
        s = a + b
        buf = C.create_string_buffer(len(s) + 1)
        self._L.orc_cofold_mfe.restype = C.c_int
        self._L.orc_cofold_mfe.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p]
        e = self._L.orc_cofold_mfe(self._P, s.encode(), len(s), len(a), buf)
        ss = buf.value.decode()
        return ss[:len(a)] + "&" + ss[len(a):], e

    def cofold_pf(self, seq_with_amp):
        """fc.pf_dimer()[1:]: (FA, FB, FcAB, FAB) in kcal/mol"""
        a, b = seq_with_amp.split("&")
        s = a + b
        out = (C.c_double * 4)()
        self._L.orc_cofold_pf.restype = None
        self._L.orc_cofold_pf.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p]
        self._L.orc_cofold_pf(self._P, s.encode(), len(s), len(a), out)
        return tuple(out)

    def two_best(self, seq):
        """the two lowest structure energies in dcal/mol (second = 10000000 if there is only one structure)"""
        e = (C.c_int * 2)()
        self._L.orc_two_best.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p]
        self._L.orc_two_best(self._P, seq.encode(), len(seq), e)
        return e[0], e[1]

    def subopt_energy(self, seq):
        self._L.orc_subopt_energy.restype = C.c_int
        self._L.orc_subopt_energy.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        return self._L.orc_subopt_energy(self._P, seq.encode(), len(seq))

    def simscore(self, ref, query):
        out = (C.c_double * 3)()
        conf = (C.c_int * 4)()
        assert len(ref) == len(query)
        self._L.orc_simscore(ref.encode(), query.encode(), len(ref), out, conf)
        return (out[0], out[1], out[2]), tuple(conf)

    def score_batch(self, seqs, targets, flags=FLAG_PF | FLAG_MFE, threads=0):
        """seqs: list of equal-length strings; targets: list of dot-bracket strings (same length)."""
        R, L = len(seqs), len(seqs[0])
        sb = "".join(seqs).encode()
        tb = "".join(targets).encode()
        Epf = np.zeros(R)
        Emfe = np.zeros(R, dtype=np.int32)
        ss = np.zeros((R, L), dtype=np.uint8)
        Ed = np.zeros((R, max(1, len(targets))), dtype=np.int32)
        self._L.orc_score_batch(self._P, R, L, sb, len(targets), tb, flags, threads,
                                Epf.ctypes.data, Emfe.ctypes.data, ss.ctypes.data, Ed.ctypes.data)
        return Epf, Emfe, [bytes(r).decode() for r in ss], Ed
