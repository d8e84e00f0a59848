"""ctypes loader for the TEST-ONLY CPU emulation of the HIP kernels (tests/emu/)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libemu.so")
_CSRC = os.path.join(_HERE, "..", "..", "desirna_amd", "csrc")


def build(flags=(), tag=""):
    """flags: extra -D options (a build of its own, libemu<tag>.so): experimental code paths of the kernels"""
    lib = _LIB if not tag else _LIB.replace(".so", tag + ".so")
    srcs = [os.path.join(_HERE, f) for f in ("emu_kernels.cpp", "hip_emu.h", "hip_emu_prims.h")]
    srcs += [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith(".hpp")]
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I", _HERE] + list(flags) + ["-o", lib,
                               os.path.join(_HERE, "emu_kernels.cpp")])
    L = C.CDLL(lib)
    vp, ci = C.c_void_p, C.c_int
    L.emu_mfe.argtypes = [vp, ci, ci, ci, C.c_char_p, ci, ci, vp, vp, vp, vp, vp]
    L.emu_pf.argtypes = [vp, ci, ci, ci, C.c_char_p, ci, vp, vp]
    L.emu_mfe_dual.argtypes = [vp, ci, ci, ci, C.c_char_p, ci, ci, ci, vp, vp, vp]
    L.emu_pf_strip.argtypes = [vp, ci, ci, ci, C.c_char_p, ci, ci, ci, vp, vp]
    L.emu_mfe_strip.argtypes = [vp, ci, ci, ci, C.c_char_p, ci, ci, ci, ci, vp, vp, vp, ci]
    L.emu_eval.argtypes = [vp, ci, ci, ci, C.c_char_p, ci, vp, vp]
    L.emu_ragged.argtypes = [vp, ci, ci, ci, vp, vp, C.c_char_p, ci, vp, vp, vp, vp]
    L.emu_cofold.argtypes = [vp, ci, ci, ci, ci, C.c_char_p, ci, vp, vp, vp, vp, vp, vp]
    L.emu_subopt.argtypes = [vp, ci, ci, ci, C.c_char_p, ci, vp, vp, vp]
    L.emu_kbest.argtypes = [vp, ci, ci, ci, C.c_char_p, ci, ci, vp, vp, vp]
    L.emu_edef.argtypes = [vp, ci, ci, ci, C.c_char_p, vp, ci, vp, vp, vp]
    return L


class Emu:
    def __init__(self, blob, flags=(), tag=""):
        self.L = build(flags, tag)
        self.blob = np.ascontiguousarray(blob, dtype=np.int32)

    def mfe(self, seqs, pk_rounds=0, nt=128, dump=False):
        R, L = len(seqs), len(seqs[0])
        sb = "".join(seqs).encode()
        E = np.zeros(R, dtype=np.int32)
        ss = np.zeros((R, L), dtype=np.uint8)
        st = np.zeros(R, dtype=np.int32)
        ld = L + 2
        Wc = np.zeros((ld, ld), dtype=np.int32) if dump else None
        F = np.zeros((ld, ld), dtype=np.int32) if dump else None
        rc = self.L.emu_mfe(self.blob.ctypes.data, self.blob.size, R, L, sb, pk_rounds, nt, E.ctypes.data,
                            ss.ctypes.data, st.ctypes.data, Wc.ctypes.data if dump else None,
                            F.ctypes.data if dump else None)
        assert rc == 0
        out = (E, [bytes(r).decode() for r in ss], st)
        return out + (Wc, F) if dump else out

    def mfe_dual(self, seqs, pk_rounds=0, nt=256, calls=1):
        """two-workgroup MFE kernel (main + helper side by side); returns (Emfe, structures, status)"""
        R, L = len(seqs), len(seqs[0])
        E = np.zeros(R, dtype=np.int32)
        ss = np.zeros((R, L), dtype=np.uint8)
        st = np.zeros(R, dtype=np.int32)
        rc = self.L.emu_mfe_dual(self.blob.ctypes.data, self.blob.size, R, L, "".join(seqs).encode(), pk_rounds, nt, calls,
                                 E.ctypes.data, ss.ctypes.data, st.ctypes.data)
        assert rc == 0
        return E, [bytes(r).decode() for r in ss], st

    def pf(self, seqs, nt=128):
        R, L = len(seqs), len(seqs[0])
        E = np.zeros(R)
        st = np.zeros(R, dtype=np.int32)
        rc = self.L.emu_pf(self.blob.ctypes.data, self.blob.size, R, L, "".join(seqs).encode(), nt,
                           E.ctypes.data, st.ctypes.data)
        assert rc == 0
        return E, st

    def mfe_strip(self, seqs, S, pk_rounds=0, nt=256, calls=1, fark=0):
        """MFE fold by S strips of columns per sequence (+ the traceback launch per round); returns (Emfe, structures, status)"""
        R, L = len(seqs), len(seqs[0])
        E = np.zeros(R, dtype=np.int32)
        ss = np.zeros((R, L), dtype=np.uint8)
        st = np.zeros(R, dtype=np.int32)
        rc = self.L.emu_mfe_strip(self.blob.ctypes.data, self.blob.size, R, L, "".join(seqs).encode(), pk_rounds, nt, S, calls,
                                  E.ctypes.data, ss.ctypes.data, st.ctypes.data, int(fark))
        assert rc == 0
        return E, [bytes(r).decode() for r in ss], st

    def pf_strip(self, seqs, S, nt=256, calls=1):
        """partition function by S strips of columns, one workgroup each, side by side; returns (Epf, status)"""
        R, L = len(seqs), len(seqs[0])
        E = np.zeros(R)
        st = np.zeros(R, dtype=np.int32)
        rc = self.L.emu_pf_strip(self.blob.ctypes.data, self.blob.size, R, L, "".join(seqs).encode(), nt, S, calls,
                                 E.ctypes.data, st.ctypes.data)
        assert rc == 0
        return E, st

    def eval(self, seqs, targets):
        R, L = len(seqs), len(seqs[0])
        pt = np.zeros((len(targets), L + 2), dtype=np.int16)
        for k, t in enumerate(targets):
            stk = []
            for i, ch in enumerate(t, 1):
                if ch == "(":
                    stk.append(i)
                elif ch == ")":
                    o = stk.pop()
                    pt[k, o] = i
                    pt[k, i] = o
        Ed = np.zeros((R, len(targets)), dtype=np.int32)
        rc = self.L.emu_eval(self.blob.ctypes.data, self.blob.size, R, L, "".join(seqs).encode(), len(targets),
                             pt.ctypes.data, Ed.ctypes.data)
        assert rc == 0
        return Ed

    def edef(self, seqs, target, nt=128, bpp=False):
        """general pf_kernel + outside_kernel: ensemble defect against `target` (and the bpp matrices)"""
        R, L = len(seqs), len(seqs[0])
        pt = np.zeros(L + 2, dtype=np.int16)
        stk = []
        for i, ch in enumerate(target, 1):
            if ch == "(":
                stk.append(i)
            elif ch == ")":
                o = stk.pop()
                pt[o] = i
                pt[i] = o
        ed = np.zeros(R)
        st = np.zeros(R, dtype=np.int32)
        B = np.zeros((R, L + 1, L + 1)) if bpp else None
        rc = self.L.emu_edef(self.blob.ctypes.data, self.blob.size, R, L, "".join(seqs).encode(), pt.ctypes.data, nt,
                             ed.ctypes.data, B.ctypes.data if bpp else None, st.ctypes.data)
        assert rc == 0
        return (ed, st, B) if bpp else (ed, st)

    def ragged(self, seqs, lds=False):
        """sequences of different lengths through one ragged 'launch' of the MFE and PF kernels"""
        R = len(seqs)
        lens = np.array([len(s) for s in seqs], dtype=np.int32)
        offs = np.concatenate(([0], np.cumsum(lens)[:-1])).astype(np.int32)
        total = int(lens.sum())
        E = np.zeros(R, dtype=np.int32)
        ss = np.zeros(total, dtype=np.uint8)
        Ep = np.zeros(R)
        st = np.zeros(2 * R, dtype=np.int32)
        rc = self.L.emu_ragged(self.blob.ctypes.data, self.blob.size, R, int(lens.max()), lens.ctypes.data, offs.ctypes.data,
                               "".join(seqs).encode(), int(lds), E.ctypes.data, ss.ctypes.data, Ep.ctypes.data, st.ctypes.data)
        assert rc == 0
        b = ss.tobytes().decode()
        return E, [b[offs[k]:offs[k] + lens[k]] for k in range(R)], Ep, st

    def cofold(self, seqs, target=None, nt=128):
        """two strands ('AAA&BBB'): co-fold MFE + PF kernels, and E(target) by the eval kernel with the nick"""
        a0, b0 = seqs[0].split("&")
        cut, L, R = len(a0), len(a0) + len(b0), len(seqs)
        flat = "".join(s.replace("&", "") for s in seqs)
        E = np.zeros(R, dtype=np.int32)
        ss = np.zeros((R, L), dtype=np.uint8)
        F4 = np.zeros((R, 4))
        st = np.zeros(2 * R, dtype=np.int32)
        pt = Ed = None
        if target is not None:
            pt = np.zeros(L + 2, dtype=np.int16)
            stk = []
            for i, ch in enumerate(target.replace("&", ""), 1):
                if ch == "(":
                    stk.append(i)
                elif ch == ")":
                    o = stk.pop()
                    pt[o] = i
                    pt[i] = o
            Ed = np.zeros(R, dtype=np.int32)
        rc = self.L.emu_cofold(self.blob.ctypes.data, self.blob.size, R, L, cut, flat.encode(), nt, E.ctypes.data, ss.ctypes.data,
                               F4.ctypes.data, st.ctypes.data, pt.ctypes.data if pt is not None else None,
                               Ed.ctypes.data if Ed is not None else None)
        assert rc == 0
        strs = [bytes(r[:cut]).decode() + "&" + bytes(r[cut:]).decode() for r in ss]
        return E, strs, F4, st, Ed

    def subopt(self, seqs, nt=128):
        """second-best structure energies: (E2 as the reference takes it, the two lowest energies)"""
        R, L = len(seqs), len(seqs[0])
        E2 = np.zeros(R, dtype=np.int32)
        E12 = np.zeros((R, 2), dtype=np.int32)
        st = np.zeros(R, dtype=np.int32)
        rc = self.L.emu_subopt(self.blob.ctypes.data, self.blob.size, R, L, "".join(seqs).encode(), nt, E2.ctypes.data,
                               E12.ctypes.data, st.ctypes.data)
        assert rc == 0
        return E2, E12, st

    def kbest(self, seqs, K, nt=128):
        """K lowest-energy structures: (R, K) energies, R lists of K strings, status"""
        R, L = len(seqs), len(seqs[0])
        E = np.zeros((R, K), dtype=np.int32)
        ss = np.zeros((R, K, L), dtype=np.uint8)
        st = np.zeros(R, dtype=np.int32)
        rc = self.L.emu_kbest(self.blob.ctypes.data, self.blob.size, R, L, "".join(seqs).encode(), nt, K, E.ctypes.data,
                              ss.ctypes.data, st.ctypes.data)
        assert rc == 0
        raw = ss.tobytes().decode("ascii")
        return E, [[raw[(r * K + k) * L:(r * K + k + 1) * L] for k in range(K)] for r in range(R)], st
