"""Nearest-neighbour energy parameters: ViennaRNA "v2.0" text file -> flat int32 blob.

Replaces ``RNA.params_load("rna_turner1999.par")`` (reference ``DesiRNA.py:455-456``,
SURVEY row a15).  The reference hands the text file to ViennaRNA, which keeps the
numbers in process-global tables; here the numbers travel as ONE flat little-endian
int32 array (the *blob*) that the C-ABI ``drna_create`` takes and that the engine
expands into device tables.  Only the 37 degC free energies are kept: the reference
never changes the folding temperature, and at T = T_measure ViennaRNA's enthalpy
rescaling is the identity.

Blob layout (all int32, C order, index 0 of every pair-type axis unused = 0):

    [0] magic 'DRNP'  [1] version  [2] total length in int32
    stack[8][8]
    mismatchH, mismatchI, mismatch1nI, mismatch23I, mismatchM, mismatchExt   each [8][5][5]  (RAW file values)
    dangle5[8][5], dangle3[8][5]                                              (RAW file values)
    int11[8][8][5][5]
    int21[8][8][5][5][5]
    int22[8][8][5][5][5][5]     (types 1..6 x nts 1..4 from the file; type 7 filled as max over 1..6)
    hairpin[31], bulge[31], interior[31]
    ninio, max_ninio, MLbase, MLclosing, MLintern, DuplexInit, TerminalAU
    lxc as IEEE double (2 x int32, little endian)
    n_tri, n_tetra, n_hexa
    per special loop: 8 bytes of NUL-padded sequence (2 x int32) + energy     (tri, then tetra, then hexa)

Pair types CG=1 GC=2 GU=3 UG=4 AU=5 UA=6 NS=7, nucleotides @=0 A=1 C=2 G=3 U=4
(SURVEY App. A.1 / App. B).
"""
import os
import struct

import numpy as np

INF = 10000000
DEF = -50
MAGIC = 0x504E5244  # b'DRNP' little endian
VERSION = 1

_DEFAULT_BLOB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "turner1999.drnp")

_SECTION_SHAPES = {
    "stack": (7, 7),
    "mismatch_hairpin": (7, 5, 5),
    "mismatch_interior": (7, 5, 5),
    "mismatch_interior_1n": (7, 5, 5),
    "mismatch_interior_23": (7, 5, 5),
    "mismatch_multi": (7, 5, 5),
    "mismatch_exterior": (7, 5, 5),
    "dangle5": (7, 5),
    "dangle3": (7, 5),
    "int11": (7, 7, 5, 5),
    "int21": (7, 7, 5, 5, 5),
    "int22": (6, 6, 4, 4, 4, 4),
    "hairpin": (31,),
    "bulge": (31,),
    "interior": (31,),
}


def _strip_comments(text):
    out = []
    i = 0
    while True:
        a = text.find("/*", i)
        if a < 0:
            out.append(text[i:])
            break
        out.append(text[i:a])
        b = text.find("*/", a + 2)
        if b < 0:
            break
        i = b + 2
    return " ".join(out)


def _tok_int(tok):
    if tok == "INF":
        return INF
    if tok == "DEF":
        return DEF
    return int(tok)


def parse_par_text(text):
    """Parse a ViennaRNA parameter file (format v2.0) into a dict of numpy arrays."""
    lines = text.splitlines()
    if not lines or "RNAfold parameter file v2.0" not in lines[0]:
        raise ValueError("not a ViennaRNA 'RNAfold parameter file v2.0'")
    sections = {}
    name = None
    buf = []
    for ln in lines[1:]:
        s = ln.strip()
        if s.startswith("#"):
            if name is not None:
                sections[name] = "\n".join(buf)
            name = s[1:].strip()
            buf = []
            if name == "END":
                break
        elif name is not None:
            buf.append(ln)
    if name is not None and name != "END":
        sections[name] = "\n".join(buf)

    P = {}
    for sec, shape in _SECTION_SHAPES.items():
        if sec not in sections:
            raise ValueError("parameter file lacks section '%s'" % sec)
        toks = _strip_comments(sections[sec]).split()
        want = int(np.prod(shape))
        if len(toks) < want:
            raise ValueError("section '%s': %d values, expected %d" % (sec, len(toks), want))
        P[sec] = np.array([_tok_int(t) for t in toks[:want]], dtype=np.int64).reshape(shape)

    t = _strip_comments(sections["NINIO"]).split()
    P["ninio"], P["max_ninio"] = int(t[0]), int(t[2])
    t = _strip_comments(sections["ML_params"]).split()
    P["MLbase"], P["MLclosing"], P["MLintern"] = int(t[0]), int(t[2]), int(t[4])
    t = _strip_comments(sections["Misc"]).split()
    P["DuplexInit"], P["TerminalAU"], P["lxc"] = int(t[0]), int(t[2]), float(t[4])
    for sec, ln in (("Triloops", 5), ("Tetraloops", 6), ("Hexaloops", 8)):
        loops = []
        for row in _strip_comments(sections.get(sec, "")).splitlines():
            f = row.split()
            if len(f) >= 2:
                if len(f[0]) != ln:
                    raise ValueError("%s entry '%s' has wrong length" % (sec, f[0]))
                loops.append((f[0], int(f[1])))
        P[sec] = loops
    return P


def _place(shape_full, arr, offs):
    out = np.zeros(shape_full, dtype=np.int64)
    sl = tuple(slice(o, o + s) for o, s in zip(offs, arr.shape))
    out[sl] = arr
    return out


def build_blob(P):
    """dict from :func:`parse_par_text` -> flat int32 blob (layout in the module docstring)."""
    parts = [np.zeros(3, dtype=np.int64)]
    parts.append(_place((8, 8), P["stack"], (1, 1)).ravel())
    for k in ("mismatch_hairpin", "mismatch_interior", "mismatch_interior_1n",
              "mismatch_interior_23", "mismatch_multi", "mismatch_exterior"):
        parts.append(_place((8, 5, 5), P[k], (1, 0, 0)).ravel())
    parts.append(_place((8, 5), P["dangle5"], (1, 0)).ravel())
    parts.append(_place((8, 5), P["dangle3"], (1, 0)).ravel())
    parts.append(_place((8, 8, 5, 5), P["int11"], (1, 1, 0, 0)).ravel())
    parts.append(_place((8, 8, 5, 5, 5), P["int21"], (1, 1, 0, 0, 0)).ravel())
    i22 = _place((8, 8, 5, 5, 5, 5), P["int22"], (1, 1, 1, 1, 1, 1))
    # non-standard pair type 7: maximum over the six canonical types (ViennaRNA fills these
    # after reading; only eval_structure of a non-canonical pair closing a 2x2 loop sees them)
    i22[1:7, 7] = i22[1:7, 1:7].max(axis=1)
    i22[7, 1:7] = i22[1:7, 1:7].max(axis=0)
    i22[7, 7] = i22[1:7, 1:7].max(axis=(0, 1))
    i22[:, :, 0] = 0
    i22[:, :, :, 0] = 0
    i22[:, :, :, :, 0] = 0
    i22[:, :, :, :, :, 0] = 0
    parts.append(i22.ravel())
    parts.append(P["hairpin"].ravel())
    parts.append(P["bulge"].ravel())
    parts.append(P["interior"].ravel())
    parts.append(np.array([P["ninio"], P["max_ninio"], P["MLbase"], P["MLclosing"], P["MLintern"],
                           P["DuplexInit"], P["TerminalAU"]], dtype=np.int64))
    lo, hi = struct.unpack("<ii", struct.pack("<d", P["lxc"]))
    parts.append(np.array([lo, hi], dtype=np.int64))
    parts.append(np.array([len(P["Triloops"]), len(P["Tetraloops"]), len(P["Hexaloops"])], dtype=np.int64))
    for sec in ("Triloops", "Tetraloops", "Hexaloops"):
        for s, e in P[sec]:
            a, b = struct.unpack("<ii", s.encode("ascii").ljust(8, b"\0"))
            parts.append(np.array([a, b, e], dtype=np.int64))
    blob = np.concatenate(parts).astype(np.int32)
    blob[0] = MAGIC
    blob[1] = VERSION
    blob[2] = blob.size
    return blob


def load_par_file(path):
    """``RNA.params_load(path)`` counterpart: text parameter file -> blob."""
    with open(path, "r") as fh:
        return build_blob(parse_par_text(fh.read()))


def load_blob(path=None):
    """Load a blob written by :func:`save_blob` (default: the shipped Turner-1999 set)."""
    blob = np.fromfile(path or _DEFAULT_BLOB, dtype="<i4")
    if blob.size < 3 or blob[0] != MAGIC or blob[1] != VERSION or blob[2] != blob.size:
        raise ValueError("bad parameter blob: %s" % (path or _DEFAULT_BLOB))
    return np.ascontiguousarray(blob.astype(np.int32))


def save_blob(blob, path):
    np.asarray(blob, dtype="<i4").tofile(path)


def load_params(spec="1999"):
    """Resolve the reference's ``-p`` option (``DesiRNA.py:163,455-456``).

    '1999' -> shipped Turner-1999 blob; a path -> parse that file.  '2004' is ViennaRNA's
    compiled-in default set, which is not part of the reference tree and cannot be reproduced.
    """
    if spec in ("1999", 1999, None):
        return load_blob()
    if spec in ("2004", 2004):
        raise NotImplementedError(
            "Turner-2004 is ViennaRNA's built-in table and is not in the reference tree; "
            "pass a parameter file path instead")
    return load_par_file(spec)
