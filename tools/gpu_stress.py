"""GPU box: randomized parity sweep of the production kernels against the CPU oracle (lengths 1 ... 200 for the LDS-resident
path, a few longer ones for the general path; uniform, GC-rich, AU-rich and low-complexity sequences)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from desirna_amd import engine as E, params  # noqa: E402
from oracle.pyoracle import Oracle, FLAG_PF, FLAG_MFE  # noqa: E402

orc = Oracle(params.load_blob())
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
eng = E.Engine(max_R=64, max_L=420, device=0)
alph = ["ACGU", "GC", "GCGCAU", "AU", "GGGGGGCCCCCCAAAU", "ACGUACGUGGCC"]
bad = 0
t0 = time.time()
lengths = list(range(1, 201)) + [201, 222, 256, 257, 300, 333, 400, 420]
for L in lengths:
    R = 12 if L <= 200 else 4
    seqs = ["".join(rng.choice(list(alph[k % len(alph)]), L)) for k in range(R)]
    tg = orc.mfe(seqs[0])[0]
    eng.set_targets([tg])
    out = eng.score_batch(seqs)
    rEpf, rEmfe, rss, rEd = orc.score_batch(seqs, [tg], FLAG_PF | FLAG_MFE, threads=0)
    for k, s in enumerate(seqs):
        ok = (out["mfe_ss"][k] == rss[k] and int(out["Emfe"][k]) == int(rEmfe[k]) and
              abs(float(out["Epf"][k]) - float(rEpf[k])) < 1e-9 and int(out["Ed"][k, 0]) == int(rEd[k, 0]))
        if not ok:
            bad += 1
            print("MISMATCH L=%d %s\n  gpu %s %d %.9f %d\n  cpu %s %d %.9f %d" % (
                L, s, out["mfe_ss"][k], out["Emfe"][k], out["Epf"][k], out["Ed"][k, 0], rss[k], rEmfe[k], rEpf[k], rEd[k, 0]))
print("lengths %d, sequences checked, mismatches %d, %.1f s" % (len(lengths), bad, time.time() - t0))

# ---- two strands, second-best energy, ensemble defect
t0 = time.time()
nco = nsub = ned = 0
for trial in range(60):
    la, lb = int(rng.integers(1, 70)), int(rng.integers(1, 70))
    al = alph[trial % len(alph)]
    seqs = ["".join(rng.choice(list(al), la)) + "&" + "".join(rng.choice(list(al), lb)) for _ in range(5)]
    if la == lb or trial % 7 == 0:
        a = "".join(rng.choice(list(al), la))
        seqs = [s.split("&")[0] + "&" + s.split("&")[0] for s in seqs[:3]] + seqs[3:] if la == lb else seqs
    tg = "." * (la + lb)
    eng.set_targets([tg])
    out = eng.cofold_batch(seqs)
    for k, s in enumerate(seqs):
        oss, oe = orc.cofold_mfe(s)
        of = orc.cofold_pf(s)
        got = [float(out[x][k]) for x in ("FA", "FB", "FcAB", "FAB")]
        if out["mfe_ss"][k] != oss or int(out["Emfe"][k]) != oe or max(abs(g - o) for g, o in zip(got, of)) > 1e-9:
            nco += 1
            print("COFOLD MISMATCH", s, out["mfe_ss"][k], oss, int(out["Emfe"][k]), oe, got, of)
for L in (7, 19, 44, 90, 150, 230):
    seqs = ["".join(rng.choice(list(alph[k % len(alph)]), L)) for k in range(8)]
    E2, E12 = eng.subopt_energy(seqs, want_both=True)
    tg = orc.mfe(seqs[0])[0]
    eng.set_targets([tg])
    ed = eng.ensemble_defect(seqs)
    for k, s in enumerate(seqs):
        if tuple(int(x) for x in E12[k]) != orc.two_best(s) or int(E2[k]) != orc.subopt_energy(s):
            nsub += 1
            print("SUBOPT MISMATCH", s, E12[k], orc.two_best(s))
        if abs(ed[k] - orc.ensemble_defect(s, tg)) > 1e-10:
            ned += 1
            print("EDEF MISMATCH", s, ed[k], orc.ensemble_defect(s, tg))
print("cofold mismatches %d, subopt %d, edef %d, %.1f s" % (nco, nsub, ned, time.time() - t0))
sys.exit(1 if (bad or nco or nsub or ned) else 0)
