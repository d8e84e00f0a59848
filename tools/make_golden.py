#!/usr/bin/env python3
"""Extract golden vectors from DATA files committed in the reference tree (never its code).

    python tools/make_golden.py /root/reference tests/golden

Sources:
  example_files/outputs/*/trajectory_files/*_traj.csv   per-sequence Epf (legacy column 'mfe', float32),
        E(target) ('edesired'), MFE / pk-annotated structure ('mfe_ss'), 1-MCC / 1-recall / 1-precision,
        mean E(alt structures) ('edesired2'); produced by the reference with ViennaRNA + Turner-1999
  example_files/inputs/*.txt                            the six example design inputs (targets)
  eterna_benchmark/Eterna100V1_benchmark_results/Eterna100V1_all_results.txt   MFE(seq) == structure
  eterna_benchmark/Eterna100V1_inputs/*.txt             target structures (bench workloads use #69/#92/#53)
Outputs: traj_golden.csv.gz, eterna_v1_solutions.csv, eterna_v1_targets.csv, example_inputs.json
"""
import csv
import glob
import gzip
import json
import os
import sys


def read_input(path):
    d = {}
    key = None
    for ln in open(path):
        ln = ln.strip()
        if not ln:
            continue
        if ln.startswith(">"):
            key = ln[1:]
            d[key] = []
        elif key:
            d[key].append(ln)
    return d


def main():
    ref, out = sys.argv[1], sys.argv[2]
    os.makedirs(out, exist_ok=True)
    runs = {}
    inputs = {}
    for p in sorted(glob.glob(os.path.join(ref, "example_files/inputs/*.txt"))):
        d = read_input(p)
        inputs[os.path.basename(p)[:-4]] = d
    json.dump(inputs, open(os.path.join(out, "example_inputs.json"), "w"), indent=1, sort_keys=True)

    rows = []
    for d in sorted(glob.glob(os.path.join(ref, "example_files/outputs/*/"))):
        base = os.path.basename(d.rstrip("/"))
        run = base.split("_R10_")[0]
        pk = 1 if "_pkon_" in base else 0
        traj = glob.glob(os.path.join(d, "trajectory_files/*_traj.csv"))[0]
        seen = set()
        for r in csv.DictReader(open(traj)):
            if r["sequence"] in seen:
                continue
            seen.add(r["sequence"])
            rows.append([run, pk, r["sequence"], r["mfe"], r["edesired"], r["mfe_ss"], r["mcc"],
                         r["recall"], r["precision"], r["edesired2"]])
        runs[run] = len(seen)
    with gzip.open(os.path.join(out, "traj_golden.csv.gz"), "wt", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["run", "pk_on", "sequence", "Epf", "edesired", "mfe_ss", "one_minus_mcc",
                    "one_minus_recall", "one_minus_precision", "edesired2"])
        w.writerows(rows)
    print("trajectory rows:", runs)

    sol = []
    for ln in open(os.path.join(ref, "eterna_benchmark/Eterna100V1_benchmark_results/Eterna100V1_all_results.txt")):
        ln = ln.strip()
        if ln.startswith(">"):
            name, seq, ss = ln[1:].split(",")[:3]
            sol.append([name, seq, ss])
    with open(os.path.join(out, "eterna_v1_solutions.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["name", "sequence", "structure"])
        w.writerows(sol)
    tg = []
    for p in sorted(glob.glob(os.path.join(ref, "eterna_benchmark/Eterna100V1_inputs/*.txt"))):
        d = read_input(p)
        tg.append([os.path.basename(p), d["sec_struct"][0]])
    with open(os.path.join(out, "eterna_v1_targets.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["name", "structure"])
        w.writerows(tg)
    print("eterna solutions:", len(sol), "targets:", len(tg))


if __name__ == "__main__":
    main()
