#!/bin/bash
# GPU box: instruction counts per sweep phase = PMC pass over builds with one phase left out (built beforehand into build/v/)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
REPO="$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for lib in build/v/*.so; do
  name=$(basename $lib .so)
  rm -rf gpurun_out/ppmc_$name
  (cd /tmp && timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d "$REPO/gpurun_out/ppmc_$name" -o p -- python3 "$REPO/tools/run_lib.py" "$REPO/$lib" > "$REPO/gpurun_out/ppmc_$name.out" 2>&1) || { tail -3 gpurun_out/ppmc_$name.out; exit 1; }
  tail -1 gpurun_out/ppmc_$name.out
done
python3 - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob("gpurun_out/ppmc_*")):
    if not os.path.isdir(d): continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = "mfe" if "mfe" in r["Kernel_Name"] else "pf" if "pf_" in r["Kernel_Name"] else None
            if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(os.path.basename(d), k, {c: round(sum(x) / len(x) / 64) for c, x in sorted(v.items())})
PY
