#!/bin/bash
# GPU box: HBM-side bytes of the partition function at R = 256 x L = 200 (one workgroup per fold, no helper) for every prebuilt
# variant in build/var/: (2 FETCH_SIZE + WRITE_SIZE) KB per launch, as the guide prescribes (separate passes per counter).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for lib in build/var/lib_*.so; do
  name=$(basename $lib .so)
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/ft_$name_$c
    (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/ft_${name}_$c" -o x -- python3 "$GRAFT_REPO_ROOT/tools/run_lib.py" "$GRAFT_REPO_ROOT/$lib" ${FT_R:-256} > /dev/null 2>&1)
  done
  python3 - <<PY
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/ft_${name}_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0][-40:]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print("$name", c, k, "avg KB per launch %.0f" % (sum(v) / len(v)), "launches", len(v))
PY
done
