#!/bin/bash
# PMC passes (own runs, --pmc only with kernel-trace): instruction mix / stalls of the fold kernels
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
run() { # name counters...
  name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc/$name" -o $name -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> "$GRAFT_REPO_ROOT/gpurun_out/pmc/$name.err")
  tail -2 gpurun_out/pmc/$name.err
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
python3 - <<'PY'
import csv,glob,collections
for name in ("sq1","sq2"):
    for f in glob.glob("gpurun_out/pmc/%s/**/*counter_collection.csv"%name, recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:40]
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
        for k,v in agg.items():
            print(name,k,{a:round(b/4) for a,b in v.items()})
PY
