#!/bin/bash
# Runs on the GPU box via gpurun: smoke, parity tests, bench (with CPU baseline), rocprofv3 kernel-trace
# stats, and two PMC passes (FETCH_SIZE, WRITE_SIZE separately) for the HBM traffic of the fold kernels.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
STEPS=${STEPS:-50}
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== pytest gpu"; timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
echo "== bench"; timeout 300 python bench.py --steps $STEPS --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -2 gpurun_out/bench.err; cat gpurun_out/bench.json
echo "== bench design-like"; timeout 200 python bench.py --steps $STEPS --warmup 5 --seqs design --no-cpu-baseline > gpurun_out/bench_design.json 2>/dev/null; cat gpurun_out/bench_design.json
echo "== rocprof kernel trace"
rm -rf gpurun_out/prof; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o r1 -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps $STEPS --warmup 5 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/bench_prof.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof.err")
for f in $(find gpurun_out/prof -name "*kernel_stats*.csv"); do cat $f; done
echo "== PMC: HBM traffic"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$c" -o $c -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> "$GRAFT_REPO_ROOT/gpurun_out/pmc_$c.err")
done
python3 - <<'PY'
import csv, glob, json, collections
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            key = "mfe" if "mfe" in k else "pf" if "pf_" in k else "eval" if "eval" in k else None
            if key:
                out.setdefault(key, {})[c + "_KB_per_launch"] = sum(v) / len(v)
res = {}
for key, d in out.items():
    fe, wr = d.get("FETCH_SIZE_KB_per_launch", 0.0), d.get("WRITE_SIZE_KB_per_launch", 0.0)
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE / WRITE_SIZE are KB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B -> x2
    res["%s_L200_R64" % key] = (2.0 * fe + wr) * 1024.0
    res["%s_detail" % key] = d
json.dump(res, open("gpurun_out/hbm_traffic.json", "w"), indent=1)
print(json.dumps(res))
PY
