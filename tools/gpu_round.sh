#!/bin/bash
# Runs on the GPU box via gpurun: parity tests, smoke, bench, rocprof kernel trace.  Outputs -> gpurun_out/
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
STEPS=${STEPS:-30}
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
echo "== bench"; timeout 900 python bench.py --steps $STEPS --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -3 gpurun_out/bench.err; cat gpurun_out/bench.json
echo "== rocprof"; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o r1 -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps $STEPS --warmup 3 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/bench_prof.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof.err"); tail -3 gpurun_out/prof.err
find gpurun_out/prof -name "*stats*" | head; for f in $(find gpurun_out/prof -name "*kernel_stats*.csv"); do head -12 $f; done
