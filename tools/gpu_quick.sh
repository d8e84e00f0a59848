#!/bin/bash
# quick GPU iteration: parity tests + bench without the CPU baseline
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
echo "== pytest gpu"; timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
echo "== bench"; timeout 120 python bench.py --steps ${STEPS:-30} --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value',round(d['value']),'ms/step',round(d['ms_per_step'],3),d['kernel_ms'],'roofline',d['roofline']['kernel'],round(d['roofline']['frac'],4))"
