#!/bin/bash
# GPU box: quick bench (no CPU baseline) for every prebuilt engine variant in build/var/ (tools/build_variants.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
for lib in build/var/lib_*.so; do
  cp "$lib" desirna_amd/libdesirna_amd.so
  timeout -k 10 200 python bench.py --steps ${STEPS:-50} --warmup 5 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-24s value %.0f  ms/step %.4f  kernel_ms %s' % ('$(basename $lib)', d['value'], d['ms_per_step'], d['kernel_ms']))"
done
