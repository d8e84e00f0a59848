#!/bin/bash
# Round-4 measurement pass on the GPU box (gpurun): everything profiles/r4/ holds.
#   SECTIONS="tests bench prof pmc floor configs" (default all); outputs under gpurun_out/r4/, copied to profiles/r4/ by hand.
#   The commit the snapshot was taken at travels in DRNA_COMMIT (the box has no .git):
#     gpurun -- "DRNA_COMMIT=$(git rev-parse --short HEAD)$(git diff --quiet || echo +dirty) tools/gpu_round4.sh"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
STEPS=${STEPS:-50}
SECTIONS=${SECTIONS:-"tests bench prof pmc floor configs"}
has() { case " $SECTIONS " in *" $1 "*) return 0;; *) return 1;; esac; }
REPO="$GRAFT_REPO_ROOT"
O=gpurun_out/r4
if has tests; then echo "== pytest gpu"; timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log; fi
if has bench; then
  echo "== bench"; timeout -k 10 400 python bench.py --steps $STEPS --warmup 5 > $O/bench_uniform.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
  cat $O/bench_uniform.json
  timeout -k 10 200 python bench.py --steps $STEPS --warmup 5 --seqs design --no-cpu-baseline --no-r-sweep > $O/bench_design.json 2>/dev/null || exit 1
  DRNA_PF_HELPER=0 timeout -k 10 200 python bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-r-sweep > $O/bench_nohelper.json 2>/dev/null || exit 1
  DRNA_FUSED=0 timeout -k 10 200 python bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-r-sweep > $O/bench_two_launches.json 2>/dev/null || exit 1
  DRNA_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 3 > $O/bench_gloo2_rehearsal.json 2> $O/bench_gloo2.err || { tail -5 $O/bench_gloo2.err; }
  DRNA_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --scaling strong --steps 20 --warmup 3 > $O/bench_gloo2_strong_rehearsal.json 2> $O/bench_gloo2_strong.err || { tail -5 $O/bench_gloo2_strong.err; }
  timeout -k 10 300 python bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-mc-loop --r-sweep > $O/bench_rsweep.json 2>/dev/null || exit 1
  python3 -c "
import json
for f in ('bench_uniform','bench_design','bench_nohelper','bench_two_launches','bench_gloo2_rehearsal','bench_gloo2_strong_rehearsal','bench_rsweep'):
    try:
        d=json.load(open('$O/'+f+'.json')); print('%-24s value %.0f n_gpus %d ms/step %.4f kernel_ms %s' % (f, d['value'], d['n_gpus'], d['ms_per_step'], d['kernel_ms']))
    except Exception as ex: print(f, 'FAILED', ex)
"
fi
if has prof; then
  echo "== rocprof kernel trace"
  rm -rf gpurun_out/prof; (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof" -o r4 -- python3 "$REPO/bench.py" --steps $STEPS --warmup 5 --no-cpu-baseline --no-mc-loop --no-r-sweep > "$REPO/$O/bench_prof.json" 2> "$REPO/$O/prof.err") || exit 1
  for f in $(find gpurun_out/prof -name "*kernel_stats*.csv"); do cp $f gpurun_out/rocprofv3_kernel_stats.csv; cp $f $O/rocprofv3_kernel_stats.csv; head -4 $f; done
fi
if has pmc; then
  echo "== PMC passes"
  pass() { # name counters...
    name=$1; shift
    rm -rf gpurun_out/pmc_$name
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$REPO/gpurun_out/pmc_$name" -o $name -- python3 "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-mc-loop --no-r-sweep > /dev/null 2> "$REPO/gpurun_out/pmc_$name.err") || { tail -3 gpurun_out/pmc_$name.err; return 1; }
    # the two-launch form (DRNA_FUSED=0) as well: its kernels' inputs stay in the file beside the one-launch kernel's
    rm -rf gpurun_out/pmc2_$name
    (cd /tmp && DRNA_FUSED=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$REPO/gpurun_out/pmc2_$name" -o $name -- python3 "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-mc-loop --no-r-sweep > /dev/null 2> "$REPO/gpurun_out/pmc2_$name.err") || { tail -3 gpurun_out/pmc2_$name.err; return 1; }
    rm -rf gpurun_out/cal_$name
    (cd /tmp && timeout -k 10 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$REPO/gpurun_out/cal_$name" -o $name -- "$REPO/tools/pmc_calib" > /dev/null 2> "$REPO/gpurun_out/cal_$name.err") || { tail -3 gpurun_out/cal_$name.err; return 1; }
  }
  pass fetch FETCH_SIZE || exit 1
  pass write WRITE_SIZE || exit 1
  pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD || exit 1
  pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS || exit 1
  pass grbm GRBM_GUI_ACTIVE || exit 1
fi
if has floor; then
  echo "== floor (every sweep phase left out, in the PRODUCTION launch configuration: two workgroups per MFE fold, helper workgroup per partition function -- finalize, hand-shakes and barriers remain)"
  timeout -k 10 300 python tools/phase_cost.py 0 15 > gpurun_out/floor.txt 2>&1 || { tail -5 gpurun_out/floor.txt; exit 1; }
  cp gpurun_out/floor.txt $O/floor.txt; cat gpurun_out/floor.txt
fi
if has configs; then echo "== configs"; timeout -k 10 600 python tools/gpu_configs.py > $O/configs.json 2> $O/configs.err || { tail -5 $O/configs.err; exit 1; }; cat $O/configs.json; fi
python3 tools/roofline_inputs.py > $O/roofline_inputs_summary.txt 2>&1 || true
cp gpurun_out/roofline_inputs.json $O/roofline_inputs.json 2>/dev/null || true
