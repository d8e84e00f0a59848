#!/bin/bash
# Runs on the GPU box via gpurun.  Sections (pick with SECTIONS="smoke tests bench prof pmc floor configs"):
#   smoke/tests: __graft_entry__.smoke(), pytest -m gpu
#   bench:  bench.py (uniform + design-like) -> gpurun_out/bench*.json
#   prof:   rocprofv3 --kernel-trace --stats of the bench command
#   pmc:    separate --pmc passes (HBM FETCH/WRITE, two SQ sets, GRBM) of the bench command + the calibration kernels
#   floor:  the fold kernels with every sweep phase left out (-DDRNA_SKIP=15): the dependency-chain floor
#   -> gpurun_out/roofline_inputs.json (copied to profiles/ by hand once the kernels are final)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
STEPS=${STEPS:-50}
SECTIONS=${SECTIONS:-"smoke tests bench prof pmc floor"}
has() { case " $SECTIONS " in *" $1 "*) return 0;; *) return 1;; esac; }
REPO="$GRAFT_REPO_ROOT"
if has smoke; then echo "== smoke"; timeout -k 10 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 || exit 1; fi
if has tests; then echo "== pytest gpu"; timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; tail -5 gpurun_out/pytest_gpu.log; [ $rc -eq 0 ] || exit 1; fi
if has bench; then
  echo "== bench"; timeout -k 10 400 python bench.py --steps $STEPS --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
  cat gpurun_out/bench.json
  echo "== bench design-like"; timeout -k 10 200 python bench.py --steps $STEPS --warmup 5 --seqs design --no-cpu-baseline > gpurun_out/bench_design.json 2>/dev/null || exit 1; cat gpurun_out/bench_design.json
fi
if has qbench; then
  echo "== quick bench (no CPU baseline)"
  timeout -k 10 200 python bench.py --steps $STEPS --warmup 5 --no-cpu-baseline > gpurun_out/qbench.json 2> gpurun_out/qbench.err || { tail -5 gpurun_out/qbench.err; exit 1; }
  python3 -c "import json; d=json.load(open('gpurun_out/qbench.json')); print('QBENCH value %.0f  ms/step %.4f  kernel_ms %s' % (d['value'], d['ms_per_step'], d['kernel_ms']))"
  timeout -k 10 200 python bench.py --steps $STEPS --warmup 5 --seqs design --no-cpu-baseline > gpurun_out/qbench_design.json 2>/dev/null || exit 1
  python3 -c "import json; d=json.load(open('gpurun_out/qbench_design.json')); print('QBENCH design value %.0f  ms/step %.4f  kernel_ms %s' % (d['value'], d['ms_per_step'], d['kernel_ms']))"
fi
if has qsingle; then
  echo "== quick bench, one workgroup per fold (DRNA_DUAL=0)"
  DRNA_DUAL=0 timeout -k 10 200 python bench.py --steps $STEPS --warmup 5 --no-cpu-baseline > gpurun_out/qbench_single.json 2>/dev/null || exit 1
  python3 -c "import json; d=json.load(open('gpurun_out/qbench_single.json')); print('QBENCH single value %.0f  ms/step %.4f  kernel_ms %s' % (d['value'], d['ms_per_step'], d['kernel_ms']))"
fi
if has qtests; then echo "== quick parity tests"; timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or config3 or edge or properties or L200 or config2 or maximum" > gpurun_out/pytest_q.log 2>&1; rc=$?; tail -3 gpurun_out/pytest_q.log; [ $rc -eq 0 ] || exit 1; fi
if has prof; then
  echo "== rocprof kernel trace"
  rm -rf gpurun_out/prof; (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof" -o r2 -- python3 "$REPO/bench.py" --steps $STEPS --warmup 5 --no-cpu-baseline > "$REPO/gpurun_out/bench_prof.json" 2> "$REPO/gpurun_out/prof.err") || exit 1
  for f in $(find gpurun_out/prof -name "*kernel_stats*.csv"); do cp $f gpurun_out/rocprofv3_kernel_stats.csv; cat $f; done
fi
if has pmc; then
  echo "== PMC passes"
  pass() { # name counters...
    name=$1; shift
    rm -rf gpurun_out/pmc_$name
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$REPO/gpurun_out/pmc_$name" -o $name -- python3 "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> "$REPO/gpurun_out/pmc_$name.err") || { tail -3 gpurun_out/pmc_$name.err; return 1; }
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
  echo "== floor (finalize-only build)"
  timeout -k 10 300 python tools/phase_cost.py 0 15 > gpurun_out/floor.txt 2>&1 || { tail -5 gpurun_out/floor.txt; exit 1; }
  cat gpurun_out/floor.txt
fi
if has configs; then echo "== configs"; timeout -k 10 600 python tools/gpu_configs.py > gpurun_out/configs.json 2> gpurun_out/configs.err || { tail -5 gpurun_out/configs.err; exit 1; }; cat gpurun_out/configs.json; fi
python3 tools/roofline_inputs.py || true
