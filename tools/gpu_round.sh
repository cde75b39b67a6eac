#!/bin/bash
# One GPU-box visit: parity tests, the bench line, a rocprofv3 kernel trace of the same
# command, and the warp+loss micro-benchmark.  Everything lands under gpurun_out/<tag>/;
# tools/rocprof_summary.py turns the trace into the text summary committed under profiles/.
# usage: tools/gpu_round.sh <tag> [steps]   (stages selected with STAGES="tests bench trace micro pmc")
set -u
TAG=${1:-r01}
STEPS=${2:-2}
STAGES=${STAGES:-"tests bench trace micro"}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
ROOT=$(pwd)
has() { [[ " $STAGES " == *" $1 "* ]]; }

if has tests; then
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
  echo "pytest exit $?" >> $OUT/pytest.log
  tail -3 $OUT/pytest.log
fi
if has micro; then
  timeout 300 python tools/microbench_warp.py > $OUT/micro_warp.log 2>&1
  timeout 300 python tools/microbench_warp.py --fwd_only >> $OUT/micro_warp.log 2>&1
  [ -f tools/microbench_mlp.py ] && timeout 300 python tools/microbench_mlp.py > $OUT/micro_mlp.log 2>&1
  tail -3 $OUT/micro_warp.log
fi
if has bench; then
  timeout 1200 python bench.py --steps $STEPS --warmup 1 > $OUT/bench.log 2> $OUT/bench.err
  echo "bench exit $?" >> $OUT/bench.err
  tail -2 $OUT/bench.log; tail -5 $OUT/bench.err
fi
if has trace; then
  ( cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $ROOT/$OUT/trace -o bench -- \
      python $ROOT/bench.py --steps $STEPS --warmup 1 --no_cpu_baseline > $ROOT/$OUT/trace.log 2>&1 )
  echo "trace exit $?" >> $OUT/trace.log
  python tools/rocprof_summary.py "$OUT/trace/**/*.db" > $OUT/trace_summary.txt 2>> $OUT/trace.log
  mkdir -p $OUT/trace_keep; find $OUT/trace -name '*stats*.csv' -exec cp {} $OUT/trace_keep/ \;
  rm -rf $OUT/trace
  head -25 $OUT/trace_summary.txt
fi
if has pmc; then
  # FETCH_SIZE costs 3 of the 4 TCC slots and WRITE_SIZE 2: separate passes (MI355X_MICROARCH.md, PMC slots)
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    name=$(echo $grp | tr ' ' '_')
    ( cd /tmp && timeout 600 rocprofv3 --pmc $grp --output-format csv -d $ROOT/$OUT/pmc_$name -o pmc -- \
        python $ROOT/tools/microbench_warp.py --iters 5 > $ROOT/$OUT/pmc_$name.log 2>&1 )
  done
  python tools/pmc_summary.py "$OUT/pmc_*/" > $OUT/pmc_summary.txt 2>&1
  rm -rf $OUT/pmc_*/
  cat $OUT/pmc_summary.txt | head -40
fi
if has tune; then
  # MIOpen find (benchmark) mode with the user find-db kept under gpurun_out/: how much do tuned solvers buy?
  mkdir -p $ROOT/$OUT/miopen_db
  MIOPEN_FIND_MODE=NORMAL MIOPEN_USER_DB_PATH=$ROOT/$OUT/miopen_db DVD_CUDNN_BENCHMARK=1 \
    timeout ${TUNE_TIMEOUT:-600} python bench.py --steps 2 --warmup 1 --no_cpu_baseline > $OUT/tune.log 2> $OUT/tune.err
  echo "tune exit $?" >> $OUT/tune.err
  tail -1 $OUT/tune.log | cut -c1-400; tail -3 $OUT/tune.err; du -sh $OUT/miopen_db
fi
du -sh $OUT; ls -la $OUT
