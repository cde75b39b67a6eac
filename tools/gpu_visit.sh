#!/bin/bash
# One GPU-box visit (parameterised by TAG and STAGES; replaces the per-call scripts of earlier rounds): full GPU test suite (with the measured-parity log), the full bench line (parity + CPU leg),
# a kernel trace of the same command, PMC traffic + SQ counters of the warp+loss launch sequence, SQ counters of the
# convolution kernel, the MLP / convolution micro-benchmarks.  STAGES selects a subset.
set -u
OUT=gpurun_out/${TAG:-visit}; mkdir -p $OUT
STAGES=${STAGES:-"tests bench trace pmc sq xsq micro"}
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
ROOT=$(pwd)
has() { [[ " $STAGES " == *" $1 "* ]]; }

if has tests; then
  DVD_PARITY_LOG=$ROOT/$OUT/parity.jsonl timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
  echo "pytest exit $?" >> $OUT/pytest.log
  grep -E "passed|failed|error" $OUT/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head
fi
if has bench; then
  timeout 1500 python bench.py > $OUT/bench.log 2> $OUT/bench.err
  echo "bench exit $?" >> $OUT/bench.err
  tail -1 $OUT/bench.log | cut -c1-250; tail -3 $OUT/bench.err
fi
if has trace; then
  ( cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $ROOT/$OUT/trace -o bench -- \
      python $ROOT/bench.py --steps 3 --warmup 1 --no_cpu_baseline > $ROOT/$OUT/trace.log 2>&1 )
  python tools/rocprof_summary.py "$OUT/trace/**/*.db" > $OUT/trace_summary.txt 2>> $OUT/trace.log
  mkdir -p $OUT/trace_keep; find $OUT/trace -name '*stats*.csv' -exec cp {} $OUT/trace_keep/ \;
  rm -rf $OUT/trace
  head -12 $OUT/trace_summary.txt | cut -c1-150
fi
if has pmc; then
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    name=$(echo $grp | tr ' ' '_')
    ( cd /tmp && timeout 300 rocprofv3 --pmc $grp --output-format csv -d $ROOT/$OUT/pmc_$name -o pmc -- \
        python $ROOT/tools/microbench_warp.py --iters 5 > $ROOT/$OUT/pmc_$name.log 2>&1 )
  done
  python tools/pmc_summary.py "$OUT/pmc_*/" > $OUT/pmc_summary.txt 2>&1
  python tools/pmc_to_json.py $OUT/pmc_summary.txt $OUT/warp_loss_pmc.json 'rocprofv3 --pmc passes'
  rm -rf $OUT/pmc_*/
  head -30 $OUT/pmc_summary.txt | cut -c1-160; cat $OUT/warp_loss_pmc.json | cut -c1-300
fi
if has sq; then
  bash tools/warp_pmc_sq.sh ${TAG:-visit}/warp_sq > /dev/null 2>&1
  cat $OUT/warp_sq/sq_summary.txt | cut -c1-200 | head -12
fi
if has xsq; then
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
             "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
    i=$((i+1))
    ( cd /tmp && XCONV_ONLY=4 timeout 300 rocprofv3 --pmc $grp --output-format csv -d $ROOT/$OUT/xsq$i -o pmc -- \
        python $ROOT/tools/microbench_xconv.py nomiopen > $ROOT/$OUT/xsq$i.log 2>&1 )
  done
  python tools/pmc_summary.py "$OUT/xsq*/" 2>&1 | grep -E "xconv_kernel|xwgrad3_kernel" > $OUT/xconv_sq_summary.txt
  rm -rf $OUT/xsq*/
  cut -c1-220 $OUT/xconv_sq_summary.txt | head -8
fi
if has micro; then
  timeout 300 python tools/microbench_warp.py > $OUT/micro_warp.log 2>&1; tail -2 $OUT/micro_warp.log | cut -c1-300
  timeout 300 python tools/microbench_mlp.py > $OUT/micro_mlp.log 2>&1; tail -3 $OUT/micro_mlp.log | cut -c1-300
  XCONV_NMUL=3 timeout 400 python tools/microbench_xconv.py nomiopen > $OUT/xconv_n48.jsonl 2> $OUT/xconv_n48.err
  python - "$OUT/xconv_n48.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    r = json.loads(l)
    print('  ', r['shape'], ' '.join('%s=%.3g' % (k, v) for k, v in r.items() if k != 'shape' and isinstance(v, (int, float))))
PY
fi
du -sh $OUT
