#!/bin/bash
# One GPU-box visit, parameterised (the ONE visit tool: replaces the per-call scripts of earlier rounds).
#   TAG=<name> STAGES="tests bench extras trace pmc sq micro a16" bash tools/gpu_visit.sh
# Everything lands under gpurun_out/$TAG/; what is to be judged is copied from there into profiles/ and committed.
#   tests   full `-m gpu` suite with the measured-parity log ($DVD_PARITY_LOG)        (PYTEST_ARGS narrows it)
#   bench   the headline line: python bench.py (parity + cpu_baseline legs included)
#   extras  the other bench lines: fp16 activations (with parity), BASELINE configs[4], frame gaps 2 / 4, hourglass, host feed,
#           two ranks over gloo on the one GPU
#   trace   rocprofv3 --kernel-trace --stats of bench.py variants (TRACES="name:args;name:args", default fp32 / fp16 / hourglass)
#   pmc     HBM-traffic counters of the warp+loss launch sequence (separate --pmc passes) -> warp_loss_pmc.json
#   sq      SQ counters of the warp+loss tile kernel
#   xsq     SQ counters of the convolution / weight-gradient kernels (3x3 256 -> 256 and the wide 1x1), fp32 and fp16 storage
#   msq     SQ counters of the scene-flow MLP kernels
#   micro   micro-benchmarks: warp+loss, scene-flow MLP, convolutions in both activation storages
#   a16     only the fp16-activation kernel tests
#   mfma    rocprofv3 --kernel-trace --stats of bench.py at TWO step counts (fp32 and fp16 activations) -> tools/mfma_roofline.py
#           -> mfma_roofline.json (per-step kernel time of every matrix-kernel class)
#   ranks   8-GPU readiness on one GPU: DVD_RESERVE_GB ballast lines (gap 1 / gap 2 / hourglass), 8 ranks over gloo,
#           one rank over RCCL with the collectives forced (bench.py --rccl_one_rank)
#   wtrace  per-kernel durations of the warp+loss launch sequence (tools/warp_trace.sh)
#   s2      the stride-2 3x3 convolutions (native strided kernels vs DVD_AB=no_s2) and the hourglass's k x k branches, fp32 / fp16
#   gaps    GPU idle time inside a steady step (rocprofv3 --kernel-trace of bench.py -> tools/step_gaps.py)
#   hgtrace per-step kernel times of the hourglass line (two step counts, differenced)
#   sched   BASELINE configs[4] at 64 pairs and frame gap 4: the recompute schedule against the late-normaliser one
set -u
OUT=gpurun_out/${TAG:-visit}; mkdir -p $OUT
STAGES=${STAGES:-"tests bench extras trace pmc sq micro"}
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
ROOT=$(pwd)
has() { [[ " $STAGES " == *" $1 "* ]]; }

if has tests; then
  DVD_PARITY_LOG=$ROOT/$OUT/parity.jsonl timeout 1500 python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} > $OUT/pytest.log 2>&1
  echo "pytest exit $?" >> $OUT/pytest.log
  grep -E "passed|failed|error" $OUT/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head
fi
if has a16; then
  DVD_PARITY_LOG=$ROOT/$OUT/parity_a16.jsonl timeout 900 python -m pytest tests/test_10_act_fp16_gpu.py -q -x ${PYTEST_ARGS:-} > $OUT/pytest_a16.log 2>&1
  tail -5 $OUT/pytest_a16.log | cut -c1-300
fi
if has bench; then
  timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err
  echo "bench exit $?" >> $OUT/bench.err
  tail -1 $OUT/bench.json | cut -c1-250; tail -2 $OUT/bench.err | cut -c1-200
fi
if has extras; then
  run() { name=$1; shift; timeout 900 python bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; tail -1 $OUT/bench_$name.json | cut -c1-230; }
  run fp16 --act_fp16 --cpu_steps 1
  run cfg4 --config 4 --no_cpu_baseline --steps 2
  run cfg4_p32 --config 4 --pairs 32 --no_cpu_baseline --steps 2
  run gap2 --gap 2 --no_cpu_baseline
  run gap4 --gap 4 --no_cpu_baseline --steps 2
  run gap4_fp16 --gap 4 --act_fp16 --no_cpu_baseline --steps 2
  run hourglass --depth hourglass --no_cpu_baseline
  run hostfeed --feed host --no_cpu_baseline
  DVD_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --pairs 8 --no_cpu_baseline --steps 2 > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err
  tail -1 $OUT/bench_2rank_gloo.json | cut -c1-230
fi
if has trace; then
  IFS=';' read -ra VS <<< "${TRACES:-fp32:;fp16:--act_fp16;hourglass:--depth hourglass}"
  for v in "${VS[@]}"; do
    name=${v%%:*}; args=${v#*:}
    ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $ROOT/$OUT/trace_$name -o bench -- \
        python $ROOT/bench.py --steps 3 --warmup 1 --no_cpu_baseline $args > $ROOT/$OUT/trace_$name.log 2>&1 )
    python tools/rocprof_summary.py "$OUT/trace_$name/**/*.db" > $OUT/trace_${name}_summary.txt 2>> $OUT/trace_$name.log
    mkdir -p $OUT/trace_keep; find $OUT/trace_$name -name '*kernel_stats*.csv' -exec cp {} $OUT/trace_keep/${name}_kernel_stats.csv \;
    rm -rf $OUT/trace_$name
    head -8 $OUT/trace_${name}_summary.txt | cut -c1-150
  done
fi
if has pmc; then
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    name=$(echo $grp | tr ' ' '_')
    ( cd /tmp && timeout 300 rocprofv3 --pmc $grp --output-format csv -d $ROOT/$OUT/pmc_$name -o pmc -- \
        python $ROOT/tools/microbench_warp.py --iters 5 > $ROOT/$OUT/pmc_$name.log 2>&1 )
  done
  python tools/pmc_summary.py "$OUT/pmc_*/" > $OUT/pmc_summary.txt 2>&1
  python tools/pmc_to_json.py $OUT/pmc_summary.txt $OUT/warp_loss_pmc.json "rocprofv3 --pmc passes, ${TAG:-visit}"
  rm -rf $OUT/pmc_*/
  head -30 $OUT/pmc_summary.txt | cut -c1-160; cat $OUT/warp_loss_pmc.json | cut -c1-300
fi
if has sq; then
  bash tools/warp_pmc_sq.sh ${TAG:-visit}/warp_sq > /dev/null 2>&1
  cat $OUT/warp_sq/sq_summary.txt | cut -c1-200 | head -12
fi
if has xsq; then
  # SQ counters of the convolution / weight-gradient kernels at 3x3 256 -> 256, 96x168 (shape 4) and the wide 1x1 (shape 8),
  # in both activation storages
  for mode in fp32 fp16; do
    i=0
    for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
               "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
               "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
      i=$((i+1))
      ( cd /tmp && XCONV_ONLY=${XSQ_SHAPES:-4,8} XCONV_NMUL=3 XCONV_FP16=$([ $mode = fp16 ] && echo 1) timeout 300 rocprofv3 --pmc $grp --output-format csv \
          -d $ROOT/$OUT/xsq_${mode}_$i -o pmc -- python $ROOT/tools/microbench_xconv.py nomiopen > $ROOT/$OUT/xsq_${mode}_$i.log 2>&1 )
    done
    python tools/pmc_summary.py "$OUT/xsq_${mode}_*/" 2>&1 | grep -E "xconv_kernel|xwgrad3_kernel|xwgrad1b_kernel" > $OUT/xconv_sq_summary_$mode.txt
    rm -rf $OUT/xsq_${mode}_*/
    cut -c1-200 $OUT/xconv_sq_summary_$mode.txt | head -6
  done
fi
if has msq; then
  # SQ counters of the scene-flow MLP kernels (tools/microbench_mlp.py)
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
             "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
    i=$((i+1))
    ( cd /tmp && timeout 300 rocprofv3 --pmc $grp --output-format csv -d $ROOT/$OUT/msq_$i -o pmc -- \
        python $ROOT/tools/microbench_mlp.py > $ROOT/$OUT/msq_$i.log 2>&1 )
  done
  python tools/pmc_summary.py "$OUT/msq_*/" 2>&1 | grep -E "mlp_" > $OUT/mlp_sq_summary.txt
  rm -rf $OUT/msq_*/
  wc -l $OUT/mlp_sq_summary.txt
fi
if has micro; then
  timeout 300 python tools/microbench_warp.py > $OUT/micro_warp.log 2>&1; tail -1 $OUT/micro_warp.log | cut -c1-300
  timeout 300 python tools/microbench_mlp.py > $OUT/micro_mlp.log 2>&1; tail -3 $OUT/micro_mlp.log | cut -c1-300
  XCONV_NMUL=3 XCONV_FP16=1 timeout 300 python tools/microbench_xconv.py nomiopen > $OUT/xconv_fp16.jsonl 2> $OUT/xconv_fp16.err
  XCONV_NMUL=3 timeout 300 python tools/microbench_xconv.py nomiopen > $OUT/xconv_fp32.jsonl 2> $OUT/xconv_fp32.err
  python - $OUT/xconv_fp16.jsonl $OUT/xconv_fp32.jsonl <<'PY'
import json, sys
for f in sys.argv[1:]:
    print(f)
    for l in open(f):
        r = json.loads(l)
        print('  ', r['shape'], ' '.join('%s=%.3g' % (k, v) for k, v in r.items() if k.endswith('_tfs')))
PY
fi
if has mfma; then
  for mode in fp32 fp16; do
    arg=$([ $mode = fp16 ] && echo --act_fp16)
    for n in 2 5; do
      ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/tr_${mode}_$n -o b -- \
          python $ROOT/bench.py --steps $n --warmup 1 --no_cpu_baseline --no_extras $arg > $ROOT/$OUT/tr_${mode}_$n.log 2>&1 )
      f=$(find $OUT/tr_${mode}_$n -name '*kernel_stats.csv' | head -1); cp $f $OUT/kernel_stats_${mode}_$n.csv; rm -rf $OUT/tr_${mode}_$n
    done
    python tools/mfma_roofline.py --a 2:$OUT/kernel_stats_${mode}_2.csv --b 5:$OUT/kernel_stats_${mode}_5.csv --mode $mode \
        --collected "rocprofv3 --kernel-trace --stats of bench.py --steps 2 / --steps 5, ${TAG:-visit}" --out $OUT/mfma_roofline.json | tee $OUT/mfma_$mode.txt
  done
fi
if has ranks; then
  runr() { name=$1; shift; timeout 1500 env "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; tail -1 $OUT/bench_$name.json | cut -c1-200; }
  runr reserve24 DVD_RESERVE_GB=24 python bench.py --no_cpu_baseline --no_extras
  runr reserve24_gap2 DVD_RESERVE_GB=24 python bench.py --no_cpu_baseline --no_extras --gap 2
  runr reserve24_hourglass DVD_RESERVE_GB=24 python bench.py --no_cpu_baseline --no_extras --depth hourglass
  runr gloo8 DVD_DIST_BACKEND=gloo python bench.py --gpus 8 --pairs 2 --no_cpu_baseline --steps 2
fi
if has s2; then
  timeout 300 python tools/microbench_s2.py > $OUT/s2_fp32.jsonl 2> $OUT/s2_fp32.err; cut -c1-260 $OUT/s2_fp32.jsonl
  XCONV_FP16=1 timeout 300 python tools/microbench_s2.py > $OUT/s2_fp16.jsonl 2> $OUT/s2_fp16.err; cut -c1-260 $OUT/s2_fp16.jsonl
  timeout 300 python tools/microbench_kxk.py > $OUT/kxk.jsonl 2> $OUT/kxk.err; cut -c1-200 $OUT/kxk.jsonl
fi
if has gaps; then
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps_tr -o b -- \
      python $ROOT/bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_extras > $ROOT/$OUT/gaps_tr.log 2>&1 )
  python tools/step_gaps.py "$(find /tmp/gaps_tr -name '*kernel_trace.csv' | head -1)" 30 > $OUT/step_gaps.txt 2>&1; head -8 $OUT/step_gaps.txt
fi
if has hgtrace; then
  for n in 2 5; do
    ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hg_tr$n -o b -- \
        python $ROOT/bench.py --depth hourglass --steps $n --warmup 1 --no_cpu_baseline --no_extras > $ROOT/$OUT/hg_tr$n.log 2>&1 )
    cp "$(find /tmp/hg_tr$n -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_hourglass_$n.csv
  done
fi
if has sched; then
  run() { name=$1; shift; timeout 900 python bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; tail -1 $OUT/bench_$name.json | cut -c1-230; }
  run cfg4_p64 --config 4 --pairs 64 --steps 2 --cfg4_parity none --no_extras --no_cpu_baseline
  run cfg4_p64_late --config 4 --pairs 64 --steps 2 --cfg4_parity none --no_extras --no_cpu_baseline --mlp_recompute 0
  run gap4 --gap 4 --no_cpu_baseline --steps 2
  run gap4_late --gap 4 --no_cpu_baseline --steps 2 --mlp_recompute 0
fi
if has wtrace; then
  bash tools/warp_trace.sh ${TAG:-visit}/wtrace | tail -5
fi
du -sh $OUT
