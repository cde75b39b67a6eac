#!/bin/bash
# Round-5 additions to tools/gpu_visit.sh (same conventions: TAG, STAGES; everything under gpurun_out/$TAG/):
#   cfg4     BASELINE configs[4]: 24 resident pairs WITH the parity leg at 768x1344 (CPU oracle if the host has the memory),
#            and the 8-GPU form's 64 pairs per GPU
#   mfma     rocprofv3 --kernel-trace --stats of bench.py at TWO step counts (fp32 and fp16 activations) -> tools/mfma_roofline.py
#            -> mfma_roofline.json (per-step kernel time of every matrix-kernel class)
#   ranks    8-GPU readiness on one GPU: DVD_RESERVE_GB ballast lines (gap 1 / gap 2 / hourglass), 8 ranks over gloo
#   wsq      warp+loss: HBM-traffic counters (pmc) and SQ counters of the current kernel -> warp_loss_pmc.json, warp_loss_sq.json
set -u
OUT=gpurun_out/${TAG:-visit5}; mkdir -p $OUT
STAGES=${STAGES:-"cfg4 mfma ranks wsq"}
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=$(pwd)
has() { [[ " $STAGES " == *" $1 "* ]]; }
run() { name=$1; shift; timeout 1500 env "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; tail -1 $OUT/bench_$name.json | cut -c1-200; }
if has cfg4; then
  run cfg4 python bench.py --config 4 --steps 2
  run cfg4_p64 python bench.py --config 4 --pairs 64 --steps 2 --cfg4_parity none
fi
if has mfma; then
  for mode in fp32 fp16; do
    arg=$([ $mode = fp16 ] && echo --act_fp16)
    for n in 2 5; do
      ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/tr_${mode}_$n -o b -- \
          python $ROOT/bench.py --steps $n --warmup 1 --no_cpu_baseline $arg > $ROOT/$OUT/tr_${mode}_$n.log 2>&1 )
      f=$(find $OUT/tr_${mode}_$n -name '*kernel_stats.csv' | head -1); cp $f $OUT/kernel_stats_${mode}_$n.csv; rm -rf $OUT/tr_${mode}_$n
    done
    python tools/mfma_roofline.py --a 2:$OUT/kernel_stats_${mode}_2.csv --b 5:$OUT/kernel_stats_${mode}_5.csv --mode $mode \
        --collected "rocprofv3 --kernel-trace --stats of bench.py --steps 2 / --steps 5, ${TAG:-visit5}" --out $OUT/mfma_roofline.json | tee $OUT/mfma_$mode.txt
  done
fi
if has ranks; then
  run reserve24 DVD_RESERVE_GB=24 python bench.py --no_cpu_baseline
  run reserve24_gap2 DVD_RESERVE_GB=24 python bench.py --no_cpu_baseline --gap 2
  run reserve24_hourglass DVD_RESERVE_GB=24 python bench.py --no_cpu_baseline --depth hourglass
  run gloo8 DVD_DIST_BACKEND=gloo python bench.py --gpus 8 --pairs 2 --no_cpu_baseline --steps 2
fi
if has wsq; then
  for grp in "FETCH_SIZE" "WRITE_SIZE"; do
    ( cd /tmp && timeout 300 rocprofv3 --pmc $grp --output-format csv -d $ROOT/$OUT/pmc_$grp -o pmc -- \
        python $ROOT/tools/microbench_warp.py --iters 5 > $ROOT/$OUT/pmc_$grp.log 2>&1 )
  done
  python tools/pmc_summary.py "$OUT/pmc_*/" > $OUT/pmc_summary.txt 2>&1
  python tools/pmc_to_json.py $OUT/pmc_summary.txt $OUT/warp_loss_pmc.json "rocprofv3 --pmc passes, ${TAG:-visit5}"
  rm -rf $OUT/pmc_*/
  SKIP_TESTS=1 SQ=1 TILES=-1 bash tools/warp_ab.sh ${TAG:-visit5}/wab > /dev/null 2>&1
  cat $OUT/wab/ab.txt
  bash tools/warp_trace.sh ${TAG:-visit5}/wtrace | tail -4
fi
du -sh $OUT
