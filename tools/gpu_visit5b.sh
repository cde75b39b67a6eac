#!/bin/bash
# Round 5, second half: the evidence visit after the MLP restructuring (profiles/README.md, visit r05f): tests, the bench lines
# of every schedule, the per-class traces, the MLP counters and micro-benchmarks.  Everything under gpurun_out/$TAG/.
export TAG=${TAG:-r05f}
OUT=gpurun_out/$TAG; mkdir -p $OUT
STAGES="tests bench" bash tools/gpu_visit.sh
run() { name=$1; shift; timeout 900 python bench.py --no_cpu_baseline "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; tail -1 $OUT/bench_$name.json | cut -c1-160; }
run fp16 --act_fp16
run cfg4 --config 4 --steps 2 --cfg4_parity none
run gap2 --gap 2
run gap4 --gap 4
run hourglass --depth hourglass
STAGES="mfma" bash tools/gpu_visit5.sh
STAGES="msq" bash tools/gpu_visit.sh
python tools/sq_ratios.py $OUT/mlp_sq_summary.txt > $OUT/mlp_sq_ratios.txt
for cfg in "4 0" "8 0" "4 1"; do set -- $cfg; MLP_NW=$1 MLP_STASH_F16=$2 timeout 200 python tools/microbench_mlp.py >> $OUT/micro_mlp.jsonl 2>/dev/null; done
cat $OUT/micro_mlp.jsonl | cut -c1-400
