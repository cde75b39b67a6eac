set -x
mkdir -p gpurun_out/r05m
timeout 600 python -m pytest tests/test_02_sf_mlp_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r05m/test02.txt
for nw in 8 4; do for f16 in 0 1; do
  MLP_NW=$nw MLP_STASH_F16=$f16 timeout 200 python tools/microbench_mlp.py >> gpurun_out/r05m/micro.jsonl 2>gpurun_out/r05m/micro_err_${nw}_${f16}.txt
done; done
cat gpurun_out/r05m/test02.txt; cat gpurun_out/r05m/micro.jsonl
