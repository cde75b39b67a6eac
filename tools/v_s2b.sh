export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/s2b
timeout 900 python -m pytest tests/test_10_act_fp16_gpu.py -q -x -k "stride_two or conv_fp16" > gpurun_out/s2b/t10.log 2>&1; tail -15 gpurun_out/s2b/t10.log
timeout 1200 python -m pytest tests/test_06_xconv_gpu.py tests/test_09_fused_joins_gpu.py tests/test_11_stem_gpu.py tests/test_30_full_step_gpu.py tests/test_03_gconv_gpu.py -q -x > gpurun_out/s2b/t.log 2>&1; tail -8 gpurun_out/s2b/t.log
XCONV_FP16=1 timeout 600 python tools/microbench_s2.py > gpurun_out/s2b/mb16.jsonl 2> gpurun_out/s2b/mb16.err; cat gpurun_out/s2b/mb16.jsonl; tail -5 gpurun_out/s2b/mb16.err
timeout 900 python bench.py --no_cpu_baseline --steps 5 > gpurun_out/s2b/bench.json 2> gpurun_out/s2b/bench.err; cut -c1-400 gpurun_out/s2b/bench.json; tail -3 gpurun_out/s2b/bench.err
DVD_AB=no_s2 timeout 900 python bench.py --no_cpu_baseline --steps 5 > gpurun_out/s2b/bench_nos2.json 2> gpurun_out/s2b/bench_nos2.err; cut -c1-400 gpurun_out/s2b/bench_nos2.json; tail -3 gpurun_out/s2b/bench_nos2.err
