export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/s2a
timeout 900 python -m pytest tests/test_06_xconv_gpu.py -q -x -k "grouped or sixteen or stride_two" > gpurun_out/s2a/t06.log 2>&1; tail -15 gpurun_out/s2a/t06.log
timeout 600 python tools/microbench_s2.py > gpurun_out/s2a/mb.jsonl 2> gpurun_out/s2a/mb.err; cat gpurun_out/s2a/mb.jsonl; tail -5 gpurun_out/s2a/mb.err
