export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/pool
timeout 900 python -m pytest tests/test_12_depth_head_gpu.py -q -x > gpurun_out/pool/t12.log 2>&1; tail -5 gpurun_out/pool/t12.log
timeout 1200 python -m pytest tests/test_30_full_step_gpu.py tests/test_20_model_surface_gpu.py -q -x -k "hourglass or cnn or surface" > gpurun_out/pool/t30.log 2>&1; tail -4 gpurun_out/pool/t30.log
timeout 900 python bench.py --depth hourglass --no_cpu_baseline > gpurun_out/pool/bench_hourglass.json 2> gpurun_out/pool/bench_hourglass.err; cut -c1-330 gpurun_out/pool/bench_hourglass.json; tail -2 gpurun_out/pool/bench_hourglass.err
