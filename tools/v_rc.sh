export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/rc
timeout 900 python -m pytest tests/test_30_full_step_gpu.py -q -x > gpurun_out/rc/t30.log 2>&1; tail -4 gpurun_out/rc/t30.log | cut -c1-300
DVD_KEEP_DEBUG=1 timeout 900 python bench.py --config 4 --pairs 64 --steps 2 --cfg4_parity none --no_extras --no_cpu_baseline > gpurun_out/rc/p64.json 2> gpurun_out/rc/p64.err; cut -c1-330 gpurun_out/rc/p64.json; grep "keep slot" gpurun_out/rc/p64.err | tail -12
timeout 900 python bench.py --config 4 --pairs 64 --steps 2 --cfg4_parity none --no_extras --no_cpu_baseline --mlp_recompute 0 > gpurun_out/rc/p64_late.json 2> gpurun_out/rc/p64_late.err; cut -c1-330 gpurun_out/rc/p64_late.json; tail -2 gpurun_out/rc/p64_late.err | cut -c1-200
