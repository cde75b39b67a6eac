export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/rc
timeout 900 python -m pytest tests/test_30_full_step_gpu.py -q -x -k "chunking" > gpurun_out/rc/t30b.log 2>&1; tail -12 gpurun_out/rc/t30b.log | cut -c1-300
timeout 900 python bench.py --gap 4 --no_cpu_baseline --steps 2 > gpurun_out/rc/gap4.json 2> gpurun_out/rc/gap4.err; cut -c1-300 gpurun_out/rc/gap4.json
timeout 900 python bench.py --gap 4 --no_cpu_baseline --steps 2 --mlp_recompute 0 > gpurun_out/rc/gap4_late.json 2> gpurun_out/rc/gap4_late.err; cut -c1-300 gpurun_out/rc/gap4_late.json
