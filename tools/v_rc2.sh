export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/rc
for dc in 16 32; do
DVD_KEEP_DEBUG=1 timeout 900 python bench.py --config 4 --pairs 64 --steps 2 --cfg4_parity none --no_extras --no_cpu_baseline --depth_chunk $dc > gpurun_out/rc/p64_dc$dc.json 2> gpurun_out/rc/p64_dc$dc.err; cut -c1-330 gpurun_out/rc/p64_dc$dc.json; grep "keep slot" gpurun_out/rc/p64_dc$dc.err | tail -16
done
