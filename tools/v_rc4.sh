export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/rc
DVD_KEEP_DEBUG=1 timeout 600 python bench.py --config 4 --pairs 64 --steps 2 --cfg4_parity none --no_extras --no_cpu_baseline --depth_keep_gb 200 > gpurun_out/rc/p64_k200.json 2> gpurun_out/rc/p64_k200.err; cut -c1-330 gpurun_out/rc/p64_k200.json; grep "keep slot" gpurun_out/rc/p64_k200.err | tail -8; grep -i "warn\|error\|failed" gpurun_out/rc/p64_k200.err | head -5
python - <<'PY'
import json
d=json.loads(open('gpurun_out/rc/p64_k200.json').read().strip().splitlines()[-1])
print(d.get('hbm_peak_reserved_GB'), d.get('hbm_peak_allocated_GB'), d.get('pairs_per_s'))
PY
