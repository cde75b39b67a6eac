export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/s2c
V=dynamic-video-depth_amd/dvd_hip/lib/variants
for name in main db1 db8; do
  if [ $name = main ]; then unset DVD_HIP_LIB; else export DVD_HIP_LIB=$PWD/$V/libdvd_hip_$name.so; fi
  echo "== $name"
  timeout 300 python tools/microbench_s2.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['shape'], 'fwd %.3f fwd+bwd %.3f'%(d['native_fwd_ms'], d['native_fwd_bwd_ms']))"
  timeout 300 python tools/microbench_kxk.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['shape'], 'fwd %.3f ms %.1f TF/s dgrad %.3f ms %.1f TF/s'%(d['fwd_ms'], d['fwd_tfs'], d['dgrad_ms'], d['dgrad_tfs']))"
done > gpurun_out/s2c/ab.txt 2>&1
cat gpurun_out/s2c/ab.txt
unset DVD_HIP_LIB
timeout 600 python -m pytest tests/test_06_xconv_gpu.py -q -x > gpurun_out/s2c/t06.log 2>&1; tail -3 gpurun_out/s2c/t06.log
