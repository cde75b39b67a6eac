mkdir -p gpurun_out/r05m
out=gpurun_out/r05m/uc.jsonl
: > $out
V=dynamic-video-depth_amd/dvd_hip/lib/variants
MLP_NW=4 timeout 200 python tools/microbench_mlp.py >> $out 2>gpurun_out/r05m/uc_err0.txt
DVD_HIP_LIB=$V/libdvd_hip_mlpko16.so MLP_NW=4 timeout 200 python tools/microbench_mlp.py >> $out 2>gpurun_out/r05m/uc_err1.txt
MLP_UNCACHED=3 MLP_NW=4 timeout 200 python tools/microbench_mlp.py >> $out 2>gpurun_out/r05m/uc_err2.txt
MLP_UNCACHED=1 MLP_NW=4 timeout 200 python tools/microbench_mlp.py >> $out 2>gpurun_out/r05m/uc_err3.txt
tail -3 gpurun_out/r05m/uc_err2.txt
python - <<'P'
import json
for l in open('gpurun_out/r05m/uc.jsonl'):
    r=json.loads(l)
    print(r['uncached'], 'fwd %.2f nostash %.2f dx %.2f dw %.2f'%(r['fwd_ms'],r['fwd_nostash_ms'],r['dx_ms'],r['dw_ms']), r['dw_bitwise_reproducible'])
P
