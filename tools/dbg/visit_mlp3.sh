mkdir -p gpurun_out/r05m
out=gpurun_out/r05m/ko.jsonl
: > $out
V=dynamic-video-depth_amd/dvd_hip/lib/variants
for lib in "" $V/libdvd_hip_mlpko1.so $V/libdvd_hip_mlpko8.so; do
  echo "{\"lib\": \"$lib\"}" >> $out
  DVD_HIP_LIB=$lib MLP_NW=4 timeout 200 python tools/microbench_mlp.py >> $out 2>/dev/null
done
python - <<'P'
import json
rows=[json.loads(l) for l in open('gpurun_out/r05m/ko.jsonl')]
for i in range(0,len(rows)-1,2):
    h,r=rows[i],rows[i+1]
    print(h, 'fwd %.2f nostash %.2f dx %.2f'%(r['fwd_ms'],r['fwd_nostash_ms'],r['dx_ms']))
P
