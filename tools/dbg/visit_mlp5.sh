mkdir -p gpurun_out/r05m
out=gpurun_out/r05m/phase2.jsonl
: > $out
V=dynamic-video-depth_amd/dvd_hip/lib/variants
run() { # nw phase lib
  echo "{\"nw\": $1, \"phase_mode\": $(( $2 & 15 )), \"delay\": $(( ($2 >> 4) & 255 )), \"spread\": $(( ($2 >> 12) & 255 )), \"lib\": \"$3\"}" >> $out
  DVD_HIP_LIB=$3 MLP_NW=$1 DVD_MLP_PHASE=$2 timeout 200 python tools/microbench_mlp.py >> $out 2>/dev/null
}
ph() { echo $(( $1 | ($2 << 4) | ($3 << 12) )); }
run 4 0 ""
run 8 0 $V/libdvd_hip_mlpocc4.so
for d in 2 4 6 10 12 14; do run 4 $(ph 3 $d 0) ""; done
python - <<'P'
import json
rows=[json.loads(l) for l in open('gpurun_out/r05m/phase2.jsonl')]
for i in range(0,len(rows)-1,2):
    h,r=rows[i],rows[i+1]
    print(h, 'fwd %.2f nostash %.2f dx %.2f'%(r['fwd_ms'],r['fwd_nostash_ms'],r['dx_ms']))
P
