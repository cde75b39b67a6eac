set -x
mkdir -p gpurun_out/r05m
out=gpurun_out/r05m/phase.jsonl
: > $out
run() { # nw phase
  echo "{\"nw\": $1, \"phase_mode\": $(( $2 & 15 )), \"delay\": $(( ($2 >> 4) & 255 )), \"spread\": $(( ($2 >> 12) & 255 ))}" >> $out
  MLP_NW=$1 DVD_MLP_PHASE=$2 timeout 200 python tools/microbench_mlp.py >> $out 2>/dev/null
}
ph() { echo $(( $1 | ($2 << 4) | ($3 << 12) )); }
run 4 0
for d in 8 16 24; do for sp in 0 3; do run 4 $(ph 3 $d $sp); done; done
run 4 $(ph 1 16 0)
run 4 $(ph 2 16 0)
run 4 $(ph 1 0 4)
run 8 $(ph 1 0 4)
run 8 $(ph 1 16 0)
python - <<'P'
import json
rows=[json.loads(l) for l in open('gpurun_out/r05m/phase.jsonl')]
for i in range(0,len(rows)-1,2):
    h,r=rows[i],rows[i+1]
    print(h, 'fwd %.2f nostash %.2f dx %.2f'%(r['fwd_ms'],r['fwd_nostash_ms'],r['dx_ms']))
P
