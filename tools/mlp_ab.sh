#!/bin/bash
# Scene-flow MLP A/B: the kernel tests (incl. the 4-wave / 8-wave bit-identity test) and the micro-benchmark in both workgroup
# shapes and both stash precisions (MLP_NW, MLP_STASH_F16: tools/microbench_mlp.py).  Output under gpurun_out/r05m/.
mkdir -p gpurun_out/r05m
timeout 900 python -m pytest tests/test_02_sf_mlp_gpu.py tests/test_10_act_fp16_gpu.py -x -q -k "mlp or sf or stash or workgroup or golden or oracle or euler" 2>&1 | tail -15 > gpurun_out/r05m/test02b.txt
cat gpurun_out/r05m/test02b.txt
out=gpurun_out/r05m/micro_t4.jsonl
: > $out
for nw in 8 4; do for f16 in 0 1; do
  MLP_NW=$nw MLP_STASH_F16=$f16 timeout 200 python tools/microbench_mlp.py >> $out 2>/dev/null
done; done
python - <<'P'
import json
for l in open('gpurun_out/r05m/micro_t4.jsonl'):
    r=json.loads(l)
    print(r['waves_per_workgroup'], r['stash_f16'], 'fwd %.2f nostash %.2f dx %.2f dw %.2f'%(r['fwd_ms'],r['fwd_nostash_ms'],r['dx_ms'],r['dw_ms']), r['dw_bitwise_reproducible'])
P
