#!/bin/bash
# rocprofv3 kernel trace of the bench in eager mode (graph replays hide the kernels from the trace): per-step kernel times
set -u
OUT=gpurun_out/${1:-r02te}; mkdir -p $OUT; ROOT=$(pwd)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
( cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/tr -o bench -- \
    python $ROOT/bench.py --steps 3 --warmup 1 --no_cpu_baseline --depth_graphs 0 > $ROOT/$OUT/trace.log 2>&1 )
f=$(find $OUT/tr -name '*kernel_stats.csv' | head -1); cp $f $OUT/kernel_stats.csv; rm -rf $OUT/tr
tail -1 $OUT/trace.log | cut -c1-200
python - <<PY
import csv
rows=[(r['Name'],int(r['Calls']),float(r['TotalDurationNs'])) for r in csv.DictReader(open('$OUT/kernel_stats.csv'))]
steps=4.0
print('launches/step %.0f  gpu ms/step %.1f' % (sum(r[1] for r in rows)/steps, sum(r[2] for r in rows)/steps/1e6))
groups={}
for n,c,t in rows:
    k='rest'
    for key in ('xconv','xwgrad','mlp_','gconv','bnrelu','upsample','warp_loss','combine','Cijk','Im2d','Col2Im','igemm','transpose','adam'):
        if key in n: k=key;break
    else:
        if 'at::native' in n or 'at6native' in n: k='aten'
    g=groups.setdefault(k,[0,0]); g[0]+=t/steps/1e6; g[1]+=c/steps
for k,v in sorted(groups.items(),key=lambda t:-t[1][0]): print('%-12s %8.1f ms %7.0f launches'%(k,v[0],v[1]))
print('--- top kernels')
for n,c,t in sorted(rows,key=lambda r:-r[2])[:14]: print('%-70s %6.0f calls/step %8.1f ms/step avg %8.1f us'%(n[:70],c/steps,t/steps/1e6,t/c/1e3))
PY
