export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/hgtrace; mkdir -p $OUT; ROOT=$(pwd)
for n in 2 5; do
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hg_tr$n -o b -- python $ROOT/bench.py --depth hourglass --steps $n --warmup 1 --no_cpu_baseline --no_extras > $ROOT/$OUT/tr$n.log 2>&1 )
f=$(find /tmp/hg_tr$n -name '*kernel_stats.csv' | head -1); cp $f $OUT/kernel_stats_hourglass_$n.csv
done
python - <<'PY'
import csv
a={r['Name']:r for r in csv.DictReader(open('gpurun_out/hgtrace/kernel_stats_hourglass_5.csv'))}
b={r['Name']:r for r in csv.DictReader(open('gpurun_out/hgtrace/kernel_stats_hourglass_2.csv'))}
rows=[]
for n,r in a.items():
    t5,c5=int(r['TotalDurationNs']),int(r['Calls'])
    t2,c2=(int(b[n]['TotalDurationNs']),int(b[n]['Calls'])) if n in b else (0,0)
    rows.append(((t5-t2)/3e6,(c5-c2)/3,n))
rows.sort(reverse=True)
print('total %.1f ms/step'%sum(r[0] for r in rows))
for t,c,n in rows[:40]: print('%8.2f ms %7.1f calls  %s'%(t,c,n[:140]))
PY
