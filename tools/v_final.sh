export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
OUT=gpurun_out/r06k; mkdir -p $OUT; ROOT=$(pwd)
timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" >> $OUT/bench.err; tail -1 $OUT/bench.json | cut -c1-250
run() { name=$1; shift; timeout 900 python bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; tail -1 $OUT/bench_$name.json | cut -c1-230; }
run cfg4 --config 4 --no_cpu_baseline --steps 2
run cfg4_p32 --config 4 --pairs 32 --no_cpu_baseline --steps 2
run cfg4_p64 --config 4 --pairs 64 --no_cpu_baseline --steps 2 --no_extras
run gap4 --gap 4 --no_cpu_baseline --steps 2
run gap4_fp16 --gap 4 --act_fp16 --no_cpu_baseline --steps 2
DVD_PARITY_LOG=$ROOT/$OUT/parity.jsonl timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
grep -E "passed|failed|error" $OUT/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
