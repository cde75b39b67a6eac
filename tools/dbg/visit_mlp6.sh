OUT=gpurun_out/r05m_tcc; mkdir -p $OUT; ROOT=$(pwd); export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  ( cd /tmp && MLP_NW=4 timeout 300 rocprofv3 --pmc $grp --output-format csv -d $ROOT/$OUT/p$i -o pmc -- \
      python $ROOT/tools/microbench_mlp.py > $ROOT/$OUT/p$i.log 2>&1 )
done
python tools/pmc_summary.py "$OUT/p*/" 2>&1 | grep -E "mlp_" > $OUT/summary.txt
rm -rf $OUT/p*/
awk '{print $2, $3, $4, $(NF-5), $(NF-4), $(NF-3), $(NF-2), $(NF-1), $NF}' $OUT/summary.txt | cut -c1-220
