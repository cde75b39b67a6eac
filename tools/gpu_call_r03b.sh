#!/bin/bash
# round 3, visit b: xconv occupancy variant (cfg 5) A/B + SQ counters, warp+loss variants A/B, xconv/warp parity tests.
set -u
OUT=gpurun_out/r03b
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
ROOT=$(pwd)
timeout 900 python -m pytest tests/test_06_xconv_gpu.py tests/test_00_warp_loss_gpu.py tests/test_01_surfaces_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
for cfg in 1 5; do
  XCONV_CFG=$cfg XCONV_NMUL=3 XCONV_NO_WGRAD=1 timeout 300 python tools/microbench_xconv.py nomiopen > $OUT/xconv_cfg$cfg.jsonl 2> $OUT/xconv_cfg$cfg.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03b/xconv_cfg*.jsonl')):
    print(f)
    for l in open(f):
        r=json.loads(l); print('  ',r['shape'],'fwd %.3f ms %.0f TF  dgrad %.3f ms %.0f TF'%(r['xconv_fwd_ms'],r['xconv_fwd_tfs'],r['xconv_dgrad_ms'],r['xconv_dgrad_tfs']))
PY
# warp+loss variants (built by tools/build_variant.sh before the visit)
for v in prod nopin nodirect neither; do
  lib=$ROOT/dynamic-video-depth_amd/dvd_hip/lib/variants/libdvd_hip_$v.so
  [ "$v" = prod ] && lib=$ROOT/dynamic-video-depth_amd/dvd_hip/lib/libdvd_hip.so
  for i in 1 2; do
    DVD_HIP_LIB=$lib timeout 200 python tools/microbench_warp.py --iters 200 2>> $OUT/warp.err | sed "s/^{/{\"variant\": \"$v\", /" >> $OUT/warp_variants.jsonl
  done
  DVD_HIP_LIB=$lib timeout 200 python tools/microbench_warp.py --iters 100 --smooth_flow --flow_sigma 10 2>> $OUT/warp.err | sed "s/^{/{\"variant\": \"$v\", /" >> $OUT/warp_variants.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r03b/warp_variants.jsonl'):
    r=json.loads(l); print(r['variant'], 'sigma', r['flow_sigma'], 'smooth', r['smooth_flow'], '%.1f us  frac %.3f'%(1e3*r['ms_per_call_incl_memset_and_reduce'], r['frac_of_8TBps']))
PY
# SQ counters of xconv at the decoder shape, 128x128 blocks at 2 (cfg 1) and 3 (cfg 5) blocks per CU
for cfg in 1 5; do
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
             "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
    i=$((i+1))
    ( cd /tmp && XCONV_CFG=$cfg XCONV_ONLY=4 XCONV_NO_WGRAD=1 timeout 300 rocprofv3 --pmc $grp --output-format csv -d $ROOT/$OUT/sq${cfg}_$i -o pmc -- \
        python $ROOT/tools/microbench_xconv.py nomiopen > $ROOT/$OUT/sq${cfg}_$i.log 2>&1 )
  done
done
python tools/pmc_summary.py "$OUT/sq*/" 2>&1 | grep -E "xconv_kernel" > $OUT/xconv_sq_summary.txt
rm -rf $OUT/sq*/
cat $OUT/xconv_sq_summary.txt | cut -c1-200
