#!/bin/bash
set -u
OUT=gpurun_out/r03g; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
for t in "tests/test_30_full_step_gpu.py::test_train_on_batch_matches_reference[fullstep_hourglass_b2_32x48_mseg_gap2]" \
         "tests/test_20_model_surface_gpu.py::test_checkpoint_round_trip_restores_weights_and_adam_state" \
         "tests/test_30_full_step_gpu.py::test_two_rank_data_parallel_step_equals_single_process" \
         "tests/test_01_surfaces_gpu.py"; do
  timeout 600 python -m pytest "$t" -m gpu -q -x 2>&1 | grep -E "passed|failed|ACTUAL|DESIRED|Error" | head -8
done
# same test twice in one process: does the second run differ?
timeout 600 python -m pytest "tests/test_30_full_step_gpu.py::test_train_on_batch_matches_reference" -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|ACTUAL|DESIRED" | head
timeout 600 python -m pytest "tests/test_30_full_step_gpu.py::test_train_on_batch_matches_reference" -m gpu -q -p no:randomly --deselect "tests/test_30_full_step_gpu.py::test_train_on_batch_matches_reference[fullstep_hourglass_b2_32x48_train]" 2>&1 | grep -E "passed|failed|^FAILED|ACTUAL|DESIRED" | head
