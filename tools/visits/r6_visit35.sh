#!/bin/bash
# Round 6, visit 35: r6fin4's full GPU suite failed test_yolo_nas_s_step_is_bit_identical_with_branch_stream once (max 9.3e-10, repeat 0); the same
# tree passed it in r6aj.  Flake rate and the parameters involved, with the pre-transposed ConvTranspose2x2 forward on and off.
TAG=${1:-r6ak}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2 3 4 5 6 7 8; do
  for cfg in "SGX_CONVT_PRETRANSPOSED=1" "SGX_CONVT_PRETRANSPOSED=0"; do
    r=$(env $cfg timeout 300 python -m pytest tests/test_yolo_nas.py -m gpu -q -x -k "bit_identical_with_branch_stream" 2>&1 | grep -E "AssertionError: repeat|passed|failed" | head -2 | tr '\n' ' ')
    echo "rep $rep [$cfg]: $r"
  done
done | tee "$OUT/branch_bit_identity_flake.txt"
