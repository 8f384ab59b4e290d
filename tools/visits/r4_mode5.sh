#!/bin/bash
# Round 4: conv math mode 5 (bf16x3 per problem, the two-source / two-output launches included, corrections in their own accumulator) against
# the default:  gpurun --timeout 700 -- 'bash tools/visits/r4_mode5.sh r4n'
# 1. kernel tests under the mode   2. step A/B (twice, interleaved)   3. the whole-model gates under the mode (flip-free element-wise, parity)
TAG=${1:-r4n}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 200 python -m pytest tests/test_kernels.py -m gpu -q -k "conv_math_patch_auto or conv_bf16x3 or dual" > "$OUT/pytest_kernels.log" 2>&1
tail -3 "$OUT/pytest_kernels.log" | cut -c1-300
bash tools/visits/r4_ab.sh "$TAG" "A=1" "SGX_CONV_MATH=patch_bf3" "SGX_CONV_MATH=patch_auto"
for mode in patch_bf3 patch_auto; do
  SGX_CONV_MATH=$mode SGX_TEST_DUMP="$OUT/dump_$mode.txt" timeout 400 python -m pytest tests/test_yolo_nas.py -m gpu -q -k "backward_exact_without or train_step_parity" > "$OUT/pytest_yolo_nas_$mode.log" 2>&1
  echo "== $mode"; tail -25 "$OUT/pytest_yolo_nas_$mode.log" | cut -c1-400
done
timeout 60 python tools/conv_error_probe.py > "$OUT/conv_error_probe.txt" 2>&1; tail -20 "$OUT/conv_error_probe.txt"
