#!/bin/bash
# Round 4: (1) 256-thread finalize kernels, (2) the weight gradients' side stream confined to part of the CUs (SGX_SIDE_CUS), (3) the
# whole-model gates under the new default conv math.   gpurun --timeout 900 -- 'bash tools/visits/r4_cus.sh r4o'
TAG=${1:-r4o}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_kernels.py tests/test_blocks.py -m gpu -q -k "bn or finalize or qarep or partial_chip or stats or colsum or reduce" > "$OUT/pytest_kernels.log" 2>&1
tail -3 "$OUT/pytest_kernels.log" | cut -c1-300
bash tools/visits/r4_ab.sh "$TAG" "A=1" "SGX_SIDE_CUS=75" "SGX_SIDE_CUS=50" "SGX_SIDE_CUS=88" "SGX_SIDE_CUS=62"
SGX_TEST_DUMP="$OUT/dump.txt" timeout 400 python -m pytest tests/test_yolo_nas.py -m gpu -q -k "backward_exact_without or train_step_parity" > "$OUT/pytest_yolo_nas.log" 2>&1
tail -5 "$OUT/pytest_yolo_nas.log" | cut -c1-400
grep "^backward\|^#" "$OUT/dump.txt"
