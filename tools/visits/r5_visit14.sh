#!/bin/bash
# Round 5, visit 14: pre-split filter planes (sgx_filter_planes_batch) - the planes parity tests on the chip, then the step A/B of the switch
# (each setting twice, interleaved, same box).
TAG=${1:-r5w}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_kernels.py tests/test_blocks.py -q -m gpu -k "filter_planes" -p no:cacheprovider > "$OUT/pytest_planes.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest_planes.log"; tail -4 "$OUT/pytest_planes.log"
bash tools/visits/r4_ab.sh $TAG "SGX_FILTER_PLANES=0" "SGX_FILTER_PLANES=1" 2>&1 | tee "$OUT/ab.txt"
