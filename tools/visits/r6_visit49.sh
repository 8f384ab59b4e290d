#!/bin/bash
# Round 6, visit 49: the new reproducibility test of sgx_dot (and its neighbours in tests/test_kernels.py)
TAG=${1:-r6az}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_kernels.py -m gpu -q 2>&1 | tail -4 | tee "$OUT/pytest_kernels.txt"
for i in 1 2 3; do timeout 200 python -m pytest tests/test_kernels.py -m gpu -q -k "dot_repeats" 2>&1 | tail -1; done | tee -a "$OUT/pytest_kernels.txt"
