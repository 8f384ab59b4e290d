#!/bin/bash
# Round 5, visit 5: conv epilogues synchronising per wave + output-row offsets after the first loads: conv / block parity under the new
# build, library A/B of the step (alt = the same sources with workgroup barriers in the epilogue and the offsets ahead of the loads).
TAG=${1:-r5g}; ALT=${2:-_alt/libsgx_alt.so}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 400 python -m pytest tests/test_kernels.py tests/test_blocks.py -m gpu -q -k "conv or pconv or qarepvgg or csp or dgrad" > "$OUT/pytest_conv.log" 2>&1
tail -3 "$OUT/pytest_conv.log" | cut -c1-300
bash tools/visits/r4_lib_ab.sh "$TAG" "$ALT"
