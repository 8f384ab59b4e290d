#!/bin/bash
# Round 5, visit 12: phase stamps of the patch conv kernel's workgroups (a -DSGX_PCONV_TIMING build swapped in for this visit only).
TAG=${1:-r5r}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cp super_gradients_amd/csrc/libsgx_hip.so /tmp/product.so
cp _alt/libsgx_timing.so super_gradients_amd/csrc/libsgx_hip.so
PCONV_TIMING_SET=${PCONV_TIMING_SET:-1} timeout 300 python tools/pconv_timing.py > "$OUT/pconv_timing.txt" 2> "$OUT/pconv_timing.err"
cp /tmp/product.so super_gradients_amd/csrc/libsgx_hip.so
cat "$OUT/pconv_timing.txt"; tail -3 "$OUT/pconv_timing.err"
