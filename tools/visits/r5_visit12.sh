#!/bin/bash
# Round 5, visit 12: phase stamps of the patch conv kernel's workgroups (a -DSGX_PCONV_TIMING build swapped in for this visit only).
TAG=${1:-r5r}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cp super_gradients_amd/csrc/libsgx_hip.so /tmp/product.so
# (the instrumented library: hipcc ... -DSGX_PCONV_TIMING[=2] -c csrc/conv.hip, linked with the other objects of csrc/_obj into _alt/libsgx_timing.so)
cp _alt/libsgx_timing.so super_gradients_amd/csrc/libsgx_hip.so
PCONV_TIMING_SET=${PCONV_TIMING_SET:-1} timeout 300 python tools/pconv_timing.py > "$OUT/pconv_timing.txt" 2> "$OUT/pconv_timing.err"
cp /tmp/product.so super_gradients_amd/csrc/libsgx_hip.so
cat "$OUT/pconv_timing.txt"; tail -3 "$OUT/pconv_timing.err"
