#!/bin/bash
# Round 6, visit 48: tools/sweep_race_probe.py with the build that still has packed fp32 instructions in bn.hip (tools/_probe/lib_old_slp.so) and with the product
TAG=${1:-r6ay}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
LIB=super_gradients_amd/csrc/libsgx_hip.so
cp $LIB /tmp/product.so
for which in old product; do
  if [ $which = old ]; then cp tools/_probe/lib_old_slp.so $LIB; else cp /tmp/product.so $LIB; fi
  echo "== $which build"
  timeout 300 python tools/sweep_race_probe.py 1500 2>&1 | grep -v amdgpu.ids | tail -6
done | tee "$OUT/sweep_race_probe.txt"
cp /tmp/product.so $LIB
