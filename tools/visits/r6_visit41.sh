#!/bin/bash
# Round 6, visit 41: the d alpha dot on the side stream itself (in order with the weight gradients it does not tolerate beside it?), and the
# weight gradients without the patch kernel / on the fp32 pipe
TAG=${1:-r6aq}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for cfg in "SGX_DALPHA_ON_SIDE=1" "SGX_WGRAD_PATCH=0" "SGX_WGRAD_MATH=0"; do
  echo "== $cfg"
  env $cfg timeout 400 python tools/branch_flake_probe.py 400 200 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-330
done | tee "$OUT/d_alpha_flips_3.txt"
