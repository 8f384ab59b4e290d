#!/bin/bash
# Round 6, visit 39: the one-ulp flips of the bottlenecks' d alpha (r6an: single-chain networks too, 11 of 2400 steps) - with the side stream off,
# and with every kernel serialised
TAG=${1:-r6ao}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for cfg in "SGX_SIDE_STREAM=0" "AMD_SERIALIZE_KERNEL=3" "SGX_FILTER_PLANES=0" "SGX_FUSE_BN_REDUCE=0"; do
  echo "== $cfg"
  env $cfg timeout 400 python tools/branch_flake_probe.py 400 200 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-400
done | tee "$OUT/d_alpha_flips.txt"
