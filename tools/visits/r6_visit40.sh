#!/bin/bash
# Round 6, visit 40: which operand of the d alpha dot is disturbed - its workspace (private buffer), its place in the chain (behind the block's
# other launches), or the side stream's overlap at that moment (main waits for the side stream first)
TAG=${1:-r6ap}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for cfg in "SGX_DOT_PRIVATE_WS=1" "SGX_DALPHA_LATE=1" "SGX_DALPHA_SYNC=1"; do
  echo "== $cfg"
  env $cfg timeout 400 python tools/branch_flake_probe.py 400 200 2>&1 | grep -v amdgpu.ids | tail -5 | cut -c1-330
done | tee "$OUT/d_alpha_flips_2.txt"
