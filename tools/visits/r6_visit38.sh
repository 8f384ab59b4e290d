#!/bin/bash
# Round 6, visit 38: tools/branch_flake_probe.py - fresh networks, shuffled allocator pools, the first two steps of single-chain / single-chain / branch-stream
TAG=${1:-r6an}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python tools/branch_flake_probe.py 400 600 2>&1 | grep -v amdgpu.ids | tail -40 | tee "$OUT/branch_flake_probe.txt"
