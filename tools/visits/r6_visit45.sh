#!/bin/bash
# Round 6, visit 45: the step probe on the final build for longer (fresh networks, shuffled pools, single-chain x2 + branch-stream, whole arena compared)
TAG=${1:-r6av}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python tools/branch_flake_probe.py 2000 600 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-330 | tee "$OUT/branch_flake_probe_final_build.txt"
timeout 200 python tools/dot_race_probe.py 4000 2>&1 | grep -v amdgpu.ids | tail -9 | tee "$OUT/dot_race_probe_final_build.txt"
