#!/bin/bash
# Round 6, visit 43: tools/dot_race_probe.py - sgx_dot on fixed operands beside other kernels
TAG=${1:-r6as}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python tools/dot_race_probe.py 4000 2>&1 | grep -v amdgpu.ids | tail -12 | tee "$OUT/dot_race_probe.txt"
