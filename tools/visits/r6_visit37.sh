#!/bin/bash
# Round 6, visit 37: tools/uninit_probe.py - recycled allocator blocks filled with NaN / 1e30 / 1e-30 before the step (see its header)
TAG=${1:-r6am}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python tools/uninit_probe.py 2>&1 | tail -20 | tee "$OUT/uninit_probe.txt"
