#!/bin/bash
# the bf16x3 weight-gradient loop on the chip for the first time: parity test, then the lab (alone) against the fp32 loop
OUT=$(pwd)/gpurun_out/${1:-r3zj}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 40 python -m pytest tests/test_kernels.py -m gpu -q -x -k "wgrad_bf16x3_loop" > "$OUT/pytest_bf16.log" 2>&1; tail -3 "$OUT/pytest_bf16.log" | cut -c1-300
timeout 40 python tools/wgrad_lab.py --configs base,bf16 --rounds 1 --iters 3 --out "$OUT/wgrad_lab_bf16.txt" > "$OUT/wgrad_lab_bf16.log" 2>&1; tail -3 "$OUT/wgrad_lab_bf16.log" | cut -c1-200
