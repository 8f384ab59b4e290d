#!/bin/bash
# Round 6, visit 8: the ping-pong GEMM loop (conv variant 14) - parity on the chip, then the lab against the one-buffer loop.
TAG=${1:-r6i}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_kernels.py -m gpu -q -x -k "filter_planes_launches" > "$OUT/pytest_pp.log" 2>&1; tail -3 "$OUT/pytest_pp.log"
P="fwd:32:40:40:384:192:1:1,fwd:32:80:80:192:384:3:2,fwd:32:20:20:1536:768:1:1,fwd:32:20:20:768:192:1:1,dgrad:32:80:80:192:96:1:1,fwd:32:80:80:192:64:1:1,fwd:32:40:40:192:192:3:2,dgrad:32:20:20:768:384:1:1,fwd:32:160:160:96:96:1:1"
timeout 500 python tools/conv_lab.py --math bf16x3 --planes 1 --tiles 64x64,128x32,64x32 --variants 0,14 --problems "$P" --rounds 5 --iters 10 --out "$OUT/pingpong_lab.txt" > "$OUT/lab.log" 2>&1
cat "$OUT/pingpong_lab.txt"; tail -3 "$OUT/lab.log"
