#!/bin/bash
# Round 6, visit 10: "fragments ahead" (conv variant 15) against the shipped loop in the lab.
TAG=${1:-r6k}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
P="fwd:32:40:40:384:192:1:1,fwd:32:80:80:192:384:3:2,fwd:32:20:20:1536:768:1:1,fwd:32:20:20:768:192:1:1,dgrad:32:80:80:192:96:1:1,fwd:32:80:80:192:64:1:1,fwd:32:40:40:192:192:3:2,dgrad:32:20:20:768:384:1:1"
timeout 500 python tools/conv_lab.py --math bf16x3 --planes 1 --tiles 64x64 --variants ${VARIANTS:-0,15} --problems "$P" --rounds 5 --iters 10 --out "$OUT/lab.txt" > "$OUT/lab.log" 2>&1
cat "$OUT/lab.txt"; tail -2 "$OUT/lab.log"
