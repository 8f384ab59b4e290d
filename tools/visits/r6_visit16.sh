#!/bin/bash
# Round 6, visit 16: tiles of the two-output / two-source GEMM launches (stride-2 QARepVGG blocks, small-map blocks): they always ran the
# heuristic's tile - the measurement override now reaches them.
TAG=${1:-r6r}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
P="fwd2:32:320:320:48:96:3:2,dgrad2:32:320:320:48:96:3:2,fwd2:32:160:160:96:192:3:2,dgrad2:32:160:160:96:192:3:2,fwd2:32:80:80:192:384:3:2,dgrad2:32:80:80:192:384:3:2,fwd2:32:40:40:384:768:3:2,dgrad2:32:40:40:384:768:3:2,fwd2:32:20:20:192:192:3:1,dgrad2:32:20:20:192:192:3:1,fwd2:32:20:20:64:64:3:1,dgrad2:32:20:20:64:64:3:1"
timeout 600 python tools/conv_lab.py --math bf16x3 --planes 1 --tiles 0x0,128x96,128x32,64x64,64x32 --problems "$P" --rounds 5 --iters 8 --out "$OUT/ph2_tiles_lab.txt" > "$OUT/lab.log" 2>&1
cat "$OUT/ph2_tiles_lab.txt"; tail -2 "$OUT/lab.log"
