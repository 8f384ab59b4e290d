#!/bin/bash
# Round 5, visit 15: pre-split filter planes per problem - the lab with the planes switch as an axis (shipped arithmetic: patch_bf3).
TAG=${1:-r5x}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
P="fwd:32:80:80:192:96:1:1,fwd:32:40:40:384:192:1:1,dgrad:32:80:80:192:192:1:1,fwd:32:20:20:768:384:1:1,fwd:32:80:80:192:384:3:2,fwd:32:160:160:96:192:3:2,fwd:32:20:20:256:256:3:1,dgrad:32:40:40:192:192:3:2"
P="$P,fwd2:32:160:160:32:32:3:1,dgrad2:32:160:160:32:32:3:1,fwd2:32:80:80:64:64:3:1,dgrad2:32:80:80:64:64:3:1,fwd:32:80:80:64:64:3:1,dgrad:32:80:80:64:64:3:1,fwd2:32:40:40:96:96:3:1,fwd2:32:80:80:96:192:3:2,dgrad2:32:80:80:96:192:3:2,fwd:32:40:40:128:128:3:1"
timeout 400 python tools/conv_lab.py --math patch_bf3 --variants 0 --planes ${PLANES:-0,1,2} --rounds 5 --iters 10 --problems $P > "$OUT/lab.txt" 2>&1
cat "$OUT/lab.txt" | cut -c1-120
