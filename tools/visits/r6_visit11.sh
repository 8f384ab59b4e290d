#!/bin/bash
# Round 6, visit 11: occupancy cap of the implicit-GEMM launches (dynamic-LDS pad) on the memory-bound problems.
TAG=${1:-r6l}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
P="fwd:32:160:160:96:96:1:1,fwd:32:160:160:64:96:1:1,dgrad:32:160:160:96:32:1:1,dgrad:32:160:160:96:96:1:1,fwd:32:80:80:96:96:1:1,dgrad:32:80:80:96:48:1:1,dgrad:32:80:80:288:96:1:1,fwd:32:80:80:192:64:1:1,dgrad:32:80:80:192:96:1:1,fwd:32:40:40:384:192:1:1,fwd:32:80:80:192:384:3:2"
for m in fp32 bf16x3; do
  timeout 400 python tools/conv_lab.py --math $m --planes 1 --ldspad 0,8,16,24,32 --problems "$P" --rounds 5 --iters 10 --out "$OUT/lab_$m.txt" > "$OUT/lab_$m.log" 2>&1
  cat "$OUT/lab_$m.txt"; tail -2 "$OUT/lab_$m.log"
done
