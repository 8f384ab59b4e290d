#!/bin/bash
# Round 5, visit 13: does the bf16x3 GEMM's time follow its vector-instruction count?  The same lab under the product library and under a
# build whose filter operand is NOT split (one truncated plane: 4 instead of 22 vector instructions per item; wrong numerics, timing only).
TAG=${1:-r5v}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
LIB=super_gradients_amd/csrc/libsgx_hip.so
cp $LIB /tmp/product.so
P="fwd:32:80:80:192:96:1:1,fwd:32:40:40:384:192:1:1,dgrad:32:80:80:192:192:1:1,fwd:32:20:20:768:384:1:1,fwd:32:80:80:192:384:3:2,fwd:32:160:160:96:192:3:2,fwd:32:20:20:256:256:3:1,dgrad:32:40:40:192:192:3:2"
for which in product alt product alt; do
  if [ $which = alt ]; then cp _alt/libsgx_alt.so $LIB; else cp /tmp/product.so $LIB; fi
  echo "== $which"; timeout 200 python tools/conv_lab.py --math bf16x3 --variants 0 --rounds 3 --iters 10 --problems $P 2>&1 | tail -12 | cut -c1-120
done > "$OUT/lab.txt" 2>&1
cp /tmp/product.so $LIB
cat "$OUT/lab.txt"
