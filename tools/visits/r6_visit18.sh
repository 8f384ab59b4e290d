#!/bin/bash
# Round 6, visit 18: ResNet blocks with the BatchNorm-backward reduce riding in the data gradients (SGX_FUSE_BN_REDUCE=0: the reduce sweeps).
TAG=${1:-r6t}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_resnet.py tests/test_blocks.py -m gpu -q -x 2>&1 | tail -3
B="python bench.py --workload resnet50 --steps 20 --warmup 5"
for rep in 1 2 3; do
  for mode in 0 1; do
    v=$(timeout 200 env SGX_FUSE_BN_REDUCE=$mode $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['final_loss'])")
    echo "rep $rep fuse=$mode: $v"
  done
done | tee "$OUT/resnet_bn_reduce_ab.txt"
