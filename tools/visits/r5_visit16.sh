#!/bin/bash
# Round 5, visit 16: pre-split filter planes on the other configurations (M, L, ResNet-50, PP-YOLOE-S): step A/B of the switch, each setting
# twice, interleaved.
TAG=${1:-r5z}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for cfg in "--model m" "--model l" "--workload resnet50" "--workload ppyoloe"; do
  for rep in 1 2; do
    for pl in 0 1; do
      v=$(SGX_FILTER_PLANES=$pl timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-predict --no-exclusive $cfg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
      echo "$cfg planes=$pl : $v"
    done
  done
done | tee "$OUT/ab_configs.txt"
