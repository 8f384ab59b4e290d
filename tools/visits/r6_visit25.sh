#!/bin/bash
# Round 6, visit 25: sgx_relu_bwd_bn_reduce (the ResNet blocks' final ReLU mask + the last BatchNorm's backward reduce in one sweep): kernel and
# model parity on the chip, ResNet-50 A/B; the S step at the new defaults twice (eager weight-gradient flush at the large maps).
TAG=${1:-r6aa}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_resnet.py tests/test_kernels.py -m gpu -q -x -k "resnet or relu_bwd or bn" 2>&1 | tail -4 | tee "$OUT/pytest_resnet_bn.txt"
B="python bench.py --workload resnet50 --steps 20 --warmup 5"
for rep in 1 2 3; do
  for mode in 0 1; do
    v=$(timeout 200 env SGX_RESNET_RELU_REDUCE=$mode $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['final_loss'])")
    echo "resnet50 rep $rep relu_reduce=$mode: $v"
  done
done | tee "$OUT/resnet_relu_reduce_ab.txt"
B="python bench.py --no-cpu-baseline --no-nms --no-predict --no-exclusive --other-configs off --steps 20 --warmup 5"
for rep in 1 2; do
  for e in 99999999999 800000; do
    v=$(timeout 200 env SGX_WGRAD_EAGER_ROWS=$e $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "S rep $rep eager_rows=$e: $v"
  done
done | tee "$OUT/s_defaults.txt"
