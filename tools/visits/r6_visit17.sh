#!/bin/bash
# Round 6, visit 17: the neck's gradients of the backbone's feature maps added in the producing data gradient's epilogue (3 accumulate
# passes per step gone): parity on the chip, step A/B (SGX_BACKBONE_ADDEND=0: the passes of rounds 1 - 5).
TAG=${1:-r6s}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_yolo_nas.py -m gpu -q -x -k "train_step_parity or golden or backward_exact or headline_config_parity" 2>&1 | tail -3
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-predict --no-exclusive"
for rep in 1 2 3; do
  for mode in 0 1; do
    v=$(timeout 200 env SGX_BACKBONE_ADDEND=$mode $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['final_loss'])")
    echo "rep $rep fold=$mode: $v"
  done
done | tee "$OUT/backbone_addend_ab.txt"
for cfg in "--model m" "--model l --size 1280 --batch 8"; do
  for mode in 0 1; do
    v=$(timeout 200 env SGX_BACKBONE_ADDEND=$mode $B $cfg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "[$cfg] fold=$mode: $v"
  done
done | tee -a "$OUT/backbone_addend_ab.txt"
