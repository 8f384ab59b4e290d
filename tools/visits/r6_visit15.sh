#!/bin/bash
# Round 6, visit 15: the RGB stem's QARepVGG block on the two-output launch (flattened K axis): parity on the chip, step A/B against the
# general sequence (SGX_QAREP_STEM_DUAL=0).
TAG=${1:-r6q}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_blocks.py tests/test_yolo_nas.py -m gpu -q -x -k "rgb_stem or two_branch or s_train_step_parity or golden or s_backward_exact or headline_config_parity" 2>&1 | tail -4
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-predict --no-exclusive"
for rep in 1 2 3; do
  for mode in 0 1; do
    v=$(timeout 200 env SGX_QAREP_STEM_DUAL=$mode $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['config']['final_loss'])")
    echo "rep $rep stem_dual=$mode: $v"
  done
done | tee "$OUT/stem_dual_ab.txt"
