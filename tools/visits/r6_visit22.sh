#!/bin/bash
# Round 6, visit 22: branch stream on by default (forward + backward, all sites, two lanes).  Parity (whole-model tests, the new bit-identity
# test, blocks, distributed), then the late join of the up stages' skip gradients and the early fork of their forward branches.
TAG=${1:-r6x}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_yolo_nas.py tests/test_blocks.py tests/test_distributed.py tests/test_trainer.py -m gpu -q -x 2>&1 | tail -4 | tee "$OUT/pytest_branch_default.txt"
B="python bench.py --no-cpu-baseline --no-nms --no-predict --no-exclusive --other-configs off --steps 20 --warmup 5"
one() { timeout 200 env SGX_BRANCH_STREAM=$1 SGX_BRANCH_LANES=$2 SGX_BRANCH_LATE_JOIN=$3 SGX_BRANCH_EARLY_FORK=$4 $B $5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('final_loss'))"; }
for rep in 1 2 3; do
  for cfg in "0 1 0 0" "3 2 0 0" "3 2 1 0" "3 2 0 1" "3 2 1 1" "3 3 1 1"; do
    echo "S rep $rep [mode lanes late early = $cfg]: $(one $cfg)"
  done
done | tee "$OUT/branch_late_early_s.txt"
for m in m l; do
  for cfg in "0 1 0 0" "3 2 1 1" "0 1 0 0" "3 2 1 1"; do
    echo "$m [mode lanes late early = $cfg]: $(one $cfg "--model $m")"
  done
done | tee "$OUT/branch_late_early_ml.txt"
