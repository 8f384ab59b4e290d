#!/bin/bash
# Round 6, visit 29: head levels started from inside the neck (SGX_HEADS_EARLY: the finest level's head beside the two down stages, forward
# and backward; four lanes), the SPP block's larger pools on lanes (site 64).  Parity first.
TAG=${1:-r6ae}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_yolo_nas.py tests/test_blocks.py tests/test_distributed.py tests/test_trainer.py -m gpu -q -x 2>&1 | tail -3 | tee "$OUT/pytest_heads_early.txt"
B="python bench.py --no-cpu-baseline --no-nms --no-predict --no-exclusive --other-configs off --steps 20 --warmup 5"
one() { timeout 200 env SGX_HEADS_EARLY=$1 SGX_BRANCH_SITES=$2 SGX_BRANCH_LANES=$3 $B $4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('final_loss'))"; }
for rep in 1 2 3; do
  for cfg in "0 63 2" "0 63 4" "0 127 4" "1 63 4" "1 127 4"; do
    echo "S rep $rep [heads_early sites lanes = $cfg]: $(one $cfg)"
  done
done | tee "$OUT/heads_early_s.txt"
for m in m l; do
  for cfg in "0 63 2" "1 127 4" "0 63 2" "1 127 4"; do
    echo "$m [heads_early sites lanes = $cfg]: $(one $cfg "--model $m")"
  done
done | tee "$OUT/heads_early_ml.txt"
