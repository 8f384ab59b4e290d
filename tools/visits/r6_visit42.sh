#!/bin/bash
# Round 6, visit 42: what the d alpha dot costs when the main chain first waits for the side stream (SGX_DALPHA_SYNC=1: no flips in 1600 steps)
TAG=${1:-r6ar}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
run() { env $1 timeout 300 python bench.py $2 --steps 20 --warmup 5 --no-cpu-baseline --no-nms --no-predict --no-exclusive --other-configs off 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('final_loss'))"; }
for rep in 1 2 3; do
  for cfg in "SGX_NONE=0" "SGX_DALPHA_SYNC=1" "SGX_DALPHA_SYNC=1 SGX_BRANCH_SITES=31"; do
    echo "S rep $rep [$cfg]: $(run "$cfg" "")"
  done
done | tee "$OUT/dalpha_sync_cost.txt"
for cfg in "SGX_NONE=0" "SGX_DALPHA_SYNC=1"; do echo "M [$cfg]: $(run "$cfg" "--model m")"; done | tee -a "$OUT/dalpha_sync_cost.txt"
