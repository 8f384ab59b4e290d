#!/bin/bash
# Round 6, visit 21: branch stream - the batch re-layout beside the filter preparations (site 8), two lanes for the head levels, and the
# weight-gradient group size with the branch stream on (its backward forks flush the queue early).
TAG=${1:-r6w}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
SGX_BRANCH_STREAM=3 SGX_BRANCH_SITES=15 SGX_BRANCH_LANES=2 timeout 900 python -m pytest tests/test_yolo_nas.py -m gpu -q -x 2>&1 | tail -3 | tee "$OUT/pytest_branch3_sites15_lanes2.txt"
B="python bench.py --no-cpu-baseline --no-nms --no-predict --no-exclusive --other-configs off --steps 20 --warmup 5"
one() { timeout 200 env SGX_BRANCH_STREAM=$1 SGX_BRANCH_SITES=$2 SGX_BRANCH_LANES=$3 SGX_WGRAD_GROUP_GFLOP=$4 $B $5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('final_loss'))"; }
for rep in 1 2; do
  for cfg in "0 7 1 160" "3 7 1 160" "3 15 1 160" "3 15 2 160" "3 15 1 80" "3 15 1 320" "0 7 1 80" "3 15 2 40"; do
    echo "S rep $rep [mode sites lanes group = $cfg]: $(one $cfg)"
  done
done | tee "$OUT/branch_lanes_group_s.txt"
