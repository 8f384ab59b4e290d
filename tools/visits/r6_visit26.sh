#!/bin/bash
# Round 6, visit 26: the bottlenecks' d alpha = <x, dz> reductions on the second lane of the branch stream (site 32).
TAG=${1:-r6ab}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_yolo_nas.py -m gpu -q -x -k "branch or train_step_parity or headline_config_parity" 2>&1 | tail -3 | tee "$OUT/pytest_site32.txt"
B="python bench.py --no-cpu-baseline --no-nms --no-predict --no-exclusive --other-configs off --steps 20 --warmup 5"
one() { timeout 200 env SGX_BRANCH_SITES=$1 $B $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('final_loss'))"; }
for rep in 1 2 3; do
  for sites in 31 63; do
    echo "S rep $rep sites=$sites: $(one $sites)"
  done
done | tee "$OUT/dalpha_lane_s.txt"
for m in m l; do
  for sites in 31 63 31 63; do
    echo "$m sites=$sites: $(one $sites "--model $m")"
  done
done | tee "$OUT/dalpha_lane_ml.txt"
