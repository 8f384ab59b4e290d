#!/bin/bash
# Round 6, visit 24: the step's tail.  r6fin2's trace: the side stream idles 2.9 ms while the queue fills at the 160 x 160 maps, then runs a
# 3.9 ms backlog that ends 0.8 ms behind the main chain's last kernel.  SGX_WGRAD_EAGER_ROWS: flush the weight-gradient queue at once for
# layers with at least that many output pixels; SGX_SIDE_PRIORITY=-1: the side stream as a high-priority HIP stream.
TAG=${1:-r6z}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
B="python bench.py --no-cpu-baseline --no-nms --no-predict --no-exclusive --other-configs off --steps 20 --warmup 5"
one() { timeout 200 env SGX_WGRAD_EAGER_ROWS=$1 SGX_SIDE_PRIORITY=$2 SGX_BRANCH_PRIORITY=$3 $B $4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('final_loss'))"; }
for rep in 1 2 3; do
  for cfg in "99999999999 0 0" "800000 0 0" "200000 0 0" "50000 0 0" "99999999999 -1 0" "800000 -1 0" "99999999999 0 -1"; do
    echo "S rep $rep [eager_rows side_prio branch_prio = $cfg]: $(one $cfg)"
  done
done | tee "$OUT/tail_eager_priority_s.txt"
for m in m l; do
  for cfg in "99999999999 0 0" "800000 0 0" "200000 0 0" "99999999999 0 0" "800000 0 0" "200000 0 0"; do
    echo "$m [eager_rows side_prio branch_prio = $cfg]: $(one $cfg "--model $m")"
  done
done | tee "$OUT/tail_eager_priority_ml.txt"
