#!/bin/bash
# Round 6, visit 19: branch stream (SGX_BRANCH_STREAM bit 0 forward, bit 1 backward): YoloNASCSPLayer's conv2 chain and the coarse head levels
# beside the main chain.  Parity of the whole-model tests with it on, then interleaved step A/Bs (S; M and L once).
TAG=${1:-r6u}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
SGX_BRANCH_STREAM=3 timeout 900 python -m pytest tests/test_yolo_nas.py -m gpu -q -x 2>&1 | tail -3 | tee "$OUT/pytest_branch3.txt"
B="python bench.py --no-cpu-baseline --no-nms --no-predict --no-exclusive --other-configs off --steps 20 --warmup 5"
one() { timeout 200 env SGX_BRANCH_STREAM=$1 $B $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('final_loss'), d.get('host_enqueue_ms_per_step'))"; }
for rep in 1 2 3; do
  for mode in 0 1 3 2; do
    echo "S rep $rep branch=$mode: $(one $mode)"
  done
done | tee "$OUT/branch_ab_s.txt"
for m in m l; do
  for mode in 0 3 0 3; do
    echo "$m branch=$mode: $(one $mode "--model $m")"
  done
done | tee "$OUT/branch_ab_ml.txt"
