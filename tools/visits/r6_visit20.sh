#!/bin/bash
# Round 6, visit 20: branch stream, which call sites and up to what launch size (SGX_BRANCH_SITES, SGX_BRANCH_MAX_TILES) - S, then M and L.
TAG=${1:-r6v}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
SGX_BRANCH_STREAM=3 timeout 900 python -m pytest tests/test_yolo_nas.py -m gpu -q -x 2>&1 | tail -3 | tee "$OUT/pytest_branch3_sites7.txt"
B="python bench.py --no-cpu-baseline --no-nms --no-predict --no-exclusive --other-configs off --steps 20 --warmup 5"
one() { timeout 200 env SGX_BRANCH_STREAM=$1 SGX_BRANCH_SITES=$2 SGX_BRANCH_MAX_TILES=$3 $B $4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('final_loss'))"; }
for rep in 1 2; do
  echo "S rep $rep off: $(one 0 7 1000000)"
  for sites in 1 2 4 7; do
    echo "S rep $rep mode 3 sites $sites: $(one 3 $sites 1000000)"
  done
  for mt in 400 1600 6400; do
    echo "S rep $rep mode 3 sites 7 max tiles $mt: $(one 3 7 $mt)"
  done
done | tee "$OUT/branch_sites_s.txt"
for m in m l; do
  for cfg in "0 7 1000000" "3 7 1000000" "3 7 1600" "3 7 400" "3 2 1000000" "0 7 1000000" "3 7 1600"; do
    echo "$m [$cfg]: $(one $cfg "--model $m")"
  done
done | tee "$OUT/branch_sites_ml.txt"
