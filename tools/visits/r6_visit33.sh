#!/bin/bash
# Round 6, visit 33: knob sweep at HEAD (the optima of rounds 3 - 5 were found before the branch stream existed): weight-gradient group size,
# branch size limit, eager flush rows, bf16x3 depth threshold, riding BatchNorm reduces.  Two interleaved repetitions, short bench form.
TAG=${1:-r6ai}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
run() { env $1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-nms --no-predict --no-exclusive --other-configs off 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('final_loss'))"; }
for rep in 1 2 3; do
  for cfg in "SGX_NONE=0" "SGX_WGRAD_GROUP_GFLOP=60" "SGX_WGRAD_GROUP_GFLOP=100" "SGX_WGRAD_GROUP_GFLOP=240" "SGX_WGRAD_GROUP_GFLOP=400" \
             "SGX_BRANCH_MAX_TILES=1500" "SGX_BRANCH_MAX_TILES=3000" "SGX_BRANCH_MAX_TILES=6000" "SGX_WGRAD_EAGER_ROWS=400000" "SGX_WGRAD_EAGER_ROWS=1600000" \
             "SGX_BF3_MIN_DEPTH=96" "SGX_BF3_MIN_DEPTH=128" "SGX_BF3_MIN_DEPTH=288" "SGX_FUSE_BN_REDUCE=0" "SGX_WT_BATCH=0" "SGX_FUSED_FINALIZE=0"; do
    echo "rep $rep [$cfg]: $(run "$cfg")"
  done
done | tee "$OUT/knob_sweep.txt"
