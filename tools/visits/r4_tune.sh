#!/bin/bash
# Round 4: the per-problem tile table re-measured under the round's default arithmetic (conv math patch_bf3, weight gradient bf16x3 + patch),
# then a step A/B: committed table / new table / no table.   gpurun --timeout 900 -- 'bash tools/visits/r4_tune.sh r4q'
TAG=${1:-r4q}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 420 python tools/conv_tune.py --wgrad --iters 5 --out "$OUT/conv_tune.txt" --emit-table "$OUT/conv_tuning_new.json" > "$OUT/conv_tune.log" 2>&1
tail -2 "$OUT/conv_tune.log" | cut -c1-300
head -30 "$OUT/conv_tune.txt" | cut -c1-200
BENCH_ARGS="" bash tools/visits/r4_ab.sh "$TAG" "A=1" "SGX_CONV_TUNING=$OUT/conv_tuning_new.json" "SGX_CONV_TUNING=0"
