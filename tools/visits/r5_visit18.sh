#!/bin/bash
# Round 5, visit 18: the per-problem (tile, variant) table re-measured as the step runs its convolutions now - pre-split filter planes,
# data gradients from pre-transposed weights - with variant 12 (filter fragments from the planes into registers) in the search; the
# committed table's weight-gradient entries are carried over.  Then the step A/B: committed table / new table, each twice.
TAG=${1:-r5ad}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 500 python tools/conv_tune.py --planes --iters 5 --out "$OUT/conv_tune.txt" --emit-table "$OUT/conv_tuning_new.json" \
    --keep-wgrad-from super_gradients_amd/csrc/conv_tuning_gfx950.json > "$OUT/conv_tune.log" 2>&1
tail -2 "$OUT/conv_tune.log" | cut -c1-300
head -24 "$OUT/conv_tune.txt" | cut -c1-200
BENCH_ARGS="" bash tools/visits/r4_ab.sh "$TAG" "A=1" "SGX_CONV_TUNING=$OUT/conv_tuning_new.json" "SGX_FILTER_PLANES=2"
