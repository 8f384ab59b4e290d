#!/bin/bash
# Round 5, visit 10: fp32 implicit GEMM with all slabs up front (variant 11): parity, per-problem search with variants 6 / 11 among the
# candidates, step A/B committed table / new table.
TAG=${1:-r5p}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_kernels.py -m gpu -q -k "deep_slabs or every_tile or tuning_table or conv_fwd or conv_bwd" > "$OUT/pytest_conv.log" 2>&1
tail -3 "$OUT/pytest_conv.log" | cut -c1-300
timeout 500 python tools/conv_tune.py --iters 5 --out "$OUT/conv_tune.txt" --emit-table "$OUT/conv_tuning_new.json" > "$OUT/conv_tune.log" 2>&1
tail -2 "$OUT/conv_tune.log" | cut -c1-300
grep -c "variant=11" "$OUT/conv_tune.txt"; grep "variant=11" "$OUT/conv_tune.txt" | head -30 | cut -c1-200
BENCH_ARGS="--no-exclusive" bash tools/visits/r4_ab.sh "$TAG" "A=1" "SGX_CONV_TUNING=$OUT/conv_tuning_new.json"
