#!/bin/bash
# Round 6, visit 13: PP-YOLOE on the half-precision path - parity on the chip, predict() bf16 against fp32.
TAG=${1:-r6o}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_half.py tests/test_predict.py tests/test_pp_yolo_e.py -m gpu -q -x --durations=5 > "$OUT/pytest_half_ppyoloe.log" 2>&1; tail -12 "$OUT/pytest_half_ppyoloe.log"
for args in "--family ppyoloe" "--family ppyoloe --fp32" "--family ppyoloe --model l" "--family ppyoloe --model l --fp32" "--family ppyoloe --model m" ""; do
  timeout 200 python tools/predict_bench.py --batches 20 $args 2>"$OUT/err.txt" | tail -1 | cut -c1-330
done | tee "$OUT/predict_ppyoloe.txt"
tail -3 "$OUT/err.txt"
