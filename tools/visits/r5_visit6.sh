#!/bin/bash
# Round 5, visit 6: sampled one-pass NMS candidate selection - GPU parity (every nms test), boxes/s under both selections (interleaved),
# kernel stats of the NMS call; the weight-gradient drift run (tools/wgrad_drift.py).
TAG=${1:-r5i}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 400 python -m pytest tests/test_kernels.py tests/test_known_answers.py -m gpu -q -k "nms" > "$OUT/pytest_nms.log" 2>&1
tail -3 "$OUT/pytest_nms.log" | cut -c1-300
timeout 200 python tools/nms_bench.py --iters 200 > "$OUT/nms_bench.txt" 2> "$OUT/nms_bench.err"; cat "$OUT/nms_bench.txt"; tail -2 "$OUT/nms_bench.err"
cd /tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/nms_stats" -o p -- bash -c "cd $REPO && python tools/nms_bench.py --iters 50 --selection 1" > "$OUT/nms_stats.log" 2>&1
(cd $REPO && python tools/prof_summary.py stats "$OUT/nms_stats" > "$OUT/nms_kernel_stats_summary.txt" 2>&1; head -14 "$OUT/nms_kernel_stats_summary.txt" | cut -c1-200)
find "$OUT/nms_stats" -name "*kernel_trace.csv" -size +8M -delete
cd $REPO
timeout 300 python tools/wgrad_drift.py --steps 200 --out "$OUT/wgrad_drift.txt" > "$OUT/wgrad_drift.log" 2>&1; cat "$OUT/wgrad_drift.txt"; tail -2 "$OUT/wgrad_drift.log"
