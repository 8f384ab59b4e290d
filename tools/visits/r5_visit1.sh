#!/bin/bash
# Round 5, visit 1 (~8 GPU-minutes): the half-precision inference path on the chip for the first time - parity tests, predict() throughput
# bf16 against fp32, tile / slab-depth sweeps of the bf16 conv kernel, a rocprofv3 kernel-stats pass of the bf16 predict loop; and a short
# train-step bench to see that the round's hygiene changes left the step where it was.
TAG=${1:-r5a}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
rocminfo 2>/dev/null | grep -m3 -E "gfx950|Compute Unit|Max Clock" > "$OUT/rocminfo.txt"
timeout 300 python -m pytest tests/test_half.py -m gpu -q > "$OUT/pytest_half.log" 2>&1
tail -5 "$OUT/pytest_half.log" | cut -c1-400
for args in "" "--fp32" "--tile 64 64 0" "--tile 128 64 0" "--tile 128 128 0" "--tile 0 0 32" "--tile 0 0 64" "--model m" "--model m --fp32"; do
  name=$(echo "predict$args" | tr ' ' '_' | tr -d '-')
  timeout 150 python tools/predict_bench.py --batches 10 $args > "$OUT/$name.json" 2> "$OUT/$name.err"
  echo "$args :: $(tail -1 "$OUT/$name.json" | cut -c1-700)"
done
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats_predict" -o p -- bash -c "cd $REPO && python tools/predict_bench.py --batches 5" > "$OUT/stats_predict.log" 2>&1
cd "$REPO"
python tools/prof_summary.py stats "$OUT/stats_predict" > "$OUT/predict_kernel_stats_summary.txt" 2>&1
head -30 "$OUT/predict_kernel_stats_summary.txt"
find "$OUT/stats_predict" -name "*kernel_trace.csv" -size +8M -delete
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-predict --no-exclusive > "$OUT/bench_short.json" 2> "$OUT/bench_short.err"
python - "$OUT/bench_short.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print("train", d["value"], "img/s", d["ms_per_step"], "ms | dtype", d["dtype"], "| conv", r["achieved"], r["frac"], r.get("frac_of_executed_pipe"), "| bound", r["per_launch_bound"]["frac"], "| host", d.get("host_enqueue_ms_per_step"))
except Exception as e:
    print("train bench FAILED", e)
PY
tail -2 "$OUT/bench_short.err"
