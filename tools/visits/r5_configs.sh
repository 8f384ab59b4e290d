#!/bin/bash
# Round 5: the other BASELINE.json configurations on the round's build - one bench line + one kernel-stats summary each.
#   gpurun --timeout 900 -- 'bash tools/visits/r4_configs.sh r5final_cfg'
TAG=${1:-r5final_cfg}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
run() {  # name, bench args
  local name=$1; shift
  timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-predict "$@" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python - "$OUT/bench_$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[2], d["value"], d["unit"], d["ms_per_step"], "ms | conv", r.get("achieved"), r.get("frac"), "| wgrad", r.get("wgrad",{}).get("achieved"), "| step_mfma_frac", r.get("step_mfma_frac"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  cd /tmp
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats_$name" -o bench -- bash -c "cd $REPO && python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-nms --no-predict --no-exclusive $*" > "$OUT/stats_$name.log" 2>&1
  cd "$REPO"
  python tools/prof_summary.py stats "$OUT/stats_$name" > "$OUT/kernel_stats_summary_$name.txt" 2>&1
  rm -rf "$OUT/stats_$name"
}
run m --model m
run l --model l
run l1280 --model l --size 1280 --batch 8
run resnet50 --workload resnet50
run ppyoloe_s --workload ppyoloe
du -sh "$OUT"
for args in "" "--fp32" "--model m" "--model m --fp32"; do
  name=$(echo "predict$args" | tr ' ' '_' | tr -d '-')
  timeout 150 python tools/predict_bench.py --batches 10 $args > "$OUT/$name.json" 2> "$OUT/$name.err"
  echo "$args :: $(tail -1 "$OUT/$name.json" | cut -c1-400)"
done
