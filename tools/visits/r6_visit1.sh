#!/bin/bash
# Round 6, first visit: the round's new GPU tests, the full default bench line (other_configs + oracle loss checks at full size under one
# clock), and a per-dispatch trace of the step condensed to one line per launch (tools/trace_compact.py) for the launch-shape analysis.
#   gpurun --timeout 1500 -- 'bash tools/visits/r6_visit1.sh r6a'
TAG=${1:-r6a}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
rocminfo 2>/dev/null | grep -m3 -E "gfx950|Compute Unit|Max Clock" > "$OUT/rocminfo.txt"
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q -x --durations=8 -k "other_baseline_configs or bs64_loss or 80_classes_640 or raw_pointer or nms_sampled or filter_planes or test_half or conv_every_tile_shape or qarepvgg" > "$OUT/pytest_new.log" 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)" >> "$OUT/pytest_new.log"; tail -14 "$OUT/pytest_new.log"
t0=$(date +%s)
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$? ($(( $(date +%s) - t0 )) s)" >> "$OUT/bench.err"; tail -2 "$OUT/bench.err"
python - "$OUT/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("S", d["value"], d["ms_per_step"], "ms | conv", r["achieved"], r["frac"], r["kernel_ms_per_step"], "| wgrad", r["wgrad"]["achieved"], r["wgrad"]["kernel_ms_per_step"], "| step_mfma_frac", r["step_mfma_frac"], "| excl", r["exclusive"]["frac"])
print("loss_check", d["config"].get("loss_check_vs_oracle",{}).get("max_rel_err"), "| nms", d.get("nms",{}).get("value"), d.get("nms",{}).get("stage2_fallbacks"), "| predict", d.get("predict",{}).get("value"))
for o in d.get("other_configs", []):
    print(o.get("config"), o.get("value"), o.get("ms_per_step"), o.get("step_mfma_frac"), (o.get("loss_check_vs_oracle") or {}).get("max_rel_err"), o.get("error"))
PY
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o bench -- bash -c "cd $REPO && python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-nms --no-predict --no-exclusive" > "$OUT/stats.log" 2>&1
cd "$REPO"
python tools/prof_summary.py stats "$OUT/stats" > "$OUT/kernel_stats_summary.txt" 2>&1
python tools/prof_summary.py timeline "$OUT/stats" 4 > "$OUT/kernel_timeline_summary.txt" 2>&1
python tools/trace_compact.py "$OUT/stats" "$OUT/dispatches.csv"
rm -rf "$OUT/stats"
head -12 "$OUT/kernel_timeline_summary.txt"
du -sh "$OUT"
