#!/bin/bash
# kernel trace of 5 steady train steps: per-kernel stats + where the wall clock goes (gaps per queue) inside whole steps
TAG=${1:-r4t}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o bench -- bash -c "cd $REPO && python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-nms --no-predict --no-exclusive ${BENCH_ARGS:-}" > "$OUT/stats.log" 2>&1
cd "$REPO"
python tools/prof_summary.py stats "$OUT/stats" > "$OUT/kernel_stats_summary.txt" 2>&1
python tools/prof_summary.py timeline "$OUT/stats" 4 > "$OUT/kernel_timeline_summary.txt" 2>&1
head -34 "$OUT/kernel_timeline_summary.txt" | cut -c1-200
find "$OUT" -name "*.csv" -size +8M -delete
