#!/bin/bash
# Round 5, visit 17: kernel trace of the step on the planes build - per-kernel stats, timeline, and who issues the runtime's copy / fill kernels.
TAG=${1:-r5aa}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
PROF_CMD="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-nms --no-predict --no-exclusive"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o bench -- bash -c "cd $REPO && $PROF_CMD" > "$OUT/stats.log" 2>&1
cd "$REPO"
python tools/prof_summary.py stats "$OUT/stats" > "$OUT/kernel_stats_summary.txt" 2>&1
python tools/prof_summary.py timeline "$OUT/stats" 4 > "$OUT/kernel_timeline_summary.txt" 2>&1
python tools/prof_summary.py neighbours "$OUT/stats" "copyBuffer|fillBuffer|elementwise|at::native" 4 > "$OUT/runtime_kernel_neighbours.txt" 2>&1
find "$OUT/stats" -name "*.csv" -size +8M -delete
head -50 "$OUT/runtime_kernel_neighbours.txt" | cut -c1-200
