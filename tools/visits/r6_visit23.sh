#!/bin/bash
# Round 6, visit 23: the rest of the GPU suite at the new defaults (branch stream on), ResNet-50 with / without the projection shortcuts on
# the branch stream, and a kernel trace of the ResNet-50 step.
TAG=${1:-r6y}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_resnet.py tests/test_pp_yolo_e.py tests/test_predict.py tests/test_half.py -m gpu -q -x 2>&1 | tail -4 | tee "$OUT/pytest_rest.txt"
B="python bench.py --workload resnet50 --steps 20 --warmup 5"
for rep in 1 2 3; do
  for sites in 15 31; do
    v=$(timeout 200 env SGX_BRANCH_SITES=$sites $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['final_loss'])")
    echo "resnet50 rep $rep sites=$sites: $v"
  done
done | tee "$OUT/resnet_shortcut_branch_ab.txt"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o bench -- bash -c "cd $REPO && python bench.py --workload resnet50 --steps 5 --warmup 3" > "$OUT/stats.log" 2>&1
cd "$REPO"
python tools/prof_summary.py stats "$OUT/stats" > "$OUT/resnet50_kernel_stats_summary.txt" 2>&1
python tools/prof_summary.py timeline "$OUT/stats" 4 > "$OUT/resnet50_kernel_timeline_summary.txt" 2>&1
head -45 "$OUT/resnet50_kernel_stats_summary.txt"; head -12 "$OUT/resnet50_kernel_timeline_summary.txt"
