#!/bin/bash
# Round 5, visit 8: per-channel constants of the sweeps hoisted out of the row loops - BatchNorm / block parity, library A/B of the step
# (alt = the build before), kernel stats of three steps for the sweep kernels' times.
TAG=${1:-r5m}; ALT=${2:-_alt/libsgx_alt.so}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 400 python -m pytest tests/test_kernels.py tests/test_blocks.py -m gpu -q -k "bn or affine or qarep or dual or axpy or relu or colsum or stats or residual or block or csp" > "$OUT/pytest_bn.log" 2>&1
tail -3 "$OUT/pytest_bn.log" | cut -c1-300
bash tools/visits/r4_lib_ab.sh "$TAG" "$ALT"
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o bench -- bash -c "cd $REPO && python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-nms --no-predict --no-exclusive" > "$OUT/stats.log" 2>&1
cd "$REPO"; python tools/prof_summary.py stats "$OUT/stats" > "$OUT/kernel_stats_summary.txt" 2>&1; grep -E "sweep|kernel  " "$OUT/kernel_stats_summary.txt" | head -14 | cut -c1-200
find "$OUT/stats" -name "*kernel_trace.csv" -size +8M -delete
