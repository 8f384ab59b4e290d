#!/bin/bash
# Round 5, visit 9: 32-channel max-pool scatter backward, one-wave-per-image targets index, LDS-staged DFL decode: parity, library A/B of the
# step (alt = the build before), kernel stats of three steps.
TAG=${1:-r5n}; ALT=${2:-_alt/libsgx_alt.so}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 500 python -m pytest tests/test_kernels.py tests/test_blocks.py tests/test_yolo_nas.py -m gpu -q -k "loss or decode or targets or maxpool or spp or atss or tal or golden or headline" > "$OUT/pytest_a.log" 2>&1
tail -3 "$OUT/pytest_a.log" | cut -c1-300
bash tools/visits/r4_lib_ab.sh "$TAG" "$ALT"
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o bench -- bash -c "cd $REPO && python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-nms --no-predict --no-exclusive" > "$OUT/stats.log" 2>&1
cd "$REPO"; python tools/prof_summary.py stats "$OUT/stats" > "$OUT/kernel_stats_summary.txt" 2>&1; grep -E "maxpool|decode|targets_index|tal_cand|box_loss|cls_loss|sigmoid|kernel  " "$OUT/kernel_stats_summary.txt" | head -14 | cut -c1-200
find "$OUT/stats" -name "*kernel_trace.csv" -size +8M -delete
