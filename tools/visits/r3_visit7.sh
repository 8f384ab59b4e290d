#!/bin/bash
TAG=${1:-r3g}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
export SGX_TEST_DUMP="$OUT/headline_backward.txt"
timeout 500 python -m pytest tests/test_yolo_nas.py -m gpu -x -q -k "headline_config_backward_exact" > "$OUT/pytest_headline.log" 2>&1; tail -3 "$OUT/pytest_headline.log" | cut -c1-900
unset SGX_TEST_DUMP
timeout 900 python -m pytest tests -m gpu -q --durations=8 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"; tail -12 "$OUT/pytest_gpu.log" | cut -c1-300
SGX_CONV_MATH=patch timeout 900 python -m pytest tests -m gpu -q -k "not headline_config_backward" > "$OUT/pytest_gpu_patch.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu_patch.log"; tail -8 "$OUT/pytest_gpu_patch.log" | cut -c1-300
SGX_WGRAD_GROUP_GFLOP=0 timeout 300 python tools/conv_bench.py --iters 6 --out "$OUT/conv_bench_fp32.txt" > "$OUT/conv_bench_fp32.log" 2>&1; head -2 "$OUT/conv_bench_fp32.txt"
SGX_WGRAD_GROUP_GFLOP=0 SGX_CONV_MATH=patch timeout 300 python tools/conv_bench.py --iters 6 --out "$OUT/conv_bench_patch.txt" > "$OUT/conv_bench_patch.log" 2>&1; head -2 "$OUT/conv_bench_patch.txt"
du -sh "$OUT"
