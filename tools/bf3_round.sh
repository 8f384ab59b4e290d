#!/bin/bash
# One short GPU visit for the bf16x3 conv arithmetic: parity subset, train-step bench and per-shape conv replay in that mode.
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/bf3_round.sh r1y'
TAG=${1:-bf3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 200 python -m pytest tests/test_kernels.py -m gpu -q -k "conv_bf16x3" > $OUT/pytest_bf16x3.log 2>&1; tail -2 $OUT/pytest_bf16x3.log
SGX_CONV_MATH=${MATH:-bf16x3} timeout 200 python bench.py --no-nms --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/bench_bf16x3.log 2> $OUT/bench_bf16x3.err
python - <<P
import json
try:
    r = json.loads(open("$OUT/bench_bf16x3.log").read().strip().splitlines()[-1])
    print("bench bf16x3:", r["value"], "img/s", r["ms_per_step"], "ms; igemm", r["roofline"]["achieved"], "TF, kernel ms/step", r["roofline"]["kernel_ms_per_step"])
except Exception as e:
    print("bench failed", e); print(open("$OUT/bench_bf16x3.err").read()[-2000:])
P
SGX_CONV_MATH=${MATH:-bf16x3} timeout 300 python tools/conv_bench.py --out $OUT/conv_bench_bf16x3.txt > $OUT/conv_bench.log 2>&1; tail -3 $OUT/conv_bench.log
