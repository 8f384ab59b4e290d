#!/bin/bash
# Round 4, first GPU visit (~10 GPU-minutes): the switches round 3 left unmeasured or un-gated, in the order that decides the most.
#   gpurun --timeout 900 -- 'bash tools/visits/r4_first_visit.sh r4a'
# 1. adoption gate of the bf16x3 weight-gradient loop: the whole GPU parity suite with SGX_WGRAD_MATH=bf16x3 (minus the 135-s full-size
#    gradient test, which gets its own line)           2. the loop's tile preference (lab, alone)        3. step A/B of the switches
TAG=${1:-r4a}
OUT=$(pwd)/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
SGX_WGRAD_MATH=bf16x3 timeout 420 python -m pytest tests -m gpu -q -k "not headline_config_backward_exact" > "$OUT/pytest_gpu_wgrad_bf16x3.log" 2>&1
tail -4 "$OUT/pytest_gpu_wgrad_bf16x3.log" | cut -c1-300
timeout 120 python tools/wgrad_lab.py --configs base,bf16,bf16+t128x128,bf16+t128x64,bf16+t64x128,bf16+t96x128,bf16+g6.16.1 --rounds 2 --iters 4 \
    --out "$OUT/wgrad_lab_bf16_tiles.txt" > "$OUT/wgrad_lab_bf16_tiles.log" 2>&1
tail -3 "$OUT/wgrad_lab_bf16_tiles.log" | cut -c1-260
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-predict"
for cfg in "A=1" "SGX_WGRAD_MATH=bf16x3" "SGX_CONV_MATH=patch_auto" "SGX_WGRAD_MATH=bf16x3 SGX_CONV_MATH=patch_auto" "A=2" "SGX_WGRAD_MATH=bf16x3 B=2"; do
  timeout 120 env $cfg $B > "$OUT/bench_$cfg.json" 2> "$OUT/bench_$cfg.err"
  python - "$OUT/bench_$cfg.json" "$cfg" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[2], d["value"], "img/s", d["ms_per_step"], "ms | igemm", r["achieved"], "| wgrad", r["wgrad"]["achieved"], r["wgrad"]["kernel_ms_per_step"], "| excl", r["exclusive"]["achieved"], r["exclusive"]["wgrad_achieved"], "loss", d["config"].get("final_loss"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
SGX_WGRAD_MATH=bf16x3 SGX_TEST_DUMP="$OUT/headline_backward_bf16x3.txt" timeout 300 python -m pytest tests/test_yolo_nas.py -m gpu -q -k "headline_config_backward_exact" > "$OUT/pytest_headline_bf16x3.log" 2>&1
tail -2 "$OUT/pytest_headline_bf16x3.log" | cut -c1-300
