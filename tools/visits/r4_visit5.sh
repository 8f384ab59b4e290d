#!/bin/bash
# Round 4, visit 5 (~3 GPU-minutes): pre-split filter planes for the patch conv kernel; patch weight gradient with 64-pixel tiles.
TAG=${1:-r4h}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 200 python -m pytest tests/test_kernels.py tests/test_blocks.py -m gpu -q -k "pconv or wgrad_patch or dgrad_carries or bn_reduce_rides or csp_layer or qarepvgg_block_two" > "$OUT/pytest_new.log" 2>&1
tail -3 "$OUT/pytest_new.log" | cut -c1-300
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-predict"
for cfg in "A=1" "SGX_WSPLIT=0" "A=2" "SGX_WSPLIT=0 B=2"; do
  timeout 120 env $cfg $B > "$OUT/bench_$cfg.json" 2> "$OUT/bench_$cfg.err"
  python - "$OUT/bench_$cfg.json" "$cfg" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; pk=r.get("patch_kernel",{})
    print(sys.argv[2], d["value"], "img/s", d["ms_per_step"], "ms | igemm+pconv", r["achieved"], r["kernel_ms_per_step"], "| pconv", pk.get("kernel_ms_per_step"), pk.get("exclusive_algorithmic_tflops"), "| wgrad", r["wgrad"]["achieved"], r["wgrad"]["kernel_ms_per_step"], "| excl", r["exclusive"]["achieved"], r["exclusive"]["wgrad_achieved"], "host", d.get("host_enqueue_ms_per_step"), "loss", d["config"].get("final_loss"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
