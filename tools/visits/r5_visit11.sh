#!/bin/bash
# Round 5, visit 11: RepVGG backward with the fused reduce, RepVGGBlock(use_alpha=True): block / PP-YOLOE parity, PP-YOLOE-S step A/B.
TAG=${1:-r5q}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 500 python -m pytest tests/test_blocks.py tests/test_pp_yolo_e.py tests/test_kernels.py -m gpu -q -k "repvgg or pp_yolo or ppyoloe or dual_affine or basic_block or csp_res" > "$OUT/pytest_a.log" 2>&1
tail -3 "$OUT/pytest_a.log" | cut -c1-300
for rep in 1 2; do for cfg in "SGX_REPVGG_FUSED_REDUCE=0" "A=1"; do
  timeout 150 env $cfg python bench.py --workload ppyoloe --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-predict --no-exclusive > "$OUT/bench_${cfg}_$rep.json" 2> "$OUT/bench_${cfg}_$rep.err"
  python - "$OUT/bench_${cfg}_$rep.json" "$cfg" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d["value"], d["unit"], d["ms_per_step"], "ms", d["roofline"].get("step_mfma_frac"))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
done; done
