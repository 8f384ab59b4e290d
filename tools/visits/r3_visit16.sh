#!/bin/bash
# true two-slab prefetch (unconditional re-issue): GPU parity of the weight-gradient tests, lab alone, step A/B
TAG=${1:-r3zc}
OUT=$(pwd)/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 120 python -m pytest tests/test_kernels.py tests/test_blocks.py -m gpu -q -x -k "wgrad or bwd_weight or conv_bwd or conv_block or csp_layer" > "$OUT/pytest_wgrad.log" 2>&1; tail -1 "$OUT/pytest_wgrad.log"
timeout 200 python tools/wgrad_lab.py --configs base,pf1,w2 --rounds 2 --iters 4 --out "$OUT/wgrad_lab.txt" > "$OUT/wgrad_lab.log" 2>&1; tail -3 "$OUT/wgrad_lab.log" | cut -c1-200
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-predict"
for cfg in "A=1" "SGX_WGRAD_PF=1" "A=2"; do
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
