#!/bin/bash
# bf16x3 weight-gradient loop inside the step
OUT=$(pwd)/gpurun_out/${1:-r3zk}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-predict"
for cfg in "SGX_WGRAD_MATH=bf16x3" "A=1"; do
  timeout 40 env $cfg $B > "$OUT/bench_$cfg.json" 2> "$OUT/bench_$cfg.err"
  python - "$OUT/bench_$cfg.json" "$cfg" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[2], d["value"], "img/s", d["ms_per_step"], "ms | igemm", r["achieved"], "| wgrad", r["wgrad"]["achieved"], r["wgrad"]["kernel_ms_per_step"], "| excl", r["exclusive"]["achieved"], r["exclusive"]["wgrad_achieved"], "loss", d["config"].get("final_loss"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
