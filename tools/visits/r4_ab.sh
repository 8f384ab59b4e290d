#!/bin/bash
# step A/B of environment switches:  gpurun --timeout 500 -- 'bash tools/visits/r4_ab.sh <tag> "A=1" "SGX_X=0" ...'   (each config twice, interleaved)
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-predict ${BENCH_ARGS:-}"
for rep in 1 2; do
for cfg in "$@"; do
  name=$(echo "$cfg" | tr ' /' '__' | tail -c 60)
  timeout 120 env $cfg $B > "$OUT/bench_${name}_$rep.json" 2> "$OUT/bench_${name}_$rep.err"
  python - "$OUT/bench_${name}_$rep.json" "$cfg" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[2], d["value"], "img/s", d["ms_per_step"], "ms | igemm", r["achieved"], r["kernel_ms_per_step"], "| wgrad", r["wgrad"]["achieved"], r["wgrad"]["kernel_ms_per_step"], "| excl", r["exclusive"]["achieved"], r["exclusive"]["wgrad_achieved"], "host", d.get("host_enqueue_ms_per_step"), "loss", d["config"].get("final_loss"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
done
