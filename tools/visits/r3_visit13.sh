#!/bin/bash
# weight-gradient loop with two slabs of loads in flight: lab (alone) + step
TAG=${1:-r3n}
OUT=$(pwd)/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
LIB=super_gradients_amd/csrc/libsgx_hip.so
cp $LIB /tmp/new.so
cp tools/_ab/libsgx_hip_lab.so $LIB
timeout 600 python tools/wgrad_lab.py --configs base,pf2,slab32,ab8,pf2+ab8,pf2+ab4,pf2+g6.8.0 --rounds 3 --iters 5 --out "$OUT/wgrad_lab_pf2.txt" > "$OUT/wgrad_lab_pf2.log" 2>&1; tail -3 "$OUT/wgrad_lab_pf2.log" | cut -c1-260
cp /tmp/new.so $LIB
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-predict"
show() {
python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[2], d["value"], "img/s", d["ms_per_step"], "ms | igemm", r["achieved"], r["kernel_ms_per_step"], "| wgrad", r["wgrad"]["achieved"], r["wgrad"]["kernel_ms_per_step"], r["wgrad"]["launches_per_step"], "| excl", r["exclusive"]["achieved"], r["exclusive"]["wgrad_achieved"])
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1][:-5]+".err").read()[-800:])
PY
}
for cfg in "A=1" "SGX_WGRAD_PF=2" "A=2" "SGX_WGRAD_PF=2 B=2" "SGX_WGRAD_PF=2 SGX_WGRAD_GROUP=6,8,0"; do
  timeout 300 env $cfg $B > "$OUT/bench_$cfg.json" 2> "$OUT/bench_$cfg.err"
  show "$OUT/bench_$cfg.json" "$cfg"
done
