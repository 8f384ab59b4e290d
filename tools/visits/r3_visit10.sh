#!/bin/bash
# weight-gradient lab: loop ablations on the -DSGX_WGRAD_LAB build, then slab depth / grouping on the product build
TAG=${1:-r3j}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
LIB=super_gradients_amd/csrc/libsgx_hip.so
cp $LIB /tmp/new.so
cp tools/_ab/libsgx_hip_lab.so $LIB
timeout 600 python tools/wgrad_lab.py --configs base,ab8,ab1,ab3,ab7,ab4,ab15,slab32,slab32+ab8 --rounds 3 --iters 5 --out "$OUT/wgrad_lab_ablation.txt" > "$OUT/wgrad_lab_ablation.log" 2>&1; tail -4 "$OUT/wgrad_lab_ablation.log" | cut -c1-260
cp /tmp/new.so $LIB
timeout 600 python tools/wgrad_lab.py --configs base,slab32,g8.4.1,g12.2.1,g6.8.0,t64x64,t128x64,t128x128 --rounds 3 --iters 5 --out "$OUT/wgrad_lab_knobs.txt" > "$OUT/wgrad_lab_knobs.log" 2>&1; tail -4 "$OUT/wgrad_lab_knobs.log" | cut -c1-260
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
for cfg in "A=1" "SGX_WGRAD_SLAB=32" "SGX_WGRAD_GROUP_GFLOP=160" "SGX_WGRAD_GROUP_GFLOP=400" "SGX_WGRAD_GROUP_GFLOP=2000" "SGX_WGRAD_GROUP_GFLOP=160 SGX_WGRAD_SLAB=32"; do
  timeout 300 env $cfg $B > "$OUT/bench_$cfg.json" 2> "$OUT/bench_$cfg.err"
  show "$OUT/bench_$cfg.json" "$cfg"
done
du -sh "$OUT"
