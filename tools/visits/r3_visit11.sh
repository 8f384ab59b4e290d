#!/bin/bash
# in-step value of the weight-gradient kernel's parts: bench under loop ablations (lab build), then knobs on the product build
TAG=${1:-r3k}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
LIB=super_gradients_amd/csrc/libsgx_hip.so
cp $LIB /tmp/new.so
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
cp tools/_ab/libsgx_hip_lab.so $LIB
for cfg in "SGX_WGRAD_ABLATE=0" "SGX_WGRAD_ABLATE=15" "SGX_WGRAD_ABLATE=8" "SGX_WGRAD_ABLATE=7" "SGX_WGRAD_ABLATE=4"; do
  timeout 300 env $cfg SGX_WGRAD_GROUP_GFLOP=160 $B > "$OUT/bench_lab_$cfg.json" 2> "$OUT/bench_lab_$cfg.err"
  show "$OUT/bench_lab_$cfg.json" "lab $cfg"
done
cp /tmp/new.so $LIB
for cfg in "A=1" "SGX_WGRAD_GROUP=6,8,0" "SGX_WGRAD_TILE=64,64" "SGX_WGRAD_TILE=64,64 SGX_WGRAD_GROUP=6,8,0" "SGX_WGRAD_TILE=64,64 SGX_WGRAD_GROUP=6,8,0 SGX_WGRAD_SLAB=32" "SGX_WGRAD_GROUP=6,16,0" "SGX_SIDE_STREAM=0"; do
  timeout 300 env $cfg SGX_WGRAD_GROUP_GFLOP=160 $B > "$OUT/bench_$cfg.json" 2> "$OUT/bench_$cfg.err"
  show "$OUT/bench_$cfg.json" "$cfg"
done
du -sh "$OUT"
