#!/bin/bash
# A/B of two BUILDS of the library on one box: gpurun --timeout 600 -- 'bash tools/visits/r4_lib_ab.sh <tag> <alternative .so>'
# (the alternative is swapped in for its runs and the product library restored afterwards; both travel with the snapshot)
TAG=$1; ALT=$2
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
LIB=super_gradients_amd/csrc/libsgx_hip.so
cp $LIB /tmp/product.so
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-predict"
for rep in 1 2; do
  for which in product alt; do
    if [ $which = alt ]; then cp "$ALT" $LIB; else cp /tmp/product.so $LIB; fi
    timeout 120 $B > "$OUT/bench_${which}_$rep.json" 2> "$OUT/bench_${which}_$rep.err"
    python - "$OUT/bench_${which}_$rep.json" "$which" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[2], d["value"], "img/s", d["ms_per_step"], "ms | conv", r["achieved"], r["kernel_ms_per_step"], "| bf16x3 GEMM", r["gemm_bf16x3"]["algorithmic_tflops"], r["gemm_bf16x3"]["exclusive_algorithmic_tflops"], "| excl", r["exclusive"]["achieved"], "loss", d["config"].get("final_loss"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  done
done
cp /tmp/product.so $LIB
