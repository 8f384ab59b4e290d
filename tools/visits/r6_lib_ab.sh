#!/bin/bash
# A/B of two BUILDS of the library on one box (round 6): lab timings of the GEMM problems + the step, interleaved.
#   gpurun --timeout 900 -- 'bash tools/visits/r6_lib_ab.sh <tag> <alternative .so> [reps]'
TAG=$1; ALT=$2; REPS=${3:-2}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
LIB=super_gradients_amd/csrc/libsgx_hip.so
cp $LIB /tmp/product.so
P="fwd:32:40:40:384:192:1:1,fwd:32:80:80:192:384:3:2,fwd:32:20:20:1536:768:1:1,fwd:32:20:20:768:192:1:1,dgrad:32:80:80:192:96:1:1,fwd:32:80:80:192:64:1:1,fwd:32:80:80:64:64:3:1,fwd2:32:80:80:96:96:3:1,dgrad2:32:80:80:96:96:3:1"
for which in alt product; do
  if [ $which = alt ]; then cp "$ALT" $LIB; else cp /tmp/product.so $LIB; fi
  timeout 300 python tools/conv_lab.py --math bf16x3 --planes 1 --problems "$P" --rounds 5 --iters 10 --out "$OUT/lab_$which.txt" > /dev/null 2>&1
done
paste -d'\n' "$OUT/lab_alt.txt" "$OUT/lab_product.txt" | awk 'NR>2' | awk '{print}' > "$OUT/lab_both.txt"
python - "$OUT" <<'PY'
import sys
o=sys.argv[1]
a=open(o+"/lab_alt.txt").read().splitlines(); b=open(o+"/lab_product.txt").read().splitlines()
print("problem                              alt us    product us   ratio")
for i in range(1,len(a),2):
    name=a[i].split()[0]; ta=float(a[i+1].split()[0]); tb=float(b[i+1].split()[0])
    print(f"{name:<34} {ta:9.1f} {tb:12.1f} {tb/ta:7.3f}")
PY
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-predict"
for rep in $(seq 1 $REPS); do
  for which in alt product; do
    if [ $which = alt ]; then cp "$ALT" $LIB; else cp /tmp/product.so $LIB; fi
    timeout 150 $B > "$OUT/bench_${which}_$rep.json" 2> "$OUT/bench_${which}_$rep.err"
    python - "$OUT/bench_${which}_$rep.json" "$which" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[2], d["value"], "img/s", d["ms_per_step"], "ms | conv", r["achieved"], r["kernel_ms_per_step"], "| bf16x3 GEMM", r["gemm_bf16x3"]["kernel_ms_per_step"], r["gemm_bf16x3"]["exclusive_algorithmic_tflops"], "| patch", r["patch_kernel"]["kernel_ms_per_step"], "| wgrad", r["wgrad"]["kernel_ms_per_step"], "| excl", r["exclusive"]["achieved"], "loss", d["config"].get("final_loss"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  done
done
cp /tmp/product.so $LIB
