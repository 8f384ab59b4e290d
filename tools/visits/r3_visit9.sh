#!/bin/bash
# A/B of two library builds (tools/_ab/libsgx_hip_prev.so = the previous commit's kernels) on one box, then knob sweeps of the new one
TAG=${1:-r3i}
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
timeout 300 python -m pytest tests/test_kernels.py -m gpu -q -x -k "wgrad or bwd_weight or conv_bwd or conv_block" > "$OUT/pytest_wgrad.log" 2>&1; tail -2 "$OUT/pytest_wgrad.log"
for rep in 1 2; do
  for which in prev new; do
    if [ $which = prev ]; then cp tools/_ab/libsgx_hip_prev.so $LIB; else cp /tmp/new.so $LIB; fi
    timeout 300 $B > "$OUT/bench_${which}_$rep.json" 2> "$OUT/bench_${which}_$rep.err"
    show "$OUT/bench_${which}_$rep.json" "${which}_$rep"
  done
done
cp /tmp/new.so $LIB
for g in "4,16,1" "8,4,1" "3,32,1" "6,8,0"; do
  SGX_WGRAD_GROUP=$g timeout 300 $B > "$OUT/bench_g$g.json" 2> "$OUT/bench_g$g.err"
  show "$OUT/bench_g$g.json" "group=$g"
done
for gf in 20 80 160; do
  SGX_WGRAD_GROUP_GFLOP=$gf timeout 300 $B > "$OUT/bench_gf$gf.json" 2> "$OUT/bench_gf$gf.err"
  show "$OUT/bench_gf$gf.json" "group_gflop=$gf"
done
du -sh "$OUT"
