#!/bin/bash
# finalize kernels: previous build vs this one (microbench + step)
TAG=${1:-r3m}
OUT=$(pwd)/gpurun_out/$TAG
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
for which in prev new prev new; do
  if [ $which = prev ]; then cp tools/_ab/libsgx_hip_prev.so $LIB; else cp /tmp/new.so $LIB; fi
  python tools/finalize_bench.py > "$OUT/finalize_$which.txt" 2>&1
  timeout 300 env SGX_WGRAD_GROUP_GFLOP=160 $B > "$OUT/bench_$which.json" 2> "$OUT/bench_$which.err"
  show "$OUT/bench_$which.json" "$which"
done
cp /tmp/new.so $LIB
paste "$OUT/finalize_prev.txt" "$OUT/finalize_new.txt" | cut -c1-200
timeout 300 python -m pytest tests/test_kernels.py tests/test_blocks.py -m gpu -q -x -k "finalize or batchnorm or qarep or bn_ or colsum or conv_block" > "$OUT/pytest_fin.log" 2>&1; tail -2 "$OUT/pytest_fin.log"
