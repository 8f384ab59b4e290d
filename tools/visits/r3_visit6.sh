#!/bin/bash
TAG=${1:-r3f}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-predict"
run() { name=$1; shift; ( timeout 300 env "$@" $B > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err" ); python - "$OUT/bench_$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[2], d["value"], "img/s", d["ms_per_step"], "ms | igemm", r["achieved"], r["kernel_ms_per_step"], "| wgrad", r["wgrad"]["achieved"], r["wgrad"]["kernel_ms_per_step"], r["wgrad"]["launches_per_step"], "| excl", r["exclusive"]["achieved"], r["exclusive"]["wgrad_achieved"], "| host", d["host_enqueue_ms_per_step"], "| loss", d["config"]["final_loss"])
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1][:-5]+".err").read()[-800:])
PY
}
timeout 300 python tools/debug_neck1.py > "$OUT/debug_neck1.log" 2>&1; tail -16 "$OUT/debug_neck1.log" | cut -c1-300
timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q -k "pconv" > "$OUT/pytest_new.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_new.log"; tail -3 "$OUT/pytest_new.log"
P=fwd:32:80:80:64:64:3:1,fwd2:32:80:80:64:64:3:1,dgrad2:32:80:80:64:64:3:1,fwd2:32:40:40:96:96:3:1,dgrad2:32:40:40:96:96:3:1,fwd2:32:160:160:32:32:3:1,dgrad2:32:160:160:32:32:3:1,fwd2:32:80:80:48:48:3:1,fwd:32:40:40:128:128:3:1,dgrad2:32:160:160:96:192:3:2,fwd2:32:20:20:192:192:3:1
timeout 400 python tools/conv_lab.py --math patch --variants 0,8 --rounds 3 --iters 8 --problems $P --out "$OUT/lab_patch.txt" > "$OUT/lab.log" 2>&1; tail -24 "$OUT/lab.log"
SGX_WGRAD_GROUP_GFLOP=0 timeout 300 python tools/conv_bench.py --iters 6 --out "$OUT/conv_bench_fp32.txt" > "$OUT/conv_bench_fp32.log" 2>&1; head -3 "$OUT/conv_bench_fp32.txt"
SGX_WGRAD_GROUP_GFLOP=0 SGX_CONV_MATH=patch timeout 300 python tools/conv_bench.py --iters 6 --out "$OUT/conv_bench_patch.txt" > "$OUT/conv_bench_patch.log" 2>&1; head -3 "$OUT/conv_bench_patch.txt"
run fp32 A=1
run patch SGX_CONV_MATH=patch
run patch_noside SGX_CONV_MATH=patch SGX_SIDE_STREAM=0
run fp32_noside SGX_SIDE_STREAM=0
du -sh "$OUT"
