#!/bin/bash
# round-3 GPU visit 2: fence-free grouped wgrad, patch kernel first contact, headline backward diagnosis
TAG=${1:-r3b}
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
timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q -k "pconv or wgrad_group or conv_bwd" > "$OUT/pytest_new.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_new.log"; tail -6 "$OUT/pytest_new.log"
run perlayer SGX_WGRAD_GROUP_GFLOP=0
run g40 SGX_WGRAD_GROUP_GFLOP=40
run g40_noxcd SGX_WGRAD_GROUP_GFLOP=40 SGX_WGRAD_GROUP=6,4,0
run g40_i8 SGX_WGRAD_GROUP_GFLOP=40 SGX_WGRAD_GROUP=6,8,1
run g40_i2 SGX_WGRAD_GROUP_GFLOP=40 SGX_WGRAD_GROUP=6,2,1
run g150 SGX_WGRAD_GROUP_GFLOP=150
run g15 SGX_WGRAD_GROUP_GFLOP=15
run patch SGX_CONV_MATH=patch
timeout 300 python tools/conv_lab.py --math fp32,patch --rounds 3 --iters 8 \
  --problems fwd:32:80:80:64:64:3:1,fwd:32:40:40:96:96:3:1,fwd:32:160:160:32:32:3:1,fwd:32:20:20:256:256:3:1,fwd:32:80:80:48:48:3:1,fwd:32:40:40:128:128:3:1,dgrad:32:80:80:64:64:3:1,dgrad:32:40:40:96:96:3:1,fwd:32:20:20:192:192:3:1 \
  --out "$OUT/lab_patch.txt" > "$OUT/lab.log" 2>&1; tail -20 "$OUT/lab.log"
export SGX_TEST_DUMP="$OUT/headline_backward.txt"
for cfg in "A=1" "SGX_CONV_TUNING=0" "SGX_WGRAD_GROUP_GFLOP=0" "SGX_SIDE_STREAM=0" "SGX_CONV_MATH=patch"; do
  echo "== $cfg" >> "$OUT/pytest_headline.log"
  timeout 400 env $cfg python -m pytest tests/test_yolo_nas.py -m gpu -x -q -k "headline_config_backward_exact" >> "$OUT/pytest_headline.log" 2>&1
  tail -3 "$OUT/pytest_headline.log" | cut -c1-400
done
timeout 600 env SGX_CONV_MATH=patch python -m pytest tests/test_yolo_nas.py tests/test_blocks.py -m gpu -x -q > "$OUT/pytest_patch.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_patch.log"; tail -5 "$OUT/pytest_patch.log" | cut -c1-600
du -sh "$OUT"
