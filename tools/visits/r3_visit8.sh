#!/bin/bash
TAG=${1:-r3h}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
export SGX_TEST_DUMP="$OUT/headline_backward.txt"
timeout 700 python -m pytest tests/test_yolo_nas.py -m gpu -x -q -k "headline_config_backward_exact" > "$OUT/pytest_headline.log" 2>&1; tail -3 "$OUT/pytest_headline.log" | cut -c1-900
unset SGX_TEST_DUMP
timeout 600 python -m pytest tests/test_kernels.py tests/test_known_answers.py tests/test_decoding.py tests/test_predict.py tests/test_oracle_vs_reference.py -m gpu -q -k "nms or pconv or wgrad or known or decod or post_prediction or predict or tuning or bf16x3" > "$OUT/pytest_new.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_new.log"; tail -4 "$OUT/pytest_new.log" | cut -c1-300
python - <<'PY' > "$OUT/nms_ab.txt" 2>&1
import json, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bench
from super_gradients_amd._lib import lib
dev = torch.device("cuda:0")
for split in (1, 0, 1, 0):
    lib().sgx_debug_set_nms_split(split)
    r = bench.nms_leg(dev, iters=200, warmup=20)
    print("split", split, r["value"], "boxes/s", r["ms_per_batch"], "ms", "kept", r["kept"], "cand", r["candidates"])
PY
cat "$OUT/nms_ab.txt" | tail -5
timeout 600 python bench.py > "$OUT/bench.log" 2> "$OUT/bench.err"; echo "bench rc=$?" >> "$OUT/bench.err"; cat "$OUT/bench.log" | cut -c1-3000; tail -2 "$OUT/bench.err"
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-predict"
for cfg in "SGX_CONV_MATH=fp32" "SGX_CONV_MATH=patch"; do
  timeout 300 env $cfg $B > "$OUT/bench_$cfg.json" 2> "$OUT/bench_$cfg.err"
  python - "$OUT/bench_$cfg.json" "$cfg" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[2], d["value"], "img/s", d["ms_per_step"], "ms | igemm", r["achieved"], r["kernel_ms_per_step"], "| wgrad", r["wgrad"]["achieved"], r["wgrad"]["kernel_ms_per_step"], r["wgrad"]["launches_per_step"], "| excl", r["exclusive"]["achieved"], r["exclusive"]["wgrad_achieved"], "| patch", r.get("patch_kernel"))
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1][:-5]+".err").read()[-800:])
PY
done
du -sh "$OUT"
