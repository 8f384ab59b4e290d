#!/bin/bash
# NMS: balanced matrix-build work items + register-resident diagonal words in the walk: GPU parity + boxes/s
TAG=${1:-r3ze}
OUT=$(pwd)/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 200 python -m pytest tests/test_kernels.py tests/test_known_answers.py tests/test_decoding.py tests/test_predict.py -m gpu -q -x -k "nms or known or post_prediction or decod or predict_pipeline" > "$OUT/pytest_nms.log" 2>&1; tail -1 "$OUT/pytest_nms.log"
python - <<'PY' > "$OUT/nms_ab.txt" 2>&1
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bench
from super_gradients_amd._lib import lib
dev = torch.device("cuda:0")
for split in (1, 0, 1):
    lib().sgx_debug_set_nms_split(split)
    r = bench.nms_leg(dev, iters=200, warmup=20)
    print("split", split, r["value"], "boxes/s", r["ms_per_batch"], "ms", "kept", r["kept"], "cand", r["candidates"])
PY
grep split "$OUT/nms_ab.txt"
cd /tmp; timeout -k 10 120 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/nms_stats" -o nms -- bash -c "cd $OLDPWD && python bench.py --only-nms 20" > "$OUT/nms_stats.log" 2>&1; cd "$OLDPWD"
python tools/prof_summary.py stats "$OUT/nms_stats" > "$OUT/nms_kernel_stats_summary.txt" 2>&1; head -9 "$OUT/nms_kernel_stats_summary.txt" | cut -c1-150
find "$OUT" -name "*.csv" -size +2M -delete
