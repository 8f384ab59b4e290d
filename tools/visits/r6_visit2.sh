#!/bin/bash
# Round 6, visit 2: (a) why YOLO-NAS-L bs32 ran at 250 ms inside `other_configs` (r6a) when the standalone line runs 108 ms: the leg alone,
# after another configuration with / without an allocator flush, with the allocator's device-malloc counts; (b) predict(): device-side
# inverse box maps + one copy against the host path (SGX_PREDICT_HOST_POST=1), cProfile of the pipeline; (c) the new GPU tests.
TAG=${1:-r6b}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_predict.py tests/test_half.py -m gpu -q -x > "$OUT/pytest_predict.log" 2>&1; tail -3 "$OUT/pytest_predict.log"
timeout 600 python - > "$OUT/l_anomaly.txt" 2>&1 <<'PY'
import gc, json, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import bench
dev = torch.device("cuda:0")
def stats():
    s = torch.cuda.memory_stats()
    return s.get("num_device_alloc", 0), s.get("num_alloc_retries", 0), round(torch.cuda.memory_reserved() / 2**30, 1)
def leg(tag, *a, **k):
    a0 = stats(); t0 = time.time()
    o = bench.other_config_leg(dev, *a, **k)
    print(tag, o["value"], o["ms_per_step"], "allocs/retries/reservedGB before", a0, "after", stats(), "wall", round(time.time() - t0, 1), flush=True)
leg("L640 alone (fresh process)", "yolo_nas", "l", 640, 32, loss_check=False)
leg("L640 again (pool warm)", "yolo_nas", "l", 640, 32, loss_check=False)
gc.collect(); torch.cuda.empty_cache()
leg("L640 after empty_cache", "yolo_nas", "l", 640, 32, loss_check=False)
leg("M640", "yolo_nas", "m", 640, 32, loss_check=False)
gc.collect(); torch.cuda.empty_cache()
leg("L640 after M + empty_cache", "yolo_nas", "l", 640, 32, loss_check=False)
leg("L640 warmup 6", "yolo_nas", "l", 640, 32, loss_check=False, warmup=6)
PY
cat "$OUT/l_anomaly.txt" | grep -v Warning | tail -12
for hp in 0 1; do
  SGX_PREDICT_HOST_POST=$hp timeout 200 python tools/predict_profile.py --batches 20 > "$OUT/predict_profile_hostpost$hp.txt" 2>&1
  tail -1 "$OUT/predict_profile_hostpost$hp.txt"
done
for hp in 0 1 0 1; do
  SGX_PREDICT_HOST_POST=$hp timeout 200 python tools/predict_bench.py --batches 20 2>/dev/null | tail -1 | cut -c1-300
done > "$OUT/predict_bench_ab.txt"; cat "$OUT/predict_bench_ab.txt"
du -sh "$OUT"
