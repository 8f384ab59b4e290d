#!/bin/bash
# Round 6, visit 34: ConvTranspose2x2 forward from the per-step transpose batch (8 transpose launches off the forward chain): parity tests that
# touch it, then the step A/B (SGX_CONVT_PRETRANSPOSED=0 is the per-call form), three interleaved repetitions, S and M.
TAG=${1:-r6aj}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_kernels.py tests/test_blocks.py -m gpu -q -x 2>&1 | tail -3 | tee "$OUT/pytest_kernels_blocks.txt"
timeout 900 python -m pytest tests/test_yolo_nas.py tests/test_api.py tests/test_distributed.py -m gpu -q -x 2>&1 | tail -3 | tee "$OUT/pytest_yolo_nas.txt"
run() { env $1 timeout 300 python bench.py $2 --steps 20 --warmup 5 --no-cpu-baseline --no-nms --no-predict --no-exclusive --other-configs off 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('final_loss'))"; }
for rep in 1 2 3; do
  for cfg in "SGX_CONVT_PRETRANSPOSED=0" "SGX_CONVT_PRETRANSPOSED=1"; do
    echo "S rep $rep [$cfg]: $(run "$cfg" "")"
  done
done | tee "$OUT/convt_ab.txt"
for rep in 1 2; do
  for cfg in "SGX_CONVT_PRETRANSPOSED=0" "SGX_CONVT_PRETRANSPOSED=1"; do
    echo "M rep $rep [$cfg]: $(run "$cfg" "--model m")"
  done
done | tee -a "$OUT/convt_ab.txt"
