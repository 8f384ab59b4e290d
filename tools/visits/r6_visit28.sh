#!/bin/bash
# Round 6, visit 28: the stem's backward in runs of images (SGX_STEM_BWD_CHUNKS): each run's weight gradients go out behind its apply sweep.
TAG=${1:-r6ad}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_blocks.py tests/test_yolo_nas.py -m gpu -q -x -k "qarep or train_step_parity or headline_config_parity or branch" 2>&1 | tail -3 | tee "$OUT/pytest_chunks.txt"
B="python bench.py --no-cpu-baseline --no-nms --no-predict --no-exclusive --other-configs off --steps 20 --warmup 5"
one() { timeout 200 env SGX_STEM_BWD_CHUNKS=$1 $B $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('final_loss'))"; }
for rep in 1 2 3; do
  for c in 1 2 4 8; do
    echo "S rep $rep stem_chunks=$c: $(one $c)"
  done
done | tee "$OUT/stem_chunks_s.txt"
for m in m l; do
  for c in 1 2 4 1 2 4; do
    echo "$m stem_chunks=$c: $(one $c "--model $m")"
  done
done | tee "$OUT/stem_chunks_ml.txt"
