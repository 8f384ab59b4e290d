#!/bin/bash
# Round 6, visit 31: bench line with the single_chain leg; the driver's torch.distributed.run form with the collectives forced on one rank
# (branch stream + bucket all-reduces from the side stream); GPU tests that changed since r6fin3 (policy test is CPU).
TAG=${1:-r6ag}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python bench.py --no-cpu-baseline --no-nms --no-predict --other-configs off > "$OUT/bench_short.json" 2> "$OUT/bench_short.err"; tail -c 3000 "$OUT/bench_short.json"
for rep in 1 2; do
  for forced in 0 1; do
    v=$(SGX_DIST_SINGLE_RANK_COLLECTIVES=$forced timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-nms --no-predict --no-exclusive --other-configs off 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('final_loss'))")
    echo "torchrun rep $rep forced_collectives=$forced: $v"
  done
done | tee "$OUT/torchrun_collectives.txt"
