#!/bin/bash
# Round 6, visit 32: r6ag - the bucket all-reduces forced on one rank cost 6.3 % with the branch stream on (0.8 % before it): RCCL's stream is a
# fifth HIP stream on four hardware queues.  Lanes 2 / 1 / branch stream off, collectives forced / not.
TAG=${1:-r6ah}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
run() { env $1 SGX_DIST_SINGLE_RANK_COLLECTIVES=$2 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-nms --no-predict --no-exclusive --other-configs off 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('final_loss'))"; }
for rep in 1 2; do
  for forced in 1 0; do
    for cfg in "SGX_BRANCH_LANES=2" "SGX_BRANCH_LANES=1" "SGX_BRANCH_STREAM=0" "SGX_BRANCH_LANES=1 SGX_BRANCH_SITES=31" "SGX_BRANCH_LANES=2 SGX_ALLREDUCE_FROM_SIDE=0" "SGX_BRANCH_LANES=2 GPU_MAX_HW_QUEUES=5"; do
      echo "rep $rep forced=$forced [$cfg]: $(run "$cfg" $forced)"
    done
  done
done | tee "$OUT/collectives_lanes.txt"
