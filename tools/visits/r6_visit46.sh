#!/bin/bash
# Round 6, visit 46: the driver's torch.distributed.run form with the collectives forced on one rank, at the defaults (GradientAllReducer now
# gives the second branch lane's queue to RCCL on its own: engine.data_parallel_streams)
TAG=${1:-r6aw}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
run() { env $1 SGX_DIST_SINGLE_RANK_COLLECTIVES=$2 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-nms --no-predict --no-exclusive --other-configs off 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('final_loss'), d['config'].get('allreduce_from_side_stream'))"; }
for rep in 1 2 3; do
  echo "rep $rep collectives off (two lanes): $(run SGX_NONE=0 0)"
  echo "rep $rep collectives forced, default policy (one lane, sites 31): $(run SGX_NONE=0 1)"
  echo "rep $rep collectives forced, two lanes kept (SGX_BRANCH_LANES=2 SGX_BRANCH_SITES=63): $(run "SGX_BRANCH_LANES=2 SGX_BRANCH_SITES=63" 1)"
done | tee "$OUT/collectives_default_policy.txt"
