#!/bin/bash
# Round 6, visit 12: bench.py through the driver's multi-GPU launch form with ONE rank, without and with the data-parallel collectives
# forced on (SGX_DIST_SINGLE_RANK_COLLECTIVES=1: a one-rank RCCL communicator; every bucket all-reduce, the loss all-reduce and the buffer
# broadcasts are issued - the price of the choreography without the transport).
TAG=${1:-r6n}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
B="bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-predict --no-exclusive"
for rep in 1 2; do
  for mode in plain torchrun forced; do
    case $mode in
      plain) cmd="python $B";;
      torchrun) cmd="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$rep $B";;
      forced) cmd="env SGX_DIST_SINGLE_RANK_COLLECTIVES=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2952$rep $B";;
    esac
    v=$(timeout 200 $cmd 2>"$OUT/err_${mode}_$rep.txt" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('allreduce_from_side_stream'), d['config']['final_loss'])")
    echo "rep $rep $mode: $v"
  done
done | tee "$OUT/single_rank_collectives.txt"
tail -3 "$OUT/err_forced_1.txt"
