#!/bin/bash
# Round 6, visit 30: r6ae measured four lanes 2 % SLOWER than two and the SPP pools on lanes 6 % slower: main + side + four lanes are six HIP
# streams on (by default) four hardware queues - streams that share a queue serialise.  GPU_MAX_HW_QUEUES=8 against the default, lanes 2 / 4,
# the head levels started from inside the neck, the SPP pools (site 64) and the backward-only filter preparations on a lane (site 128).
TAG=${1:-r6af}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
B="python bench.py --no-cpu-baseline --no-nms --no-predict --no-exclusive --other-configs off --steps 20 --warmup 5"
one() { timeout 200 env $1 SGX_HEADS_EARLY=$2 SGX_BRANCH_SITES=$3 SGX_BRANCH_LANES=$4 $B $5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('final_loss'))"; }
for rep in 1 2 3; do
  for q in "X=0" "GPU_MAX_HW_QUEUES=8"; do
    for cfg in "0 63 2" "0 191 2" "0 127 2" "1 63 4" "1 255 4"; do
      echo "S rep $rep $q [heads_early sites lanes = $cfg]: $(one $q $cfg)"
    done
  done
done | tee "$OUT/hw_queues_s.txt"
