#!/bin/bash
# Round 6, visit 14: the stem's analytically-zero bias gradient no longer summed (a 629 MB column sum in the step's tail); A/B against the
# build before (SGX_QAREP_OLD_BIAS_SUM=1 re-enables it) and with the 1x1 weight gradient flushed early (SGX_QAREP_FLUSH_1X1=1).
TAG=${1:-r6p}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-predict --no-exclusive"
for rep in 1 2 3; do
  for mode in old new flush; do
    case $mode in
      old) e="SGX_QAREP_OLD_BIAS_SUM=1";;
      new) e="SGX_X=0";;
      flush) e="SGX_QAREP_FLUSH_1X1=1";;
    esac
    v=$(timeout 200 env $e $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['final_loss'])")
    echo "rep $rep $mode: $v"
  done
done | tee "$OUT/stem_bias_sum_ab.txt"
timeout 300 python -m pytest tests/test_yolo_nas.py -m gpu -q -x -k "s_train_step_parity or golden or s_backward_exact" 2>&1 | tail -3
