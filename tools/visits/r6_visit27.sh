#!/bin/bash
# Round 6, visit 27: wave issue priority (s_setprio) of the main chain's kernels over the side stream's weight-gradient waves:
# SGX_WAVE_PRIO bit 0 the finalize kernels (3), bit 1 the sweeps (2), bit 2 forward / data-gradient conv kernels (1).
TAG=${1:-r6ac}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
B="python bench.py --no-cpu-baseline --no-nms --no-predict --no-exclusive --other-configs off --steps 20 --warmup 5"
one() { timeout 200 env SGX_WAVE_PRIO=$1 $B $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('final_loss'))"; }
for rep in 1 2 3; do
  for mode in 0 1 3 7 4; do
    echo "S rep $rep wave_prio=$mode: $(one $mode)"
  done
done | tee "$OUT/wave_prio_s.txt"
for m in m l; do
  for mode in 0 3 7 0 3 7; do
    echo "$m wave_prio=$mode: $(one $mode "--model $m")"
  done
done | tee "$OUT/wave_prio_ml.txt"
for mode in 0 3 7 0 3 7; do
  v=$(timeout 200 env SGX_WAVE_PRIO=$mode python bench.py --workload resnet50 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "resnet50 wave_prio=$mode: $v"
done | tee "$OUT/wave_prio_resnet.txt"
