#!/bin/bash
# Round 6, visit 44: the build without packed fp32 instructions: the d alpha probe, the step against the previous build
# (tools/_probe/lib_old_slp.so: bn.hip / pool / se / loss / nms / optim with the SLP vectoriser), the whole GPU suite
TAG=${1:-r6au}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
LIB=super_gradients_amd/csrc/libsgx_hip.so
cp $LIB /tmp/product.so
timeout 400 python tools/branch_flake_probe.py 400 200 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-330 | tee "$OUT/branch_flake_probe_no_packed.txt"
run() { timeout 300 python bench.py $1 --steps 20 --warmup 5 --no-cpu-baseline --no-nms --no-predict --no-exclusive --other-configs off 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('final_loss'))"; }
for rep in 1 2 3; do
  for which in old product; do
    if [ $which = old ]; then cp tools/_probe/lib_old_slp.so $LIB; else cp /tmp/product.so $LIB; fi
    echo "S rep $rep [$which]: $(run "")"
  done
done | tee "$OUT/packed_ab.txt"
for which in old product; do
  if [ $which = old ]; then cp tools/_probe/lib_old_slp.so $LIB; else cp /tmp/product.so $LIB; fi
  echo "M [$which]: $(run "--model m")"; echo "resnet50 [$which]: $(run "--workload resnet50")"
done | tee -a "$OUT/packed_ab.txt"
cp /tmp/product.so $LIB
timeout 900 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; tail -3 "$OUT/pytest_gpu.log"
