#!/bin/bash
# the ill-conditioned loss-gradient L2 figure of the whole-model tests under the build's switches; the tests the -x stop did not reach
TAG=${1:-r3zb}
OUT=$(pwd)/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
export SGX_TEST_DUMP="$OUT/backward_b.txt"
for cfg in "A=1" "SGX_WGRAD_GROUP_GFLOP=40" "SGX_CONV_MATH=fp32" "SGX_CONV_MATH=fp32 SGX_WGRAD_GROUP_GFLOP=40" "SGX_WGRAD_PF=1" "SGX_WGRAD_GROUP=6,1,1"; do
  echo "== $cfg" >> "$SGX_TEST_DUMP"
  timeout 200 env $cfg python -m pytest tests/test_yolo_nas.py -m gpu -q -k "train_step_parity and not headline and not atss" > "$OUT/pytest_$cfg.log" 2>&1
  tail -1 "$OUT/pytest_$cfg.log"
done
cat "$SGX_TEST_DUMP"
unset SGX_TEST_DUMP
timeout 300 python -m pytest tests/test_yolo_nas.py -m gpu -q -k "not headline_config_backward_exact" > "$OUT/pytest_yolo_nas_rest.log" 2>&1; tail -4 "$OUT/pytest_yolo_nas_rest.log" | cut -c1-300
