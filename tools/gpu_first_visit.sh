#!/bin/bash
# First GPU visit of a round: (1) the tests that were written without GPU time (green on the host emulation only), non-fatally;
# (2) A/B of the conv-kernel experiment switches on the train-step bench; (3) the per-problem tuner over every tile x variant.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_first_visit.sh r2a'      (~25-30 GPU-minutes with every stage)
#   STAGES="tests ab" ...                                                                  (subset: tests | ab | sweeps | tune | tune_bf3)
# Everything lands under gpurun_out/$TAG/; copy what is to be judged into profiles/.
TAG=${1:-first}
STAGES=${STAGES:-"tests ab sweeps tune"}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
has() { [[ " $STAGES " == *" $1 "* ]]; }
if has tests; then
SGX_GPU_UNVALIDATED=1 timeout 400 python -m pytest tests/test_kernels.py tests/test_blocks.py tests/test_decoding.py tests/test_tools.py -m gpu -q \
  -k "any_class_count or assignment_adversarial or distance_tie_policy or nms_degenerate_boxes or conv_every_tile_shape or conv_deep_slabs or stem_with_custom_in_channels or decod or fused_bn_backward or fused_finalize or tuning_table or learnable_alpha or round_trip or switches_compose" \
  > "$OUT/pytest_first_gpu_run.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest_first_gpu_run.log"; tail -4 "$OUT/pytest_first_gpu_run.log"
fi
if has ab; then
for m in fp32 auto; do
  for v in 0 5 6; do
    f="$OUT/bench_${m}_variant$v"
    SGX_CONV_MATH=$m SGX_CONV_VARIANT=$v timeout 200 python bench.py --no-nms --no-cpu-baseline ${BENCH_ARGS:-} > "$f.json" 2> "$f.err"
    echo "math $m variant $v rc=$?: $(python -c "import json,sys; r=json.loads(open('$f.json').read().strip().splitlines()[-1]); print(r['value'],'img/s', r['ms_per_step'],'ms; igemm', r['roofline']['achieved'], 'TF concurrent,', r['roofline']['exclusive']['achieved'], 'TF exclusive')" 2>&1 | tail -1)"
  done
done
fi
if has sweeps; then
f="$OUT/bench_fused_finalize"
SGX_FUSED_FINALIZE=1 timeout 200 python bench.py --no-nms --no-cpu-baseline ${BENCH_ARGS:-} > "$f.json" 2> "$f.err"
echo "SGX_FUSED_FINALIZE=1 rc=$?: $(python -c "import json; r=json.loads(open('$f.json').read().strip().splitlines()[-1]); print(r['value'],'img/s', r['ms_per_step'],'ms')" 2>&1 | tail -1)"
f="$OUT/bench_fused_bn_reduce"
SGX_FUSE_BN_REDUCE=1 timeout 200 python bench.py --no-nms --no-cpu-baseline ${BENCH_ARGS:-} > "$f.json" 2> "$f.err"
echo "SGX_FUSE_BN_REDUCE=1 rc=$?: $(python -c "import json; r=json.loads(open('$f.json').read().strip().splitlines()[-1]); print(r['value'],'img/s', r['ms_per_step'],'ms')" 2>&1 | tail -1)"
fi
if has tune; then
  timeout 900 python tools/conv_tune.py ${TUNE_ARGS:---wgrad} --out "$OUT/conv_tune_variants.txt" --emit-table "$OUT/conv_tuning_gfx950.json" > "$OUT/conv_tune.log" 2>&1
  echo "conv_tune rc=$?"; head -12 "$OUT/conv_tune_variants.txt"; tail -1 "$OUT/conv_tune.log"
  # the table's end-to-end effect (commit it as super_gradients_amd/csrc/conv_tuning_gfx950.json to make it the default)
  f="$OUT/bench_tuning_table"
  SGX_CONV_TUNING="$OUT/conv_tuning_gfx950.json" timeout 200 python bench.py --no-nms --no-cpu-baseline ${BENCH_ARGS:-} > "$f.json" 2> "$f.err"
  echo "tuning table rc=$?: $(python -c "import json; r=json.loads(open('$f.json').read().strip().splitlines()[-1]); print(r['value'],'img/s', r['ms_per_step'],'ms; entries', r['config']['conv_tuning_entries'])" 2>&1 | tail -1)"
fi
if has tune_bf3; then
  SGX_CONV_MATH=bf16x3 timeout 700 python tools/conv_tune.py --out "$OUT/conv_tune_variants_bf16x3.txt" > "$OUT/conv_tune_bf16x3.log" 2>&1
  echo "conv_tune bf16x3 rc=$?"; head -12 "$OUT/conv_tune_variants_bf16x3.txt"
fi
