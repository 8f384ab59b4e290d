export TMPDIR=/tmp
OUT=gpurun_out/r2e; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 600 python tools/conv_tune.py --wgrad --iters 5 --out $OUT/conv_tune_variants.txt --emit-table $OUT/conv_tuning_gfx950.json > $OUT/conv_tune.log 2>&1; tail -1 $OUT/conv_tune.log
for t in 0 $OUT/conv_tuning_gfx950.json; do
SGX_CONV_TUNING=$t SGX_NO_PROF=1 python bench.py --no-nms --no-cpu-baseline --no-exclusive 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('table $t:', r['value'],'img/s', r['ms_per_step'],'ms', r['config']['conv_tuning_entries'])"
done
