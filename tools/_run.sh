export TMPDIR=/tmp
OUT=gpurun_out/r2s; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
python bench.py > $OUT/bench.json 2>$OUT/bench.err; python -c "
import json; r=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], 'host', r['host_enqueue_ms_per_step'], 'roofline', r['roofline']['achieved'], r['roofline']['frac'], 'excl', r['roofline']['exclusive']['achieved'], 'wgrad', r['roofline']['wgrad']['achieved'], 'step frac', r['roofline']['step_mfma_frac'], 'cpu', r['cpu_baseline']['value'], r['cpu_baseline']['seconds_per_step'], 'nms', r['nms']['value'])"
