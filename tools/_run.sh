timeout 600 python -m pytest tests/test_kernels.py tests/test_decoding.py tests/test_oracle_vs_reference.py tests/test_yolo_nas.py tests/test_detection_metrics.py tests/test_pp_yolo_e.py -m gpu -x -q -k "nms or decod or post_prediction or headline or eval_and_nms or metrics or eval" 2>&1 | tail -2
export TMPDIR=/tmp; REPO=$(pwd); cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $REPO/gpurun_out/nmsprof -o nms -- bash -c "cd $REPO && python tools/_nmsprof.py" > /dev/null 2>&1
cd $REPO; python tools/prof_summary.py stats gpurun_out/nmsprof | cut -c1-150 | head -8
python bench.py --no-cpu-baseline --no-exclusive --steps 3 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); n=r['nms']; print(n['value'], n['ms_per_batch'], n['roofline']['achieved'])"
