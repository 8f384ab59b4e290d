#!/bin/bash
# Round 4, visit 4 (~5 GPU-minutes): BatchNorm-backward reduce carried by the data gradients; patch weight gradient with one filter block.
#   gpurun --timeout 700 -- 'bash tools/visits/r4_visit4.sh r4d'
TAG=${1:-r4d}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 200 python -m pytest tests/test_kernels.py tests/test_blocks.py -m gpu -q -k "wgrad or dgrad_carries or bn_reduce_rides or csp_layer or conv_block" > "$OUT/pytest_new.log" 2>&1
tail -3 "$OUT/pytest_new.log" | cut -c1-300
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-predict"
for cfg in "A=1" "SGX_FUSE_BN_REDUCE=0" "SGX_WGRAD_MATH=bf16x3" "A=2" "SGX_FUSE_BN_REDUCE=0 B=2"; do
  timeout 120 env $cfg $B > "$OUT/bench_$cfg.json" 2> "$OUT/bench_$cfg.err"
  python - "$OUT/bench_$cfg.json" "$cfg" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[2], d["value"], "img/s", d["ms_per_step"], "ms | igemm", r["achieved"], r["kernel_ms_per_step"], "| wgrad", r["wgrad"]["achieved"], r["wgrad"]["kernel_ms_per_step"], "| excl", r["exclusive"]["achieved"], r["exclusive"]["wgrad_achieved"], "host", d.get("host_enqueue_ms_per_step"), "loss", d["config"].get("final_loss"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
timeout 100 python tools/wgrad_lab.py --configs nopatch,base --rounds 2 --iters 4 --out "$OUT/wgrad_lab.txt" > "$OUT/wgrad_lab.log" 2>&1
tail -2 "$OUT/wgrad_lab.log" | cut -c1-200
SGX_TEST_DUMP="$OUT/backward_b.txt" timeout 300 python -m pytest tests/test_yolo_nas.py -m gpu -q -k "train_step_parity or backward_exact_without or eval_and_nms" > "$OUT/pytest_yolo_nas.log" 2>&1
tail -3 "$OUT/pytest_yolo_nas.log" | cut -c1-300
grep "backward B" "$OUT/backward_b.txt" 2>/dev/null
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o bench -- bash -c "cd $REPO && python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-nms --no-predict --no-exclusive" > "$OUT/stats.log" 2>&1
cd "$REPO"
python tools/prof_summary.py stats "$OUT/stats" > "$OUT/kernel_stats_summary.txt" 2>&1
head -30 "$OUT/kernel_stats_summary.txt" | cut -c1-170
find "$OUT" -name "*.csv" -size +8M -delete
