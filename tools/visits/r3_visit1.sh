#!/bin/bash
# round-3 GPU visit 1: parity tests, grouped-wgrad knob sweep, bf16x3 tile lab, kernel stats
TAG=${1:-r3a}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-predict"
run() { name=$1; shift; ( timeout 300 env "$@" $B > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err" ); python - "$OUT/bench_$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[2], d["value"], "img/s", d["ms_per_step"], "ms | igemm", r["achieved"], r["kernel_ms_per_step"], "| wgrad", r["wgrad"]["achieved"], r["wgrad"]["kernel_ms_per_step"], r["wgrad"]["launches_per_step"], "| excl", r["exclusive"]["achieved"], r["exclusive"]["wgrad_achieved"], "| host", d["host_enqueue_ms_per_step"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run perlayer SGX_WGRAD_GROUP_GFLOP=0
run g40 SGX_WGRAD_GROUP_GFLOP=40
run g10 SGX_WGRAD_GROUP_GFLOP=10
run g40_noxcd SGX_WGRAD_GROUP_GFLOP=40 SGX_WGRAD_GROUP=6,8,0
run g40_r12 SGX_WGRAD_GROUP_GFLOP=40 SGX_WGRAD_GROUP=12,4,1
timeout 900 python -m pytest tests -m gpu -x -q --durations=12 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"; tail -4 "$OUT/pytest_gpu.log"
timeout 300 python tools/conv_lab.py --math fp32,bf16x3 --tiles 64x64,128x64,128x128 --rounds 3 --iters 8 \
  --problems fwd:32:80:80:64:64:3:1,fwd:32:40:40:96:96:3:1,fwd:32:160:160:32:32:3:1,fwd:32:20:20:256:256:3:1,dgrad:32:80:80:64:64:3:1,fwd:32:80:80:192:384:3:2 \
  --out "$OUT/lab_bf16x3_tiles.txt" > "$OUT/lab.log" 2>&1; tail -14 "$OUT/lab.log"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o bench -- bash -c "cd $REPO && python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-nms --no-predict --no-exclusive" > "$OUT/stats.log" 2>&1
cd "$REPO"
python tools/prof_summary.py stats "$OUT/stats" > "$OUT/kernel_stats_summary.txt" 2>&1; head -30 "$OUT/kernel_stats_summary.txt"
python tools/prof_summary.py timeline "$OUT/stats" > "$OUT/kernel_timeline_summary.txt" 2>&1; head -8 "$OUT/kernel_timeline_summary.txt"
find "$OUT/stats" -name "*kernel_trace.csv" -size +8M -delete
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; tail -2 "$OUT/smoke.log"
du -sh "$OUT"
