#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprofv3 kernel stats + PMC passes.  Outputs under gpurun_out/$TAG/.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_round.sh r1a'
# Stages are individually time-boxed so that one hang cannot eat the box budget (rocprofv3 gets SIGKILL 10 s after SIGTERM: after an
# aborted counter configuration it catches SIGTERM and never exits - r3o lost 25 GPU-minutes that way).  STAGES env selects a subset.
TAG=${1:-r1}
STAGES=${STAGES:-"bench stats pmc tests smoke"}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
has() { [[ " $STAGES " == *" $1 "* ]]; }

rocminfo 2>/dev/null | grep -m3 -E "gfx950|Compute Unit|Max Clock" > "$OUT/rocminfo.txt"
nproc > "$OUT/nproc.txt"

if has bench; then
  timeout 600 python bench.py ${BENCH_ARGS:-} > "$OUT/bench.log" 2> "$OUT/bench.err"
  echo "bench rc=$?" >> "$OUT/bench.err"
  cat "$OUT/bench.log"; tail -3 "$OUT/bench.err"
fi
if has convbench; then
  timeout 400 python tools/conv_bench.py --out "$OUT/conv_bench.txt" > "$OUT/conv_bench.log" 2>&1
  echo "convbench rc=$?" >> "$OUT/conv_bench.log"
  tail -4 "$OUT/conv_bench.log"
fi
PROF_CMD="python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-nms --no-predict --no-exclusive"
if has stats; then
  cd /tmp
  timeout -k 10 600 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o bench -- bash -c "cd $REPO && $PROF_CMD" > "$OUT/stats.log" 2>&1
  echo "stats rc=$?" >> "$OUT/stats.log"
  cd "$REPO"
  python tools/prof_summary.py stats "$OUT/stats" > "$OUT/kernel_stats_summary.txt" 2>&1
  head -40 "$OUT/kernel_stats_summary.txt"
  python tools/prof_summary.py timeline "$OUT/stats" 4 > "$OUT/kernel_timeline_summary.txt" 2>&1   # idle / overlap / per-queue gaps over the last 4 train steps
  head -12 "$OUT/kernel_timeline_summary.txt"
  # the raw per-dispatch trace is large; keep the stats csv only
  find "$OUT/stats" -name "*kernel_trace.csv" -size +8M -delete
fi
if has pmc; then
  # (rocprofv3 --pmc serialises the dispatches: every kernel of these passes has the chip to itself - the per-kernel figures are EXCLUSIVE ones,
  # whatever stream the kernel was launched on; r4zz: a pass with the side stream off measured the same durations)
  PMC_CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-nms --no-predict --no-exclusive"
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    cd /tmp
    timeout -k 10 420 rocprofv3 --pmc $set --kernel-trace -f csv -d "$OUT/pmc$i" -o bench -- bash -c "cd $REPO && $PMC_CMD" > "$OUT/pmc$i.log" 2>&1
    echo "pmc$i ($set) rc=$?" >> "$OUT/pmc$i.log"
    cd "$REPO"
    python tools/prof_summary.py pmc "$OUT/pmc$i" > "$OUT/pmc${i}_summary.txt" 2>&1
    head -12 "$OUT/pmc${i}_summary.txt"
  done
  python tools/pmc_traffic.py "$OUT/pmc1" "$OUT/pmc2" "$OUT/igemm_traffic.json" "$OUT/pmc3" > "$OUT/pmc_traffic.log" 2>&1
  cat "$OUT/pmc_traffic.log" | head -5
  python tools/prof_summary.py hbm "$OUT/pmc1" "$OUT/pmc2" > "$OUT/hbm_rate_per_kernel.txt" 2>&1
  for i in 1 2 3; do find "$OUT/pmc$i" -name "*.csv" -size +8M -delete; done
fi
if has nms; then
  # the post-prediction kernels alone: kernel stats + FETCH_SIZE / WRITE_SIZE passes of 20 calls -> profiles/nms_traffic.json
  cd /tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/nms_stats" -o nms -- bash -c "cd $REPO && python bench.py --only-nms 20" > "$OUT/nms_stats.log" 2>&1
  cd "$REPO"; python tools/prof_summary.py stats "$OUT/nms_stats" > "$OUT/nms_kernel_stats_summary.txt" 2>&1; head -12 "$OUT/nms_kernel_stats_summary.txt"
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1)); cd /tmp
    timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace -f csv -d "$OUT/nms_pmc$i" -o nms -- bash -c "cd $REPO && python bench.py --only-nms 20" > "$OUT/nms_pmc$i.log" 2>&1
    cd "$REPO"
  done
  python tools/pmc_traffic.py "$OUT/nms_pmc1" "$OUT/nms_pmc2" "$OUT/nms_traffic.json" - 20 > "$OUT/nms_traffic.log" 2>&1; head -4 "$OUT/nms_traffic.json"
  find "$OUT" -name "*.csv" -size +8M -delete
fi
if has subset; then  # a short parity pass on this build: kernels, blocks, the S model's whole-step checks
  timeout 400 python -m pytest tests/test_kernels.py tests/test_blocks.py -m gpu -q > "$OUT/pytest_gpu_subset.log" 2>&1
  timeout 300 python -m pytest tests/test_yolo_nas.py -m gpu -q -k "s_train_step_parity or s_backward_exact or l_backward_exact" >> "$OUT/pytest_gpu_subset.log" 2>&1
  grep -E "passed|failed" "$OUT/pytest_gpu_subset.log"
fi
if has tests; then
  SGX_TEST_DUMP="$OUT/test_dump.txt" timeout 900 python -m pytest tests -m gpu -q --durations=15 > "$OUT/pytest_gpu.log" 2>&1
  echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"
  tail -5 "$OUT/pytest_gpu.log"
fi
if has smoke; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
  echo "smoke rc=$?" >> "$OUT/smoke.log"
  tail -2 "$OUT/smoke.log"
fi
du -sh "$OUT"
