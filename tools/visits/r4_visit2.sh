#!/bin/bash
# Round 4, visit 2 (~9 GPU-minutes): the weight gradient's PATCH kernel (wgrad_patch.hip) meets the chip.
#   gpurun --timeout 800 -- 'bash tools/visits/r4_visit2.sh r4b'
# 1. its parity test + the other weight-gradient kernel tests     2. lab: alone, against the slab loops, item size / filter-block choice
# 3. kernel trace of the lab (which kernel form takes what)       4. counters of the lab: matrix pipe busy, LDS conflicts, waits
# 5. step A/B (default = patch, bf16x3 slab loop, fp32)           6. S-model train-step parity under the new default
TAG=${1:-r4b}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 200 python -m pytest tests/test_kernels.py -m gpu -q -k "wgrad" > "$OUT/pytest_wgrad.log" 2>&1
tail -4 "$OUT/pytest_wgrad.log" | cut -c1-300
timeout 150 python tools/wgrad_lab.py --configs fp32,nopatch,base,base+p24.0.0,base+p96.0.0,base+p48.3.0,base+p48.1.0,base+g6.8.0 --rounds 2 --iters 4 \
    --out "$OUT/wgrad_lab_patch.txt" > "$OUT/wgrad_lab_patch.log" 2>&1
tail -4 "$OUT/wgrad_lab_patch.log" | cut -c1-260
cd /tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/lab_stats" -o lab -- bash -c "cd $REPO && python tools/wgrad_lab.py --configs base --rounds 1 --iters 3" > "$OUT/lab_stats.log" 2>&1
cd "$REPO"
python tools/prof_summary.py stats "$OUT/lab_stats" > "$OUT/lab_kernel_stats_summary.txt" 2>&1
grep -E "wpatch|wgrad" "$OUT/lab_kernel_stats_summary.txt" | cut -c1-200 | head -30
cd /tmp
timeout -k 10 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -f csv \
    -d "$OUT/lab_pmc" -o lab -- bash -c "cd $REPO && python tools/wgrad_lab.py --configs base --rounds 1 --iters 2" > "$OUT/lab_pmc.log" 2>&1
cd "$REPO"
python tools/prof_summary.py pmc "$OUT/lab_pmc" > "$OUT/lab_pmc_summary.txt" 2>&1
grep -E "wpatch|wgrad|kernel" "$OUT/lab_pmc_summary.txt" | cut -c1-260 | head -30
find "$OUT" -name "*.csv" -size +8M -delete
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-predict"
for cfg in "A=1" "SGX_WGRAD_MATH=bf16x3" "SGX_WGRAD_MATH=fp32" "A=2"; do
  timeout 120 env $cfg $B > "$OUT/bench_$cfg.json" 2> "$OUT/bench_$cfg.err"
  python - "$OUT/bench_$cfg.json" "$cfg" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[2], d["value"], "img/s", d["ms_per_step"], "ms | igemm", r["achieved"], "| wgrad", r["wgrad"]["achieved"], r["wgrad"]["kernel_ms_per_step"], "| excl", r["exclusive"]["achieved"], r["exclusive"]["wgrad_achieved"], "loss", d["config"].get("final_loss"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
SGX_TEST_DUMP="$OUT/backward_b.txt" timeout 200 python -m pytest tests/test_yolo_nas.py -m gpu -q -k "train_step_parity or backward_exact_without" > "$OUT/pytest_yolo_nas.log" 2>&1
tail -3 "$OUT/pytest_yolo_nas.log" | cut -c1-300
cat "$OUT/backward_b.txt" 2>/dev/null
