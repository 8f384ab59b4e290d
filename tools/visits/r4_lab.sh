#!/bin/bash
# Round 4: a short weight-gradient lab visit (~1 GPU-minute):  gpurun --timeout 400 -- 'bash tools/visits/r4_lab.sh <tag> "<configs>" [pmc]'
TAG=${1:-r4lab}
CFG=${2:-nopatch,base}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 100 python -m pytest tests/test_kernels.py -m gpu -q -k "wgrad_patch" > "$OUT/pytest_wgrad.log" 2>&1
tail -2 "$OUT/pytest_wgrad.log" | cut -c1-200
timeout 150 python tools/wgrad_lab.py --configs "$CFG" --rounds 2 --iters 4 --out "$OUT/wgrad_lab.txt" > "$OUT/wgrad_lab.log" 2>&1
cut -c1-200 "$OUT/wgrad_lab.txt"
if [ -n "$3" ]; then
  cd /tmp
  timeout -k 10 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -f csv \
      -d "$OUT/lab_pmc" -o lab -- bash -c "cd $REPO && python tools/wgrad_lab.py --configs base --rounds 1 --iters 2" > "$OUT/lab_pmc.log" 2>&1
  cd "$REPO"
  python tools/prof_summary.py pmc "$OUT/lab_pmc" > "$OUT/lab_pmc_summary.txt" 2>&1
  grep -E "wpatch|kernel  " "$OUT/lab_pmc_summary.txt" | cut -c1-330 | head -10
  find "$OUT" -name "*.csv" -size +4M -delete
fi
