#!/bin/bash
# Round 5, visit 3: the stride-2 patch weight gradient with the even staging pitch (LDS store conflicts) - parity, the bias pin test,
# a library A/B of the step (product = new, alt = the previous wgrad_patch.hip), one counter pass of the weight-gradient lab for the conflicts.
TAG=${1:-r5e}; ALT=${2:-_alt/libsgx_prev_wpatch.so}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_kernels.py -m gpu -q -s -k "wgrad_patch or wgrad_group or wgrad_bf16x3" > "$OUT/pytest_wgrad.log" 2>&1
tail -4 "$OUT/pytest_wgrad.log" | cut -c1-300
grep "signed offsets" "$OUT/pytest_wgrad.log" | cut -c1-300
bash tools/visits/r4_lib_ab.sh "$TAG" "$ALT"
cd /tmp
timeout -k 10 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d "$OUT/pmc_wg" -o p -- \
    bash -c "cd $REPO && python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-nms --no-predict --no-exclusive" > "$OUT/pmc_wg.log" 2>&1
(cd $REPO && python tools/prof_summary.py pmc "$OUT/pmc_wg" > "$OUT/pmc_wg_summary.txt" 2>&1; grep -E "kernel  |wpatch" "$OUT/pmc_wg_summary.txt" | cut -c1-260)
find "$OUT/pmc_wg" -name "*.csv" -size +4M -delete
