#!/bin/bash
# Round 5, visit 4: the stride-2 patch weight gradient with the rotated staging enumeration - parity, library A/B of the step (alt = the previous
# wgrad_patch.hip), conflicts counter; the patch conv kernel's 32-filter tiles with the second fragment set (SGX_PCONV_PIPE=1) - parity + step A/B.
TAG=${1:-r5f}; ALT=${2:-_alt/libsgx_prev_wpatch.so}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_kernels.py -m gpu -q -k "wgrad_patch or wgrad_group or pconv or math_patch" > "$OUT/pytest_a.log" 2>&1
tail -3 "$OUT/pytest_a.log" | cut -c1-300
SGX_PCONV_PIPE=1 timeout 300 python -m pytest tests/test_kernels.py tests/test_blocks.py -m gpu -q -k "pconv or math_patch or qarepvgg" > "$OUT/pytest_pipe.log" 2>&1
tail -3 "$OUT/pytest_pipe.log" | cut -c1-300
bash tools/visits/r4_lib_ab.sh "$TAG" "$ALT"
BENCH_ARGS="--no-exclusive" bash tools/visits/r4_ab.sh "$TAG" "A=1" "SGX_PCONV_PIPE=1"
cd /tmp
timeout -k 10 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d "$OUT/pmc_wg" -o p -- \
    bash -c "cd $REPO && python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-nms --no-predict --no-exclusive" > "$OUT/pmc_wg.log" 2>&1
(cd $REPO && python tools/prof_summary.py pmc "$OUT/pmc_wg" > "$OUT/pmc_wg_summary.txt" 2>&1; grep -E "wpatch|pconv" "$OUT/pmc_wg_summary.txt" | cut -c1-260)
find "$OUT/pmc_wg" -name "*.csv" -size +4M -delete
