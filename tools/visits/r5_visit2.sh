#!/bin/bash
# Round 5, visit 2 (~8 GPU-minutes): the pipelined bf16x3 GEMM loop (igemm_kernel<..., KD = 32, NBUF = 2>, variant 6) on the chip:
# parity of the variant, the per-problem (tile, variant) search with variant 6 among the candidates, a step A/B committed table / new table,
# and one counter pass (matrix-pipe busy cycles) of a lab that runs the dominant 1x1 / stride-2 problems under variant 0 and 6.
TAG=${1:-r5c}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_kernels.py -m gpu -q -k "deep_slabs or bf16x3 or every_tile or tuning_table" > "$OUT/pytest_conv.log" 2>&1
tail -3 "$OUT/pytest_conv.log" | cut -c1-300
timeout 420 python tools/conv_tune.py --iters 5 --out "$OUT/conv_tune.txt" --emit-table "$OUT/conv_tuning_new.json" > "$OUT/conv_tune.log" 2>&1
tail -2 "$OUT/conv_tune.log" | cut -c1-300
head -40 "$OUT/conv_tune.txt" | cut -c1-200
grep -c "variant=6" "$OUT/conv_tune.txt"
BENCH_ARGS="--no-exclusive" bash tools/visits/r4_ab.sh "$TAG" "A=1" "SGX_CONV_TUNING=$OUT/conv_tuning_new.json"
# counters: the same problems under both loops, one process per variant (kernel names differ by the NBUF template argument)
cd /tmp
for v in 0 6; do
  timeout -k 10 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -f csv -d "$OUT/pmc_v$v" -o p -- \
    bash -c "cd $REPO && python tools/conv_lab.py --math bf16x3 --variants $v --rounds 2 --iters 5 --problems fwd:32:80:80:192:96:1:1,fwd:32:40:40:384:192:1:1,dgrad:32:80:80:192:192:1:1,fwd:32:20:20:768:384:1:1,fwd:32:80:80:192:384:3:2,fwd:32:160:160:96:192:3:2,fwd:32:20:20:256:256:3:1,dgrad:32:40:40:192:192:3:2" > "$OUT/pmc_v$v.log" 2>&1
  (cd $REPO && python tools/prof_summary.py pmc "$OUT/pmc_v$v" > "$OUT/pmc_v${v}_summary.txt" 2>&1; head -12 "$OUT/pmc_v${v}_summary.txt" | cut -c1-300)
  find "$OUT/pmc_v$v" -name "*.csv" -size +4M -delete
done
