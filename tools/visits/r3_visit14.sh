#!/bin/bash
# PMC passes over the weight-gradient lab (product build): where the loop's cycles go
TAG=${1:-r3o}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp
timeout -k 10 120 rocprofv3 --list-avail > "$OUT/counters_avail.txt" 2>&1
CMD="cd $REPO && python tools/wgrad_lab.py --configs base --rounds 1 --iters 2"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" \
           "FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "wgrad_kernel" -f csv -d "$OUT/pmc$i" -o lab -- bash -c "$CMD" > "$OUT/pmc$i.log" 2>&1
  echo "pmc$i ($set) rc=$?" >> "$OUT/pmc$i.log"
  (cd $REPO && python tools/prof_summary.py pmc "$OUT/pmc$i" > "$OUT/pmc${i}_summary.txt" 2>&1)
  grep -E "^kernel|wgrad_kernel|TOTAL" "$OUT/pmc${i}_summary.txt" | cut -c1-330 | head -12
  tail -2 "$OUT/pmc$i.log" | cut -c1-200
  find "$OUT/pmc$i" -name "*.csv" -size +4M -delete
done
du -sh "$OUT"
