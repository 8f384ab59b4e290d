export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_lab2; mkdir -p $OUT
cd /tmp
for v in 0 5 14; do
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -f csv -d $OUT/v$v -o lab -- bash -c "cd $REPO && python tools/conv_lab.py --variants $v --problems fwd:32:80:80:64:64:3:1 --rounds 1 --iters 5" > $OUT/v$v.log 2>&1
  echo "variant $v rc=$?"
  cd $REPO; python tools/prof_summary.py pmc $OUT/v$v 2>&1 | grep -E "kernel  |igemm" | cut -c1-260; cd /tmp
done
