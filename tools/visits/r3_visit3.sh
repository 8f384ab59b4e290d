#!/bin/bash
# round-3 GPU visit 3: BN hi/lo mean fix vs headline backward, tree-fold wgrad, pconv tuning + PMC
TAG=${1:-r3c}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-predict"
run() { name=$1; shift; ( timeout 300 env "$@" $B > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err" ); python - "$OUT/bench_$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[2], d["value"], "img/s", d["ms_per_step"], "ms | igemm", r["achieved"], r["kernel_ms_per_step"], "| wgrad", r["wgrad"]["achieved"], r["wgrad"]["kernel_ms_per_step"], r["wgrad"]["launches_per_step"], "| excl", r["exclusive"]["achieved"], r["exclusive"]["wgrad_achieved"], "| host", d["host_enqueue_ms_per_step"], "| loss", d["config"]["final_loss"])
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1][:-5]+".err").read()[-800:])
PY
}
timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q -k "pconv or wgrad_group or conv_bwd or sums_to_zero or batchnorm" > "$OUT/pytest_new.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_new.log"; tail -4 "$OUT/pytest_new.log"
export SGX_TEST_DUMP="$OUT/headline_backward.txt"
timeout 400 python -m pytest tests/test_yolo_nas.py -m gpu -x -q -k "headline_config_backward_exact" > "$OUT/pytest_headline.log" 2>&1; tail -3 "$OUT/pytest_headline.log" | cut -c1-700
unset SGX_TEST_DUMP
run g40 SGX_WGRAD_GROUP_GFLOP=40
run g40_i8 SGX_WGRAD_GROUP_GFLOP=40 SGX_WGRAD_GROUP=6,8,1
run g40_i2 SGX_WGRAD_GROUP_GFLOP=40 SGX_WGRAD_GROUP=6,2,1
run g150_i8 SGX_WGRAD_GROUP_GFLOP=150 SGX_WGRAD_GROUP=6,8,1
run perlayer SGX_WGRAD_GROUP_GFLOP=0
run patch SGX_CONV_MATH=patch
timeout 300 python tools/conv_lab.py --math fp32,patch --rounds 3 --iters 8 \
  --problems fwd:32:80:80:64:64:3:1,fwd:32:40:40:96:96:3:1,fwd:32:160:160:32:32:3:1,fwd:32:20:20:256:256:3:1,fwd:32:80:80:48:48:3:1,fwd:32:40:40:128:128:3:1,dgrad:32:80:80:64:64:3:1,dgrad:32:160:160:96:192:3:2,dgrad:32:320:320:48:96:3:2,fwd:32:20:20:192:192:3:1 \
  --out "$OUT/lab_patch.txt" > "$OUT/lab.log" 2>&1; tail -22 "$OUT/lab.log"
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -f csv -d "$OUT/pmc$i" -o lab -- bash -c "cd $REPO && python tools/conv_lab.py --math patch --rounds 1 --iters 3 --problems fwd:32:80:80:64:64:3:1,fwd:32:40:40:96:96:3:1" > "$OUT/pmc$i.log" 2>&1
  echo "pmc$i rc=$?" >> "$OUT/pmc$i.log"
  cd "$REPO"; python tools/prof_summary.py pmc "$OUT/pmc$i" > "$OUT/pmc${i}_summary.txt" 2>&1; grep -i "pconv\|kernel " "$OUT/pmc${i}_summary.txt" | head -8 | cut -c1-400; cd /tmp
  find "$OUT/pmc$i" -name "*.csv" -size +8M -delete
done
cd "$REPO"
du -sh "$OUT"
