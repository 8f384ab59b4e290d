#!/bin/bash
# Round 6, visit 47: the whole GPU suite and smoke() at the round's last commit
TAG=${1:-r6ax}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"; tail -3 "$OUT/pytest_gpu.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" >> "$OUT/smoke.log"; tail -2 "$OUT/smoke.log"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-nms --no-predict --other-configs off 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['step_mfma_frac'], d['config'].get('final_loss'))" | tee "$OUT/bench_short.txt"
