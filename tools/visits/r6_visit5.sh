#!/bin/bash
# Round 6, visit 5: bisect the slow YOLO-NAS-L bs32 leg inside bench.py's own process (r6a, r6c): with the host profile of the leg, and with
# legs of the headline run left out.
TAG=${1:-r6e}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "S", d["value"], "|", " ".join(f"{o.get('config')}={o.get('value')}/host{o.get('host_enqueue_ms_per_step')}/allocs{o.get('device_allocs_in_timed_steps')}" for o in d.get("other_configs", [])))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
B="python bench.py --no-cpu-baseline --no-nms --no-predict --no-exclusive --other-configs on"
SGX_BENCH_LEG_PROFILE=1 timeout 300 $B > "$OUT/bench_min_profiled.json" 2> "$OUT/bench_min_profiled.err"; show "$OUT/bench_min_profiled.json"
grep -A22 "YOLO-NAS-L synthetic COCO 640" "$OUT/bench_min_profiled.err" | head -30
grep -A22 "YOLO-NAS-M synthetic" "$OUT/bench_min_profiled.err" | head -26
SGX_BENCH_SKIP=prof timeout 300 $B > "$OUT/bench_min_noprof.json" 2> "$OUT/bench_min_noprof.err"; show "$OUT/bench_min_noprof.json"
SGX_BENCH_SKIP=prof,host timeout 300 $B > "$OUT/bench_min_noprof_nohost.json" 2> "$OUT/bench_min_noprof_nohost.err"; show "$OUT/bench_min_noprof_nohost.json"
timeout 300 $B --steps 3 --warmup 1 > "$OUT/bench_min_short.json" 2> "$OUT/bench_min_short.err"; show "$OUT/bench_min_short.json"
du -sh "$OUT"
