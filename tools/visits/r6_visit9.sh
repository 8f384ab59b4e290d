#!/bin/bash
# Round 6, visit 9: tile / variant search for YOLO-NAS-S with the ping-pong loop (variant 14) in the search, then the step A/B of the tables.
TAG=${1:-r6j}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 700 python tools/conv_tune.py --model s --planes --iters 5 --keep-wgrad-from super_gradients_amd/csrc/conv_tuning_gfx950.json --emit-table "$OUT/tune_s.json" --out "$OUT/conv_tune_s.txt" > "$OUT/conv_tune_s.log" 2>&1
tail -1 "$OUT/conv_tune_s.log"; head -40 "$OUT/conv_tune_s.txt"; grep -c "variant=14" "$OUT/conv_tune_s.txt"
python - "$OUT" <<'PY'
import json, sys, os
out = sys.argv[1]
base = json.load(open("super_gradients_amd/csrc/conv_tuning_gfx950.json"))
new = json.load(open(os.path.join(out, "tune_s.json")))
key = lambda e: (e["kind"], e["N"], e["H"], e["W"], e["C"], e["K"], e["R"], e["stride"], e["pad"])
newk = {key(e): e for e in new["entries"] if e["kind"] != "wgrad"}
# the S problems: every fwd / dgrad key of the new search replaces the old entry (or removes it when the heuristic won)
s_keys = set()
for line in open(os.path.join(out, "conv_tune_s.txt")):
    pass
merged = [e for e in base["entries"] if e["kind"] == "wgrad" or key(e) not in newk] + list(newk.values())
base["entries"] = merged
json.dump(base, open(os.path.join(out, "conv_tuning_s_v14.json"), "w"), indent=1)
print("entries", len(merged), "new S fwd/dgrad entries", len(newk), "with variant 14:", sum(1 for e in newk.values() if e["variant"] == 14))
PY
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-predict --no-exclusive"
for rep in 1 2 3; do
  for tab in default v14; do
    if [ $tab = v14 ]; then export SGX_CONV_TUNING="$OUT/conv_tuning_s_v14.json"; else unset SGX_CONV_TUNING; fi
    v=$(timeout 200 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['wgrad']['kernel_ms_per_step'])")
    echo "rep $rep $tab: $v"
  done
done | tee "$OUT/step_ab.txt"
