#!/bin/bash
# Round 6, visit 6: per-problem tile / variant search for the OTHER BASELINE configurations (the committed table holds YOLO-NAS-S problems
# only: M, L and L@1280 run the built-in heuristics), the tables merged into one, and the step A/B per configuration.
TAG=${1:-r6f}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tune() {  # name, conv_tune args
  local name=$1; shift
  timeout 700 python tools/conv_tune.py "$@" --planes --wgrad --iters 4 --emit-table "$OUT/tune_$name.json" --out "$OUT/conv_tune_$name.txt" > "$OUT/conv_tune_$name.log" 2>&1
  echo "tune $name rc=$?"; tail -1 "$OUT/conv_tune_$name.log"; head -1 "$OUT/conv_tune_$name.txt"
}
tune m --model m
tune l1280 --model l --size 1280 --batch 8
tune l640 --model l
python - "$OUT" <<'PY'
import json, sys, os
out = sys.argv[1]
base = json.load(open("super_gradients_amd/csrc/conv_tuning_gfx950.json"))
key = lambda e: (e["kind"], e["N"], e["H"], e["W"], e["C"], e["K"], e["R"], e["stride"], e["pad"])
seen = {key(e) for e in base["entries"]}
merged, added = list(base["entries"]), {}
for name in ("m", "l1280", "l640"):
    p = os.path.join(out, f"tune_{name}.json")
    if not os.path.exists(p):
        continue
    t = json.load(open(p))
    n = 0
    for e in t["entries"]:
        if key(e) not in seen:
            seen.add(key(e)); merged.append(e); n += 1
    added[name] = dict(entries=n, ms_per_step_heuristic=t["meta"]["ms_per_step_heuristic"], ms_per_step_table=t["meta"]["ms_per_step_table"])
base["meta"]["other_configurations"] = added
base["entries"] = merged
json.dump(base, open(os.path.join(out, "conv_tuning_merged.json"), "w"), indent=1)
print("merged:", len(merged), "entries;", added)
PY
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-predict --no-exclusive"
for rep in 1 2; do
  for cfg in "--model m" "--model l --size 1280 --batch 8" "--model l" ""; do
    for tab in default merged; do
      if [ $tab = merged ]; then export SGX_CONV_TUNING="$OUT/conv_tuning_merged.json"; else unset SGX_CONV_TUNING; fi
      v=$(timeout 200 $B $cfg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['wgrad']['kernel_ms_per_step'])")
      echo "rep $rep [$cfg] $tab: $v"
    done
  done
done | tee "$OUT/step_ab.txt"
unset SGX_CONV_TUNING
du -sh "$OUT"
