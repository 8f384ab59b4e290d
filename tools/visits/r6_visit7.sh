#!/bin/bash
# Round 6, visit 7: tile / variant search for ResNet-50 (BASELINE configs[1]) and PP-YOLOE-S (tools/conv_bench.py's recorder takes them now),
# step A/B of the merged table, then the full default bench line with the allocator / memory instrumentation.
TAG=${1:-r6g}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tune() {
  local name=$1; shift
  timeout 600 python tools/conv_tune.py "$@" --planes --wgrad --iters 4 --emit-table "$OUT/tune_$name.json" --out "$OUT/conv_tune_$name.txt" > "$OUT/conv_tune_$name.log" 2>&1
  echo "tune $name rc=$?"; tail -1 "$OUT/conv_tune_$name.log"; head -1 "$OUT/conv_tune_$name.txt"
}
tune resnet50 --model resnet50 --batch 64 --size 224
tune ppyoloe_s --model ppyoloe_s
python - "$OUT" <<'PY'
import json, sys, os
out = sys.argv[1]
base = json.load(open("super_gradients_amd/csrc/conv_tuning_gfx950.json"))
key = lambda e: (e["kind"], e["N"], e["H"], e["W"], e["C"], e["K"], e["R"], e["stride"], e["pad"])
seen = {key(e) for e in base["entries"]}
merged, added = list(base["entries"]), {}
for name in ("resnet50", "ppyoloe_s"):
    p = os.path.join(out, f"tune_{name}.json")
    if not os.path.exists(p):
        continue
    t = json.load(open(p)); n = 0
    for e in t["entries"]:
        if key(e) not in seen:
            seen.add(key(e)); merged.append(e); n += 1
    added[name] = dict(entries=n, ms_per_step_heuristic=t["meta"]["ms_per_step_heuristic"], ms_per_step_table=t["meta"]["ms_per_step_table"])
base["meta"].setdefault("other_configurations", {}).update(added)
base["entries"] = merged
json.dump(base, open(os.path.join(out, "conv_tuning_merged.json"), "w"), indent=1)
print("merged:", len(merged), "entries;", added)
PY
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-predict --no-exclusive"
for rep in 1 2; do
  for cfg in "--workload resnet50" "--workload ppyoloe"; do
    for tab in default merged; do
      if [ $tab = merged ]; then export SGX_CONV_TUNING="$OUT/conv_tuning_merged.json"; else unset SGX_CONV_TUNING; fi
      v=$(timeout 200 $B $cfg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['wgrad']['kernel_ms_per_step'])")
      echo "rep $rep [$cfg] $tab: $v"
    done
  done
done | tee "$OUT/step_ab.txt"
unset SGX_CONV_TUNING
t0=$(date +%s)
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$? ($(( $(date +%s) - t0 )) s)"
python - "$OUT/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("S", d["value"], d["ms_per_step"], "allocs", d["config"].get("device_allocs_in_timed_steps"), d["config"].get("hbm_gb"), "| predict", d.get("predict",{}).get("value"), "| nms", d.get("nms",{}).get("value"))
for o in d.get("other_configs", []):
    print(o.get("config"), o.get("value"), o.get("ms_per_step"), o.get("step_mfma_frac"), "host", o.get("host_enqueue_ms_per_step"), "allocs", o.get("device_allocs_in_timed_steps"), "retries", o.get("alloc_retries_in_timed_steps"), o.get("hbm_gb"), (o.get("loss_check_vs_oracle") or {}).get("max_rel_err"), o.get("error"))
PY
du -sh "$OUT"
