#!/bin/bash
# Round 6, visit 3: (a) ablation lab of the bf16x3 implicit-GEMM loop (a -DSGX_IGEMM_LAB build swapped in: tools/_ab/libsgx_lab.so) - what a
# launch costs without its global loads / LDS stores / MFMAs / split / epilogue stores; (b) does the 250 ms YOLO-NAS-L bs32 leg of r6a recur
# inside the full bench line, and with which legs; (c) predict() after the eval()/train() walks were removed.
TAG=${1:-r6c}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
LIB=super_gradients_amd/csrc/libsgx_hip.so
cp $LIB /tmp/product.so
cp tools/_ab/libsgx_lab.so $LIB
P="fwd:32:40:40:384:192:1:1,fwd:32:80:80:192:384:3:2,fwd:32:20:20:1536:768:1:1,fwd:32:20:20:768:192:1:1,dgrad:32:80:80:192:96:1:1,fwd:32:80:80:192:64:1:1"
timeout 500 python tools/conv_lab.py --math bf16x3 --planes 1 --tiles 64x64 --ablate 0,1,2,3,4,8,16,5,7,23,31 --problems "$P" --rounds 3 --iters 10 --out "$OUT/igemm_ablation_6464.txt" > "$OUT/lab1.log" 2>&1
tail -3 "$OUT/lab1.log"
timeout 300 python tools/conv_lab.py --math bf16x3 --planes 1 --tiles 128x96,128x64 --ablate 0,1,2,4,16,31 --problems "fwd:32:40:40:384:192:1:1,fwd:32:80:80:192:384:3:2" --rounds 3 --iters 10 --out "$OUT/igemm_ablation_big_tiles.txt" > "$OUT/lab2.log" 2>&1
cp /tmp/product.so $LIB
cat "$OUT/igemm_ablation_6464.txt" "$OUT/igemm_ablation_big_tiles.txt"
for hp in 0 1 0; do
  SGX_PREDICT_HOST_POST=$hp timeout 200 python tools/predict_bench.py --batches 20 2>/dev/null | tail -1 | cut -c1-400
done > "$OUT/predict_bench_ab.txt"; cat "$OUT/predict_bench_ab.txt"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "S", d["value"], "|", " ".join(f"{o.get('config')}={o.get('value')}" for o in d.get("other_configs", [])), "| predict", d.get("predict",{}).get("value"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
timeout 600 python bench.py > "$OUT/bench_full.json" 2> "$OUT/bench_full.err"; show "$OUT/bench_full.json"
timeout 400 python bench.py --no-cpu-baseline --other-configs on > "$OUT/bench_nocpu.json" 2> "$OUT/bench_nocpu.err"; show "$OUT/bench_nocpu.json"
timeout 400 python bench.py --no-cpu-baseline --no-nms --no-predict --no-exclusive --other-configs on > "$OUT/bench_min.json" 2> "$OUT/bench_min.err"; show "$OUT/bench_min.json"
du -sh "$OUT"
