#!/bin/bash
# Round 6, visit 4: the YOLO-NAS-L bs32 leg runs 250 ms inside bench.py's process (r6a, r6c: all three invocations) and 109 ms alone (r6b).
# What of the headline run's state does it: the S network alive / its cached allocator blocks / the flush between legs / the host?
TAG=${1:-r6d}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
run() {  # name, python snippet
  timeout 300 python - "$1" > "$OUT/$1.txt" 2>&1 <<PY
import gc, json, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import bench
dev = torch.device("cuda:0")
def stats():
    s = torch.cuda.memory_stats()
    return s.get("num_device_alloc", 0), s.get("num_device_free", 0), s.get("num_alloc_retries", 0), round(torch.cuda.memory_reserved() / 2**30, 1), round(torch.cuda.memory_allocated() / 2**30, 1)
def leg(tag, *a, **k):
    a0 = stats(); t0 = time.time()
    o = bench.other_config_leg(dev, *a, **k)
    print(tag, o["value"], o["ms_per_step"], "host", o.get("host_enqueue_ms_per_step"), "allocs in timed steps", o.get("device_allocs_in_timed_steps"), "| before", a0, "after", stats(), "wall", round(time.time() - t0, 1), flush=True)
def s_run(steps=13):
    from super_gradients_amd.training import models
    from super_gradients_amd.training.losses import PPYoloELoss
    from super_gradients_amd.training.utils.ema import ModelEMA
    from super_gradients_amd.training.utils.optimizers import ArenaAdamW
    torch.manual_seed(42)
    net = models.get("yolo_nas_s", num_classes=80).materialize(dev).train()
    crit = PPYoloELoss(num_classes=80, use_static_assigner=False)
    opt = ArenaAdamW(net, lr=2e-4, weight_decay=1e-5, zero_weight_decay_on_bias_and_bn=True)
    ema = ModelEMA.from_params(net, decay=0.9997, decay_type="threshold")
    x, t = bench.synthetic_batch(32, 640, 42, dev)
    for i in range(steps):
        loss, _ = crit(net(x), t); loss.backward(); opt.step(); opt.zero_grad(); ema.update(net, i, 100000)
    torch.cuda.synchronize()
    return net, crit, opt, ema, x, t, loss
$2
PY
  grep -v -i "warning\|amdgpu.ids\|detach\|step_mfma" "$OUT/$1.txt" | tail -8
}
run a_S_alive_then_L 'keep = s_run(); print("S done", stats()); leg("L640 (S alive, its cache kept)", "yolo_nas", "l", 640, 32, loss_check=False)'
run b_S_alive_flush_then_L 'keep = s_run(); gc.collect(); torch.cuda.empty_cache(); print("S done + flush", stats()); leg("L640 (S alive, flushed)", "yolo_nas", "l", 640, 32, loss_check=False)'
run c_S_deleted_then_L 'keep = s_run(); del keep; gc.collect(); torch.cuda.empty_cache(); print("S deleted + flush", stats()); leg("L640 (S deleted)", "yolo_nas", "l", 640, 32, loss_check=False)'
run d_S_M_flush_L 'keep = s_run(); leg("M640 (S alive)", "yolo_nas", "m", 640, 32, loss_check=False); gc.collect(); torch.cuda.empty_cache(); leg("L640 (S alive, after M + flush)", "yolo_nas", "l", 640, 32, loss_check=False); leg("L640 again", "yolo_nas", "l", 640, 32, loss_check=False)'
du -sh "$OUT"
