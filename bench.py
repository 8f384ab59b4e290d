"""bench.py - the north-star measurement: YOLO-NAS-S 640x640 train-step throughput (images/s) on N MI355X.

One "step" = one pass of the hot path over one synthetic batch that is already resident in HBM:
    forward (libsgx_hip conv/BN kernels) -> PPYoloELoss (TaskAligned assigner, VFL+GIoU+DFL) -> backward ->
    [gradient all-reduce over RCCL/xGMI, overlapped with backward, N > 1] -> AdamW (wd 1e-5, zero-WD on bias/BN) -> EMA.
Precision: fp32 end to end (dtype "fp32": the parity mode, conv on v_mfma_f32_32x32x2_f32; roofline peak 157.3 TFLOP/s).
Weak scaling: 32 images per GPU, global batch 32*N.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 32] [--size 640] [--model s] [--no-cpu-baseline]
N > 1 is launched by the driver as  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see the driver contract).  Extra objects:
  roofline      dominant kernel class = the implicit-GEMM conv kernel (forward + data gradient launches): algorithmic
                FLOPs (2*M*N*K of the real problem) / HIP-event time of those launches, measured live in the timed steps
                on the launch stream; peak = fp32 MFMA 157.3 TFLOP/s (MI355X_MICROARCH.md).  `wgrad` gives the same
                for the weight-gradient kernel, `step_mfma_frac` the whole-step figure SURVEY 8(d) defines
                (images/s x 101.634 GFLOP / peak).
  cpu_baseline  the CPU oracle (the reference's arithmetic, oracle/yolo_nas.py + oracle/ppyolo_loss.py, ATen/oneDNN
                kernels) timed on this box's host cores on a bounded sample of the same workload (kind "port").
"""
import argparse
import json
import os
import sys
import time

# the GPU box's driver only supports dmabuf IPC: RCCL / cross-process tensor sharing fail without this (must be set before HIP initialises)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

TRAIN_GFLOP_PER_IMG = {"s": 101.634, "m": 282.556, "l": 386.959}  # 3 x forward conv FLOPs @640^2
PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense (MI355X_MICROARCH.md; the headline figures with 2:1 sparsity are not used)
# a launch in bf16x3 arithmetic executes six bf16 MFMA products per algorithmic fp32 product: its matrix-pipe ceiling in algorithmic FLOPs
PEAK_BF16X3_EQUIV_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0


def conv_bound_ms(K, classes=(0, 2, 3)):
    """Per-launch roofline time of the recorded conv launches, every class priced on the pipe it EXECUTES on: class 0 = fp32
    matrix pipe, classes 2 / 3 (patch kernel, bf16x3 implicit GEMM) = the bf16 pipe at six products per algorithmic one."""
    return sum(K.prof_bound_ms(c, (PEAK_FP32_MFMA_TFLOPS if c == 0 else PEAK_BF16X3_EQUIV_TFLOPS) * 1e12, HBM_ACHIEVABLE_TBS * 1e12) for c in classes)


def dtype_label(K):
    """Storage and accumulation are fp32 everywhere; what differs is the pipe the products run on."""
    return "fp32" if K.get_conv_math() == "fp32" else "fp32 (bf16x3 MFMA)"
HBM_ACHIEVABLE_TBS = 6.3  # MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable


def synthetic_batch(batch, size, seed, device):
    import torch
    from util import synthetic_targets

    g = torch.Generator().manual_seed(seed)
    x = torch.rand(batch, 3, size, size, generator=g)
    t = synthetic_targets(batch, seed=seed, kmax=20, size=size, num_classes=80)
    return x.to(device), t.to(device)


def cpu_baseline(model, size, batch=32, seconds_budget=45.0, family="yolo_nas"):
    """the reference's arithmetic (CPU oracle: oracle/yolo_nas.py + oracle/ppyolo_loss.py on ATen / oneDNN kernels) on
    THIS box's host cores - the config batch size, every core (torch.set_num_threads(os.cpu_count())), AdamW lr 2e-4 wd 1e-5, fp32,
    forward / loss / backward / optimizer timed separately.  Bounded: 1 warm-up step, then timed steps until `seconds_budget` is spent
    (at most 5, at least 1) so that the default bench run stays within minutes.  The oracle loss check also runs with this thread count."""
    import torch
    from oracle.ppyolo_loss import PPYoloELossOracle
    from oracle.yolo_nas import YoloNAS as OracleYoloNAS
    from util import synthetic_targets

    cores = os.cpu_count() or 1
    # BASELINE.md asks for every core; on the GPU box os.cpu_count() is 256 and ATen / oneDNN with 256 threads oversubscribes badly (r2s:
    # 717 s per step = 0.045 images/s, against 1.5 images/s with 64 threads) - the baseline is meant to be the CPU path at its best
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    if family == "ppyoloe":
        from oracle.pp_yolo_e import PPYoloE as OraclePPYoloE

        net = OraclePPYoloE(model, num_classes=80).train()
    else:
        net = OracleYoloNAS(model, num_classes=80).train()
    opt = torch.optim.AdamW(net.parameters(), lr=2e-4, weight_decay=1e-5)
    crit = PPYoloELossOracle(80, use_static_assigner=False)
    x = torch.rand(batch, 3, size, size)
    t = synthetic_targets(batch, seed=42, kmax=20, size=size)
    split = {"fwd": 0.0, "loss": 0.0, "bwd": 0.0, "opt": 0.0}

    def step(record):
        t0 = time.perf_counter()
        out = net(x)
        t1 = time.perf_counter()
        loss, _ = crit(out, t)
        t2 = time.perf_counter()
        loss.backward()
        t3 = time.perf_counter()
        opt.step()
        opt.zero_grad()
        t4 = time.perf_counter()
        if record:
            for k, v in zip(split, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                split[k] += v

    step(False)
    t0 = time.perf_counter()
    n = 0
    while n < 5 and (n == 0 or time.perf_counter() - t0 < seconds_budget):
        step(True)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(batch * n / dt, 3), "unit": "images/s", "cores": threads, "kind": "port",
            "seconds_per_step": {k: round(v / n, 3) for k, v in split.items()},
            "sample": f"oracle {'PP-YOLOE' if family == 'ppyoloe' else 'YOLO-NAS'}-{model.upper()} {size}x{size} fp32 train step (fwd + PPYoloELoss + bwd + AdamW), batch {batch}, "
                      f"{n} timed step(s) after 1 warm-up, {threads} threads of {cores} host cores"}


def oracle_loss_check(net, crit, x, targets, model, family):
    """Parity of the benchmarked workload itself (outside the timed region): the HIP forward + PPYoloELoss of THIS batch at the
    initial weights against the CPU oracle (the reference's arithmetic) with the same weights.  Bar: 1e-4 relative on the loss items."""
    import torch
    from oracle.ppyolo_loss import PPYoloELossOracle

    if family == "ppyoloe":
        from oracle.pp_yolo_e import PPYoloE as Oracle
    else:
        from oracle.yolo_nas import YoloNAS as Oracle
    t0 = time.time()
    torch.set_num_threads(min(os.cpu_count() or 1, 64))  # (256 threads oversubscribe ATen on the GPU box, see cpu_baseline)
    ref = Oracle(model, num_classes=80)
    ref.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()}, strict=True)
    ref.train()
    with torch.no_grad():
        _, items_ref = PPYoloELossOracle(80, use_static_assigner=False)(ref(x.cpu()), targets.cpu())
        _, items = crit(net(x), targets)
    items = items.cpu()
    err = float((items - items_ref).abs().max() / items_ref.abs().max())
    rec = {"hip": [round(float(v), 6) for v in items], "oracle": [round(float(v), 6) for v in items_ref], "max_rel_err": float(f"{err:.3e}"),
           "tolerance": 1e-4, "seconds": round(time.time() - t0, 1),
           "what": "loss items [cls, iou, dfl, total] of the benchmarked batch at the initial weights: HIP path vs CPU oracle (same weights, same batch)"}
    if not err <= 1e-4:
        raise RuntimeError(f"bench: HIP loss differs from the CPU oracle on the benchmarked batch: {rec}")
    return rec


NMS_SCORE_PASSES = 1  # passes of the current kernels over the score tensor (csrc/nms.hip, stage 1: one append pass behind a threshold estimated from a 1/32 line sample; sgx_debug_set_nms_selection(0): the exact three-pass selection)


def nms_leg(device, iters=100, warmup=10):
    """BASELINE.json's second metric: NMS boxes/s.  SURVEY 8(d) config-5 style input: B=32 images, L=8400 anchors, 80 classes, scores
    ~ Beta(0.5,0.5)^4 and boxes clustered around 30 centres so that every image has >= 1000 candidates above the recipe's
    thresholds (score 0.01, top-k 1000, IoU 0.7, max 300, class-agnostic).  boxes/s = candidates entering NMS (after threshold +
    top-k, summed over the batch) / device time of the whole post-prediction call, HIP-event timed.  CPU beside it: the C
    restatement of torchvision's kernel (oracle/nms.c, single thread, as torchvision's CPU kernel is) on the same candidates."""
    import numpy as np
    import torch

    from oracle import nms as onms
    from super_gradients_amd import kernels as K

    B, L, C = 32, 8400, 80
    g = np.random.RandomState(0)
    cen = g.uniform(96, 544, (B, 30, 2))
    which = g.randint(0, 30, (B, L))
    c = np.take_along_axis(cen, which[..., None].repeat(2, -1), 1) + g.normal(0, 8, (B, L, 2))
    wh = g.uniform(20, 160, (B, L, 2))
    boxes = torch.from_numpy(np.concatenate([c - wh / 2, c + wh / 2], -1).astype(np.float32)).to(device)
    scores = torch.from_numpy((g.beta(0.5, 0.5, (B, L, C)) ** 4).astype(np.float32)).to(device)
    args = (0.01, 0.7, 1000, 300)

    def run():
        return K.nms(boxes, scores, *args, multi_label=True, class_mode=0)

    for _ in range(warmup):
        out = run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        out = run()
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / iters
    ncand = int(torch.clamp(out[3], max=1000).sum())
    kept = int(out[1].sum())
    fallbacks = K.nms_fallbacks()  # images whose stage 2 streamed the raw scores instead of stage 1's list (exact, many times slower): 0 here
    # CPU: the NMS proper on the same top-k candidates of 4 images (bounded sample)
    t0 = time.perf_counter()
    n_cpu = 0
    for b in range(4):
        sc, bx = scores[b].cpu(), boxes[b].cpu()
        i, j = (sc > 0.01).nonzero(as_tuple=False).T
        conf = sc[i, j]
        top = torch.topk(conf, 1000).indices
        onms.nms(bx[i][top], conf[top], 0.7)
        n_cpu += 1000
    cpu_s = time.perf_counter() - t0
    # HBM roofline of the call: the ALGORITHMIC traffic of post-prediction is one read of the B*L*C scores plus the selected
    # candidates' boxes and the output rows (K * 20 B per image) - whatever the implementation re-reads on top of that (its selection passes
    # over the scores) is reported as `traffic` / `passes_over_scores`, not credited
    algo_bytes = 1.0 * B * L * C * 4 + B * 1000 * 20.0
    return {"value": round(ncand / (ms * 1e-3), 1), "unit": "boxes/s", "ms_per_batch": round(ms, 4), "candidates": ncand, "kept": kept, "batch": B, "stage2_fallbacks": fallbacks,
            "roofline": {"bound": "hbm", "achieved": round(algo_bytes / (ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(algo_bytes / (ms * 1e-3) / 8e12, 4), "traffic": measured_nms_traffic().get("nms_bytes_per_call"),
                         "traffic_unit": "bytes/call (HBM, PMC: FETCH_SIZE x2 + WRITE_SIZE over the kernels of one post-prediction call; from the "
                                         "committed profiles/nms_traffic.json of the round's profiling visit, not re-measured in this run)",
                         "algorithmic_bytes_per_call": round(algo_bytes), "passes_over_scores": NMS_SCORE_PASSES,
                         "note": "algorithmic bytes = ONE read of the fp32 scores (86 MB per batch) + 20 B per selected candidate / time of the WHOLE "
                                 "post-prediction call (selection + per-image sort + suppression scan; the last two are latency-bound and move no HBM bytes)"},
            "config": "B=32 L=8400 C=80 multi-label, score>0.01, top-k 1000, IoU 0.7, max 300, class-agnostic",
            "cpu_baseline": {"value": round(n_cpu / cpu_s, 1), "unit": "boxes/s", "cores": 1, "kind": "port",
                             "sample": "4 images x 1000 candidates: threshold + top-k (ATen) + oracle/nms.c"}}


def predict_leg(device, model="s", batch=32, batches=10):
    """Inference side of the same model, reported next to the headline number: `model.predict()` end to end on `batch`
    synthetic 480x640 uint8 images resident in HBM with the reference's default YOLO-NAS COCO processing (longest side -> 636, centre pad to
    640x640, /255): one device pre-processing launch, the fused eval forward (bf16 kernels for predict's default fp16=True; the fp32 path
    is timed beside it), the NMS kernels, one device-to-host copy of the kept
    rows, the reference's box maps and result objects.  A failure is reported in the object, it does not take the bench line down."""
    import torch

    try:
        from super_gradients_amd.training import models
        from super_gradients_amd.training.processing import default_yolo_nas_coco_processing_params

        net = models.get(f"yolo_nas_{model}", num_classes=80).materialize(device)
        net.set_dataset_processing_params(**default_yolo_nas_coco_processing_params())
        g = torch.Generator().manual_seed(0)
        images = [torch.randint(0, 256, (480, 640, 3), generator=g, dtype=torch.uint8).to(device) for _ in range(batch)]
        def run(fp16):
            pipe = net._get_pipeline(conf=0.01, fp16=fp16)
            pipe(images, batch_size=batch)  # warm-up; takes the fused copy
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(batches):
                res = pipe(images, batch_size=batch)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / batches, len(res[0].prediction), pipe.half

        dt, ndet, half = run(True)        # the reference's default: predict(fp16=True) -> the bf16 kernels (csrc/half.hip)
        dt32, ndet32, _ = run(False)      # the fp32 path, for the price of the precision
        return {"value": round(batch / dt, 1), "unit": "images/s", "ms_per_batch": round(1e3 * dt, 3), "batch": batch,
                "dtype": "bf16 (bf16 activations / filters, fp32 accumulate, fp32 prediction outputs)" if half else "fp32",
                "detections_first_image": ndet,
                "fp32_path": {"value": round(batch / dt32, 1), "ms_per_batch": round(1e3 * dt32, 3), "detections_first_image": ndet32},
                "config": f"YOLO-NAS-{model.upper()} predict(): 480x640 uint8 -> 640x640, conf 0.01, iou 0.7, fused copy, random-init weights"}
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


OTHER_CONFIGS = (  # BASELINE.json configs 4, 5 and 2, plus YOLO-NAS-L at the headline shape: (key, workload, model, size, batch, oracle loss check)
    ("yolo_nas_m_640_bs32", "yolo_nas", "m", 640, 32, True),
    ("yolo_nas_l_640_bs32", "yolo_nas", "l", 640, 32, False),
    ("yolo_nas_l_1280_bs8", "yolo_nas", "l", 1280, 8, True),
    ("resnet50_224_bs64", "resnet50", None, 224, 64, True),
)


def other_config_leg(device, workload, model, size, batch, steps=10, warmup=3, loss_check=True):
    """One of the other BASELINE.json configurations under the same clock as the headline line: the same step (detection: forward,
    PPYoloELoss, backward, AdamW, EMA; ResNet-50: forward + cross-entropy + backward, as configs[1] says), `steps` timed steps between
    device synchronisations after `warmup` untimed ones, and - outside the timed region - the loss of the benchmarked batch at the
    initial weights against the CPU oracle at the configuration's FULL size (bar 1e-4 relative)."""
    import torch

    from super_gradients_amd.training import models

    torch.manual_seed(42)
    rec = {}
    if workload == "resnet50":
        from super_gradients_amd.training.losses import CrossEntropyLoss

        net = models.get("resnet50", num_classes=1000).materialize(device).train()
        x = torch.randn(batch, 3, size, size, device=device)
        y = torch.randint(0, 1000, (batch,), device=device)
        crit = CrossEntropyLoss()
        if loss_check:
            import torch.nn.functional as F

            from oracle.resnet import build

            t0 = time.time()
            torch.set_num_threads(min(os.cpu_count() or 1, 64))
            ref = build("resnet50", 1000)
            ref.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()}, strict=True)
            ref.train()
            with torch.no_grad():
                l_ref = float(F.cross_entropy(ref(x.cpu()), y.cpu()))
                l_hip = float(crit(net(x), y))
            err = abs(l_hip - l_ref) / abs(l_ref)
            rec["loss_check_vs_oracle"] = {"hip": round(l_hip, 6), "oracle": round(l_ref, 6), "max_rel_err": float(f"{err:.3e}"), "tolerance": 1e-4,
                                           "seconds": round(time.time() - t0, 1), "what": f"cross-entropy of the benchmarked batch ({batch} x {size}x{size}) at the initial weights, training-mode BatchNorm: HIP path vs CPU oracle"}
            if not err <= 1e-4:
                raise RuntimeError(f"bench: HIP loss differs from the CPU oracle: {rec}")

        def step():
            loss = crit(net(x), y)
            loss.backward()
            net.zero_grad()
            return loss

        gflop, label = 24.54, f"ResNet-50 synthetic ImageNet-shape {size}x{size}, bs={batch}, forward+backward only, random-init weights"
    else:
        from super_gradients_amd.training.losses import PPYoloELoss
        from super_gradients_amd.training.utils.ema import ModelEMA
        from super_gradients_amd.training.utils.optimizers import ArenaAdamW

        net = models.get(f"yolo_nas_{model}", num_classes=80).materialize(device).train()
        crit = PPYoloELoss(num_classes=80, use_static_assigner=False)
        opt = ArenaAdamW(net, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, zero_weight_decay_on_bias_and_bn=True)
        ema = ModelEMA.from_params(net, decay=0.9997, decay_type="threshold")
        x, targets = synthetic_batch(batch, size, 42, device)
        if loss_check:
            rec["loss_check_vs_oracle"] = oracle_loss_check(net, crit, x, targets, model, "yolo_nas")
        state = {"step": 0}

        def step():
            loss, _ = crit(net(x), targets)
            loss.backward()
            opt.step()
            opt.zero_grad()
            ema.update(net, state["step"], 100000)
            state["step"] += 1
            return loss

        gflop = TRAIN_GFLOP_PER_IMG[model] * (size / 640.0) ** 2
        label = f"YOLO-NAS-{model.upper()} synthetic COCO {size}x{size}, bs={batch}/GPU, PPYoloELoss(TAL)+AdamW+EMA, random-init weights"
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    allocs0, retries0 = torch.cuda.memory_stats().get("num_device_alloc", 0), torch.cuda.memory_stats().get("num_alloc_retries", 0)
    prof = None
    if os.environ.get("SGX_BENCH_LEG_PROFILE"):  # diagnosis aid: where the host spends a leg's timed steps (stderr)
        import cProfile

        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if prof is not None:
        import io
        import pstats

        prof.disable()
        buf = io.StringIO()
        pstats.Stats(prof, stream=buf).sort_stats("tottime").print_stats(14)
        print(f"---- {label}: {dt / steps * 1e3:.1f} ms per step, host loop {t_host / steps * 1e3:.1f} ms\n" + buf.getvalue()[:3500], file=sys.stderr, flush=True)
    value = batch * steps / dt
    rec["host_enqueue_ms_per_step"] = round(t_host / steps * 1e3, 3)  # (the loop returns before the device is done when the host is ahead)
    rec["device_allocs_in_timed_steps"] = torch.cuda.memory_stats().get("num_device_alloc", 0) - allocs0  # hipMalloc calls of the caching allocator: 0 in steady state
    # (r6a / r6c: two boxes ran the two 86 GB configurations 2.3x slower inside this process and three boxes did not; an allocator retry -
    # hipMalloc failed, every cached block was released, the request repeated - is what memory pressure on a shared box looks like)
    rec["alloc_retries_in_timed_steps"] = torch.cuda.memory_stats().get("num_alloc_retries", 0) - retries0
    free_b, total_b = torch.cuda.mem_get_info()
    rec["hbm_gb"] = {"reserved_by_this_process": round(torch.cuda.memory_reserved() / 2**30, 1), "free_on_device": round(free_b / 2**30, 1), "total": round(total_b / 2**30, 1)}
    out = {"workload": label, "value": round(value, 2), "unit": "images/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps, "warmup": warmup,
           "step_mfma_frac": round(value * gflop / 1e3 / PEAK_FP32_MFMA_TFLOPS, 4), "gflop_per_image": round(gflop, 3), "final_loss": round(float(loss), 5)}
    out.update(rec)
    return out


def other_configs_leg(device, steps=10, warmup=3, loss_check=True):
    """-> list of other_config_leg objects; a failing configuration is reported in its object and does not take the bench line down."""
    import gc

    import torch

    res = []
    for key, workload, model, size, batch, check in OTHER_CONFIGS:
        try:
            o = other_config_leg(device, workload, model, size, batch, steps, warmup, loss_check and check)
        except Exception as e:  # noqa: BLE001
            o = {"error": repr(e)}
        o["config"] = key
        res.append(o)
        gc.collect()
        torch.cuda.empty_cache()
    return res


def measured_traffic():
    """HBM bytes per conv-kernel launch (forward / data gradient, and weight gradient) and the loaded shader clock, from the committed
    rocprofv3 PMC passes of this same command (profiles/igemm_traffic.json, written by tools/pmc_traffic.py from the FETCH_SIZE /
    WRITE_SIZE / GRBM_GUI_ACTIVE passes, with MI355X_MICROARCH.md's gfx950 correction).  -> dict (empty when no pass is committed)"""
    f = os.path.join(ROOT, "profiles", "igemm_traffic.json")
    if not os.path.exists(f):
        return {}
    return json.load(open(f))


def measured_nms_traffic():
    f = os.path.join(ROOT, "profiles", "nms_traffic.json")
    return json.load(open(f)) if os.path.exists(f) else {}


def resnet50_main(args):
    """BASELINE.json configs[1]: ResNet-50, synthetic 224x224, bs 64, forward + backward only (no optimizer), 1 GPU."""
    import torch

    from super_gradients_amd.training import models
    from super_gradients_amd.training.losses import CrossEntropyLoss

    dev = torch.device("cuda:0")
    torch.manual_seed(42)
    net = models.get("resnet50", num_classes=1000).materialize(dev).train()
    x = torch.randn(64, 3, 224, 224, device=dev)
    y = torch.randint(0, 1000, (64,), device=dev)
    crit = CrossEntropyLoss()

    def step():
        loss = crit(net(x), y)
        loss.backward()
        net.zero_grad()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    value = 64 * args.steps / dt
    # kernel-level roofline: the same steps once more with a HIP event pair around every conv launch on its launch stream (as the
    # detection workloads do): class 0 = fp32-MFMA forward / data gradient, 2 = the bf16x3 patch kernel, 1 = weight gradients
    from super_gradients_amd import kernels as K
    from super_gradients_amd._lib import lib

    K.prof_enable(True)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    pc_ms, pc_fl, pc_n = (a + b for a, b in zip(K.prof_summary(2), K.prof_summary(3)))  # launches on the bf16 pipe: patch kernel + bf16x3 GEMM
    ig_ms, ig_fl, ig_n = (a + b for a, b in zip(K.prof_summary(0), (pc_ms, pc_fl, pc_n)))
    wg_ms, wg_fl, wg_n = K.prof_summary(1)
    ig_bytes = K.prof_bytes(0) + K.prof_bytes(2) + K.prof_bytes(3)
    ig_bound_ms = conv_bound_ms(K)
    K.prof_enable(False)
    ig_tf = ig_fl / (ig_ms * 1e-3) / 1e12 if ig_ms > 0 else 0.0
    wg_tf = wg_fl / (wg_ms * 1e-3) / 1e12 if wg_ms > 0 else 0.0
    print(json.dumps({"metric": "images/sec ResNet-50 224x224 fwd+bwd", "value": round(value, 2), "unit": "images/s", "n_gpus": 1, "steps": args.steps,
                      "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                      "vs_baseline": None, "dtype": dtype_label(K), "data": "synthetic",
                      "config": {"workload": "ResNet-50 synthetic ImageNet-shape 224x224, bs=64, forward+backward only, random-init weights",
                                 "final_loss": round(float(loss), 5), "conv_math": K.get_conv_math()},
                      "roofline": {"bound": "mfma", "kernel": "igemm_kernel / pconv_kernel (conv forward + data gradient)", "achieved": round(ig_tf, 2),
                                   "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ig_tf / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None,
                                   "timed_over": f"{args.steps} further steps with a HIP event pair around every launch of the kernel on its launch stream",
                                   "launches_per_step": ig_n // max(args.steps, 1), "kernel_ms_per_step": round(ig_ms / args.steps, 3),
                                   "algorithmic_bytes_per_launch": round(ig_bytes / max(ig_n, 1)),
                                   "fp32_pipe": {"achieved": round((ig_fl - pc_fl) / ((ig_ms - pc_ms) * 1e-3) / 1e12, 2) if ig_ms > pc_ms else None,
                                                 "peak": PEAK_FP32_MFMA_TFLOPS, "launches_per_step": (ig_n - pc_n) // max(args.steps, 1),
                                                 "kernel_ms_per_step": round((ig_ms - pc_ms) / args.steps, 3)},
                                   "bf16_pipe": None if pc_n == 0 else {"achieved": round(6.0 * pc_fl / (pc_ms * 1e-3) / 1e12, 1), "peak": PEAK_BF16_MFMA_TFLOPS,
                                                                        "frac": round(6.0 * pc_fl / (pc_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
                                                                        "launches_per_step": pc_n // max(args.steps, 1),
                                                                        "kernel_ms_per_step": round(pc_ms / args.steps, 3),
                                                                        "note": "executed bf16 MFMA FLOPs = 6 x algorithmic"},
                                   "per_launch_bound": {"frac": round(ig_bound_ms / ig_ms, 4) if ig_ms > 0 else None,
                                                        "note": f"sum over launches of max(FLOPs / the peak of the pipe the launch executes on ({PEAK_FP32_MFMA_TFLOPS} TFLOP/s fp32 MFMA; {PEAK_BF16X3_EQUIV_TFLOPS:.1f} = {PEAK_BF16_MFMA_TFLOPS:.0f} / 6 for bf16x3 launches), algorithmic bytes / {HBM_ACHIEVABLE_TBS} TB/s) / measured kernel time"},
                                   "wgrad": {"achieved": round(wg_tf, 2), "frac": round(wg_tf / PEAK_FP32_MFMA_TFLOPS, 4), "launches_per_step": wg_n // max(args.steps, 1),
                                             "kernel_ms_per_step": round(wg_ms / args.steps, 3),
                                             "math": {0: "fp32", 1: "bf16x3", 2: "bf16x3+patch"}[int(lib().sgx_conv_get_wgrad_math())]},
                                   "step_mfma_frac": round(value * 24.54 / 1e3 / PEAK_FP32_MFMA_TFLOPS, 4),
                                   "note": "step_mfma_frac: images/s x 24.54 GFLOP (SURVEY 8d) / peak"}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--model", default="s", choices=["s", "m", "l", "x"], help="x: PP-YOLOE only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ema", action="store_true")
    ap.add_argument("--no-nms", action="store_true")
    ap.add_argument("--no-predict", action="store_true", help="skip the predict() leg (YOLO-NAS only)")
    ap.add_argument("--only-nms", type=int, default=0, metavar="CALLS", help="run ONLY the NMS leg with that many timed calls and print its object "
                    "(what the rocprofv3 passes of the post-prediction kernels trace: tools/gpu_round.sh nms stage)")
    ap.add_argument("--other-configs", default="auto", choices=["auto", "on", "off"], help="the `other_configs` array: the other BASELINE.json configurations "
                    "(YOLO-NAS-M bs32, L bs32, L@1280 bs8, ResNet-50 bs64), 10 timed steps each + the oracle loss check at the configuration's full size; "
                    "auto = with the full default line (i.e. unless --no-cpu-baseline), so that the quick A/B and profiling invocations stay what they were")
    ap.add_argument("--loss-check-only", action="store_true", help="run the oracle loss check of the benchmarked batch but not the timed CPU baseline")
    ap.add_argument("--no-exclusive", action="store_true", help="skip the 3 extra untimed steps that time the conv kernels without the side stream")
    ap.add_argument("--sync-bn", action="store_true", help="synchronised BatchNorm across ranks (recipe setting; off in the reference's own benchmark)")
    ap.add_argument("--workload", default="yolo_nas", choices=["yolo_nas", "resnet50", "ppyoloe"],
                    help="yolo_nas = BASELINE.json's headline config; resnet50 = configs[1]; ppyoloe = SURVEY 8f-1 (same loss / step, CSPResNet model)")
    args = ap.parse_args()
    if args.only_nms:
        import torch

        print(json.dumps(nms_leg(torch.device("cuda:0"), iters=args.only_nms, warmup=0)), flush=True)
        return
    if args.workload == "resnet50":
        if args.gpus != 1:
            raise RuntimeError("the ResNet-50 workload (BASELINE.json configs[1]) is single-GPU")
        return resnet50_main(args)

    import torch
    import torch.distributed as dist

    from super_gradients_amd import kernels as K
    from super_gradients_amd._lib import lib
    from super_gradients_amd.training import models
    from super_gradients_amd.training.losses import PPYoloELoss
    from super_gradients_amd.training.utils.distributed_training_utils import GradientAllReducer, setup_device_from_env
    from super_gradients_amd.training.utils.distributed_training_utils import barrier as dist_barrier
    from super_gradients_amd.training.utils.ema import ModelEMA
    from super_gradients_amd.training.utils.optimizers import ArenaAdamW

    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a HIP GPU (the product has no CPU path)")
    if os.environ.get("SGX_WGRAD_SPLIT"):  # experiment switch: weight-gradient split target (waves), see csrc/conv.hip wgrad_plan
        lib().sgx_debug_set_tiles(0, 0, 0, 0, int(os.environ["SGX_WGRAD_SPLIT"]))
    rank, world, device = setup_device_from_env()
    if world != args.gpus:
        raise RuntimeError(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with python -m torch.distributed.run --nproc-per-node {args.gpus}")

    torch.manual_seed(42)  # identical initial weights on every rank (the reference's default seed)
    family = "PP-YOLOE" if args.workload == "ppyoloe" else "YOLO-NAS"
    net = models.get(f"{'ppyoloe' if args.workload == 'ppyoloe' else 'yolo_nas'}_{args.model}", num_classes=80)
    net.materialize(device)
    net.train()
    if args.sync_bn:
        net.set_sync_bn(True)
    reducer = GradientAllReducer(net, net.gradient_buckets())
    reducer.broadcast_parameters(0)
    crit = PPYoloELoss(num_classes=80, use_static_assigner=False)
    opt = ArenaAdamW(net, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, zero_weight_decay_on_bias_and_bn=True)
    ema = None if args.no_ema else ModelEMA.from_params(net, decay=0.9997, decay_type="threshold")
    x, targets = synthetic_batch(args.batch, args.size, 42 + rank, device)

    # parity of the benchmarked workload itself (rank 0 of a single-GPU run; skipped together with the CPU baseline)
    loss_check = None
    if world == 1 and (args.loss_check_only or not args.no_cpu_baseline):
        loss_check = oracle_loss_check(net, crit, x, targets, args.model, "ppyoloe" if args.workload == "ppyoloe" else "yolo_nas")

    state = {"step": 0}

    def step():
        if world > 1:
            reducer.broadcast_buffers(0)  # what DDP(broadcast_buffers=True), the reference's wrapping, does at every forward
        out = net(x)
        loss, _ = crit(out, targets)
        loss.backward()
        opt.step(grad_scale=reducer.grad_scale if world > 1 else None)
        opt.zero_grad()
        if ema is not None:
            ema.update(net, state["step"], 100000)
        state["step"] += 1
        return loss

    def fence():
        dist_barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    allocs0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    allocs_timed = torch.cuda.memory_stats().get("num_device_alloc", 0) - allocs0
    free_b, total_b = torch.cuda.mem_get_info()
    # Roofline leg: the SAME K steps once more with a HIP event pair around every conv launch on its launch stream.  Recording ~900 events
    # per step costs ~3 % of the step (r2f: 577 vs 558 images/s), so it is kept out of the K steps `value` is measured on; the per-kernel
    # durations are what rocprofv3 --kernel-trace reports for the same command (profiles/).
    skip = set((os.environ.get("SGX_BENCH_SKIP") or "").split(","))  # diagnosis aid: legs left out ("prof", "host")
    K.prof_enable(True)
    for _ in range(0 if "prof" in skip else args.steps):
        step()
    fence()
    # class 0 = fp32-MFMA implicit GEMM, class 2 = the bf16x3 patch kernel, class 3 = the implicit GEMM in bf16x3 arithmetic (conv math
    # "patch_bf3": the deep problems the patch kernel does not take): forward / data gradient together
    pc_ms, pc_fl, pc_n = K.prof_summary(2)
    g3_ms, g3_fl, g3_n = K.prof_summary(3)
    bf_ms, bf_fl, bf_n = pc_ms + g3_ms, pc_fl + g3_fl, pc_n + g3_n  # everything on the bf16 matrix pipe
    ig_bytes, wg_bytes = K.prof_bytes(0) + K.prof_bytes(2) + K.prof_bytes(3), K.prof_bytes(1)
    ig_ms, ig_fl, ig_n = (a + b for a, b in zip(K.prof_summary(0), (bf_ms, bf_fl, bf_n)))
    wg_ms, wg_fl, wg_n = K.prof_summary(1)
    # per-launch roofline time: max(FLOPs / MFMA peak, algorithmic bytes / achievable HBM rate) summed over the same launches
    ig_bound_ms = conv_bound_ms(K)
    wg_bound_ms = K.prof_bound_ms(1, (PEAK_BF16X3_EQUIV_TFLOPS if lib().sgx_conv_get_wgrad_math() else PEAK_FP32_MFMA_TFLOPS) * 1e12, HBM_ACHIEVABLE_TBS * 1e12)
    K.prof_enable(False)
    # The per-launch figures above are taken while the weight-gradient kernels run concurrently on the side HIP stream (they share the
    # CUs, which is what makes the step faster but stretches every launch).  For the kernel's own efficiency: 3 extra, untimed steps
    # with the side stream off (every kernel has the GPU to itself), same HIP-event timing.
    fence()
    # (round 6: the branch stream - sub-chains of the model on lanes beside the main chain, modules/engine.py fork_branch - stretches the
    # launches the same way; `single_chain` = 3 extra steps with the lanes off and the side stream on, the concurrency rounds 1 - 5's `frac`
    # was measured under, for the trend across rounds; `exclusive` has lanes AND side stream off)
    branch_mode = getattr(net, "branch_mode", 0)
    sc_ms = sc_fl = scw_ms = scw_fl = 0.0
    if branch_mode:
        net.branch_mode = 0
        K.prof_enable(True)
        for _ in range(0 if args.no_exclusive else 3):
            step()
        fence()
        sc_ms, sc_fl = (sum(K.prof_summary(c)[i] for c in (0, 2, 3)) for i in (0, 1))
        scw_ms, scw_fl, _ = K.prof_summary(1)
        K.prof_enable(False)
    side, net.side_stream = getattr(net, "side_stream", None), None
    K.prof_enable(True)
    for _ in range(0 if args.no_exclusive else 3):
        step()
    fence()
    expc_ms, expc_fl, expc_n = K.prof_summary(2)
    exg3_ms, exg3_fl, exg3_n = K.prof_summary(3)
    ex_ms, ex_fl, ex_n = (a + b + c for a, b, c in zip(K.prof_summary(0), (expc_ms, expc_fl, expc_n), (exg3_ms, exg3_fl, exg3_n)))
    exw_ms, exw_fl, exw_n = K.prof_summary(1)
    ex_bound_ms = conv_bound_ms(K)
    K.prof_enable(False)
    net.side_stream = side
    if branch_mode:
        net.branch_mode = branch_mode
    # host side of one step: enqueue time of a step with the device idle at the start (no sync inside)
    fence()
    h0 = time.perf_counter()
    for _ in range(0 if "host" in skip else 2):
        step()
    host_ms = (time.perf_counter() - h0) / 2 * 1e3
    fence()
    tmax = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    loss_val = float(loss.detach())

    if rank == 0:
        imgs = args.batch * world * args.steps
        value = imgs / dt
        ig_tf = ig_fl / (ig_ms * 1e-3) / 1e12 if ig_ms > 0 else 0.0
        wg_tf = wg_fl / (wg_ms * 1e-3) / 1e12 if wg_ms > 0 else 0.0
        per_gpu = value / world
        # the committed PMC passes were taken on the headline workload only
        pmc = measured_traffic() if (args.workload == "yolo_nas" and args.model == "s") else {}
        traffic, traffic_src = pmc.get("bytes_per_launch"), pmc.get("source", "no PMC pass committed for this workload")
        rec = {
            "metric": f"images/sec/node {family}-{args.model.upper()} {args.size}x{args.size} train-step",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "host_enqueue_ms_per_step": round(host_ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype_label(K), "data": "synthetic",
            "config": {"workload": f"{family}-{args.model.upper()} synthetic COCO {args.size}x{args.size}, bs={args.batch}/GPU, PPYoloELoss(TAL)+AdamW"
                                   + ("" if args.no_ema else "+EMA") + ", random-init weights",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}", "final_loss": round(loss_val, 5),
                       # conv_math: "patch_bf3" (default) = bf16x3 patch kernel + per-problem bf16x3 / fp32 implicit GEMM; SGX_CONV_MATH=fp32|patch|... ; conv_variant /
                       # conv_tuning_entries: experiment switch and per-problem (tile, variant) table (tools/conv_tune.py --emit-table), 0 = heuristics
                       "conv_math": K.get_conv_math(), "conv_variant": int(os.environ.get("SGX_CONV_VARIANT") or 0),
                       "bn_reduce_in_data_gradients": bool(getattr(net, "fuse_bn_reduce", False)),
                       "conv_tuning_entries": int(lib().sgx_conv_tuning_size()),
                       # pre-split filter planes: the step's filters split into bf16x3 pieces once per step (SGX_FILTER_PLANES=0: off)
                       "filter_planes": getattr(net, "_fp_jobs", None) is not None,
                       "allreduce_from_side_stream": bool(reducer.from_side) if world > 1 else None,
                       # hipMalloc calls of torch's caching allocator inside the timed steps (0 once the pool has its steady shape) and the device's memory
                       "device_allocs_in_timed_steps": allocs_timed,
                       "hbm_gb": {"reserved_by_this_process": round(torch.cuda.memory_reserved() / 2**30, 1), "free_on_device": round(free_b / 2**30, 1),
                                  "total": round(total_b / 2**30, 1)}},
            "roofline": {"bound": "mfma", "kernel": "igemm_kernel (implicit-GEMM conv forward + data gradient, "
                                                    + ("v_mfma_f32_32x32x2_f32)" if K.get_conv_math() == "fp32" else
                                                       "v_mfma_f32_32x32x2_f32) + pconv_kernel (3x3 problems from an LDS patch, v_mfma_f32_32x32x16_bf16 x6)" if K.get_conv_math() == "patch"
                                                       else "v_mfma_f32_32x32x16_bf16 x6 where taps x channels >= 192, v_mfma_f32_32x32x2_f32 below) + pconv_kernel (3x3 stride-1 "
                                                            "problems from an LDS patch, v_mfma_f32_32x32x16_bf16 x6)" if K.get_conv_math() == "patch_bf3"
                                                       else "v_mfma_f32_32x32x16_bf16 x6 / v_mfma_f32_32x32x2_f32 per problem)"),
                         "achieved": round(ig_tf, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ig_tf / PEAK_FP32_MFMA_TFLOPS, 4),
                         # the same launches priced on the pipe most of them EXECUTE on: bf16 MFMA FLOPs issued (6 x algorithmic) / the dense bf16 peak
                         "frac_of_executed_pipe": None if bf_n == 0 else round(6.0 * bf_fl / (bf_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
                         "frac_note": "frac = algorithmic fp32 FLOPs / the fp32-MFMA peak BASELINE.md prices the target against; "
                                      "frac_of_executed_pipe = the bf16x3 launches (most of the step) against the bf16 pipe they run on.  Launch durations are "
                                      "taken in the benchmarked configuration, i.e. with the weight-gradient stream and (round 6) the branch-stream lanes sharing the "
                                      "chip - see single_chain (lanes off: the conditions of earlier rounds' frac) and exclusive (nothing concurrent); the step-level "
                                      "figure that follows the wall clock is step_mfma_frac",
                         "timed_over": f"{args.steps} further steps of the same loop with a HIP event pair around every launch of the kernel on its launch stream",
                         "traffic": traffic, "traffic_unit": "bytes/launch (HBM, PMC)", "traffic_source": traffic_src, "traffic_measured_in_this_run": False,
                         # the two matrix pipes apart (round 4): launches on the fp32 pipe against the fp32 peak, launches of the bf16x3 patch kernel
                         # by the bf16 MFMA work they EXECUTE (six products per algorithmic one) against the dense bf16 peak; `achieved` above is
                         # the combined fp32-equivalent figure
                         "fp32_pipe": {"achieved": round((ig_fl - bf_fl) / ((ig_ms - bf_ms) * 1e-3) / 1e12, 2) if ig_ms > bf_ms else None, "peak": PEAK_FP32_MFMA_TFLOPS,
                                       "frac": round((ig_fl - bf_fl) / ((ig_ms - bf_ms) * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) if ig_ms > bf_ms else None,
                                       "launches_per_step": (ig_n - bf_n) // max(args.steps, 1), "kernel_ms_per_step": round((ig_ms - bf_ms) / args.steps, 3)},
                         "bf16_pipe": None if bf_n == 0 else {"achieved": round(6.0 * bf_fl / (bf_ms * 1e-3) / 1e12, 1), "peak": PEAK_BF16_MFMA_TFLOPS,
                                                              "frac": round(6.0 * bf_fl / (bf_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
                                                              "launches_per_step": bf_n // max(args.steps, 1), "kernel_ms_per_step": round(bf_ms / args.steps, 3),
                                                              "note": "patch kernel + implicit-GEMM launches in bf16x3 arithmetic; executed bf16 MFMA FLOPs = 6 x algorithmic"},
                         # the implicit GEMM's launches that run in bf16x3 arithmetic (conv math "patch_bf3": stride-2 3x3, deep 1x1, QARepVGG two-branch forms)
                         "gemm_bf16x3": None if g3_n == 0 else {
                             "launches_per_step": g3_n // max(args.steps, 1), "kernel_ms_per_step": round(g3_ms / args.steps, 3),
                             "algorithmic_tflops": round(g3_fl / (g3_ms * 1e-3) / 1e12, 2), "executed_bf16_tflops": round(6.0 * g3_fl / (g3_ms * 1e-3) / 1e12, 1),
                             "frac_of_bf16_mfma_peak": round(6.0 * g3_fl / (g3_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
                             "exclusive_algorithmic_tflops": round(exg3_fl / (exg3_ms * 1e-3) / 1e12, 2) if exg3_ms > 0 else None},
                         "algorithmic_bytes_per_launch": round(ig_bytes / max(ig_n, 1)), "launches_per_step": ig_n // max(args.steps, 1), "avg_launch_us": round(ig_ms * 1e3 / max(ig_n, 1), 2),
                         "gflop_per_launch": round(ig_fl / max(ig_n, 1) / 1e9, 3), "kernel_ms_per_step": round(ig_ms / args.steps, 3),
                         # the launch mix against the bound that applies to EACH launch's shape (shallow 1x1 layers are nearer the HBM bound than the MFMA one)
                         "per_launch_bound": {"bound_ms_per_step": round(ig_bound_ms / args.steps, 3), "frac": round(ig_bound_ms / ig_ms, 4) if ig_ms > 0 else None,
                                              "exclusive_frac": round(ex_bound_ms / ex_ms, 4) if ex_ms > 0 else None,
                                              "wgrad_frac": round(wg_bound_ms / wg_ms, 4) if wg_ms > 0 else None,
                                              "note": f"sum over launches of max(FLOPs / the peak of the pipe the launch executes on ({PEAK_FP32_MFMA_TFLOPS} TFLOP/s fp32 MFMA; {PEAK_BF16X3_EQUIV_TFLOPS:.1f} = {PEAK_BF16_MFMA_TFLOPS:.0f} / 6 for bf16x3 launches), algorithmic bytes / {HBM_ACHIEVABLE_TBS} TB/s) / measured kernel time"},
                         "exclusive": {"achieved": round(ex_fl / (ex_ms * 1e-3) / 1e12, 2) if ex_ms > 0 else None,
                                       "frac": round(ex_fl / (ex_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) if ex_ms > 0 else None,
                                       "wgrad_achieved": round(exw_fl / (exw_ms * 1e-3) / 1e12, 2) if exw_ms > 0 else None,
                                       "note": "same kernels and launches with the side HIP stream and the branch-stream lanes disabled (every kernel has the "
                                               "chip to itself): 3 extra untimed steps after the timed region"},
                         "single_chain": None if sc_ms <= 0 else {
                             "achieved": round(sc_fl / (sc_ms * 1e-3) / 1e12, 2), "frac": round(sc_fl / (sc_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                             "wgrad_achieved": round(scw_fl / (scw_ms * 1e-3) / 1e12, 2) if scw_ms > 0 else None,
                             "note": "same kernels and launches with the branch-stream lanes off and the side stream on - the concurrency `frac` was measured under "
                                     "in rounds 1 - 5 (3 extra untimed steps).  `frac` itself is taken in the benchmarked configuration: since round 6 sub-chains of "
                                     "the model run on two more HIP streams beside the main chain, which shortens the step and stretches every launch that shares "
                                     "the chip with them (sum of kernel durations per step up, wall clock per step down)"},
                         "wgrad": {"achieved": round(wg_tf, 2), "frac": round(wg_tf / PEAK_FP32_MFMA_TFLOPS, 4), "launches_per_step": wg_n // max(args.steps, 1),
                                   "kernel_ms_per_step": round(wg_ms / args.steps, 3),
                                   # arithmetic of the weight gradient (sgx_conv_get_wgrad_math: 0 fp32 pipe, 1 bf16x3 slab loop, 2 + patch kernel): in
                                   # modes 1 / 2 the algorithmic FLOPs run as six bf16 MFMA products each - priced against the dense bf16 peak too
                                   "math": {0: "fp32", 1: "bf16x3", 2: "bf16x3+patch"}[int(lib().sgx_conv_get_wgrad_math())],
                                   "executed_bf16_tflops": round(6.0 * wg_tf, 1) if lib().sgx_conv_get_wgrad_math() else None,
                                   "frac_of_bf16_mfma_peak": round(6.0 * wg_tf / PEAK_BF16_MFMA_TFLOPS, 4) if lib().sgx_conv_get_wgrad_math() else None,
                                   "traffic": pmc.get("wgrad_bytes_per_launch"), "traffic_unit": "bytes/launch (HBM, PMC; a launch = one group of weight gradients)",
                                   "algorithmic_bytes_per_launch": round(wg_bytes / max(wg_n, 1)), "gflop_per_launch": round(wg_fl / max(wg_n, 1) / 1e9, 3)},
                         # the patch kernel alone, priced against BOTH pipes: algorithmic (fp32-equivalent) FLOPs against the fp32 matrix peak, and the
                         # bf16 MFMA FLOPs it actually executes (six products per algorithmic one) against the dense bf16 peak
                         "patch_kernel": None if pc_n == 0 else {
                             "launches_per_step": pc_n // max(args.steps, 1), "kernel_ms_per_step": round(pc_ms / args.steps, 3),
                             "algorithmic_tflops": round(pc_fl / (pc_ms * 1e-3) / 1e12, 2), "frac_of_fp32_mfma_peak": round(pc_fl / (pc_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                             "executed_bf16_tflops": round(6.0 * pc_fl / (pc_ms * 1e-3) / 1e12, 1), "frac_of_bf16_mfma_peak": round(6.0 * pc_fl / (pc_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
                             "exclusive_algorithmic_tflops": round(expc_fl / (expc_ms * 1e-3) / 1e12, 2) if expc_ms > 0 else None},
                         "loaded_clock_ghz": pmc.get("loaded_clock_ghz"),
                         # whole-step MFMA utilisation: algorithmic conv FLOPs of one step (model table; measured launches for PP-YOLOE) / step time
                         "step_mfma_frac": round((per_gpu * TRAIN_GFLOP_PER_IMG[args.model] * (args.size / 640.0) ** 2 / 1e3 if args.workload == "yolo_nas"
                                                  else (ig_fl + wg_fl) / args.steps / (dt / args.steps) / 1e12) / PEAK_FP32_MFMA_TFLOPS, 4)},
        }
        if loss_check is not None:
            rec["config"]["loss_check_vs_oracle"] = loss_check
        if not args.no_cpu_baseline and world == 1:
            rec["cpu_baseline"] = cpu_baseline(args.model, args.size, batch=args.batch, family="ppyoloe" if args.workload == "ppyoloe" else "yolo_nas")
        if not args.no_nms and world == 1:
            rec["nms"] = nms_leg(device)
        if not args.no_predict and world == 1 and args.workload == "yolo_nas":
            rec["predict"] = predict_leg(device, args.model, min(args.batch, 32))
        # the other BASELINE.json configurations under the same clock (headline invocation only: S at the default shape on one GPU)
        headline = world == 1 and args.workload == "yolo_nas" and args.model == "s" and args.size == 640 and args.batch == 32
        if args.other_configs == "on" or (args.other_configs == "auto" and headline and not args.no_cpu_baseline):
            # the headline model is done: its network, optimizer state, saved activations and the allocator's cached blocks (56 GB after the S
            # steps) are released before the larger configurations take their 50 - 90 GB each
            import gc

            step = fence = None
            del net, reducer, crit, opt, ema, x, targets, loss, side
            gc.collect()
            torch.cuda.empty_cache()
            rec["other_configs"] = other_configs_leg(device, loss_check=not args.no_cpu_baseline or args.loss_check_only)
        print(json.dumps(rec), flush=True)
    if dist.is_initialized():  # (world > 1, or the one-rank communicator of SGX_DIST_SINGLE_RANK_COLLECTIVES=1)
        dist_barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
