"""Data-parallel path with world_size 2 over gloo on CPU (the kernels run on the host emulation in each process):
  * GradientAllReducer: bucketed all-reduce of the gradient arena == sum of the per-rank gradients, parameters broadcast from rank 0,
    the mean folded into the optimizer (grad_scale) -> identical parameters on both ranks;
  * PPYoloELoss: the four sums exchanged as ONE collective, normaliser = clip(sum of assigned scores / world, 1)
    (reference: training/losses/ppyolo_loss.py:971-979), local gradients scaled by the global normaliser.
On the GPU box the same code runs over RCCL (backend "nccl"); bench.py --gpus N is the multi-GPU measurement."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, outdir):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(HERE, "emu"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    import emu_env

    emu_env.activate()
    import torch.distributed as dist

    from super_gradients_amd import kernels as K
    from super_gradients_amd.training.losses import PPYoloELoss
    from super_gradients_amd.training.utils import distributed_training_utils as DU
    from super_gradients_amd.training.utils.distributed_training_utils import GradientAllReducer, setup_device_from_env
    from super_gradients_amd.training.utils.optimizers import ArenaSGD
    from test_trainer import _tiny_models
    from oracle import golden_util as G
    from oracle.yolo_nas import make_anchors

    r, w, dev = setup_device_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dist.is_initialized()
    out = {}
    # ---- gradient all-reduce --------------------------------------------------------------------------------------
    torch.manual_seed(100 + rank)  # different initial weights per rank: broadcast must equalise them
    _, net = _tiny_models(dev)
    for p in net.parameters():
        p.data.add_(torch.randn_like(p) * 0.01 * (rank + 1))
    net.materialize(dev).train()
    g = torch.Generator().manual_seed(10 + rank)
    x, y = torch.randn(4, 4, 8, 8, generator=g), torch.randint(0, 4, (4,), generator=g)
    from super_gradients_amd.training.losses import CrossEntropyLoss

    crit = CrossEntropyLoss()
    crit(net(x), y).backward()  # no reducer yet: purely local gradients (of rank-local weights)
    reducer = GradientAllReducer(net, net.gradient_buckets())
    reducer.broadcast_parameters(0)
    out["params_after_broadcast"] = net.p_arena.buf.clone()
    net.zero_grad()
    crit(net(x), y).backward()
    out["reduced"] = net.g_arena.buf.clone()
    # local gradient at the broadcast weights, computed without communication
    net._grad_ready, net._post_backward_hook = None, None
    net.zero_grad()
    crit(net(x), y).backward()
    out["local"] = net.g_arena.buf.clone()
    net.g_arena.buf.copy_(out["reduced"])
    opt = ArenaSGD(net, lr=0.1, momentum=0.0)
    K.sgd_step  # noqa: B018  (the arena SGD has no grad_scale argument: scale the arena like Trainer does)
    net.g_arena.buf.mul_(1.0 / world)
    opt.step()
    out["params_after_step"] = net.p_arena.buf.clone()
    # ---- gradient accumulation under data parallelism (Trainer batch_accumulate = 2): the first micro-batch is NOT exchanged, the
    # second one's backward exchanges the locally accumulated arena once -> sum over ranks of (g1 + g2), not world * g1 + g2
    net._grad_ready, net._post_backward_hook = reducer.ready, reducer.finish
    x2, y2 = torch.randn(4, 4, 8, 8, generator=g), torch.randint(0, 4, (4,), generator=g)
    net.p_arena.buf.copy_(out["params_after_broadcast"])
    net.zero_grad()
    reducer.sync = False
    crit(net(x), y).backward()
    out["accum_after_first"] = net.g_arena.buf.clone()
    reducer.sync = True
    # the second micro-batch takes the collective from a (stand-in) side stream: the branch the single-GPU tests cannot reach
    class _Stream:
        calls = []

        def wait_stream(self, other):
            self.calls.append(("wait", other))

    import contextlib

    @contextlib.contextmanager
    def on_stream(s_):
        _Stream.calls.append(("enter", s_))
        yield

    fake = _Stream()
    reducer.from_side = True
    reducer._current_stream, reducer._on_stream, reducer._side_stream = (lambda: "main"), on_stream, (lambda: fake)
    crit(net(x2), y2).backward()
    out["accum_reduced"] = net.g_arena.buf.clone()
    out["side_calls"] = [c[0] for c in _Stream.calls]
    net._grad_ready, net._post_backward_hook = None, None
    net.zero_grad()
    crit(net(x), y).backward()
    crit(net(x2), y2).backward()
    out["accum_local"] = net.g_arena.buf.clone()
    # ---- loss sums exchanged as one collective -----------------------------------------------------------------------
    def anchors(hw, strides):
        a, pts, _pg, counts, strd = make_anchors(hw, strides)
        return a, pts, counts, strd

    preds = G.synthetic_predictions(2, [8, 4, 3], 8, 16, seed=30 + rank, make_anchors=anchors)
    t = G.detection_targets(2, 64, seed=40 + rank, kmax=3, num_classes=8, empty_last=False)
    logits = preds[0].clone().requires_grad_(True)
    distri = preds[1].clone().requires_grad_(True)
    loss, items = PPYoloELoss(8, use_static_assigner=False)((None, (logits, distri) + tuple(preds[2:])), t)
    loss.backward()
    out["items"], out["g_logits"] = items.clone(), logits.grad.clone()
    loc = K.ppyoloe_loss_fwd(preds[0], preds[1], preds[2], preds[3], preds[5], t, preds[4], False, True, (1.0, 2.5, 0.5))
    out["local_sums"], out["local_g_logits"] = loc["sums"].clone(), loc["g_logits"].clone()
    # ---- synchronised BatchNorm: two ranks x half a batch == one process x the whole batch ----------------------------------
    ref, snet = _tiny_models(dev)  # identical weights on both ranks (seeded inside)
    snet.materialize(dev).train()
    snet.set_sync_bn(True)
    sred = GradientAllReducer(snet, snet.gradient_buckets())
    gfull = torch.Generator().manual_seed(77)
    xfull, wfull = torch.randn(8, 4, 8, 8, generator=gfull), torch.randn(8, 6, generator=gfull)
    xs, ws_ = xfull[rank * 4:(rank + 1) * 4], wfull[rank * 4:(rank + 1) * 4]
    snet.zero_grad()
    ys = snet(xs)
    (ys * ws_).sum().backward()
    out["sync_y"], out["sync_grad"] = ys.detach().clone(), snet.g_arena.buf.clone()
    out["sync_running"] = snet.b_arena.buf.clone()
    # ---- DDP(broadcast_buffers=True): rank 0's BatchNorm statistics overwrite everyone's before a forward ----------------------------
    net.b_arena.buf.add_(float(rank + 1))
    net.i_arena.add_(rank + 5)
    out["buffers_before"] = (net.b_arena.buf.clone(), net.i_arena.clone())
    reducer.broadcast_buffers(0)
    out["buffers_after"] = (net.b_arena.buf.clone(), net.i_arena.clone())
    # ---- validation meters: a rank that saw no batch still takes part in the exchange (and learns the item count from the others) ---------
    from super_gradients_amd.training.sg_trainer.sg_trainer import AverageMeter

    meter = AverageMeter()
    if rank == 0:
        meter.update(torch.tensor([1.0, 2.0, 3.0]), 4)
        meter.update(torch.tensor([3.0, 2.0, 1.0]), 4)
    meter.all_reduce(device=dev)
    out["meter"] = (meter.average, meter.n)
    if rank == 0:
        ref.train()
        yf = ref(xfull)
        (yf * wfull).sum().backward()
        out["full_y"] = yf.detach().clone()
        out["full_grads"] = {k: p.grad.clone() for k, p in ref.named_parameters()}
        out["full_buffers"] = {k: b.clone() for k, b in ref.named_buffers()}
        out["slot_layout"] = [(s.name, s.start, s.numel, tuple(s.param.shape)) for s in snet.slots]
        out["buffer_state"] = {k: v.detach().clone() for k, v in snet.state_dict().items() if "running" in k}
    torch.save(out, os.path.join(outdir, f"rank{rank}.pt"))
    DU.barrier()  # the helper bench.py fences with (names the GPU for RCCL, plain barrier for gloo)
    dist.destroy_process_group()


def test_world_size_2_gloo(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), f"rank{i}.pt")) for i in range(world)]
    # broadcast: both ranks hold rank 0's parameters
    assert torch.equal(r[0]["params_after_broadcast"], r[1]["params_after_broadcast"])
    # all-reduce: every rank's arena == sum of the local gradients
    total = r[0]["local"] + r[1]["local"]
    for i in range(world):
        assert torch.allclose(r[i]["reduced"], total, rtol=1e-6, atol=1e-7)
    assert torch.equal(r[0]["params_after_step"], r[1]["params_after_step"])
    expect = r[0]["params_after_broadcast"] - 0.1 * total / world
    assert torch.allclose(r[0]["params_after_step"], expect, rtol=1e-6, atol=1e-7)
    # gradient accumulation: nothing exchanged after the first micro-batch, one exchange of the accumulated arena after the second, issued
    # through the side-stream branch of GradientAllReducer.ready (stream waits + collective under the side stream's context, per bucket)
    for i in range(world):
        assert not torch.allclose(r[i]["accum_after_first"], r[1 - i]["accum_after_first"]), "first micro-batch must stay local"
        assert torch.allclose(r[i]["accum_reduced"], r[0]["accum_local"] + r[1]["accum_local"], rtol=1e-5, atol=1e-7), "accumulated arena exchanged once"
        assert r[i]["side_calls"] and r[i]["side_calls"].count("wait") == r[i]["side_calls"].count("enter") >= 1
    # loss: items from the global sums, normaliser = clip(score_sum / world, 1); gradients = local gradient of the weighted sums / normaliser
    s = r[0]["local_sums"] + r[1]["local_sums"]
    norm = max(float(s[3]) / world, 1.0)
    items = torch.tensor([1.0 * float(s[0]) / norm, 2.5 * float(s[1]) / norm, 0.5 * float(s[2]) / norm])
    for i in range(world):
        assert torch.allclose(r[i]["items"][:3], items, rtol=2e-5), (r[i]["items"], items)
        assert abs(float(r[i]["items"][3]) - float(items.sum())) <= 2e-5 * float(items.sum())
        assert torch.allclose(r[i]["g_logits"], r[i]["local_g_logits"] / norm, rtol=2e-5, atol=1e-8)
    # synchronised BatchNorm: outputs equal the full-batch run's halves; all-reduced gradients equal the full-batch gradients; the running
    # statistics are the full-batch ones on both ranks
    for i in range(world):
        assert torch.allclose(r[i]["sync_y"], r[0]["full_y"][i * 4:(i + 1) * 4], rtol=1e-4, atol=1e-5), f"rank {i} sync-BN forward"
    assert torch.allclose(r[0]["sync_grad"], r[1]["sync_grad"], rtol=1e-6, atol=1e-7)
    for name, start, numel, shape in r[0]["slot_layout"]:
        g = r[0]["sync_grad"][start:start + numel]
        ref = r[0]["full_grads"][name]
        if len(shape) == 4:  # conv weight: arena holds OHWI with C padded to 4
            K_, C_, R_, S_ = shape
            cp = (C_ + 3) // 4 * 4
            g = g.view(K_, R_, S_, cp)[..., :C_].permute(0, 3, 1, 2)
        else:
            g = g.view(shape)
        e = float((g - ref).abs().max()) / max(float(ref.abs().max()), 1e-3)
        assert e <= 2e-4, f"sync-BN gradient of {name}: {e:.2e}"
    for k, v in r[0]["buffer_state"].items():
        assert torch.allclose(v, r[0]["full_buffers"][k], rtol=1e-5, atol=1e-6), k
    assert torch.equal(r[0]["sync_running"], r[1]["sync_running"])
    # DDP(broadcast_buffers=True): after the broadcast every rank holds what rank 0 held before it (float statistics and step counters)
    assert not torch.equal(r[0]["buffers_before"][0], r[1]["buffers_before"][0])
    for i in range(world):
        assert torch.equal(r[i]["buffers_after"][0], r[0]["buffers_before"][0]) and torch.equal(r[i]["buffers_after"][1], r[0]["buffers_before"][1])
        # both ranks report rank 0's two batches (rank 1 had none): mean over 8 samples
        assert r[i]["meter"][1] == 8 and r[i]["meter"][0] == pytest.approx((2.0, 2.0, 2.0))


def _worker_rccl(rank, world, port, outdir):
    """The same exchange over RCCL on two MI355X: real HIP streams, the default side-stream issue of the bucket collectives, the fused
    16-byte loss reduction, synchronised BatchNorm.  One process per GPU."""
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    from super_gradients_amd import kernels as K
    from super_gradients_amd.training.losses import CrossEntropyLoss, PPYoloELoss
    from super_gradients_amd.training.utils import distributed_training_utils as DU
    from super_gradients_amd.training.utils.distributed_training_utils import GradientAllReducer, setup_device_from_env
    from test_trainer import _tiny_models
    from oracle import golden_util as G
    from oracle.yolo_nas import make_anchors

    r, w, dev = setup_device_from_env()  # backend "nccl" = RCCL
    assert (r, w) == (rank, world) and dev.type == "cuda"
    out = {}
    ref, net = _tiny_models(dev)
    net.materialize(dev).train()
    net.p_arena.buf.add_(0.01 * (rank + 1))           # ranks start apart: the broadcast must equalise them
    reducer = GradientAllReducer(net, net.gradient_buckets())
    assert reducer.from_side and net.side_stream is not None
    reducer.broadcast_parameters(0)
    out["params"] = net.p_arena.buf.cpu().clone()
    g = torch.Generator().manual_seed(10 + rank)
    x, y = torch.randn(4, 4, 8, 8, generator=g).to(dev), torch.randint(0, 4, (4,), generator=g).to(dev)
    crit = CrossEntropyLoss()
    net.zero_grad()
    crit(net(x), y).backward()
    torch.cuda.synchronize()
    out["reduced"] = net.g_arena.buf.cpu().clone()
    net._grad_ready, net._post_backward_hook = None, None
    net.zero_grad()
    crit(net(x), y).backward()
    torch.cuda.synchronize()
    out["local"] = net.g_arena.buf.cpu().clone()

    def anchors(hw, strides):
        a, pts, _pg, counts, strd = make_anchors(hw, strides)
        return a, pts, counts, strd

    preds = G.synthetic_predictions(2, [8, 4, 3], 8, 16, seed=30 + rank, make_anchors=anchors)
    t = G.detection_targets(2, 64, seed=40 + rank, kmax=3, num_classes=8, empty_last=False)
    dp = [p.to(dev) if torch.is_tensor(p) else p for p in preds]
    loss, items = PPYoloELoss(8, use_static_assigner=False)((None, tuple(dp)), t.to(dev))
    out["items"] = items.cpu().clone()
    loc = K.ppyoloe_loss_fwd(dp[0], dp[1], dp[2], dp[3], dp[5], t.to(dev), dp[4], False, True, (1.0, 2.5, 0.5))
    out["local_sums"] = loc["sums"].cpu().clone()
    # synchronised BatchNorm: two ranks x half a batch == the whole batch
    _, snet = _tiny_models(dev)
    snet.materialize(dev).train()
    snet.set_sync_bn(True)
    GradientAllReducer(snet, snet.gradient_buckets())
    gfull = torch.Generator().manual_seed(77)
    xfull, wfull = torch.randn(8, 4, 8, 8, generator=gfull), torch.randn(8, 6, generator=gfull)
    ys = snet(xfull[rank * 4:(rank + 1) * 4].to(dev))
    (ys * wfull[rank * 4:(rank + 1) * 4].to(dev)).sum().backward()
    torch.cuda.synchronize()
    out["sync_y"], out["sync_grad"] = ys.detach().cpu().clone(), snet.g_arena.buf.cpu().clone()
    if rank == 0:
        ref.train()
        yf = ref(xfull)
        out["full_y"] = yf.detach().clone()
    torch.save(out, os.path.join(outdir, f"rccl{rank}.pt"))
    DU.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_world_size_2_rccl(tmp_path):
    """N > 1 on real hardware (reference: sg_trainer.py:452-459 DDP wrapping, ppyolo_loss.py:971-977).  Needs two GPUs: skipped on the
    single-GPU test box, runs wherever `bench.py --gpus N` can."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 HIP GPUs")
    world = 2
    port = 29700 + (os.getpid() % 2000)
    mp.spawn(_worker_rccl, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), f"rccl{i}.pt")) for i in range(world)]
    assert torch.equal(r[0]["params"], r[1]["params"])
    total = r[0]["local"] + r[1]["local"]
    for i in range(world):
        assert torch.allclose(r[i]["reduced"], total, rtol=1e-5, atol=1e-7)
    s = r[0]["local_sums"] + r[1]["local_sums"]
    norm = max(float(s[3]) / world, 1.0)
    items = torch.tensor([1.0 * float(s[0]) / norm, 2.5 * float(s[1]) / norm, 0.5 * float(s[2]) / norm])
    for i in range(world):
        assert torch.allclose(r[i]["items"][:3], items, rtol=2e-5)
        assert torch.allclose(r[i]["sync_y"], r[0]["full_y"][i * 4:(i + 1) * 4], rtol=1e-4, atol=1e-5)
    assert torch.allclose(r[0]["sync_grad"], r[1]["sync_grad"], rtol=1e-6, atol=1e-7)


def _worker_rccl_single(rank, world, port, outdir):
    """ONE rank, a real RCCL communicator (SGX_DIST_SINGLE_RANK_COLLECTIVES=1 makes a one-rank group issue its collectives): what the
    single-GPU test box can run of the data-parallel path - communicator creation, the parameter / buffer broadcasts, the bucket all-reduces
    issued from the side stream while backward runs on the main one, the wait at the end of backward, the loss's 16-byte all-reduce,
    synchronised BatchNorm's statistics exchange - with every collective an identity."""
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0",
                      SGX_DIST_SINGLE_RANK_COLLECTIVES="1")
    import torch.distributed as dist

    from super_gradients_amd import kernels as K
    from super_gradients_amd.training.losses import CrossEntropyLoss, PPYoloELoss
    from super_gradients_amd.training.utils import distributed_training_utils as DU
    from super_gradients_amd.training.utils.distributed_training_utils import GradientAllReducer, setup_device_from_env
    from test_trainer import _tiny_models
    from oracle import golden_util as G
    from oracle.yolo_nas import make_anchors

    r, w, dev = setup_device_from_env()
    assert (r, w) == (0, 1) and dev.type == "cuda" and dist.is_initialized() and dist.get_backend() == "nccl" and DU.collectives_active()
    out = {}
    ref, net = _tiny_models(dev)
    net.materialize(dev).train()
    g = torch.Generator().manual_seed(10)
    x, y = torch.randn(4, 4, 8, 8, generator=g).to(dev), torch.randint(0, 4, (4,), generator=g).to(dev)
    crit = CrossEntropyLoss()
    net.zero_grad()
    crit(net(x), y).backward()
    torch.cuda.synchronize()
    out["local"] = net.g_arena.buf.cpu().clone()
    reducer = GradientAllReducer(net, net.gradient_buckets())
    assert reducer.from_side and net.side_stream is not None
    p0 = net.p_arena.buf.cpu().clone()
    reducer.broadcast_parameters(0)
    reducer.broadcast_buffers(0)
    out["params_kept"] = bool(torch.equal(net.p_arena.buf.cpu(), p0))
    issued = []
    orig = dist.all_reduce
    dist.all_reduce = lambda t, *a, **k: (issued.append((t.numel(), torch.cuda.current_stream().cuda_stream)), orig(t, *a, **k))[1]
    try:
        net.zero_grad()
        crit(net(x), y).backward()
    finally:
        dist.all_reduce = orig
    torch.cuda.synchronize()
    out["reduced"] = net.g_arena.buf.cpu().clone()
    out["collectives"] = len(issued)
    out["from_side_stream"] = all(s == net.side_stream.cuda_stream for _, s in issued)
    out["covered"] = sum(n for n, _ in issued) == net.g_arena.size

    def anchors(hw, strides):
        a, pts, _pg, counts, strd = make_anchors(hw, strides)
        return a, pts, counts, strd

    preds = G.synthetic_predictions(2, [8, 4, 3], 8, 16, seed=30, make_anchors=anchors)
    t = G.detection_targets(2, 64, seed=40, kmax=3, num_classes=8, empty_last=False)
    dp = [p.to(dev) if torch.is_tensor(p) else p for p in preds]
    _, items = PPYoloELoss(8, use_static_assigner=False)((None, tuple(dp)), t.to(dev))
    out["items"] = items.cpu().clone()
    os.environ["SGX_DIST_SINGLE_RANK_COLLECTIVES"] = "0"
    _, items_local = PPYoloELoss(8, use_static_assigner=False)((None, tuple(dp)), t.to(dev))
    out["items_local"] = items_local.cpu().clone()
    os.environ["SGX_DIST_SINGLE_RANK_COLLECTIVES"] = "1"
    # synchronised BatchNorm over one rank == the plain BatchNorm of the whole batch
    _, snet = _tiny_models(dev)
    snet.materialize(dev).train()
    snet.set_sync_bn(True)
    GradientAllReducer(snet, snet.gradient_buckets())
    gfull = torch.Generator().manual_seed(77)
    xfull, wfull = torch.randn(8, 4, 8, 8, generator=gfull), torch.randn(8, 6, generator=gfull)
    ys = snet(xfull.to(dev))
    (ys * wfull.to(dev)).sum().backward()
    torch.cuda.synchronize()
    out["sync_y"], out["sync_grad"] = ys.detach().cpu().clone(), snet.g_arena.buf.cpu().clone()
    _, pnet = _tiny_models(dev)
    pnet.materialize(dev).train()
    yp = pnet(xfull.to(dev))
    (yp * wfull.to(dev)).sum().backward()
    torch.cuda.synchronize()
    out["plain_y"], out["plain_grad"] = yp.detach().cpu().clone(), pnet.g_arena.buf.cpu().clone()
    torch.save(out, os.path.join(outdir, "rccl_single.pt"))
    DU.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_single_rank_communicator(tmp_path):
    """Rows a18 / e / f4 on the hardware a one-GPU box has: the data-parallel choreography over a REAL RCCL communicator of one rank (round
    6; rounds 1 - 5 had only the gloo world-2 tests on the host emulation, and `test_world_size_2_rccl` skips without a second GPU).  Every
    collective is an identity, so: the reduced gradient arena equals the local one bit for bit, the bucket all-reduces cover the arena and
    were all issued from the side stream, the parameter / buffer broadcasts leave the arenas alone, the loss items equal the
    non-distributed ones, synchronised BatchNorm equals plain BatchNorm.  Reference: sg_trainer.py:452-459, ppyolo_loss.py:971-977,
    sg_trainer.py:1344-1350."""
    port = 29900 + (os.getpid() % 2000)
    mp.spawn(_worker_rccl_single, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    r = torch.load(os.path.join(str(tmp_path), "rccl_single.pt"))
    assert r["params_kept"]
    assert r["collectives"] >= 2 and r["covered"] and r["from_side_stream"], r
    assert torch.equal(r["reduced"], r["local"])
    assert torch.allclose(r["items"], r["items_local"], rtol=1e-6, atol=0)
    assert torch.allclose(r["sync_y"], r["plain_y"], rtol=1e-5, atol=1e-6)
    assert torch.allclose(r["sync_grad"], r["plain_grad"], rtol=1e-4, atol=1e-6)
