"""Per-shape timing of the three conv kernels over the convolution problems of one real train step.

    python tools/conv_bench.py [--model s] [--batch 32] [--size 640] [--iters 10] [--out gpurun_out/conv_bench.txt]

Records every K.conv2d_fwd / conv2d_bwd_data / conv2d_bwd_weight call of one YOLO-NAS train step (shape, strides,
epilogue options), then replays each distinct problem `iters` times between HIP events and prints, per problem: calls per
step, average microseconds, algorithmic TFLOP/s (2*M*K*C*R*S) and its share of the per-step conv time.  Measurement tool:
uses only the product library.
"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def record_problems(model, batch, size, dev):
    """One real train step; returns OrderedDict problem-key -> calls per step."""
    import torch

    from super_gradients_amd import kernels as K
    from super_gradients_amd.training import models
    from super_gradients_amd.training.losses import PPYoloELoss
    from util import synthetic_targets

    torch.manual_seed(0)
    # (weight gradients one call per layer for the recording: the grouped launches of the product go through conv2d_bwd_weight_group)
    group_env = os.environ.get("SGX_WGRAD_GROUP_GFLOP")
    os.environ["SGX_WGRAD_GROUP_GFLOP"] = "0"
    # model: s / m / l = YOLO-NAS; "resnet50" (BASELINE.json configs[1]: batch x 3 x size x size, cross-entropy); "ppyoloe_s" ...
    name = model if model.startswith(("resnet", "ppyoloe")) else f"yolo_nas_{model}"
    try:
        net = models.get(name, num_classes=1000 if name.startswith("resnet") else 80).materialize(dev).train()
    finally:
        if group_env is None:
            del os.environ["SGX_WGRAD_GROUP_GFLOP"]
        else:
            os.environ["SGX_WGRAD_GROUP_GFLOP"] = group_env
    if name.startswith("resnet"):
        from super_gradients_amd.training.losses import CrossEntropyLoss

        x = torch.randn(batch, 3, size, size, device=dev)
        labels = torch.randint(0, 1000, (batch,), device=dev)
        ce = CrossEntropyLoss()
        t = None
        crit = lambda out, _t: (ce(out, labels), None)
    else:
        x = torch.rand(batch, 3, size, size, device=dev)
        t = synthetic_targets(batch, seed=0, kmax=20, size=size).to(dev)
        crit = PPYoloELoss(80, use_static_assigner=False)
    rec = collections.OrderedDict()
    orig = (K.conv2d_fwd, K.conv2d_bwd_data, K.conv2d_bwd_weight, K.conv2d_fwd_dual, K.conv2d_bwd_data_dual, K.conv2d_bwd_data_wt)

    def key_of(kind, xs, K_, R, stride, pad, xl, yl, extra):
        return (kind,) + tuple(xs) + (K_, R, stride, pad, xl, yl) + extra

    def fwd(x, w, bias=None, addend=None, out=None, act=None, stride=1, pad=0, stat_partials=False):
        y = orig[0](x, w, bias=bias, addend=addend, out=out, act=act, stride=stride, pad=pad, stat_partials=stat_partials)
        yo = y[0] if stat_partials else y
        k = key_of("fwd", x.shape, w.shape[0], w.shape[2], stride, pad, x.stride(2), yo.stride(2), (bias is not None, addend is not None, act, stat_partials))
        rec[k] = rec.get(k, 0) + 1
        return y

    def dgrad(dy, w, x_shape, stride=1, pad=0, addend=None, out=None, accumulate=False):
        o = orig[1](dy, w, x_shape, stride=stride, pad=pad, addend=addend, out=out, accumulate=accumulate)
        k = key_of("dgrad", x_shape, w.shape[0], w.shape[2], stride, pad, o.stride(2), dy.stride(2), (addend is not None, bool(accumulate)))
        rec[k] = rec.get(k, 0) + 1
        return o

    def wgrad(x, dy, dw, dbias=None, stride=1, pad=0):
        orig[2](x, dy, dw, dbias, stride=stride, pad=pad)
        k = key_of("wgrad", x.shape, dw.shape[0], dw.shape[2], stride, pad, x.stride(2), dy.stride(2), (dbias is not None,))
        rec[k] = rec.get(k, 0) + 1

    def fwd2(x, w, w1p, bias1, stride=1):
        r = orig[3](x, w, w1p, bias1, stride=stride)
        k = key_of("fwd2", x.shape, w.shape[0], w.shape[2], stride, w.shape[2] // 2, x.stride(2), r[0].stride(2), ())
        rec[k] = rec.get(k, 0) + 1
        return r

    def dgrad2(dy, w, wt, ds, w1pt, x_shape, stride=1, addend=None, out=None, accumulate=False, addend2=None, addend2_scale=None, reqs=None):
        o = orig[4](dy, w, wt, ds, w1pt, x_shape, stride=stride, addend=addend, out=out, accumulate=accumulate, addend2=addend2, addend2_scale=addend2_scale,
                    reqs=reqs)
        k = key_of("dgrad2", x_shape, w.shape[0], w.shape[2], stride, w.shape[2] // 2, o.stride(2), dy.stride(2), (addend is not None, bool(accumulate)))
        rec[k] = rec.get(k, 0) + 1
        return o

    def dgrad_wt(dy, w, wt, x_shape, stride=1, pad=0, addend=None, out=None, accumulate=False, reqs=None):
        o = orig[5](dy, w, wt, x_shape, stride=stride, pad=pad, addend=addend, out=out, accumulate=accumulate, reqs=reqs)
        k = key_of("dgrad", x_shape, w.shape[0], w.shape[2], stride, pad, o.stride(2), dy.stride(2), (addend is not None, bool(accumulate)))
        rec[k] = rec.get(k, 0) + 1
        return o

    K.conv2d_fwd, K.conv2d_bwd_data, K.conv2d_bwd_weight, K.conv2d_fwd_dual, K.conv2d_bwd_data_dual, K.conv2d_bwd_data_wt = fwd, dgrad, wgrad, fwd2, dgrad2, dgrad_wt
    try:
        loss, _ = crit(net(x), t)
        loss.backward()
    finally:
        K.conv2d_fwd, K.conv2d_bwd_data, K.conv2d_bwd_weight, K.conv2d_fwd_dual, K.conv2d_bwd_data_dual, K.conv2d_bwd_data_wt = orig
    torch.cuda.synchronize()
    del net
    return rec


def _with_planes(fn, filters, dev):
    """Pre-split filter planes for the runner's filters ((ptr, rows, taps, ch) records), kept alive with the callable; the caller opens the
    step scope (kernels.filter_planes_scope) for as long as it replays planes-aware launches."""
    import torch

    from super_gradients_amd import kernels as K

    plan, total = K.filter_planes_plan(filters)
    if plan:
        buf = torch.empty(total, dtype=torch.uint8, device=dev)
        jobs, jdev = K.filter_planes_table(plan, buf)
        K.filter_planes_batch(jobs, jdev)
        fn._planes = (buf, jobs, jdev)
    return fn


def _wt_filters(wt, wtt, stride, pad):
    import ctypes

    from super_gradients_amd import _lib
    from super_gradients_amd import kernels as K

    raw = K.conv2d_transpose_jobs(wt, wtt, stride=stride, pad=pad)
    return [(q.wt, q.C, q.T, q.K) for q in (_lib.WtransJob * (len(raw) // ctypes.sizeof(_lib.WtransJob))).from_buffer_copy(raw)]


def make_runner(k, dev, planes=False):
    """-> (callable replaying the problem once, its algorithmic FLOPs).  planes: the problem's filters get pre-split planes (as the step's
    prefetch makes them) and a data gradient runs from pre-transposed weights, as it does in the step."""
    import torch

    from super_gradients_amd import kernels as K

    def buf(n, h, w, c, ld):
        return torch.randn(n, h, w, ld, device=dev)[..., :c]

    kind, n, h, w, c, K_, R, stride, pad, xl, yl = k[:11]
    extra = k[11:]
    ho, wo = (h + 2 * pad - R) // stride + 1, (w + 2 * pad - R) // stride + 1
    xin = buf(n, h, w, c, xl)
    yout = buf(n, ho, wo, K_, yl)
    wt = K.to_ohwi(torch.randn(K_, c, R, R, device=dev) / (c * R * R) ** 0.5)
    flops = 2.0 * n * ho * wo * K_ * c * R * R
    if kind in ("fwd2", "dgrad2"):  # QARepVGG block: RxS conv + 1x1 conv per launch
        flops += 2.0 * n * ho * wo * K_ * c
        w1 = K.to_ohwi(torch.randn(K_, c, 1, 1, device=dev) / c ** 0.5)
        if kind == "fwd2":
            b1 = torch.randn(K_, device=dev)
            fn = lambda: K.conv2d_fwd_dual(xin, wt, w1, b1, stride=stride)
            return (_with_planes(fn, [(wt.data_ptr(), K_, R * R, c), (w1.data_ptr(), K_, 1, c)], dev) if planes else fn), flops
        has_add, acc = extra
        add = buf(n, h, w, c, xl) if has_add else None
        wtt = K.conv2d_wt_buffer(wt, dev)
        K.conv2d_transpose_weights(wt, wtt, stride=stride, pad=pad)
        w1t = w1.reshape(K_, c).t().contiguous()
        ds = buf(n, ho, wo, K_, yl)
        fn = lambda: K.conv2d_bwd_data_dual(yout, wt, wtt, ds, w1t, (n, h, w, c), stride=stride, addend=add, out=xin, accumulate=acc)
        return (_with_planes(fn, _wt_filters(wt, wtt, stride, pad) + [(w1t.data_ptr(), c, 1, K_)], dev) if planes else fn), flops
    if kind == "fwd":
        has_b, has_add, act, stats = extra
        b = torch.randn(K_, device=dev) if has_b else None
        add = buf(n, ho, wo, K_, yl) if has_add else None
        fn = lambda: K.conv2d_fwd(xin, wt, bias=b, addend=add, out=yout, act=act, stride=stride, pad=pad, stat_partials=stats)
        return (_with_planes(fn, [(wt.data_ptr(), K_, R * R, c)], dev) if planes else fn), flops
    if kind == "dgrad":
        has_add, acc = extra
        add = buf(n, h, w, c, xl) if has_add else None
        if planes:
            wtt = K.conv2d_wt_buffer(wt, dev)
            K.conv2d_transpose_weights(wt, wtt, stride=stride, pad=pad)
            fn = lambda: K.conv2d_bwd_data_wt(yout, wt, wtt, (n, h, w, c), stride=stride, pad=pad, addend=add, out=xin, accumulate=acc)
            return _with_planes(fn, _wt_filters(wt, wtt, stride, pad), dev), flops
        return (lambda: K.conv2d_bwd_data(yout, wt, (n, h, w, c), stride=stride, pad=pad, addend=add, out=xin, accumulate=acc)), flops
    (has_b,) = extra
    dw = torch.zeros_like(wt)
    db = torch.zeros(K_, device=dev) if has_b else None
    return (lambda: K.conv2d_bwd_weight(xin, yout, dw, db, stride=stride, pad=pad)), flops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="s")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    import torch

    os.environ["SGX_SIDE_STREAM"] = "0"  # isolated per-kernel timing
    dev = torch.device("cuda:0")
    rec = record_problems(args.model, args.batch, args.size, dev)
    rows = []
    for k, calls in rec.items():
        kind, n, h, w, c, K_, R, stride, pad, xl, yl = k[:11]
        extra = k[11:]
        fn, flops = make_runner(k, dev)
        fn()
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.iters
        rows.append((kind, n, h, w, c, K_, R, stride, xl, yl, extra, calls, us, flops / us / 1e6))
    tot = sum(r[11] * r[12] for r in rows)
    lines = [f"# YOLO-NAS-{args.model.upper()} bs={args.batch} {args.size}x{args.size}: {len(rows)} distinct conv problems, {sum(r[11] for r in rows)} conv calls/step, "
             f"{tot / 1e3:.2f} ms/step of conv kernels when replayed back to back; total {sum(r[11] * r[13] * r[12] * 1e6 for r in rows) / 1e12:.3f} TFLOP",
             f"{'kind':<6}{'N':>3}{'H':>5}{'W':>5}{'C':>6}{'K':>6}{'R':>3}{'s':>3}{'x_ld':>6}{'y_ld':>6} {'calls':>5}{'us':>10}{'TFLOP/s':>9}{'step%':>7}  options"]
    for r in sorted(rows, key=lambda r: -r[11] * r[12]):
        kind, n, h, w, c, K_, R, stride, xl, yl, extra, calls, us, tf = r
        lines.append(f"{kind:<6}{n:>3}{h:>5}{w:>5}{c:>6}{K_:>6}{R:>3}{stride:>3}{xl:>6}{yl:>6} {calls:>5}{us:>10.1f}{tf:>9.1f}{100 * calls * us / tot:>7.2f}  {extra}")
    for kind in ("fwd", "fwd2", "dgrad", "dgrad2", "wgrad"):
        sel = [r for r in rows if r[0] == kind]
        if not sel:
            continue
        t_us = sum(r[11] * r[12] for r in sel)
        fl = sum(r[11] * r[13] * r[12] * 1e6 for r in sel)
        lines.append(f"# {kind}: {t_us / 1e3:.2f} ms/step, {fl / t_us / 1e6:.1f} TFLOP/s aggregate")
    text = "\n".join(lines)
    print(text)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
