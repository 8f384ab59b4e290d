"""Weight-gradient lab: replay the grouped weight-gradient launches of one real train step, alone on the chip, under several settings of
the loop / the grouping - interleaved in ONE process (rounds x configs, median).

    python tools/wgrad_lab.py [--model s] [--batch 32] [--size 640] [--configs base,slab32,ab1,ab3,ab7,ab8,ab15] [--rounds 3] [--iters 5]

A config is a '+'-joined list of:  base (the product default: bf16x3 arithmetic, patch kernel on the 3x3 problems) | nopatch (bf16x3 slab loop
everywhere) | fp32 (the fp32 slab loop everywhere) | slab32 | pf1 (one slab of loads in flight instead of two) | w2 (64x64 tile on two
waves) | bf16 (force the bf16x3 slab loop) | abN (ablation bits, needs a library built with -DSGX_WGRAD_LAB: 1 no global loads,
2 no LDS stores, 4 no MFMAs, 8 no fold / dW) | gR.I.X (sgx_debug_set_wgrad_group rounds.item_mflop.xcd) | tBxJ (tile override of the slab
loop) | pI.K.F (sgx_debug_set_wgrad_patch: largest item MFLOP . filter blocks . least fill percent).
Per group (= one K.conv2d_bwd_weight_group call of the step): jobs, GFLOP, then microseconds per config; last line: ms per step and
algorithmic TFLOP/s.  Measurement tool: product library only.
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def record_groups(model, batch, size, dev):
    """One real train step -> [[(x shape, x pixel stride, dy shape, dy pixel stride, dw shape, stride, pad), ...], ...] in launch order"""
    import torch

    from super_gradients_amd import kernels as K
    from super_gradients_amd.training import models
    from super_gradients_amd.training.losses import PPYoloELoss
    from util import synthetic_targets

    torch.manual_seed(0)
    net = models.get(f"yolo_nas_{model}", num_classes=80).materialize(dev).train()
    x = torch.rand(batch, 3, size, size, device=dev)
    t = synthetic_targets(batch, seed=0, kmax=20, size=size).to(dev)
    crit = PPYoloELoss(80, use_static_assigner=False)
    groups = []
    orig = K.conv2d_bwd_weight_group

    def rec(entries):
        groups.append([(tuple(xx.shape), xx.stride(2), tuple(dy.shape), dy.stride(2), tuple(dw.shape), st, pd) for xx, dy, dw, st, pd in entries])
        return orig(entries)

    K.conv2d_bwd_weight_group = rec
    try:
        loss, _ = crit(net(x), t)
        loss.backward()
    finally:
        K.conv2d_bwd_weight_group = orig
    torch.cuda.synchronize()
    del net
    return groups


def apply_config(cfg, lib):
    lib.sgx_debug_set_wgrad_group(0, 0, 1)
    lib.sgx_debug_set_tiles(0, 0, 0, 0, 0)
    lib.sgx_debug_set_wgrad_patch(0, 0, 0)
    deep = ab = 0
    for part in cfg.split("+"):
        if part == "base":
            pass
        elif part == "nopatch":
            deep |= 16
        elif part == "fp32":
            deep |= 16 + 32
        elif part == "slab32":
            deep |= 1
        elif part == "pf1":
            deep |= 2
        elif part == "w2":
            deep |= 4
        elif part == "bf16":
            deep |= 8
        elif part.startswith("ab"):
            ab = int(part[2:])
        elif part.startswith("g"):
            r, i, xo = (int(v) for v in part[1:].split("."))
            lib.sgx_debug_set_wgrad_group(r, i, xo)
        elif part.startswith("p"):
            lib.sgx_debug_set_wgrad_patch(*[int(v) for v in part[1:].split(".")])
        elif part.startswith("t"):
            b, j = (int(v) for v in part[1:].split("x"))
            lib.sgx_debug_set_tiles(0, 0, b, j, 0)
        else:
            raise SystemExit(f"unknown config part {part!r}")
    lib.sgx_debug_set_wgrad_loop(deep, ab)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="s")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--configs", default="base,slab32")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch

    from super_gradients_amd import kernels as K
    from super_gradients_amd._lib import lib

    dev = torch.device("cuda:0")
    groups = record_groups(args.model, args.batch, args.size, dev)
    configs = args.configs.split(",")

    def buf(shape, ld):
        n, h, w, c = shape
        return torch.randn(n, h, w, ld, device=dev)[..., :c]

    lines = [f"# YOLO-NAS-{args.model.upper()} bs={args.batch} {args.size}x{args.size}: {len(groups)} grouped weight-gradient calls per step, replayed alone "
             f"(median of {args.rounds} rounds x {args.iters} launches); columns: microseconds per call",
             f"{'group':>5} {'jobs':>4} {'GFLOP':>8} {'largest job (N H W C K R s)':<34}" + "".join(f"{c:>16}" for c in configs)]
    tot = {c: 0.0 for c in configs}
    tot_flops = 0.0
    for gi, g in enumerate(groups):
        entries, flops, big = [], 0.0, (0.0, None)
        for xs, xl, ys, yl, ws, st, pd in g:
            k_, c_, r_, s_ = ws
            dw = torch.zeros_like(K.to_ohwi(torch.empty(k_, c_, r_, s_, device=dev)))
            entries.append((buf(xs, xl), buf(ys, yl), dw, st, pd))
            f = 2.0 * ys[0] * ys[1] * ys[2] * k_ * c_ * r_ * s_
            flops += f
            if f > big[0]:
                big = (f, f"{xs[0]} {xs[1]} {xs[2]} {xs[3]} {k_} {r_} {st}")
        times = {c: [] for c in configs}
        for _ in range(args.rounds):
            for c in configs:
                apply_config(c, lib())
                K.conv2d_bwd_weight_group(entries)  # warm (workspace growth, ticket buffer)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    K.conv2d_bwd_weight_group(entries)
                e1.record()
                torch.cuda.synchronize()
                times[c].append(e0.elapsed_time(e1) * 1e3 / args.iters)
        med = {c: statistics.median(times[c]) for c in configs}
        for c in configs:
            tot[c] += med[c]
        tot_flops += flops
        lines.append(f"{gi:>5} {len(g):>4} {flops / 1e9:>8.1f} {big[1]:<34}" + "".join(f"{med[c]:>16.1f}" for c in configs))
        del entries
    lines.append(f"{'total ms/step':<54}" + "".join(f"{tot[c] / 1e3:>16.2f}" for c in configs))
    lines.append(f"{'algorithmic TFLOP/s':<54}" + "".join(f"{tot_flops / tot[c] / 1e6:>16.1f}" for c in configs))
    apply_config("base", lib())
    text = "\n".join(lines)
    print(text)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
