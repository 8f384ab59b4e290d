"""Time of the statistics finalize kernels (the few-workgroup launches between a convolution and the sweep that needs its statistics).

    python tools/finalize_bench.py [--iters 50]
Rows = partial rows the producing kernel left (one per 64- or 128-pixel M tile of a conv, <= 1024 for a sweep).  Measurement tool."""
import argparse
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    import torch

    from super_gradients_amd import kernels as K

    dev = torch.device("cuda:0")

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / args.iters

    def bn(c):
        return types.SimpleNamespace(weight=torch.ones(c, device=dev), bias=torch.zeros(c, device=dev), eps=1e-5, momentum=0.1,
                                     running_mean=torch.zeros(c, device=dev), running_var=torch.ones(c, device=dev))

    print(f"{'rows':>7} {'C':>5} | {'bn_finalize':>12} {'qarep_fwd_finalize':>20}   (us per call, back to back on one stream)")
    for nblk, c in [(25600, 48), (12800, 96), (12800, 32), (6400, 96), (3200, 64), (3200, 192), (1600, 192), (1024, 96), (800, 96), (800, 384), (400, 384), (200, 768), (100, 768)]:
        parts = torch.rand(2, nblk, c, device=dev)
        s5 = torch.rand(5, nblk, c, device=dev)
        b = bn(c)
        b2 = bn(c)
        bias1 = torch.zeros(c, device=dev)
        M = nblk * 64
        t1 = timed(lambda: K.bn_finalize(parts, M, b.weight, b.bias, b.eps, b.momentum, b.running_mean, b.running_var))
        t2 = timed(lambda: K.qarep_fwd_finalize(s5, M, bias1, b, b2))
        print(f"{nblk:>7} {c:>5} | {t1:>12.1f} {t2:>20.1f}")


if __name__ == "__main__":
    main()
