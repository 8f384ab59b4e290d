"""Achieved HBM bandwidth of the element sweeps (BatchNorm apply / backward, QARepVGG sweeps, axpy) on representative YOLO-NAS-S maps.

    python tools/sweep_bench.py [--iters 20]
Bytes counted per element: the tensors a sweep must read and write once (4 B each).  Measurement tool: product library only."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    import torch

    from super_gradients_amd import kernels as K

    dev = torch.device("cuda:0")
    shapes = [(32, 320, 320, 48), (32, 160, 160, 96), (32, 160, 160, 32), (32, 80, 80, 192), (32, 80, 80, 64), (32, 40, 40, 384), (32, 40, 40, 96), (32, 20, 20, 768)]

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

    print(f"{'shape':<22}{'MB/tensor':>10} | " + " | ".join(f"{n:>24}" for n in ("affine_act (r+w)", "bn_bwd (2r, 2r+w)", "dual_affine (2r+w)", "qarep_bwd (3r, 3r+2w)", "axpy acc (2r+w)")) + "   us, TB/s")
    for n, h, w, c in shapes:
        x = torch.randn(n, h, w, c, device=dev)
        y = torch.empty_like(x)
        u = torch.randn_like(x)
        dy = torch.randn_like(x)
        sc, sh = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
        mean, inv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        gam = torch.ones(c, device=dev)
        dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
        nb = x.numel() * 4
        cols = []
        t = timed(lambda: K.affine_act(x, sc, sh, act="relu", out=y))
        cols.append((t, 2 * nb))
        t = timed(lambda: K.bn_bwd(dy, x, sc, sh, gam, mean, inv, dg, db, act="relu", dx_out=y))
        cols.append((t, 5 * nb))
        t = timed(lambda: K.dual_affine_act(x, sc, sh, u, sc, sh, act="relu", out=y))
        cols.append((t, 3 * nb))
        cols.append((float("nan"), 0))
        t = timed(lambda: K.axpy(x, out=y, accumulate=True))
        cols.append((t, 3 * nb))
        print(f"{str((n, h, w, c)):<22}{nb / 1e6:>10.1f} | " + " | ".join(f"{t:>14.1f} {b / t / 1e6 if t == t else 0:>9.2f}" for t, b in cols))


if __name__ == "__main__":
    main()
