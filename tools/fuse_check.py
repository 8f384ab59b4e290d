"""Whole-model A/B of a switch on ONE batch: every parameter gradient with the switch on (twice: run-to-run determinism) and off.

    python tools/fuse_check.py [--model s] [--batch 2] [--size 256] [--env SGX_FUSE_BN_REDUCE] [--attr fuse_bn_reduce]

Prints the largest relative differences per parameter (max |a - b| / max |b|).  Measurement tool: product library only."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="s")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--attr", default="fuse_bn_reduce")
    ap.add_argument("--wgrad-math", type=int, default=None, help="second arm: sgx_conv_set_wgrad_math(mode) instead of the attribute switch")
    args = ap.parse_args()
    import torch

    from super_gradients_amd import kernels as K
    from super_gradients_amd._lib import lib
    from super_gradients_amd.training import models

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = models.get(f"yolo_nas_{args.model}", num_classes=80).materialize(dev).train()
    g = torch.Generator().manual_seed(7)
    x = torch.rand(args.batch, 3, args.size, args.size, generator=g).to(dev)

    def run(on):
        if args.wgrad_math is None:
            setattr(net, args.attr, on)
        else:
            lib().sgx_conv_set_wgrad_math(2 if on else args.wgrad_math)
        net.zero_grad()
        K.BN_REQ_STATS["taken"] = K.BN_REQ_STATS["declined"] = 0
        out = net(x)
        lg, ds = out[1][0], out[1][1]
        gg = torch.Generator().manual_seed(21)
        up_l, up_d = torch.randn(lg.shape, generator=gg).to(dev), torch.randn(ds.shape, generator=gg).to(dev)
        torch.autograd.backward([lg, ds], [up_l, up_d])
        torch.cuda.synchronize()
        return {n: p.grad.detach().cpu().clone() for n, p in net.named_parameters() if p.grad is not None}, dict(K.BN_REQ_STATS)

    a, sa = run(True)
    b, _ = run(True)
    c, sc = run(False)
    print("requests (on):", sa, " (off):", sc)

    def report(u, v, what):
        rows = []
        for n in u:
            den = float(v[n].abs().max())
            rows.append((float((u[n] - v[n]).abs().max()) / max(den, 1e-30), n, den))
        rows.sort(reverse=True)
        print(f"--- {what}: largest relative differences")
        for e, n, den in rows[:12]:
            print(f"{e:10.3e}  {n}  (max |g| {den:.3e})")

    report(a, b, "on vs on (determinism)")
    report(a, c, "on vs off")


if __name__ == "__main__":
    main()
