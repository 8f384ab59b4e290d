"""Dump every parameter gradient (and the forward outputs) of one seeded YOLO-NAS train-step backward to a .pt file - run it from two
checkouts / under two settings and compare with --compare.  Measurement tool: product library only.

    python tools/grad_dump.py out.pt [--model s] [--batch 2] [--size 256] [--seed 0]
    python tools/grad_dump.py --compare a.pt b.pt"""
import os
import sys

ROOT = os.getcwd()
sys.path.insert(0, ROOT)


def main():
    import torch

    if sys.argv[1] == "--compare":
        a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
        for key in ("out", "grad"):
            rows = []
            for n in a[key]:
                den = float(b[key][n].abs().max())
                rows.append((float((a[key][n] - b[key][n]).abs().max()) / max(den, 1e-30), n, den))
            rows.sort(reverse=True)
            print(f"--- {key}: largest relative differences (of {len(rows)})")
            for e, n, den in rows[:14]:
                print(f"{e:10.3e}  {n}  (max {den:.3e})")
        return
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--model", default="s")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    from super_gradients_amd.training import models

    dev = torch.device("cuda:0")
    torch.manual_seed(args.seed)
    net = models.get(f"yolo_nas_{args.model}", num_classes=80).materialize(dev).train()
    g = torch.Generator().manual_seed(7)
    x = torch.rand(args.batch, 3, args.size, args.size, generator=g).to(dev)
    net.zero_grad()
    out = net(x)
    lg, ds = out[1][0], out[1][1]
    gg = torch.Generator().manual_seed(21)
    up_l, up_d = torch.randn(lg.shape, generator=gg).to(dev), torch.randn(ds.shape, generator=gg).to(dev)
    torch.autograd.backward([lg, ds], [up_l, up_d])
    torch.cuda.synchronize()
    torch.save({"out": {"logits": lg.detach().cpu(), "distri": ds.detach().cpu()},
                "grad": {n: p.grad.detach().cpu().clone() for n, p in net.named_parameters() if p.grad is not None}}, args.out)


if __name__ == "__main__":
    main()
