"""One-off diagnosis (r3): where does the gradient error of neck.neck1 at 32 x 640^2 (flip-free form) enter?
Captures, on the HIP model, the operands of neck2.conv's backward (dy in, dt = BatchNorm backward output, dx out) and compares
  * dx against a float64 transposed convolution of the captured dt with the layer's own weights (is the data-gradient kernel right?),
  * dy / dx against the oracle's autograd gradients at the same places (where does the deviation start?).
    python tools/debug_neck1.py [--batch 32] [--size 640]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=640)
    args = ap.parse_args()
    import torch
    import torch.nn.functional as F

    from oracle.yolo_nas import YoloNAS as Oracle
    from super_gradients_amd import kernels as K
    from super_gradients_amd.training import models

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    torch.set_num_threads(64)
    ref = Oracle("s", num_classes=80)
    g = torch.Generator().manual_seed(5)
    for name, p in ref.named_parameters():
        if name.endswith("bn.weight") or name.endswith("post_bn.weight"):
            p.data.uniform_(0.5, 1.0, generator=g)
        elif (name.endswith("bn.bias") and "branch_3x3" not in name) or name.endswith("post_bn.bias"):
            p.data.fill_(4.0)
    net = models.get("yolo_nas_s", num_classes=80)
    net.load_state_dict(ref.state_dict(), strict=True)
    ref.train()
    net.train()
    x = torch.rand(args.batch, 3, args.size, args.size, generator=torch.Generator().manual_seed(8))
    grads = {}

    def keep(name):
        def hook(gr):
            grads[name] = gr.detach().clone()
        return hook

    # oracle: gradients at neck1.out (= neck2.conv input) and at neck2.conv's output / conv output
    n2 = ref.neck.neck2
    orig_fwd = n2.conv.forward

    def conv_fwd(inp):
        inp.register_hook(keep("neck1_out"))
        t = n2.conv.conv(inp)
        t.register_hook(keep("n2conv_t"))
        y = F.relu(n2.conv.bn(t))
        y.register_hook(keep("n2conv_y"))
        return y

    n2.conv.forward = conv_fwd
    out_ref = ref(x)
    lg_r, ds_r = out_ref[1][:2]
    gg = torch.Generator().manual_seed(21)
    up_l, up_d = torch.randn(lg_r.shape, generator=gg), torch.randn(ds_r.shape, generator=gg)
    torch.autograd.backward([lg_r, ds_r], [up_l, up_d])

    cap = {}
    blk = net.neck.neck2.conv
    conv = blk.conv
    orig_bwd, orig_dgrad = blk.bwd, conv.dgrad

    def bwd(dy, **kw):
        cap["dy"] = dy.detach().clone()
        return orig_bwd(dy, **kw)

    def dgrad(dt, shape, **kw):
        cap["dt"] = dt.detach().clone()
        cap["kw"] = {k: (None if v is None else (tuple(v.shape), v.stride()) if torch.is_tensor(v) else v) for k, v in kw.items()}
        dx = orig_dgrad(dt, shape, **kw)
        cap["dx"] = dx.detach().clone()
        return dx

    blk.bwd, conv.dgrad = bwd, dgrad
    # the two terms of neck2's g_inter: upsample.bwd(dcat[..., :oc]) and the axpy of neck3's skip gradient
    up = net.neck.neck2.upsample
    orig_up = up.bwd

    def up_bwd(dy, **kw):
        cap["up_dy"] = dy.detach().clone()
        cap["up_dy_meta"] = (tuple(dy.shape), dy.stride())
        dx = orig_up(dy, **kw)
        cap["up_dx"] = dx.detach().clone()
        return dx

    up.bwd = up_bwd
    orig_axpy = K.axpy
    axpys = []

    def axpy(xx, a=1.0, a_dev=None, out=None, accumulate=False):
        rec = None
        if tuple(xx.shape) == (args.batch, args.size // 16, args.size // 16, 96) and accumulate:
            rec = dict(x=xx.detach().clone(), x_meta=(tuple(xx.shape), xx.stride()), before=out.detach().clone())
        r = orig_axpy(xx, a=a, a_dev=a_dev, out=out, accumulate=accumulate)
        if rec is not None:
            rec["after"] = r.detach().clone()
            axpys.append(rec)
        return r

    K.axpy = axpy
    import super_gradients_amd.training.models.detection_models.yolo_nas.yolo_stages as YS
    YS.K.axpy = axpy
    out = net(x.to(dev))
    lg, ds = out[1][:2]
    torch.autograd.backward([lg, ds], [up_l.to(dev), up_d.to(dev)])
    torch.cuda.synchronize()
    nchw = lambda t: t.permute(0, 3, 1, 2).cpu()  # noqa: E731
    print("dgrad call:", cap["kw"], "dt", tuple(cap["dt"].shape), cap["dt"].stride(), "dx", tuple(cap["dx"].shape), cap["dx"].stride())
    print("dy   (gradient at neck2.conv output)   hip vs oracle:", rel(nchw(cap["dy"]), grads["n2conv_y"]))
    print("dt   (BatchNorm backward output)       hip vs oracle:", rel(nchw(cap["dt"]), grads["n2conv_t"]))
    print("dx   (gradient at neck1.out)           hip vs oracle:", rel(nchw(cap["dx"]), grads["neck1_out"]))
    w = conv.weight.detach().cpu().double()  # [K, C, 1, 1]
    dx64 = F.conv_transpose2d(nchw(cap["dt"]).double(), w)
    print("dx   hip kernel vs float64 conv_transpose of ITS OWN dt:", rel(nchw(cap["dx"]), dx64))
    e = (nchw(cap["dx"]).double() - dx64)
    print("     error per-channel mean / rms:", float(e.mean((0, 2, 3)).abs().max()), float(e.pow(2).mean().sqrt()), " dx rms", float(dx64.pow(2).mean().sqrt()))
    eo = nchw(cap["dx"]).double() - grads["neck1_out"].double()
    print("     vs oracle: error per-channel mean max", float(eo.mean((0, 2, 3)).abs().max()), "rms", float(eo.pow(2).mean().sqrt()))
    wt = up.weight.detach().cpu().double()  # ConvTranspose2d weight [C_in, C_out, 2, 2]: d/dx = conv2d(dy, W, stride 2)
    updx64 = F.conv2d(nchw(cap["up_dy"]).double(), wt, stride=2)
    print("upsample.bwd: operand", cap["up_dy_meta"], " hip vs float64 conv of ITS OWN dy:", rel(nchw(cap["up_dx"]), updx64))
    for i, r in enumerate(axpys):
        want = r["before"].double() + r["x"].double()
        print(f"axpy[{i}] x {r['x_meta']}: after vs before + x:", rel(r["after"], want), " |x| max", float(r["x"].abs().max()))
    if axpys:
        tot = nchw(axpys[-1]["after"])
        print("g_inter after the axpy vs oracle:", rel(tot, grads["n2conv_y"]), "   upsample term alone vs oracle total:", rel(nchw(cap["up_dx"]), grads["n2conv_y"]))
        e = (tot.double() - grads["n2conv_y"].double()).abs()
        idx = torch.nonzero(e > 0.25 * e.max())
        print("   large-error elements:", idx.shape[0], "first", idx[:8].tolist(), " of shape", tuple(e.shape))
    # the same kernel again, stand-alone, on the captured operand
    dx2 = K.conv2d_bwd_data(cap["dt"], conv._w, tuple(cap["dx"].shape), stride=1, pad=0)
    print("dx   stand-alone kernel call vs float64:", rel(nchw(dx2), dx64))


if __name__ == "__main__":
    main()
