"""Error statistics of the conv arithmetic modes against fp64 on inputs with a large positive mean (the flip-free regime of the whole-model
gradient checks): mean signed error (bias) and rms, relative to the rms of the exact result.  Measurement tool: product library only.

    python tools/conv_error_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.nn.functional as F

    from super_gradients_amd import kernels as K
    from util import to_nhwc, to_nchw_cpu

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    print(f"{'case':<34}{'mode':<12}{'bias/rms(y)':>14}{'rms err/rms(y)':>16}{'max err/rms(y)':>16}")
    for (n, h, w, c, k, r, s, p, mean) in [(2, 40, 40, 256, 256, 1, 1, 0, 4.0), (2, 40, 40, 256, 256, 1, 1, 0, 0.0), (2, 40, 40, 64, 128, 3, 2, 1, 4.0),
                                           (2, 20, 20, 768, 384, 1, 1, 0, 4.0), (2, 40, 40, 192, 384, 3, 2, 1, 4.0)]:
        x = torch.randn(n, c, h, w, generator=g) + mean
        wt = torch.randn(k, c, r, r, generator=g) / (c * r * r) ** 0.5
        ref = F.conv2d(x.double(), wt.double(), None, stride=s, padding=p)
        cpu = F.conv2d(x, wt, None, stride=s, padding=p).double()
        sc = float(ref.pow(2).mean().sqrt())
        rows = [("ATen cpu fp32", cpu)]
        xd, wd = to_nhwc(x, dev), K.to_ohwi(wt.to(dev))
        for mode in ("fp32", "bf16x3", "patch_bf3"):
            K.set_conv_math(mode)
            rows.append((mode, to_nchw_cpu(K.conv2d_fwd(xd, wd, stride=s, pad=p)).double()))
        K.set_conv_math(K.DEFAULT_CONV_MATH)
        for name, y in rows:
            e = y - ref
            print(f"{str((n, h, w, c, k, r, s, mean)):<34}{name:<12}{float(e.mean()) / sc:>14.3e}{float(e.pow(2).mean().sqrt()) / sc:>16.3e}{float(e.abs().max()) / sc:>16.3e}")

    # weight gradients (leaf results: their error does not travel through further layers): fp32 slab loop / bf16x3 slab loop (two accumulators) /
    # + patch kernel for the 3x3 layers (ONE accumulator per block: the six products of a 16-pixel step add into the same registers)
    from super_gradients_amd._lib import lib

    print()
    print(f"{'weight gradient case':<34}{'mode':<12}{'bias/rms(dw)':>14}{'rms err/rms(dw)':>16}{'max err/rms(dw)':>16}")
    for (n, h, w, c, k, r, s, p, mean) in [(8, 80, 80, 64, 64, 3, 1, 1, 4.0), (8, 80, 80, 64, 64, 3, 1, 1, 0.0), (8, 80, 80, 64, 128, 3, 2, 1, 4.0), (8, 40, 40, 192, 192, 3, 1, 1, 4.0)]:
        x = torch.randn(n, c, h, w, generator=g) + mean
        ho, wo = (h + 2 * p - r) // s + 1, (w + 2 * p - r) // s + 1
        dy = torch.randn(n, k, ho, wo, generator=g)
        xr = x.double().requires_grad_(False)
        wref = torch.zeros(k, c, r, r, dtype=torch.float64, requires_grad=True)
        (F.conv2d(xr, wref, None, stride=s, padding=p) * dy.double()).sum().backward()
        ref = wref.grad
        w32 = torch.zeros(k, c, r, r, requires_grad=True)
        (F.conv2d(x, w32, None, stride=s, padding=p) * dy).sum().backward()
        sc = float(ref.pow(2).mean().sqrt())
        rows = [("ATen cpu fp32", w32.grad.double())]
        xd, dyd = to_nhwc(x, dev), to_nhwc(dy, dev)
        for mode, name in ((0, "fp32"), (1, "bf16x3"), (2, "patch")):
            lib().sgx_conv_set_wgrad_math(mode)
            dw = K.to_ohwi(torch.zeros(k, c, r, r, device=dev))
            K.conv2d_bwd_weight_group([(xd, dyd, dw, s, p)])
            rows.append((name, dw.cpu().double()))
        lib().sgx_conv_set_wgrad_math(2)
        for name, y in rows:
            e = y - ref
            print(f"{str((n, h, w, c, k, r, s, mean)):<34}{name:<12}{float(e.mean()) / sc:>14.3e}{float(e.pow(2).mean().sqrt()) / sc:>16.3e}{float(e.abs().max()) / sc:>16.3e}")


if __name__ == "__main__":
    main()
