"""Companion of dot_race_probe.py: do ELEMENTWISE sweeps built with packed fp32 instructions (the build before DESIGN.md 11.12) change output bits
beside a weight-gradient kernel?  Victims: sgx_axpy (y = a x), sgx_bn_bwd_apply (the BatchNorm-backward apply sweep); outputs compared as raw bits
with the same call made alone."""
import sys
import time

import torch

sys.path.insert(0, ".")
from super_gradients_amd import kernels as K  # noqa: E402
from super_gradients_amd._lib import lib, ptr, check, stream  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
g = torch.Generator().manual_seed(0)
x = torch.randn(4, 40, 40, 64, generator=g).to(dev)
dy = torch.randn(4, 40, 40, 64, generator=g).to(dev)
C = 64
scale, shift = (torch.rand(C, generator=g) + 0.5).to(dev), torch.randn(C, generator=g).to(dev)
coef = torch.randn(5, C, generator=g).to(dev) * 0.1
side = torch.cuda.Stream(device=dev)
bx = torch.randn(8, 80, 80, 96, generator=g).to(dev)
bdy = torch.randn(8, 80, 80, 96, generator=g).to(dev)
dw3 = K.ohwi_empty(96, 96, 3, 3, dev)


def co_wgrad3():
    dw3.zero_()
    K.conv2d_bwd_weight(bx, bdy, dw3, stride=1, pad=1)


def v_axpy():
    return K.axpy(x, a=1.0001).view(torch.int32)


def v_bn_apply():
    out = torch.empty_like(x)
    M = x.shape[0] * x.shape[1] * x.shape[2]
    check(lib().sgx_bn_bwd_apply(ptr(dy), C, ptr(x), C, ptr(scale), ptr(shift), ptr(coef), ptr(out), C, None, 0, M, C, K.ACT["relu"], stream()), "sgx_bn_bwd_apply")
    return out.view(torch.int32)


for vname, victim in (("axpy", v_axpy), ("bn_bwd_apply", v_bn_apply)):
    torch.cuda.synchronize()
    ref = victim().clone()
    for cname, co in (("nothing", None), ("3x3 weight gradient", co_wgrad3)):
        bad_calls, bad_elems, t0 = 0, 0, time.time()
        for i in range(N):
            if co is not None:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    co()
                    co()
            d = int((victim() != ref).sum())
            bad_calls += d > 0
            bad_elems += d
            if co is not None and i % 16 == 15:
                torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        print(f"{vname} beside {cname}: {bad_calls} of {N} calls differ from the call alone ({bad_elems} elements in all)  [{time.time() - t0:.0f} s]", flush=True)
