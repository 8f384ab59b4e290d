"""sgx_dot beside another kernel: does its result move?  (the bottlenecks' d alpha flips its last bit in ~0.5 % of the steps, only
while a weight-gradient kernel of the side stream is resident.)  One fixed pair of operands with the cancellation of the real dot; the dot runs on
the main stream while a second stream runs, in turn: nothing, 3x3 weight gradients (patch kernel), 1x1 weight gradients, forward convolutions,
a BatchNorm sweep, a torch matmul.  Counts how many of N results differ from the dot alone, and by how much (fp64 of the two launches' output)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from super_gradients_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
g = torch.Generator().manual_seed(0)
n, h, w, c = 4, 40, 40, 64
x = torch.randn(n, h, w, c, generator=g).to(dev)
dz = (torch.randn(n, h, w, c, generator=g) * 1e-3).to(dev)
out = torch.zeros(1, device=dev)
side = torch.cuda.Stream(device=dev)

# the co-runners' operands
bx = torch.randn(8, 80, 80, 96, generator=g).to(dev)
bdy3 = torch.randn(8, 80, 80, 96, generator=g).to(dev)
dw3 = K.ohwi_empty(96, 96, 3, 3, dev)
dw1 = K.ohwi_empty(96, 96, 1, 1, dev)
w3 = K.to_ohwi(torch.randn(96, 96, 3, 3, generator=g).to(dev) * 0.05)
a = torch.randn(4096, 4096, device=dev)


def co_wgrad3():
    dw3.zero_()
    K.conv2d_bwd_weight(bx, bdy3, dw3, stride=1, pad=1)


def co_wgrad1():
    dw1.zero_()
    K.conv2d_bwd_weight(bx, bdy3, dw1, stride=1, pad=0)


def co_fwd():
    K.conv2d_fwd(bx, w3, stride=1, pad=1)


def co_sweep():
    K.axpy(bx, a=1.0001)


def co_matmul():
    torch.mm(a, a)


NPART = K.stats_blocks(n * h * w) * ((c // 4 + 15) // 16)  # workgroups of the first stage = fp64 partial rows in the stream's workspace


def dot():
    """-> (fp32 result, the first stage's fp64 partials as raw bits: any change of a single addend shows, not only a flipped last bit)"""
    K.dot_sum(x, dz, out, accumulate=False)
    ws = K.WORKSPACE.get(1, dev)
    return torch.cat([out.view(torch.int32).to(torch.int64), ws[: NPART * 8].view(torch.int64)])


torch.cuda.synchronize()
ref = dot().cpu()
print(f"reference {float(out)!r}, {NPART} partial rows", flush=True)
for name, co in (("nothing", None), ("3x3 weight gradient", co_wgrad3), ("1x1 weight gradient", co_wgrad1), ("forward convolution", co_fwd), ("axpy sweep", co_sweep),
                 ("torch.mm", co_matmul), ("3x3 weight gradient", co_wgrad3)):
    vals, t0 = [], time.time()
    for i in range(N):
        if co is not None:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                co()
                co()
        vals.append(dot())
        if co is not None and i % 16 == 15:
            torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    v = torch.stack(vals).cpu()
    bad_out = v[:, 0] != ref[0]
    bad_part = (v[:, 1:] != ref[1:]).any(dim=1)
    rows = sorted({int(j) for j in (v[:, 1:] != ref[1:]).nonzero()[:, 1].tolist()})[:8]
    if rows:  # how far: relative distance of the fp64 rows, rows per call
        d64, r64 = v[:, 1:].contiguous().view(torch.float64), ref[1:].contiguous().view(torch.float64)
        rel = ((d64 - r64).abs() / r64.abs().clamp_min(1e-300))
        per_call = (v[:, 1:] != ref[1:]).sum(dim=1)
        print(f"    rows per differing call: min {int(per_call[bad_part].min())} max {int(per_call[bad_part].max())}; relative distance of a row: max {float(rel.max()):.3e}, "
              f"median of the differing ones {float(rel[rel > 0].median()):.3e}; total (sum of rows) moves by at most {float((d64.sum(dim=1) - r64.sum()).abs().max() / r64.sum().abs()):.3e}", flush=True)
    print(f"beside {name}: fp32 result differs in {int(bad_out.sum())} of {N} calls, fp64 partial rows in {int(bad_part.sum())}" + (f" (rows {rows})" if rows else "")
          + f"  [{time.time() - t0:.0f} s]", flush=True)
