"""Debug aid: re-derive every _ConvBN.bwd with torch fp64 ops from the block's own saved tensors and compare with the kernels."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import torch.nn.functional as F
from test_yolo_nas import _build_pair
from super_gradients_amd.modules import conv_bn_act_block as cb

variant, B, size = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
ref, net = _build_pair(variant, 80, dev)
names = {id(m): n for n, m in net.named_modules()}
orig = cb._ConvBN.bwd
def bwd(self, dy, dx_out=None, accumulate=False, addend=None, need_dx=True):
    conv, bn = self._parts()
    x, t, scale, shift, mean, invstd = self._ctx
    td, dyd = t.double().clone(), dy.double().clone()
    pre = td * scale.double() + shift.double()
    g = dyd * (pre > 0).double() if self.act == "relu" else dyd
    M = t.shape[0] * t.shape[1] * t.shape[2]
    mg = g.sum((0, 1, 2)) / M
    xc = td - mean.double()
    k = (g * xc).sum((0, 1, 2)) / M * invstd.double() ** 2
    dt_ref = bn.weight.double() * invstd.double() * ((g - mg) - xc * k)
    w = conv.weight.double()
    dx_ref = torch.nn.grad.conv2d_input((x.shape[0], x.shape[3], x.shape[1], x.shape[2]), w if w.shape[1] == x.shape[3] else F.pad(w, (0, 0, 0, 0, 0, x.shape[3] - w.shape[1])),
                                        dt_ref.permute(0, 3, 1, 2), stride=conv.stride, padding=conv.padding).permute(0, 2, 3, 1)
    prev = dx_out.double().clone() if (dx_out is not None and accumulate) else 0
    r = orig(self, dy, dx_out=dx_out, accumulate=accumulate, addend=addend, need_dx=need_dx)
    e_dt = float((t.double() - dt_ref).abs().max() / dt_ref.abs().max())   # t now holds dt (in place)
    msg = ""
    if need_dx:
        exp = dx_ref + prev + (addend.double() if addend is not None else 0)
        e_dx = float((r.double() - exp).abs().max() / exp.abs().max())
        msg = f" dx {e_dx:.2e}"
    print(f"{names[id(self)]:45s} dt {e_dt:.2e}{msg}  dy_strides {dy.stride()} acc {accumulate} addend {addend is not None}")
    return r
cb._ConvBN.bwd = bwd
net.train()
x = torch.rand(B, 3, size, size, generator=torch.Generator().manual_seed(7))
out = net(x.to(dev))
gg = torch.Generator().manual_seed(21)
up_l, up_d = torch.randn(out[1][0].shape, generator=gg), torch.randn(out[1][1].shape, generator=gg)
torch.autograd.backward([out[1][0], out[1][1]], [up_l.to(dev), up_d.to(dev)])
