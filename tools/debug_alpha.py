import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from test_yolo_nas import _build_pair
from super_gradients_amd import kernels as K
from super_gradients_amd.training.models.detection_models.yolo_nas import yolo_stages as ys

orig = ys.YoloNASBottleneck.bwd
names = {}
def bwd(self, dz, dx_out=None, accumulate=False, addend=None, need_dx=True):
    if self.add:
        x = self._x
        before = float(self.alpha.grad)
        truth = float((x.double() * dz.double()).sum())
        tmp = torch.zeros(1, device=x.device)
        K.dot_sum(x, dz, tmp, accumulate=False)
        print(f"{names[id(self)]:50s} x{tuple(x.shape)} xs{x.stride()} dzs{dz.stride()} truth {truth:+.6e} kernel {float(tmp):+.6e} before {before:+.3e}")
    return orig(self, dz, dx_out=dx_out, accumulate=accumulate, addend=addend, need_dx=need_dx)
ys.YoloNASBottleneck.bwd = bwd

variant, B, size = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
ref, net = _build_pair(variant, 80, dev)
for n, m in net.named_modules():
    names[id(m)] = n
net.train()
x = torch.rand(B, 3, size, size, generator=torch.Generator().manual_seed(7))
out = net(x.to(dev))
gg = torch.Generator().manual_seed(21)
up_l, up_d = torch.randn(out[1][0].shape, generator=gg), torch.randn(out[1][1].shape, generator=gg)
torch.autograd.backward([out[1][0], out[1][1]], [up_l.to(dev), up_d.to(dev)])
for n, p in net.named_parameters():
    if n.endswith("alpha"): print(n, float(p.grad))
