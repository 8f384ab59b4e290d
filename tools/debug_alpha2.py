import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, copy
from test_yolo_nas import _build_pair
from super_gradients_amd.training.models.detection_models.yolo_nas import yolo_stages as ys

variant, B, size = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
ref, net = _build_pair(variant, 80, dev)
ref = ref.double()
names = {id(m): n for n, m in net.named_modules()}
cap = {}
orig = ys.YoloNASBottleneck.bwd
def bwd(self, dz, **kw):
    if self.add:
        cap[names[id(self)]] = (self._x.detach().clone(), dz.detach().clone())
    return orig(self, dz, **kw)
ys.YoloNASBottleneck.bwd = bwd
rcap = {}
for n, m in ref.named_modules():
    if type(m).__name__ == "Bottleneck":
        def fh(mod, inp, out, n=n): rcap.setdefault(n, {})["x"] = inp[0].detach()
        def bh(mod, gin, gout, n=n): rcap.setdefault(n, {})["dz"] = gout[0].detach()
        m.register_forward_hook(fh); m.register_full_backward_hook(bh)
net.train(); ref.train()
x = torch.rand(B, 3, size, size, generator=torch.Generator().manual_seed(7))
out = net(x.to(dev)); oref = ref(x.double())
gg = torch.Generator().manual_seed(21)
up_l, up_d = torch.randn(out[1][0].shape, generator=gg), torch.randn(out[1][1].shape, generator=gg)
torch.autograd.backward([out[1][0], out[1][1]], [up_l.to(dev), up_d.to(dev)])
torch.autograd.backward([oref[1][0], oref[1][1]], [up_l.double(), up_d.double()])
rp = dict(ref.named_parameters())
for n in cap:
    xh, dzh = cap[n]
    xr, dzr = rcap[n]["x"], rcap[n]["dz"]
    xh = xh.cpu().permute(0, 3, 1, 2).double(); dzh = dzh.cpu().permute(0, 3, 1, 2).double()
    ex = float((xh - xr).abs().max() / xr.abs().max()); ed = float((dzh - dzr).abs().max() / dzr.abs().max())
    print(f"{n:45s} x err {ex:.2e} dz err {ed:.2e}  sum|x dz| {float((xr*dzr).abs().sum()):.3e} sum {float((xr*dzr).sum()):+.4e} alpha.grad ref {float(rp[n+'.alpha'].grad):+.4e} hip {float(dict(net.named_parameters())[n+'.alpha'].grad):+.4e}")
