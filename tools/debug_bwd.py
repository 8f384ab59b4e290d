"""Debug aid: compare the gradient arriving at every block (dy) between the HIP model and the fp64 oracle."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from test_yolo_nas import _build_pair
from super_gradients_amd.modules.engine import SgxBlock

variant, B, size = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
ref, net = _build_pair(variant, 80, dev)
ref = ref.double()
cap, order, capo = {}, [], {}
for n, m in net.named_modules():
    if isinstance(m, SgxBlock) and n:
        ob = m.bwd
        def bwd(dy, *a, _ob=ob, _n=n, **kw):
            if torch.is_tensor(dy) and dy.dim() == 4:
                cap[_n] = dy.detach().clone(); order.append(_n)
            r = _ob(dy, *a, **kw)
            if torch.is_tensor(r) and r.dim() == 4: capo[_n] = r.detach().clone()
            return r
        m.bwd = bwd
rcap = {}; rcapi = {}
for n, m in ref.named_modules():
    if n:
        def bh(mod, gin, gout, n=n):
            if gout[0] is not None: rcap[n] = gout[0].detach()
            if gin[0] is not None: rcapi[n] = gin[0].detach()
        m.register_full_backward_hook(bh)
net.train(); ref.train()
x = torch.rand(B, 3, size, size, generator=torch.Generator().manual_seed(7))
out = net(x.to(dev)); oref = ref(x.double())
gg = torch.Generator().manual_seed(21)
up_l, up_d = torch.randn(out[1][0].shape, generator=gg), torch.randn(out[1][1].shape, generator=gg)
torch.autograd.backward([out[1][0], out[1][1]], [up_l.to(dev), up_d.to(dev)])
torch.autograd.backward([oref[1][0], oref[1][1]], [up_l.double(), up_d.double()])
for n in order:
    if n not in rcap: continue
    a = cap[n].cpu().permute(0, 3, 1, 2).double(); b = rcap[n]
    if a.shape != b.shape: print(n, "shape", a.shape, b.shape); continue
    d = (a - b).abs()
    e = float(d.max() / b.abs().max())
    bad = (d > 1e-3 * b.abs().max())
    msg = ""
    if e > 1e-4:
        idx = bad.nonzero()
        msg = f" nbad {int(bad.sum())}/{bad.numel()} first {idx[:4].tolist()}"
    print(f"{e:9.2e} {n}{msg}")

print("---- returned dx vs oracle grad_input (no-accumulate blocks only meaningful)")
for n in order[:12]:
    if n in capo and n in rcapi:
        a = capo[n].cpu().permute(0, 3, 1, 2).double(); b = rcapi[n]
        if a.shape != b.shape: continue
        d = (a - b).abs(); e = float(d.max() / b.abs().max()); bad = d > 1e-3 * b.abs().max()
        print(f"{e:9.2e} {n} nbad {int(bad.sum())} first {bad.nonzero()[:6].tolist()}")
