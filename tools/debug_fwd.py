"""Debug aid: compare every block's forward output between the HIP model and the fp64 oracle."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from test_yolo_nas import _build_pair
from super_gradients_amd.modules.engine import SgxBlock

variant, B, size = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
ref, net = _build_pair(variant, 80, dev)
import copy
ref32 = ref; ref = copy.deepcopy(ref).double()
rcap32 = {}
for n, m in ref32.named_modules():
    if n:
        def fh32(mod, inp, out, n=n):
            if torch.is_tensor(out): rcap32[n] = out.detach()
        m.register_forward_hook(fh32)
cap, order = {}, []
for n, m in net.named_modules():
    if isinstance(m, SgxBlock) and n:
        of = m.fwd
        def fwd(x, *a, _of=of, _n=n, **kw):
            r = _of(x, *a, **kw)
            if torch.is_tensor(r) and r.dim() == 4:
                cap[_n] = r.detach().clone(); order.append(_n)
            return r
        m.fwd = fwd
rcap = {}
for n, m in ref.named_modules():
    if n:
        def fh(mod, inp, out, n=n):
            if torch.is_tensor(out): rcap[n] = out.detach()
        m.register_forward_hook(fh)
net.train(); ref.train()
x = torch.rand(B, 3, size, size, generator=torch.Generator().manual_seed(7))
out = net(x.to(dev)); oref = ref(x.double()); ref32.train(); ref32(x)
for n in order:
    if n not in rcap: continue
    a = cap[n].cpu().permute(0, 3, 1, 2).double(); b = rcap[n]
    if a.shape != b.shape: print(n, "shape", a.shape, b.shape); continue
    d = (a - b).abs(); e = float(d.max() / b.abs().max())
    bad = d > 1e-4 * b.abs().max()
    flips = int(((a > 0) != (b > 0)).sum())
    c = rcap32[n].double(); e32 = float((c - b).abs().max() / b.abs().max()); flips32 = int(((c > 0) != (b > 0)).sum())
    print(f"hip {e:9.2e} flips {flips:4d} | cpu32 {e32:9.2e} flips {flips32:4d} /{a.numel()} {n}" + (f" nbad {int(bad.sum())} first {bad.nonzero()[:4].tolist()}" if e > 1e-4 else ""))
