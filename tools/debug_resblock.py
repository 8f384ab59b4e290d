import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import to_nhwc, to_nchw_cpu
from oracle.resnet import BasicBlock
from super_gradients_amd.training.models.classification_models.resnet import BasicResNetBlock
from super_gradients_amd.modules.layers import BatchNorm
import test_blocks as TB
dev = torch.device("cuda:0")
for seed in range(4):
    ref, blk = BasicBlock(64, 64, 1, 1), BasicResNetBlock(64, 64, 1, 1)
    TB._randomize(ref, 1)
    for m in blk.modules():
        if isinstance(m, BatchNorm): m.eps, m.momentum = 1e-3, 0.03
    net = TB._wrap(blk, dev)
    blk.load_state_dict(ref.state_dict(), strict=True)
    ref.train(); net.train()
    x = torch.randn(2, 64, 28, 28, generator=torch.Generator().manual_seed(seed))
    xr = x.clone().requires_grad_(True)
    y = ref(xr)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5))
    y.backward(dy)
    net.zero_grad()
    yd = blk.fwd(to_nhwc(x, dev))
    ey = (to_nchw_cpu(yd) - y).abs()
    dx = to_nchw_cpu(blk.bwd(to_nhwc(dy, dev)))
    e = (dx - xr.grad).abs()
    scale = float(xr.grad.abs().max())
    bad = e > 1e-4 * scale
    idx = bad.nonzero()
    mask_flip = ((to_nchw_cpu(yd) > 0) != (y > 0)).sum()
    print(f"seed {seed}: fwd max err {float(ey.max()):.2e}; dx max rel err {float(e.max())/scale:.2e}; bad elems {int(bad.sum())}; mask flips {int(mask_flip)}; "
          f"bad h range {idx[:,2].min().item() if len(idx) else -1}-{idx[:,2].max().item() if len(idx) else -1} w range {idx[:,3].min().item() if len(idx) else -1}-{idx[:,3].max().item() if len(idx) else -1} imgs {idx[:,0].unique().tolist() if len(idx) else []}")
