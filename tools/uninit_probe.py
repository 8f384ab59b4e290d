"""Does any kernel of the YOLO-NAS-S train step read memory it did not write?  (r6fin4: the branch-stream bit-identity test failed once inside the
whole GPU suite - one ulp in one place - and never alone: in a fresh process torch.empty hands out zeroed pages, inside the suite recycled ones.)
Recycled blocks are filled with a pattern (NaN, 1e30, 1e-30, the previous contents), then two single-chain networks and a branch-stream network
with equal weights run the same step: NaN anywhere, or gradients that depend on the pattern, name the reader."""
import sys

import torch

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
from util import synthetic_targets  # noqa: E402

from super_gradients_amd.training import models  # noqa: E402
from super_gradients_amd.training.losses import PPYoloELoss  # noqa: E402

dev = torch.device("cuda:0")


def build(branch):
    torch.manual_seed(3)
    net = models.get("yolo_nas_s", num_classes=80).materialize(dev).train()
    net.wg_group_flops = 0.0
    if not branch:
        net.branch_mode = 0
    return net


def pollute(value, streams):
    """fill the caching allocator's free blocks (every stream's pool) with `value`"""
    for st in streams:
        with torch.cuda.stream(st):
            junk = []
            for mb in (1, 2, 3, 5, 8, 16, 24, 40, 64, 100, 160, 256, 400, 640) * 3:
                t = torch.empty(mb << 18, device=dev)
                t.fill_(value)
                junk.append(t)
            for n in (128, 512, 2048, 8192, 32768, 131072):  # the small-block pool (< 1 MB: partial rows, coefficient rows, counters)
                for _ in range(300):
                    t = torch.empty(n, device=dev)
                    t.fill_(value)
                    junk.append(t)
            del junk
    torch.cuda.synchronize()


x = torch.rand(4, 3, 320, 320, generator=torch.Generator().manual_seed(1)).to(dev)
t = synthetic_targets(4, seed=2, kmax=6, size=320).to(dev)
crit = PPYoloELoss(num_classes=80, use_static_assigner=False)


def step(net):
    net.zero_grad()
    loss, _ = crit(net(x), t)
    loss.backward()
    net.join_side()
    torch.cuda.synchronize()
    return loss.detach().cpu().clone(), net.g_arena.buf.cpu().clone()


def names(net, idx):
    return sorted({next((s.name for s in net.slots if s.start <= int(i) < s.start + max(s.numel, 1)), "?") for i in idx[:2000]})[:10]


plain, plain2, forked = build(False), build(False), build(True)
plain2.load_state_dict(plain.state_dict())
forked.load_state_dict(plain.state_dict())
streams = [torch.cuda.current_stream(), plain.side_stream, forked.side_stream] + list(forked.branch_lanes)
step(plain), step(plain2), step(forked)  # first steps: lazy buffers exist from here on
ref = None
for value in (0.0, float("nan"), 1e30, 1e-30, -3.7e-9, float("nan")):
    row = []
    for name, net in (("plain", plain), ("plain2", plain2), ("forked", forked)):
        pollute(value, [s for s in streams if s is not None])
        l, g = step(net)
        if ref is None:
            ref = (l, g)
        bad = (g != ref[1]) | torch.isnan(g)
        row.append(f"{name}: loss {'==' if torch.equal(l, ref[0]) else repr(float(l))} grads "
                   + ("==" if not bool(bad.any()) else f"{int(bad.sum())} differ / nan {int(torch.isnan(g).sum())} max {float((g - ref[1]).abs().nan_to_num(1e9).max()):.2e} in {names(net, bad.nonzero().flatten())}"))
    print(f"pattern {value!r}: " + " | ".join(row), flush=True)
