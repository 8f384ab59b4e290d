"""How often, and where, does a first step of a fresh branch-stream network differ from the single-chain one?  (r6fin4: once, one ulp, inside the
whole GPU suite.)  Fresh networks every round, allocator state shuffled between rounds, a second single-chain network as the control."""
import random
import sys
import time

import torch

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
from util import synthetic_targets  # noqa: E402

from super_gradients_amd.training import models  # noqa: E402
from super_gradients_amd.training.losses import PPYoloELoss  # noqa: E402

dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 600.0


def build(branch, state=None):
    torch.manual_seed(3)
    net = models.get("yolo_nas_s", num_classes=80).materialize(dev).train()
    net.wg_group_flops = 0.0
    if not branch:
        net.branch_mode = 0
    if state is not None:
        net.load_state_dict(state)
    return net


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


rng = random.Random(0)
t0, fails = time.time(), {"forked": 0, "plain2": 0}
ref = None
keep = []
for r in range(rounds):
    if time.time() - t0 > budget:
        break
    keep = [torch.full((rng.randrange(1 << 8, 1 << 24),), float("nan"), device=dev) for _ in range(rng.randrange(0, 12))]  # shuffle the pools
    del keep[::2]
    plain = build(False)
    state = plain.state_dict()
    nets = {"plain": plain, "plain2": build(False, state), "forked": build(True, state)}
    for rep in range(2):
        for name, net in nets.items():
            l, g = step(net)
            if ref is None:
                ref = (l, g)
            if not (torch.equal(l, ref[0]) and torch.equal(g, ref[1])):
                bad = ((g != ref[1]) | torch.isnan(g)).nonzero().flatten()
                fails[name if name != "plain" else "plain2"] += 1
                print(f"round {r} step {rep} {name}: loss {float(l)!r} vs {float(ref[0])!r}; {bad.numel()} gradient elements differ, max {float((g - ref[1]).abs().nan_to_num(1e9).max()):.3e}, "
                      f"largest |g| among them {float(ref[1][bad].abs().max()) if bad.numel() else 0.0:.3e}, in {names(net, bad)}", flush=True)
    del nets, plain, state
print(f"{r + 1} rounds x 2 steps x 3 networks in {time.time() - t0:.0f} s: mismatching steps {fails}", flush=True)
