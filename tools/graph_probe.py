"""Probe: how fast does the train step replay as ONE hipGraph (torch.cuda.graph capture of forward + loss + backward + AdamW + EMA)?
Hyper-parameters are frozen at capture time here (probe only)."""
import os, sys, time, json
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from super_gradients_amd.training import models
from super_gradients_amd.training.losses import PPYoloELoss
from super_gradients_amd.training.utils.ema import ModelEMA
from super_gradients_amd.training.utils.optimizers import ArenaAdamW
from bench import synthetic_batch

dev = torch.device("cuda:0")
torch.manual_seed(42)
net = models.get("yolo_nas_s", num_classes=80); net.materialize(dev); net.train()
crit = PPYoloELoss(num_classes=80, use_static_assigner=False)
opt = ArenaAdamW(net, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, zero_weight_decay_on_bias_and_bn=True)
ema = ModelEMA.from_params(net, decay=0.9997, decay_type="threshold")
x, targets = synthetic_batch(32, 640, 42, dev)
state = {"step": 0}
def step():
    out = net(x); loss, _ = crit(out, targets); loss.backward(); opt.step(); opt.zero_grad(); ema.update(net, state["step"], 100000); state["step"] += 1; return loss
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 10 * 1e3
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        loss = step()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 10 * 1e3
    print(json.dumps({"eager_ms": round(eager, 3), "graph_ms": round(graph, 3), "loss": float(loss)}))
except Exception as e:
    import traceback; traceback.print_exc()
    print(json.dumps({"eager_ms": round(eager, 3), "graph_error": str(e)[:300]}))
