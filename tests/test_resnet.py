"""ResNet-18-CIFAR / ResNet-50 (BASELINE.json configs[0], configs[1]) parity.

  oracle (oracle/resnet.py)  <- golden fixtures generated from the reference's own resnet.py (oracle/make_golden.py)   CPU
  oracle                     <- live reference through the import shim, where /root/reference exists                    CPU
  product (HIP)              <- golden fixtures and the oracle on identical weights / inputs                            GPU
Tolerance: 1e-4 relative on logits / loss (BASELINE.json); gradients judged against the fp64 run of the same modules.
"""
import os

import pytest
import torch
import torch.nn.functional as F

from oracle import golden_util as G
from oracle import ref_shim
from util import rel_err


def _fx(name):
    return torch.load(os.path.join(G.GOLDEN_DIR, f"{name}.pt"), weights_only=False)


def _inputs(fx):
    x = torch.randn(fx["batch"], 3, fx["size"], fx["size"], generator=torch.Generator().manual_seed(5))
    return x, fx["labels"]


def _grad_check(norms, fx, what):
    t64, ref = fx["grad_norms_f64"], fx["grad_norms"]
    big = ref > 1e-3 * ref.max()
    e_hip = ((norms - t64).abs() / t64.clamp_min(1e-30))[big]
    e_ref = ((ref - t64).abs() / t64.clamp_min(1e-30))[big]
    msg = f"{what}: gradient norms vs fp64: worst {float(e_hip.max()):.2e} mean {float(e_hip.mean()):.2e}; reference fp32 worst {float(e_ref.max()):.2e} mean {float(e_ref.mean()):.2e}"
    assert float(e_hip.max()) <= max(5e-3, 3.0 * float(e_ref.max())) and float(e_hip.mean()) <= max(1e-3, 3.0 * float(e_ref.mean())), msg


@pytest.mark.parametrize("name", ["resnet18_cifar", "resnet50"])
def test_oracle_resnet_golden(name):
    from oracle.resnet import build

    fx = _fx(name)
    net = build(name, fx["classes"])
    assert list(net.state_dict().keys()) == fx["state_keys"]
    assert [tuple(v.shape) for v in net.state_dict().values()] == fx["state_shapes"]
    G.deterministic_fill(net, seed=4)
    net.train()
    x, y = _inputs(fx)
    logits = net(x)
    loss = F.cross_entropy(logits, y)
    loss.backward()
    assert rel_err(logits, fx["logits"]) <= 2e-5
    assert abs(float(loss) - float(fx["loss"])) <= 2e-5 * abs(float(fx["loss"]))
    _grad_check(torch.tensor([float(p.grad.double().norm()) for p in net.parameters()], dtype=torch.float64), fx, name)
    for k, v in fx["bn_running_checksum"].items():
        assert abs(float(net.state_dict()[k].double().sum()) - v) <= 2e-5 * max(abs(v), 1.0), k


@pytest.mark.skipif(not ref_shim.available(), reason="/root/reference not present")
@pytest.mark.parametrize("name,cls", [("resnet18", "ResNet18"), ("resnet18_cifar", "ResNet18Cifar"), ("resnet50", "ResNet50")])
def test_oracle_resnet_live(name, cls):
    from oracle.resnet import build

    torch.manual_seed(3)
    ref = ref_shim.reference_resnet(cls, 10).train()
    net = build(name, 10).train()
    net.load_state_dict(ref.state_dict(), strict=True)
    x = torch.randn(2, 3, 64, 64)
    a, b = ref(x), net(x)
    assert torch.equal(a, b)


def test_state_dict_matches_oracle():
    from oracle.resnet import build
    from super_gradients_amd.training import models

    for name in ("resnet18", "resnet34", "resnet50", "resnet18_cifar"):
        a = build(name, 10).state_dict()
        b = models.get(name, num_classes=10).state_dict()
        assert list(a.keys()) == list(b.keys()), name
        assert [tuple(v.shape) for v in a.values()] == [tuple(v.shape) for v in b.values()], name


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["resnet18_cifar", "resnet50"])
def test_product_resnet_golden(gpu_device, name):
    from super_gradients_amd.training import models
    from super_gradients_amd.training.losses import CrossEntropyLoss

    fx = _fx(name)
    net = models.get(name, num_classes=fx["classes"])
    assert list(net.state_dict().keys()) == fx["state_keys"]
    G.deterministic_fill(net, seed=4)
    net.materialize(gpu_device).train()
    x, y = _inputs(fx)
    logits = net(x.to(gpu_device))
    loss = CrossEntropyLoss()(logits, y.to(gpu_device))
    loss.backward()
    e_pair = rel_err(logits.cpu(), fx["logits"])
    e_hip, e_cpu = rel_err(logits.cpu().double(), fx["logits_f64"]), rel_err(fx["logits"].double(), fx["logits_f64"])
    assert e_pair <= 1e-4 or e_hip <= max(1e-4, 2.0 * e_cpu), f"logits: hip-ref32 {e_pair:.2e} hip-ref64 {e_hip:.2e} ref32-ref64 {e_cpu:.2e}"
    assert abs(float(loss) - float(fx["loss"])) <= 1e-4 * abs(float(fx["loss"]))
    params = dict(net.named_parameters())
    _grad_check(torch.tensor([float(params[n].grad.double().norm()) for n in fx["grad_names"]], dtype=torch.float64), fx, name)
    for k, v in fx["bn_running_checksum"].items():
        assert abs(float(net.state_dict()[k].double().sum()) - v) <= 1e-4 * max(abs(v), 1.0), k


@pytest.mark.gpu
def test_product_resnet50_imagenet_shape_fwd_bwd(gpu_device):
    """BASELINE.json configs[1] shape (224x224), reduced batch for the CPU oracle: every parameter gradient against the oracle."""
    from oracle.resnet import build
    from super_gradients_amd.training import models
    from super_gradients_amd.training.losses import CrossEntropyLoss

    torch.manual_seed(0)
    ref = build("resnet50", 1000).train()
    net = models.get("resnet50", num_classes=1000)
    net.load_state_dict(ref.state_dict(), strict=True)
    net.materialize(gpu_device).train()
    x = torch.randn(4, 3, 224, 224)
    y = torch.randint(0, 1000, (4,))
    lr = F.cross_entropy(ref(x), y)
    lr.backward()
    lh = CrossEntropyLoss()(net(x.to(gpu_device)), y.to(gpu_device))
    lh.backward()
    assert abs(float(lh) - float(lr)) <= 1e-4 * abs(float(lr))
    # the same modules in fp64 are the truth: at random init (batch 4, 7x7 maps in layer4) the fp32 gradients of BOTH paths carry
    # round-off amplified by the training-mode BatchNorms; bar = no further from fp64 than 3x the CPU fp32 path (whole-gradient
    # relative L2, and per tensor with a floor)
    import copy

    ref64 = copy.deepcopy(ref).double()
    ref64.zero_grad()
    F.cross_entropy(ref64(x.double()), y).backward()
    rp, tp64 = dict(ref.named_parameters()), dict(ref64.named_parameters())
    num_h = num_c = den = 0.0
    floor = 1e-3 * max(float(p.grad.norm()) for p in tp64.values())
    for n, p in net.named_parameters():
        t = tp64[n].grad
        eh, ec = float((p.grad.cpu().double() - t).norm()), float((rp[n].grad.double() - t).norm())
        num_h, num_c, den = num_h + eh ** 2, num_c + ec ** 2, den + float(t.norm()) ** 2
        sc = max(float(t.norm()), floor)
        assert eh / sc <= max(1e-3, 3.0 * ec / sc, 2e-2), f"{n}: hip {eh / sc:.2e} vs cpu fp32 {ec / sc:.2e} (relative L2 against fp64)"
    l2_h, l2_c = (num_h / den) ** 0.5, (num_c / den) ** 0.5
    assert l2_h <= max(1e-3, 3.0 * l2_c), f"whole-gradient L2 error vs fp64: hip {l2_h:.2e}, cpu fp32 {l2_c:.2e}"
    print(f"resnet50@224: gradient L2 error vs fp64: hip {l2_h:.2e}, cpu fp32 {l2_c:.2e}")


@pytest.mark.gpu
def test_product_resnet50_imagenet_shape_bs64_loss(gpu_device):
    """BASELINE.json configs[1] AT ITS OWN SIZE (64 x 3 x 224 x 224; the gradient test above runs batch 4 for the CPU oracle's backward):
    training-mode forward + cross-entropy against the oracle - logits within 1e-4 (relative, max-norm), loss within 1e-4."""
    from oracle.resnet import build
    from super_gradients_amd.training import models
    from super_gradients_amd.training.losses import CrossEntropyLoss

    torch.manual_seed(0)
    ref = build("resnet50", 1000).train()
    net = models.get("resnet50", num_classes=1000)
    net.load_state_dict(ref.state_dict(), strict=True)
    net.materialize(gpu_device).train()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(64, 3, 224, 224, generator=g)
    y = torch.randint(0, 1000, (64,), generator=g)
    torch.set_num_threads(min(64, torch.get_num_threads() * 4))
    with torch.no_grad():
        lo_r = ref(x)
        lo = net(x.to(gpu_device)).cpu()
    assert rel_err(lo, lo_r) <= 1e-4, f"logits @ bs64: {rel_err(lo, lo_r):.2e}"
    lr, lh = float(F.cross_entropy(lo_r, y)), float(CrossEntropyLoss()(lo.to(gpu_device), y.to(gpu_device)))
    assert abs(lh - lr) <= 1e-4 * abs(lr)
