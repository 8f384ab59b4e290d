"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.pt by running the REAL reference source files
(/root/reference/src/super_gradients/..., unmodified, through oracle/ref_shim.py) on CPU fp32.

    python oracle/make_golden.py          # only works in the build container, where /root/reference exists

Fixtures (all inputs are regenerated from seeds by oracle/golden_util.py; only reference OUTPUTS are stored):
  yolo_nas_{s,m,l}.pt  reference YoloNAS (customizable_detector.py:30, yolo_nas_variants.py:75-146) with
                       deterministic_fill weights, train-mode forward on a seeded image batch: decoded boxes / scores,
                       raw logits / distribution logits, anchors, points, strides; reference PPYoloELoss (TAL and ATSS)
                       loss items; per-parameter gradient L2 norms and sums of the TAL loss; BN running-stat checksums.
  ppyoloe_s.pt         reference PPYoloE-S (pp_yolo_e/pp_yolo_e.py:95, csp_resnet.py, pan.py, pp_yolo_head.py), deterministic_fill weights:
                       train-mode raw outputs (fp32 + the same modules in fp64), PPYoloELoss (TAL / ATSS) items, per-parameter gradient
                       norms (fp32 / fp64), BN running-stat checksums, eval-mode decoded boxes / scores and raw outputs.
  resnet18_cifar.pt, resnet50.pt   reference CifarResNet / ResNet (classification_models/resnet.py) with deterministic_fill
                       weights, train-mode forward on a seeded batch: logits, mean cross-entropy, per-parameter gradient norms
                       (fp32 and the same modules in fp64), BN running-stat checksums.
  ppyoloe_loss.pt      reference PPYoloELoss on random head outputs: {ATSS,TAL} x {varifocal,focal} x {batched,
                       sequential}, with the reference unit test's own fixed target tensor and with a seeded target set
                       that contains an empty image, plus the all-empty case: loss, items, d loss / d logits, d loss / d distri.
  detection_metrics.pt reference compute_detection_matching (IoUMatching, crowd targets, per-class top-k, normalised targets) and
                       compute_detection_metrics (training/utils/detection_utils.py:880-1580) on the seeded cases of
                       tests/test_detection_metrics.py: matched / ignore flags per image, AP / precision / recall / F1 / best thresholds.
  post_prediction.pt   reference PPYoloEPostPredictionCallback.forward (post_prediction_callback.py:42-123) with
                       torchvision.ops.boxes.{nms,batched_nms} bound to oracle/nms.py (torchvision is not installed and
                       not vendored: the NMS arithmetic itself stays "parity unpinned", the code around it is pinned).
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import golden_util as G  # noqa: E402
from oracle import nms as onms  # noqa: E402
from oracle import ref_shim  # noqa: E402

MODEL_CASES = {"s": dict(batch=2, size=128), "m": dict(batch=1, size=128), "l": dict(batch=1, size=128)}


def _ref_anchors(hw, strides):
    ref_shim.install()
    from super_gradients.training.models.detection_models.pp_yolo_e.pp_yolo_head import generate_anchors_for_grid_cell

    feats = [torch.zeros(1, 1, h, w) for h, w in hw]
    return generate_anchors_for_grid_cell(feats, strides)


def make_model_fixture(variant):
    cfg = MODEL_CASES[variant]
    torch.manual_seed(0)
    net = ref_shim.build_reference_yolo_nas(variant, num_classes=80)
    G.deterministic_fill(net, seed=1)
    net.train()
    x = G.seeded_input(cfg["batch"], 3, cfg["size"], seed=2)
    targets = G.detection_targets(cfg["batch"], cfg["size"], seed=3, kmax=3, empty_last=False)
    out = net(x)
    (boxes, scores), (logits, distri, anchors, points, counts, strides) = out
    fx = dict(variant=variant, batch=cfg["batch"], size=cfg["size"], state_keys=list(net.state_dict().keys()),
              state_shapes=[tuple(v.shape) for v in net.state_dict().values()],
              boxes=boxes.detach().clone(), scores=scores.detach().clone(), logits=logits.detach().clone(), distri=distri.detach().clone(),
              anchors=anchors.clone(), points=points.clone(), counts=list(counts), strides=strides.clone(), targets=targets)
    # the same reference modules in fp64 (the "truth" the fp32 paths are judged against where fp32 round-off through ~60
    # training-mode BatchNorms exceeds the 1e-4 bar on its own)
    import copy

    net64 = copy.deepcopy(net).double()
    net64_eval = copy.deepcopy(net64)  # same running statistics as `net` after its single training-mode forward
    out64 = net64(x.double())
    (b64, s64), (l64, d64, *_rest) = out64
    fx.update(boxes_f64=b64.detach().clone(), scores_f64=s64.detach().clone(), logits_f64=l64.detach().clone(), distri_f64=d64.detach().clone())
    # fp64 gradients of the TAL loss: at seeded weights the loss gradient is dominated by a per-channel constant that the
    # training-mode BatchNorms annihilate, so the fp32 gradients (reference and ours alike) carry amplified round-off; the
    # product is judged against this truth relative to the reference's own fp32 deviation from it
    torch.set_default_dtype(torch.float64)  # the loss builds its DFL projection with the default dtype (ppyolo_loss.py:688-696)
    try:
        loss64, _ = ref_shim.reference_ppyolo_loss(num_classes=80, use_static_assigner=False)(out64, targets.double())
    finally:
        torch.set_default_dtype(torch.float32)
    loss64.backward()
    fx["grad_norms_f64"] = torch.tensor([float(p.grad.norm()) for n, p in net64.named_parameters() if p.grad is not None], dtype=torch.float64)
    # 64 seeded elements of every parameter gradient (fp64 truth here, the reference's fp32 values below): an element-wise check of the
    # product's gradients against the reference that a norm comparison cannot give (a permuted or sign-flipped gradient has the right norm)
    gs = torch.Generator().manual_seed(7)
    fx["grad_sample_index"] = {n: torch.randperm(p.numel(), generator=gs)[:64].clone() for n, p in net64.named_parameters() if p.grad is not None}
    fx["grad_samples_f64"] = {n: p.grad.reshape(-1)[fx["grad_sample_index"][n]].clone() for n, p in net64.named_parameters() if p.grad is not None}
    del net64
    net64_eval.eval()
    with torch.no_grad():
        _, (el64, ed64, *_r) = net64_eval(x.double())
    fx.update(eval_logits_f64=el64.clone(), eval_distri_f64=ed64.clone())
    del net64_eval
    for static in (False, True):
        crit = ref_shim.reference_ppyolo_loss(num_classes=80, use_static_assigner=static)
        loss, items = crit(out, targets)
        fx["loss_items_atss" if static else "loss_items_tal"] = items.detach().clone()
        if not static:
            net.zero_grad()
            loss.backward(retain_graph=True)
            names, norms, sums = [], [], []
            for n, p in net.named_parameters():
                if p.grad is None:
                    continue
                names.append(n)
                norms.append(float(p.grad.double().norm()))
                sums.append(float(p.grad.double().sum()))
            fx["grad_names"], fx["grad_norms"], fx["grad_sums"] = names, torch.tensor(norms, dtype=torch.float64), torch.tensor(sums, dtype=torch.float64)
            fx["grad_samples"] = {n: p.grad.reshape(-1)[fx["grad_sample_index"][n]].clone() for n, p in net.named_parameters() if p.grad is not None}
    bn = {k: float(v.double().sum()) for k, v in net.state_dict().items() if k.endswith("running_mean") or k.endswith("running_var")}
    fx["bn_running_checksum"] = bn
    # eval-mode forward returns the same 2-tuple (SURVEY 8c edge case)
    net.eval()
    with torch.no_grad():
        (eb, es), (el, ed, *_r) = net(x)
    fx["eval_boxes"], fx["eval_scores"], fx["eval_logits"], fx["eval_distri"] = eb.clone(), es.clone(), el.clone(), ed.clone()
    return fx


def make_ppyoloe_fixture(variant="s", batch=2, size=128):
    """Reference PPYoloE (CSPResNet + CSPPAN + PPYOLOEHead): train-mode raw 6-tuple, PPYoloELoss items, gradient norms (fp32 / fp64),
    BN running-stat checksums, eval-mode decoded + raw outputs."""
    import copy

    torch.manual_seed(0)
    net = ref_shim.build_reference_ppyoloe(variant, num_classes=80)
    G.deterministic_fill(net, seed=1)
    net.train()
    x = G.seeded_input(batch, 3, size, seed=2)
    targets = G.detection_targets(batch, size, seed=3, kmax=3, empty_last=False)
    net64 = copy.deepcopy(net).double()
    out = net(x)
    logits, distri, anchors, points, counts, strides = out
    fx = dict(variant=variant, batch=batch, size=size, state_keys=list(net.state_dict().keys()),
              state_shapes=[tuple(v.shape) for v in net.state_dict().values()], logits=logits.detach().clone(), distri=distri.detach().clone(),
              anchors=anchors.clone(), points=points.clone(), counts=list(counts), strides=strides.clone(), targets=targets)
    net64_eval = copy.deepcopy(net64)
    out64 = net64(x.double())
    fx.update(logits_f64=out64[0].detach().clone(), distri_f64=out64[1].detach().clone())
    torch.set_default_dtype(torch.float64)
    try:
        loss64, _ = ref_shim.reference_ppyolo_loss(num_classes=80, use_static_assigner=False)(out64, targets.double())
    finally:
        torch.set_default_dtype(torch.float32)
    loss64.backward()
    fx["grad_norms_f64"] = torch.tensor([float(p.grad.norm()) for n, p in net64.named_parameters()], dtype=torch.float64)
    for static in (False, True):
        loss, items = ref_shim.reference_ppyolo_loss(num_classes=80, use_static_assigner=static)(out, targets)
        fx["loss_items_atss" if static else "loss_items_tal"] = items.detach().clone()
        if not static:
            loss.backward(retain_graph=True)
            fx["grad_names"] = [n for n, p in net.named_parameters()]
            fx["grad_norms"] = torch.tensor([float(p.grad.double().norm()) for p in net.parameters()], dtype=torch.float64)
    fx["bn_running_checksum"] = {k: float(v.double().sum()) for k, v in net.state_dict().items() if k.endswith("running_mean") or k.endswith("running_var")}
    # eval on the seeded running statistics (a fresh copy: the training forward above updated them); fp64 twin for the three-way bar
    ev = ref_shim.build_reference_ppyoloe(variant, num_classes=80)
    G.deterministic_fill(ev, seed=1)
    ev.eval()
    ev64 = copy.deepcopy(ev).double()
    with torch.no_grad():
        (eb, es), (el, ed, *_r) = ev(x)
        (eb64, es64), (el64, ed64, *_r) = ev64(x.double())
    fx.update(eval_boxes=eb.clone(), eval_scores=es.clone(), eval_logits=el.clone(), eval_distri=ed.clone(), eval_logits_f64=el64.clone(),
              eval_distri_f64=ed64.clone(), eval_boxes_f64=eb64.clone())
    del net64_eval
    return fx


RESNET_CASES = {"resnet18_cifar": dict(cls="ResNet18Cifar", batch=8, size=32, classes=10), "resnet50": dict(cls="ResNet50", batch=4, size=64, classes=100)}


def make_resnet_fixture(name):
    import copy

    cfg = RESNET_CASES[name]
    net = ref_shim.reference_resnet(cfg["cls"], cfg["classes"])
    G.deterministic_fill(net, seed=4)
    net.train()
    x = torch.randn(cfg["batch"], 3, cfg["size"], cfg["size"], generator=torch.Generator().manual_seed(5))
    y = torch.randint(0, cfg["classes"], (cfg["batch"],), generator=torch.Generator().manual_seed(6))
    net64 = copy.deepcopy(net).double()
    logits = net(x)
    loss = torch.nn.functional.cross_entropy(logits, y)
    loss.backward()
    l64 = net64(x.double())
    loss64 = torch.nn.functional.cross_entropy(l64, y)
    loss64.backward()
    names = [n for n, p in net.named_parameters()]
    return dict(name=name, batch=cfg["batch"], size=cfg["size"], classes=cfg["classes"], state_keys=list(net.state_dict().keys()),
                state_shapes=[tuple(v.shape) for v in net.state_dict().values()], logits=logits.detach().clone(), logits_f64=l64.detach().clone(),
                loss=loss.detach().clone(), loss_f64=loss64.detach().clone(), labels=y, grad_names=names,
                grad_norms=torch.tensor([float(p.grad.double().norm()) for p in net.parameters()], dtype=torch.float64),
                grad_norms_f64=torch.tensor([float(p.grad.norm()) for p in net64.parameters()], dtype=torch.float64),
                bn_running_checksum={k: float(v.double().sum()) for k, v in net.state_dict().items() if k.endswith("running_mean") or k.endswith("running_var")})


def make_loss_fixture():
    cases = []
    sizes = [8, 4, 3]  # 89 anchors: keeps the fixture small; level 3 still holds the 9 anchors ATSS's per-level top-9 needs
    preds = G.synthetic_predictions(3, sizes, 80, 16, seed=5, make_anchors=_ref_anchors)
    tsets = {
        "reference_unit_test": G.REFERENCE_UNIT_TEST_TARGETS * torch.tensor([1, 1, 0.125, 0.125, 0.125, 0.125]),  # scaled into the 64-px canvas
        "seeded_with_empty_image": G.detection_targets(3, 64, seed=6, kmax=5, empty_last=True),
        "no_targets": torch.zeros(0, 6),
    }
    for tname, t in tsets.items():
        for static in (True, False):
            for vfl in (True, False):
                for batched in (True, False):
                    logits = preds[0].clone().requires_grad_(True)
                    distri = preds[1].clone().requires_grad_(True)
                    crit = ref_shim.reference_ppyolo_loss(num_classes=80, use_varifocal_loss=vfl, use_static_assigner=static, reg_max=16,
                                                          use_batched_assignment=batched)
                    loss, items = crit((None, (logits, distri) + tuple(preds[2:])), t)
                    loss.backward()
                    case = dict(targets=tname, static=static, vfl=vfl, batched=batched, loss=loss.detach().clone(), items=items.detach().clone())
                    if batched:  # the sequential path must reproduce the batched one (the reference's own unit test); its gradients are not stored
                        case.update(g_logits=logits.grad.clone(), g_distri=distri.grad.clone())
                    cases.append(case)
    return dict(sizes=sizes, batch=3, seed=5, target_sets=tsets, cases=cases)


def make_post_prediction_fixture():
    out = []
    for case in G.nms_cases():
        for multi_label in (True, False):
            for agnostic in (True, False):
                cb = ref_shim.reference_post_prediction_callback(onms.nms, onms.batched_nms, score_threshold=case["score_threshold"],
                                                                 nms_threshold=case["nms_threshold"], nms_top_k=case["nms_top_k"],
                                                                 max_predictions=case["max_predictions"], multi_label_per_box=multi_label,
                                                                 class_agnostic_nms=agnostic)
                res = cb(((case["boxes"], case["scores"]), None))
                out.append(dict(name=case["name"], multi_label=multi_label, class_agnostic=agnostic, rows=[r.clone() for r in res]))
    return out


def make_detection_metrics_fixture():
    import importlib

    T = importlib.import_module("tests.test_detection_metrics")
    recs = []
    for case in T.CASES:
        preds, t, c, size = T._case(case["seed"], normalized=case["normalized"], crowd=case.get("crowd", True))
        out, flat, met = T._reference(preds, t, c, size, case["top_k"], case["normalized"])
        recs.append(dict(case=case, matched=[o[0].clone() for o in out], ignore=[o[1].clone() for o in out], scores=flat[2].clone(), pred_cls=flat[3].clone(),
                         target_cls=flat[4].clone(), ap=met[0].clone(), precision=met[1].clone(), recall=met[2].clone(), f1=met[3].clone(),
                         best_score_threshold=met[5].clone(), best_per_class=met[6].clone()))
    return recs


def main():
    if not ref_shim.available():
        raise SystemExit("reference tree not found: make_golden.py runs in the build container only")
    os.makedirs(G.GOLDEN_DIR, exist_ok=True)
    torch.set_num_threads(8)
    only = set(sys.argv[1:])   # e.g. `python oracle/make_golden.py ppyoloe_s` regenerates just that fixture
    for v in MODEL_CASES:
        if only and f"yolo_nas_{v}" not in only:
            continue
        fx = make_model_fixture(v)
        torch.save(fx, os.path.join(G.GOLDEN_DIR, f"yolo_nas_{v}.pt"))
        print(v, "loss items TAL", fx["loss_items_tal"].tolist(), "ATSS", fx["loss_items_atss"].tolist())
    if not only or "ppyoloe_s" in only:
        fx = make_ppyoloe_fixture("s")
        torch.save(fx, os.path.join(G.GOLDEN_DIR, "ppyoloe_s.pt"))
        print("ppyoloe_s loss items TAL", fx["loss_items_tal"].tolist(), "ATSS", fx["loss_items_atss"].tolist())
    if only:
        return
    for name in RESNET_CASES:
        fx = make_resnet_fixture(name)
        torch.save(fx, os.path.join(G.GOLDEN_DIR, f"{name}.pt"))
        print(name, "CE", float(fx["loss"]), "fp64", float(fx["loss_f64"]))
    torch.save(make_loss_fixture(), os.path.join(G.GOLDEN_DIR, "ppyoloe_loss.pt"))
    torch.save(make_post_prediction_fixture(), os.path.join(G.GOLDEN_DIR, "post_prediction.pt"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.save(make_detection_metrics_fixture(), os.path.join(G.GOLDEN_DIR, "detection_metrics.pt"))
    for f in sorted(os.listdir(G.GOLDEN_DIR)):
        print(f, os.path.getsize(os.path.join(G.GOLDEN_DIR, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
