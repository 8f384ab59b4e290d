"""TEST INFRASTRUCTURE.  Pins oracle/nms.c (the C restatement of torchvision's CPU NMS) to REAL torchvision output.

torchvision (requirements.txt:12, `torchvision>=0.10.0`) is a dependency the reference calls for the NMS arithmetic
(pp_yolo_e/post_prediction_callback.py:85,87) and is not installed in the build container, so the restatement's parity with it is
"unpinned" until this script has been run somewhere torchvision is importable:

    python oracle/make_nms_golden.py            # writes tests/golden/nms_torchvision.pt

It generates seeded cases (clustered boxes, engineered score ties and duplicate boxes, degenerate zero-area boxes, per-class cases on both
sides of torchvision's 4000-element batched_nms switch) and stores torchvision.ops.nms / batched_nms index outputs next to the inputs.
tests/test_kernels.py::test_nms_against_torchvision_fixture then holds BOTH the oracle and the HIP kernel to those indices (bit-exact) - and
skips, saying so, while the fixture does not exist.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "nms_torchvision.pt")


def cases():
    out = []
    for seed, (n, clusters, ncls) in enumerate([(300, 6, 1), (1000, 12, 1), (900, 10, 20), (1100, 10, 5), (64, 2, 3), (1, 1, 1), (0, 1, 1)]):
        g = np.random.RandomState(100 + seed)
        cen = g.uniform(60, 580, (max(clusters, 1), 2))
        c = cen[g.randint(0, max(clusters, 1), n)] + g.normal(0, 7, (n, 2))
        wh = g.uniform(15, 140, (n, 2))
        boxes = np.concatenate([c - wh / 2, c + wh / 2], -1).astype(np.float32)
        scores = (g.beta(0.5, 0.5, n) ** 2).astype(np.float32)
        if n >= 20:
            scores[1::7] = scores[0:-1:7][: scores[1::7].shape[0]]      # exact score ties
            boxes[2::11] = boxes[0:-2:11][: boxes[2::11].shape[0]]      # duplicated boxes
            boxes[5, 2:] = boxes[5, :2]                                  # zero-area box
        out.append(dict(boxes=torch.from_numpy(boxes), scores=torch.from_numpy(scores), classes=torch.from_numpy(g.randint(0, ncls, n)).long(),
                        iou=float([0.5, 0.65, 0.7, 0.6, 0.3, 0.5, 0.5][seed])))
    return out


def main():
    try:
        import torchvision
        from torchvision.ops import batched_nms, nms
    except ImportError:
        print("torchvision is not importable here: the fixture cannot be generated (NMS parity stays unpinned)", file=sys.stderr)
        return 1
    fx = dict(torchvision_version=torchvision.__version__, torch_version=torch.__version__, cases=[])
    for c in cases():
        keep = nms(c["boxes"], c["scores"], c["iou"]) if c["boxes"].shape[0] else torch.zeros(0, dtype=torch.long)
        keep_b = batched_nms(c["boxes"], c["scores"], c["classes"], c["iou"]) if c["boxes"].shape[0] else torch.zeros(0, dtype=torch.long)
        fx["cases"].append(dict(c, keep=keep.clone(), keep_batched=keep_b.clone()))
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    torch.save(fx, OUT)
    print(f"wrote {OUT}: {len(fx['cases'])} cases from torchvision {torchvision.__version__}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
