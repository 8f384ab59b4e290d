"""Where a predict() batch's wall time goes on the host: cProfile over N calls of the pipeline (YOLO-NAS-S, 32 x 480x640 uint8 images resident
in HBM, the reference's default COCO processing), top functions by cumulative time, then the end-to-end rate.
    python tools/predict_profile.py [--batches 20] [--fp32]"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=20)
    ap.add_argument("--fp32", action="store_true")
    a = ap.parse_args()
    from super_gradients_amd.training import models
    from super_gradients_amd.training.processing import default_yolo_nas_coco_processing_params

    dev = torch.device("cuda:0")
    net = models.get("yolo_nas_s", num_classes=80).materialize(dev)
    net.set_dataset_processing_params(**default_yolo_nas_coco_processing_params())
    g = torch.Generator().manual_seed(0)
    images = [torch.randint(0, 256, (480, 640, 3), generator=g, dtype=torch.uint8).to(dev) for _ in range(32)]
    pipe = net._get_pipeline(conf=0.01, fp16=not a.fp32)
    for _ in range(3):
        pipe(images, batch_size=32)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.batches):
        pipe(images, batch_size=32)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.batches
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.batches):
        pipe(images, batch_size=32)
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(32)
    print(s.getvalue()[:6000])
    print(f"end to end (unprofiled): {dt * 1e3:.3f} ms per 32-image batch = {32 / dt:.1f} images/s; host_post={os.environ.get('SGX_PREDICT_HOST_POST', '0')}")


if __name__ == "__main__":
    main()
