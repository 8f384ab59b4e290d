"""predict() throughput on one GPU: YOLO-NAS-S (random-init, 80 classes), the reference's default COCO processing (longest side -> 636, centre pad
to 640x640 with 114, /255), batches of 32 synthetic 480x640 uint8 images already resident in HBM.  Prints one JSON line with the end-to-end
rate and the split pre-processing launch / fused eval forward / NMS (HIP events).  Usage: python tools/predict_bench.py [--batches 10]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--model", default="s", choices=["s", "m", "l"])
    ap.add_argument("--family", default="yolo_nas", choices=["yolo_nas", "ppyoloe"], help="ppyoloe: PP-YOLOE with the reference's default COCO processing "
                    "(reverse channels, rescale to 640x640, normalise)")
    ap.add_argument("--fp32", action="store_true", help="predict(fp16=False): the fp32 path (default: the reference's default fp16=True -> bf16 kernels)")
    ap.add_argument("--tile", type=int, nargs=3, default=None, metavar=("BM", "BN", "KD"), help="force the bf16 conv kernel's tile / slab depth (0 = heuristic)")
    a = ap.parse_args()
    from super_gradients_amd.training import models
    from super_gradients_amd.training.processing import default_ppyoloe_coco_processing_params, default_yolo_nas_coco_processing_params

    dev = torch.device("cuda:0")
    net = models.get(f"{a.family}_{a.model}", num_classes=80).materialize(dev)
    if a.tile:
        from super_gradients_amd._lib import check, lib

        check(lib().sgx_hconv_debug_set_tile(*a.tile), "sgx_hconv_debug_set_tile")
    net.set_dataset_processing_params(**(default_ppyoloe_coco_processing_params() if a.family == "ppyoloe" else default_yolo_nas_coco_processing_params()))
    g = torch.Generator(device="cpu").manual_seed(0)
    images = [torch.randint(0, 256, (480, 640, 3), generator=g, dtype=torch.uint8).to(dev) for _ in range(a.batch)]
    pipe = net._get_pipeline(conf=0.01, fp16=not a.fp32)
    pipe(images, batch_size=a.batch)  # warm-up: fuses the model copy
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    split = [0.0, 0.0, 0.0]
    t0 = time.perf_counter()
    for _ in range(a.batches):
        ev[0].record()
        batch, metas = pipe.image_processor.preprocess_batch(images, device=dev)
        ev[1].record()
        with torch.no_grad():
            out = pipe.model(batch)
        ev[2].record()
        rows = pipe.post_prediction_callback(out, device=dev)
        ev[3].record()
        torch.cuda.synchronize()
        for i in range(3):
            split[i] += ev[i].elapsed_time(ev[i + 1])
    t_stage = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(a.batches):
        res = pipe(images, batch_size=a.batch)
    torch.cuda.synchronize()
    t_e2e = time.perf_counter() - t0
    n = a.batches * a.batch
    gmac = {"s": 16.939, "m": 47.093, "l": 64.493}[a.model] if a.family == "yolo_nas" else {"s": 8.7, "m": 24.9, "l": 55.0}[a.model]  # forward GMAC per 640 x 640 image (PP-YOLOE: the paper's FLOPs / 2)
    fwd_ms = split[1] / a.batches
    print(json.dumps({"metric": "images/s %s-%s predict() 480x640 -> 640x640, bs=%d, %s, fused blocks" % ("PP-YOLOE" if a.family == "ppyoloe" else "YOLO-NAS", a.model.upper(), a.batch, "bf16" if pipe.half else "fp32"),
                      "value": round(n / t_e2e, 1), "dtype": "bf16 (fp32 accumulate, fp32 prediction outputs)" if pipe.half else "fp32", "tile_override": a.tile,
                      "forward_tflops": round(2 * gmac * a.batch / fwd_ms, 1),
                      "end_to_end_includes": "device pre-processing, eval forward, NMS, D2H of the kept rows, host box post-processing, result objects",
                      "ms_per_batch": {"preprocess": round(split[0] / a.batches, 3), "forward": round(split[1] / a.batches, 3),
                                       "nms": round(split[2] / a.batches, 3), "stages_wall": round(1e3 * t_stage / a.batches, 3),
                                       "end_to_end": round(1e3 * t_e2e / a.batches, 3)},
                      "detections_first_image": len(res[0].prediction), "data": "synthetic uint8 images resident in HBM, random-init weights"}))


if __name__ == "__main__":
    main()
