"""NMS leg of bench.py alone, under both candidate selections (sgx_debug_set_nms_selection): boxes/s and ms per batch.

    python tools/nms_bench.py [--iters 200] [--selection 1|0|both]
Measurement tool: product library only (bench.nms_leg builds the inputs and the CPU comparison)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--selection", default="both")
    args = ap.parse_args()
    import torch

    import bench
    from super_gradients_amd._lib import lib

    dev = torch.device("cuda:0")
    for sel in ((1, 0, 1, 0) if args.selection == "both" else (int(args.selection),)):
        lib().sgx_debug_set_nms_selection(sel)
        r = bench.nms_leg(dev, iters=args.iters)
        print(json.dumps({"selection": "sampled" if sel else "exact three-pass", "boxes_per_s": r["value"], "ms_per_batch": r["ms_per_batch"], "candidates": r["candidates"],
                          "kept": r["kept"]}))
    lib().sgx_debug_set_nms_selection(1)


if __name__ == "__main__":
    main()
