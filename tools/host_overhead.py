"""Host-side cost of one train step with the GPU taken out of the picture (development aid, runs on a GPU-less box).

Every entry point of include/sgx_hip.h is replaced by a C function that returns immediately (a generated `libsgx_null.so`: same symbols,
same ctypes marshalling, no work), tensors live on the CPU, and the bench.py step (forward, PPYoloELoss, backward, AdamW, EMA) runs at a toy
size.  What is left is exactly the per-step Python + ctypes + allocator time that bench.py reports as `host_enqueue_ms_per_step` minus the HIP
launch cost itself - the part that can be profiled and shaved without a GPU.  Results computed by the null kernels are garbage; nothing reads
them.

    python tools/host_overhead.py [--steps 20] [--profile] [--model s]
"""
import argparse
import cProfile
import ctypes
import os
import pstats
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build_null_library():
    from super_gradients_amd import _lib

    lines = ["#include <stdint.h>", "static const char empty[1] = {0};"]
    for name, (res, _args) in _lib.PROTOTYPES.items():
        if name == "sgx_conv2d_transpose_jobs":  # one record per convolution so that the batched launch has a table
            lines.append("int64_t sgx_conv2d_transpose_jobs(void* d, void* w, void* wt, int64_t b, void* jobs, int32_t cap, int32_t* n) { *n = 1; return 0; }")
        elif res is ctypes.c_char_p:
            lines.append(f"const char* {name}() {{ return empty; }}")
        elif "workspace" in name:
            lines.append(f"int64_t {name}() {{ return 4096; }}")
        elif name.endswith("_blocks") or name.endswith("_size"):
            lines.append(f"int32_t {name}() {{ return 1; }}")
        else:
            lines.append(f"int64_t {name}() {{ return 0; }}")
    d = tempfile.mkdtemp(prefix="sgx_null_")
    src, so = os.path.join(d, "null.c"), os.path.join(d, "libsgx_null.so")
    with open(src, "w") as f:
        f.write("\n".join(lines) + "\n")
    subprocess.run(["gcc", "-O1", "-shared", "-fPIC", "-Wno-implicit-function-declaration", "-o", so, src], check=True)
    return so


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--model", default="s")
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    import torch

    from super_gradients_amd import _lib

    _lib._LIB = _lib.bind(ctypes.CDLL(build_null_library()))
    _lib._TEST_HOST_MODE = True
    from super_gradients_amd.training import models
    from super_gradients_amd.training.losses import PPYoloELoss
    from super_gradients_amd.training.utils.ema import ModelEMA
    from super_gradients_amd.training.utils.optimizers import ArenaAdamW
    from util import synthetic_targets

    torch.manual_seed(0)
    dev = torch.device("cpu")
    net = models.get(f"yolo_nas_{a.model}", num_classes=80)
    net.materialize(dev)
    net.train()
    crit = PPYoloELoss(num_classes=80, use_static_assigner=False)
    opt = ArenaAdamW(net, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, zero_weight_decay_on_bias_and_bn=True)
    ema = ModelEMA.from_params(net, decay=0.9997, decay_type="threshold")
    x = torch.rand(2, 3, 64, 64)
    targets = synthetic_targets(2, seed=0, size=64)
    state = {"step": 0}

    def step():
        out = net(x)
        loss, _ = crit(out, targets)
        loss.backward()
        opt.step(grad_scale=None)
        opt.zero_grad()
        ema.update(net, state["step"], 100000)
        state["step"] += 1

    for _ in range(3):
        step()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    dt = (time.perf_counter() - t0) / a.steps
    print(f"host time per step without kernels: {1e3 * dt:.2f} ms  (YOLO-NAS-{a.model.upper()}, {a.steps} steps)")
    if a.profile:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(a.steps):
            step()
        pr.disable()
        st = pstats.Stats(pr)
        st.sort_stats("tottime").print_stats(a.top)


if __name__ == "__main__":
    main()
