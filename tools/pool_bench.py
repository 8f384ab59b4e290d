"""Max-pool forward / backward of the SPP's map (32 x 20 x 20 x 384, k = 5 / 9 / 13, stride 1) timed alone.   python tools/pool_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from super_gradients_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
n, h, w, c = [int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (32, 20, 20, 384))]
x = torch.randn(n, h, w, c, device=dev)
for k in (5, 9, 13):
    y, am = K.maxpool_fwd(x, k, 1, k // 2)
    dy = torch.randn_like(y)
    dx = torch.zeros_like(x)
    for name, fn in (("fwd", lambda: K.maxpool_fwd(x, k, 1, k // 2, out=y)), ("bwd", lambda: K.maxpool_bwd(dy, am, tuple(x.shape), k, 1, k // 2, out=dx)),
                     ("bwd+acc", lambda: K.maxpool_bwd(dy, am, tuple(x.shape), k, 1, k // 2, out=dx, accumulate=True))):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        e1.synchronize()
        print(f"k={k:2d} {name:8s} {e0.elapsed_time(e1) * 1e3 / 20:8.1f} us")
