"""Where a patch-kernel workgroup's cycles go: phase stamps (s_memtime) of every workgroup of a few QARepVGG forward / data-gradient launches.

    python tools/pconv_timing.py            (needs a library built with -DSGX_PCONV_TIMING: tools/visits/r5_visit12.sh swaps one in)
Stamps (lane 0 of wave 0): 0 entry, 1 source state set up, 2 first chunk's loads issued (+ row offsets), 3 first chunk split and stored
(= its loads arrived), 4 barrier passed, 5 chunk 0's MFMAs issued, 6 second barrier passed, 7 chunk 1 split and stored, 8 chunk loop done,
9 epilogue done.  Measurement tool: product library only."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    from super_gradients_amd import kernels as K
    from super_gradients_amd._lib import lib

    try:
        setter = getattr(lib(), "sgx_debug_set_pconv_timing")
    except AttributeError:
        raise SystemExit("this library was not built with -DSGX_PCONV_TIMING")
    setter.restype = ctypes.c_int32
    setter.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda:0")
    names = ["setup", "issue+rowoff", "first data+split+store", "barrier", "MFMA chunk 0", "barrier 2", "chunk 1 wait+split+store", "rest of loop", "epilogue"]
    for (n, h, w, c, k, kind) in [(32, 160, 160, 32, 32, "fwd2"), (32, 80, 80, 64, 64, "fwd2"), (32, 40, 40, 96, 96, "fwd2"), (32, 160, 160, 32, 32, "dgrad2"),
                                  (32, 80, 80, 64, 64, "fwd")]:
        x = torch.randn(n, h, w, c, device=dev)
        wt = K.to_ohwi(torch.randn(k, c, 3, 3, device=dev) / (c * 9) ** 0.5)
        w1 = K.to_ohwi(torch.randn(k, c, 1, 1, device=dev) / c ** 0.5)
        b1 = torch.randn(k, device=dev)
        y = torch.randn(n, h, w, k, device=dev)
        if kind == "fwd2":
            fn = lambda: K.conv2d_fwd_dual(x, wt, w1, b1, stride=1)
        elif kind == "dgrad2":
            wtb = K.conv2d_wt_buffer(wt, dev)
            K.conv2d_transpose_weights(wt, wtb, stride=1, pad=1)
            w1t = w1.reshape(k, c).t().contiguous()
            ds = torch.randn(n, h, w, k, device=dev)
            fn = lambda: K.conv2d_bwd_data_dual(y, wt, wtb, ds, w1t, (n, h, w, c), stride=1, out=x)
        else:
            fn = lambda: K.conv2d_fwd(x, wt, out=y, stride=1, pad=1, stat_partials=True)
        nwg = 65536
        buf = torch.zeros(nwg, 16, dtype=torch.int64, device=dev)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        buf.zero_()
        assert setter(buf.data_ptr()) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        setter(None)
        t = buf.cpu()
        live = t[:, 9] > 0
        t = t[live].double()
        if t.shape[0] == 0:
            print(kind, (n, h, w, c, k), "no stamps (the launch did not run the patch kernel)")
            continue
        d = torch.stack([t[:, i + 1] - t[:, i] for i in range(9)], 1)
        if c // 16 < 2:
            d[:, 6] = 0
        if os.environ.get("PCONV_TIMING_SET") == "2" and float(t[:, 10].max()) > 0:
            e = [t[:, 1] - t[:, 11], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3], t[:, 5] - t[:, 4], t[:, 10] - t[:, 5], t[:, 9] - t[:, 10]]
            print("    two-output epilogue (means): sums %.0f | stage y %.0f | stores y %.0f | stage u %.0f | stores u %.0f | to the barrier %.0f | barrier + statistic rows %.0f"
                  % tuple(float(v.mean()) for v in e))
            continue
        if float(t[:, 10].max()) > 0:  # two-output epilogue: 8 -> 11 accumulator merge, 11 -> 10 sums + transposes + stores, 10 -> 9 last barrier + statistic rows
            e = torch.stack([t[:, 11] - t[:, 8], t[:, 10] - t[:, 11], t[:, 9] - t[:, 10]], 1)
            print("    epilogue split: merge %.0f, sums + transposes + stores %.0f, last barrier + statistic rows %.0f (means)" % tuple(float(v) for v in e.mean(0)))
        life = t[:, 9] - t[:, 0]
        span = float(t[:, 9].max() - t[:, 0].min())
        start = t[:, 0] - t[:, 0].min()
        print(f"{kind} {(n, h, w, c, k)}: {t.shape[0]} workgroups, launch {e0.elapsed_time(e1) * 1e3:.1f} us (events), stamps span {span:.0f} ticks; workgroup lifetime mean {float(life.mean()):.0f} "
              f"p10 {float(life.quantile(0.1)):.0f} p90 {float(life.quantile(0.9)):.0f} ticks; starts: p50 {float(start.quantile(0.5)):.0f} p90 {float(start.quantile(0.9)):.0f}")
        for i, nm in enumerate(names):
            col = d[:, i]
            print(f"    {nm:<28} mean {float(col.mean()):8.0f}  p10 {float(col.quantile(0.1)):8.0f}  p50 {float(col.quantile(0.5)):8.0f}  p90 {float(col.quantile(0.9)):8.0f}  ({100 * float(col.mean() / life.mean()):4.1f} % of the lifetime)")


if __name__ == "__main__":
    main()
