"""Kernel-design lab: time a few representative convolution problems under several kernel variants, interleaved in ONE process
(rounds x variants, median), so that small deltas are comparable (cdna_hip_programming.md 5.4 rule 24).

    python tools/conv_lab.py --variants 0,5,11,12 [--tiles 64x64,128x64] [--problems fwd:32:80:80:64:64:3:1,...] [--rounds 5] [--iters 10]

variant = sgx_debug_set_variant code (0 = shipped kernel).  Measurement tool: product library only.
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DEFAULT = [
    "fwd:32:80:80:64:64:3:1", "fwd:32:160:160:32:32:3:1", "fwd:32:40:40:96:96:3:1", "fwd:32:80:80:192:384:3:2", "fwd:32:20:20:256:256:3:1",
    "fwd:32:80:80:64:64:1:1", "fwd:32:160:160:96:32:1:1", "fwd:32:40:40:96:96:1:1", "fwd:32:20:20:1536:768:1:1", "fwd:32:320:320:48:96:3:2",
    "dgrad:32:320:320:48:96:3:2", "dgrad:32:80:80:64:64:3:1", "wgrad:32:80:80:64:64:3:1", "wgrad:32:160:160:32:32:3:1", "wgrad:32:40:40:96:96:1:1",
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="0")
    ap.add_argument("--tiles", default="0x0")
    ap.add_argument("--math", default="fp32", help="comma list of fp32 | bf16x3 (kernels.set_conv_math) crossed with the variants / tiles")
    ap.add_argument("--problems", default=",".join(DEFAULT))
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--planes", default=None, help="comma list of 0 | 1: crossed with the other axes, pre-split filter planes off / on (the problem's "
                    "filters are split once, the step's scope is open for the whole run; data gradients then use the pre-transposed-weights entry point)")
    ap.add_argument("--ablate", default=None, help="comma list of ablation bit sets of a -DSGX_IGEMM_LAB build (sgx_debug_set_igemm_lab: 1 no global "
                    "loads, 2 no LDS stores, 4 no MFMAs, 8 no split, 16 no epilogue stores), crossed with the other axes; 0 = the whole kernel")
    ap.add_argument("--ldspad", default=None, help="comma list of dynamic-LDS pads in KB (sgx_debug_set_igemm_lds_pad) crossed with the other axes: an occupancy cap")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch

    from super_gradients_amd import kernels as K
    from super_gradients_amd._lib import lib

    dev = torch.device("cuda:0")
    variants = [int(v) for v in args.variants.split(",")]
    tiles = [tuple(int(a) for a in t.split("x")) for t in args.tiles.split(",")]
    maths = args.math.split(",")
    planes = [int(v) for v in args.planes.split(",")] if args.planes else [None]
    ablate = [int(v) for v in args.ablate.split(",")] if args.ablate else [None]
    if args.ldspad:  # (shares the ablation axis' slot: the two are not combined)
        ablate = [-1024 * int(v) - 1 for v in args.ldspad.split(",")]
    configs = [(v, t, m, (pl, ab)) for m in maths for t in tiles for v in variants for pl in planes for ab in ablate]
    lines = [f"{'problem':<34}" + "".join(f"{f'{m[:2]}{v}/{t[0]}x{t[1]}' + ('' if pl[0] is None else f'p{pl[0]}') + ('' if pl[1] is None else (f'L{-(pl[1] + 1) // 1024}' if pl[1] < 0 else f'a{pl[1]}')):>16}" for v, t, m, pl in configs) + "   (TFLOP/s, median of rounds; us below)"]
    for spec in args.problems.split(","):
        kind, n, h, w, c, k, r, s = spec.split(":")
        n, h, w, c, k, r, s = (int(a) for a in (n, h, w, c, k, r, s))
        pad = r // 2
        ho, wo = (h + 2 * pad - r) // s + 1, (w + 2 * pad - r) // s + 1
        x = torch.randn(n, h, w, c, device=dev)
        y = torch.randn(n, ho, wo, k, device=dev)
        wt = K.to_ohwi(torch.randn(k, c, r, r, device=dev) / (c * r * r) ** 0.5)
        dw = torch.zeros_like(wt)
        flops = 2.0 * n * ho * wo * k * c * r * r
        if kind in ("fwd2", "dgrad2"):  # the QARepVGG pair: 3x3 + 1x1 branch per launch (two outputs / two sources)
            w1 = K.to_ohwi(torch.randn(k, c, 1, 1, device=dev) / c ** 0.5)
            b1 = torch.randn(k, device=dev)
            wtb = K.conv2d_wt_buffer(wt, dev)
            K.conv2d_transpose_weights(wt, wtb, stride=s, pad=pad)
            w1t = w1.reshape(k, c).t().contiguous()
            ds = torch.randn(n, ho, wo, k, device=dev)
            flops += 2.0 * n * ho * wo * k * c
        keep = None
        if args.planes:
            import ctypes

            from super_gradients_amd import _lib

            wtb = K.conv2d_wt_buffer(wt, dev)
            K.conv2d_transpose_weights(wt, wtb, stride=s, pad=pad)
            raw = K.conv2d_transpose_jobs(wt, wtb, stride=s, pad=pad)
            recs = (_lib.WtransJob * (len(raw) // ctypes.sizeof(_lib.WtransJob))).from_buffer_copy(raw)
            filters = [(wt.data_ptr(), k, r * r, c)] + [(q.wt, q.C, q.T, q.K) for q in recs]
            if kind in ("fwd2", "dgrad2"):
                filters += [(w1.data_ptr(), k, 1, c), (w1t.data_ptr(), c, 1, k)]
            plan, total = K.filter_planes_plan(filters)
            if plan:
                buf = torch.empty(total, dtype=torch.uint8, device=dev)
                jobs, jdev = K.filter_planes_table(plan, buf)
                K.filter_planes_batch(jobs, jdev)
                keep = (buf, jobs, jdev)
            K.filter_planes_scope(True)
        if kind == "fwd2":
            fn = lambda: K.conv2d_fwd_dual(x, wt, w1, b1, stride=s)
        elif kind == "dgrad2":
            fn = lambda: K.conv2d_bwd_data_dual(y, wt, wtb, ds, w1t, (n, h, w, c), stride=s, out=x)
        elif kind == "fwd":
            fn = lambda: K.conv2d_fwd(x, wt, out=y, stride=s, pad=pad, stat_partials=True)
        elif kind == "dgrad" and args.planes:
            fn = lambda: K.conv2d_bwd_data_wt(y, wt, wtb, (n, h, w, c), stride=s, pad=pad, out=x)
        elif kind == "dgrad":
            fn = lambda: K.conv2d_bwd_data(y, wt, (n, h, w, c), stride=s, pad=pad, out=x)
        else:
            fn = lambda: K.conv2d_bwd_weight(x, y, dw, None, stride=s, pad=pad)
        res = {cfg: [] for cfg in configs}
        for _ in range(args.rounds):
            for cfg in configs:
                v, (bm, bn), m, (pl, ab) = cfg
                if pl is not None:
                    lib().sgx_debug_set_filter_planes(pl)
                if ab is not None and ab < 0:
                    lib().sgx_debug_set_igemm_lds_pad(-(ab + 1))
                elif ab is not None:
                    lib().sgx_debug_set_igemm_lab(ab)
                K.set_conv_math(m)
                lib().sgx_debug_set_variant(v)
                lib().sgx_debug_set_tiles(bm, bn, 0, 0, 0)
                K.clear_desc_cache()
                try:
                    fn()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(args.iters):
                        fn()
                    e1.record()
                    e1.synchronize()
                    res[cfg].append(e0.elapsed_time(e1) * 1e3 / args.iters)
                except Exception as ex:  # a variant that has no kernel for this tile
                    res[cfg].append(float("nan"))
        lib().sgx_debug_set_variant(0)
        lib().sgx_debug_set_tiles(0, 0, 0, 0, 0)
        if args.ldspad:
            lib().sgx_debug_set_igemm_lds_pad(0)
        elif args.ablate:
            lib().sgx_debug_set_igemm_lab(0)
        if args.planes:
            lib().sgx_debug_set_filter_planes(1)
            K.filter_planes_scope(False)
            K.filter_planes_invalidate(None)
        K.set_conv_math(K.DEFAULT_CONV_MATH)
        K.clear_desc_cache()
        med = {cfg: statistics.median(v) for cfg, v in res.items()}
        lines.append(f"{spec:<34}" + "".join(f"{flops / med[cfg] / 1e6:>16.1f}" for cfg in configs))
        lines.append(f"{'':<34}" + "".join(f"{med[cfg]:>16.1f}" for cfg in configs))
    text = "\n".join(lines)
    print(text)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
