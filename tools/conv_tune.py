"""Exhaustive tile search over the conv problems of one train step (measurement tool; uses sgx_debug_set_tiles).

    python tools/conv_tune.py [--model s] [--batch 32] [--size 640] [--out gpurun_out/conv_tune.txt]
For every distinct problem recorded by tools/conv_bench.py's recorder: time the heuristic choice and every legal override,
print the best and the gain.  Output feeds the tile heuristics in csrc/conv.hip (pick_tile_heuristic / wgrad_plan)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def build_table(agg, min_gain):
    """agg: {(kind, N, H, W, C, K, R, stride, pad): {(bm, bn, variant): calls x us per step}} with (0, 0, 0) = the built-in heuristic.
    -> (entries for _lib.load_conv_tuning, meta): the fastest configuration of every problem that beats the heuristic by min_gain."""
    entries, t_heur, t_tab = [], 0.0, 0.0
    for pk, cfgs in agg.items():
        cfg, t = min(cfgs.items(), key=lambda kv: kv[1])
        heur = cfgs[(0, 0, 0)]
        t_heur += heur
        if t < (1.0 - min_gain) * heur:
            kind, n, h, w, c, k_, r, stride, pad = pk
            entries.append(dict(kind=kind, N=n, H=h, W=w, C=c, K=k_, R=r, stride=stride, pad=pad, bm=cfg[0], bn=cfg[1], variant=cfg[2],
                                us_heuristic=round(heur, 1), us_table=round(t, 1)))
            t_tab += t
        else:
            t_tab += heur
    return entries, dict(ms_per_step_heuristic=round(t_heur / 1e3, 3), ms_per_step_table=round(t_tab / 1e3, 3), min_gain=min_gain)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="s", help="s / m / l (YOLO-NAS), resnet50 (use --batch 64 --size 224), ppyoloe_s ...")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--out", default=None)
    ap.add_argument("--wgrad", action="store_true", help="also search the weight-gradient tiles")
    ap.add_argument("--emit-table", default=None, help="write the per-problem winners (forward / data gradient) as a tuning table for "
                    "sgx_conv_tuning_load (super_gradients_amd/csrc/conv_tuning_gfx950.json is loaded by default when present)")
    ap.add_argument("--planes", action="store_true", help="replay with pre-split filter planes, as the step runs (data gradients from pre-transposed "
                    "weights), and also search variant 12 - the GEMM loop's filter fragments straight from the planes into registers")
    ap.add_argument("--keep-wgrad-from", default=None, help="a committed table whose weight-gradient entries are carried over (with --emit-table, without --wgrad)")
    ap.add_argument("--min-gain", type=float, default=0.02, help="a winner enters the table only if it beats the heuristic by this fraction")
    args = ap.parse_args()
    import torch

    import conv_bench as CB
    from super_gradients_amd import kernels as K
    from super_gradients_amd._lib import lib, load_conv_tuning

    os.environ["SGX_SIDE_STREAM"] = "0"
    dev = torch.device("cuda:0")
    load_conv_tuning([])  # measure the built-in heuristic, not a previously committed table
    rec = CB.record_problems(args.model, args.batch, args.size, dev)
    agg = {}  # (kind, N, H, W, C, K, R, stride, pad) -> {(bm, bn, variant): calls x us summed over the stride / option variants of the problem}

    def note(key, calls, cfg, t):
        d = agg.setdefault(tuple(key[:9]), {})
        d[cfg] = d.get(cfg, 0.0) + calls * t

    def timeit(fn):
        try:
            fn()
        except Exception:
            return float("inf")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e3 / args.iters

    lines = []
    tot_base = tot_best = 0.0
    for key, calls in rec.items():
        kind = key[0]
        K.filter_planes_scope(False)
        K.filter_planes_invalidate(None)  # (the previous problem's buffers are about to be freed)
        fn, flops = CB.make_runner(key, dev, planes=args.planes and kind != "wgrad")
        if args.planes:
            K.filter_planes_scope(True)
        lib().sgx_debug_set_tiles(0, 0, 0, 0, 0)
        K.clear_desc_cache()
        base = timeit(fn)
        best, best_cfg = base, "heuristic"
        if kind in ("fwd2", "dgrad2"):
            continue  # the two-source QARepVGG launches run the heuristic's tile (no overrides): nothing to search
        if kind in ("fwd", "dgrad"):
            note(key, calls, (0, 0, 0), base)
            stats = kind == "fwd" and key[11:][3]
            for bm in (64, 128):
                for bn in (32, 64, 96, 128):
                    lib().sgx_debug_set_tiles(bm, bn, 0, 0, 0)
                    K.clear_desc_cache()
                    t = timeit(fn)
                    note(key, calls, (bm, bn, 0), t)
                    if t < best:
                        best, best_cfg = t, f"bm={bm} bn={bn}"
            # the 16-deep loop (variant 7): the default is 32-deep slabs with one LDS buffer wherever C % 32 == 0
            lib().sgx_debug_set_variant(7)
            for bm in (0, 64, 128):
                for bn in ((0,) if bm == 0 else (32, 64, 96, 128)):
                    lib().sgx_debug_set_tiles(bm, bn, 0, 0, 0)
                    K.clear_desc_cache()
                    t = timeit(fn)
                    note(key, calls, (bm, bn, 7), t)
                    if t < best:
                        best, best_cfg = t, ("heuristic tile" if bm == 0 else f"bm={bm} bn={bn}") + " variant=7"
            # the pipelined bf16x3 loop (variant 6, round 5: two LDS buffers, two register stages, the split of slab k + 1 between the MFMAs of
            # slab k) exists for the 64x64 / 128x32 / 64x32 tiles; on a problem that does not run the bf16x3 32-deep loop it is the default kernel
            lib().sgx_debug_set_variant(6)
            for bm, bn in ((0, 0), (64, 64), (128, 32), (64, 32)):
                lib().sgx_debug_set_tiles(bm, bn, 0, 0, 0)
                K.clear_desc_cache()
                t = timeit(fn)
                note(key, calls, (bm, bn, 6), t)
                if t < best:
                    best, best_cfg = t, ("heuristic tile" if bm == 0 else f"bm={bm} bn={bn}") + " variant=6"
            # all slabs up front (variant 11, round 5: fp32 launches whose reduction is at most four 32-deep slabs)
            lib().sgx_debug_set_variant(11)
            for bm, bn in ((0, 0), (64, 64), (128, 32), (64, 32), (128, 64)):
                lib().sgx_debug_set_tiles(bm, bn, 0, 0, 0)
                K.clear_desc_cache()
                t = timeit(fn)
                note(key, calls, (bm, bn, 11), t)
                if t < best:
                    best, best_cfg = t, ("heuristic tile" if bm == 0 else f"bm={bm} bn={bn}") + " variant=11"
            if args.planes:  # variant 12: filter fragments from the planes into registers (tiles of one 32-filter block per wave)
                lib().sgx_debug_set_variant(12)
                for bm, bn in ((0, 0), (64, 64), (128, 32), (64, 32), (128, 64)):
                    lib().sgx_debug_set_tiles(bm, bn, 0, 0, 0)
                    K.clear_desc_cache()
                    t = timeit(fn)
                    note(key, calls, (bm, bn, 12), t)
                    if t < best:
                        best, best_cfg = t, ("heuristic tile" if bm == 0 else f"bm={bm} bn={bn}") + " variant=12"
                # variant 14 (round 6): the ping-pong loop - two tiles per 512-thread workgroup, staging and matrix phases half a period apart
                lib().sgx_debug_set_variant(14)
                for bm, bn in ((64, 64), (128, 32), (64, 32)):
                    lib().sgx_debug_set_tiles(bm, bn, 0, 0, 0)
                    K.clear_desc_cache()
                    t = timeit(fn)
                    note(key, calls, (bm, bn, 14), t)
                    if t < best:
                        best, best_cfg = t, f"bm={bm} bn={bn} variant=14"
            lib().sgx_debug_set_variant(0)
        elif args.wgrad:
            note(key, calls, (0, 0, 0), base)
            for bnk in (32, 64, 96, 128):
                for bj in (32, 64, 96, 128):
                    for split in (2048, 4096, 8192):
                        lib().sgx_debug_set_tiles(0, 0, bnk, bj, split)
                        K.clear_desc_cache()
                        t = timeit(fn)
                        note(key, calls, (bnk, bj, split), t)
                        if t < best:
                            best, best_cfg = t, f"bnk={bnk} bj={bj} split={split}"
        lib().sgx_debug_set_tiles(0, 0, 0, 0, 0)
        K.clear_desc_cache()
        tot_base += calls * base
        tot_best += calls * best
        lines.append((calls * (base - best), f"{kind:<6}{key[1:11]} x{calls:<3} heuristic {base:8.1f} us  best {best:8.1f} us ({flops / best / 1e6:6.1f} TF)  {best_cfg}"))
    lines.sort(key=lambda t: -t[0])
    text = "\n".join([f"# conv tile search: heuristic {tot_base / 1e3:.2f} ms/step -> best-per-problem {tot_best / 1e3:.2f} ms/step"] + [l for _, l in lines])
    K.filter_planes_scope(False)
    K.filter_planes_invalidate(None)
    print(text)
    if args.out:
        open(args.out, "w").write(text + "\n")
    if args.emit_table:
        import json

        entries, meta = build_table(agg, args.min_gain)
        if args.keep_wgrad_from and not args.wgrad:
            old = json.load(open(args.keep_wgrad_from))
            entries += [e for e in old.get("entries", []) if e.get("kind") == "wgrad"]
            meta["wgrad_entries_from"] = os.path.basename(args.keep_wgrad_from)
        meta["filter_planes"] = bool(args.planes)
        meta.update(model=args.model if args.model.startswith(("resnet", "ppyoloe")) else f"yolo_nas_{args.model}", batch=args.batch, size=args.size, conv_math=K.get_conv_math())
        json.dump(dict(meta=meta, entries=entries), open(args.emit_table, "w"), indent=1)
        print(f"# tuning table: {len(entries)} of {len(agg)} problems, {meta['ms_per_step_heuristic']} -> {meta['ms_per_step_table']} ms/step -> {args.emit_table}")

if __name__ == "__main__":
    main()
