"""Builds libsgx_hip.so (gfx950 device code + C-ABI host code) in-tree with hipcc.

    python super_gradients_amd/csrc/build.py            # incremental
    python super_gradients_amd/csrc/build.py --force

hipcc cross-compiles gfx950 code objects without a GPU.  The library is linked against libamdhip64.so.7;
when it is loaded into a process that already imported torch, the dynamic loader resolves that SONAME
to the runtime torch already loaded (torch/lib/libamdhip64.so has the same SONAME), so kernels run on
torch's streams and allocations - see super_gradients_amd/_lib.py.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "libsgx_hip.so")
SOURCES = ["conv.hip", "wgrad_patch.hip", "bn.hip", "pool.hip", "se.hip", "loss.hip", "nms.hip", "optim.hip", "image.hip", "half.hip", "api.cpp"]
# decisions in loss/nms must round like the CPU op-by-op arithmetic: no fma contraction there
NO_CONTRACT = {"loss.hip", "nms.hip"}
# No SLP vectorisation in ANY device code: -O3 turns adjacent fp32 operations (a float4's four lanes) into packed fp32 VALU (v_pk_add_f32,
# v_pk_fma_f32).  (1) Beside MFMAs a pair costs ~26 cycles (MI355X_MICROARCH.md: "an anti-lever beside MFMAs") - the reason conv.hip,
# wgrad_patch.hip and half.hip had the flag since round 5.  (2) Round 6 (tools/dot_race_probe.py): the results of packed fp32
# instructions are NOT REPRODUCIBLE while a weight-gradient kernel of another stream is resident on the chip - sgx_dot's dependent TwoSum
# chains moved their fp64 partial rows by 1e-9 (up to 2e-7) in 8 - 14 % of 4 000 calls beside wgrad_kernel / wpatch_kernel, never alone,
# never beside a forward convolution, a sweep or a rocBLAS GEMM, and never once the packed instructions were gone (0 of 12 000).  In the
# train step that was one ulp of a bottleneck's d alpha in ~0.5 % of the steps.  Bit-exact index work (nms.hip, loss.hip) must not depend on it.
EXTRA = {"loss.hip": ["-fno-vectorize"]}  # (atss_candidates_kernel: the LOOP vectoriser made 45 packed operations of its distance loop)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mcode-object-version=5", "-Wno-unused-result", "-fno-slp-vectorize"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hdrs = [os.path.join(HERE, h) for h in ("sgx_common.h", "conv_mma.h", "wgrad_patch.h")] + [os.path.join(HERE, "..", "..", "include", "sgx_hip.h")]
    objdir = os.path.join(HERE, "_obj")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(HERE, src)
        obj = os.path.join(objdir, src.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [sp] + hdrs + [os.path.abspath(__file__)]):  # (the flags live in this file: an edit here rebuilds everything)
            cmd = [HIPCC] + COMMON + (["-ffp-contract=off"] if src in NO_CONTRACT else []) + EXTRA.get(src, []) + ["-x", "hip", "-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if force or procs or _stale(OUT, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
