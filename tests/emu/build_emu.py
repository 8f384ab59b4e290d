"""TEST INFRASTRUCTURE ONLY.  Builds tests/emu/_build/libsgx_emu.so: the SAME kernel sources as libsgx_hip.so,
compiled for the host against the HIP emulation in hip_emu.h (one fiber per HIP thread, workgroup by workgroup), so that
kernel logic can be checked against the oracle in a container without a GPU.  Never used by the product."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(HERE, "..", "..", "super_gradients_amd", "csrc"))
OUTDIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUTDIR, "libsgx_emu.so")
SOURCES = ["conv.hip", "wgrad_patch.hip", "bn.hip", "pool.hip", "se.hip", "loss.hip", "nms.hip", "optim.hip", "image.hip", "half.hip", "api.cpp"]
CXX = os.environ.get("EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")


def build(force=False, verbose=False):
    os.makedirs(OUTDIR, exist_ok=True)
    deps = [os.path.join(HERE, "hip_emu.h"), os.path.join(HERE, "hip_emu.cpp"), os.path.join(CSRC, "sgx_common.h"), os.path.join(CSRC, "conv_mma.h"), os.path.join(CSRC, "wgrad_patch.h"),
            os.path.join(CSRC, "..", "..", "include", "sgx_hip.h")]
    objs, procs = [], []
    for src in SOURCES + ["hip_emu.cpp"]:
        sp = os.path.join(HERE if src == "hip_emu.cpp" else CSRC, src)
        obj = os.path.join(OUTDIR, src.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in [sp] + deps):
            cmd = [CXX, "-DSGX_EMU", "-O1", "-g", "-std=c++17", "-fPIC", "-ffp-contract=off", "-pthread", "-I", HERE, "-I", CSRC, "-x", "c++", "-c", sp,
                   "-o", obj, "-Wno-unused-value"]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"emu compile failed on {src}")
    if force or procs or not os.path.exists(OUT):
        # -Bsymbolic: the emulation library's own sgx_* references must never bind to a product library loaded earlier in the process
        subprocess.check_call([CXX, "-shared", "-fPIC", "-pthread", "-Wl,-Bsymbolic", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
