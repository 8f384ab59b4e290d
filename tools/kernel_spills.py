"""Fail when a kernel of the BUILT product spills vector registers to scratch memory.

    python tools/kernel_spills.py [objdir]        # default: super_gradients_amd/csrc/_obj; exit 1 if any kernel spills

Reads what ships: every object's gfx950 code object is taken out of its .hip_fatbin section (llvm-objcopy + clang-offload-bundler) and the
kernels' metadata notes (.vgpr_spill_count / .sgpr_spill_count / .private_segment_fixed_size) are parsed - seconds, no recompile.  Round 5
shipped instantiations with 1-50 spilled dwords while DESIGN said "no spills"; __graft_entry__.build() and tests/test_tools.py run this
check now.  (tools/kernel_regs.py is the per-source report with occupancy; it recompiles the source.)
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
DEFAULT_OBJDIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "super_gradients_amd", "csrc", "_obj")
_FIELDS = {"vgpr_count": "vgpr", "sgpr_count": "sgpr", "vgpr_spill_count": "vgpr_spill", "sgpr_spill_count": "sgpr_spill",
           "private_segment_fixed_size": "scratch", "group_segment_fixed_size": "lds"}


def code_object_kernels(obj):
    """-> [{name, vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds}] of the gfx950 code object embedded in a hipcc host object."""
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "k.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
        if not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return []  # host-only object (api.cpp)
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    kernels, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s*(?:- )?\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip().strip("'")
        if k == "agpr_count":  # first key of a kernel's record (the keys of a record are sorted)
            cur = {"agpr": int(v)}
            kernels.append(cur)
        elif cur is not None:
            if k == "name":
                cur["name"] = v
            elif k in _FIELDS:
                cur[_FIELDS[k]] = int(v)
    return [k for k in kernels if "name" in k]


def packed_fp32_instructions(obj):
    """-> number of packed fp32 VALU instructions (v_pk_add_f32, v_pk_mul_f32, v_pk_fma_f32, ...) in the gfx950 code object of a hipcc host
    object.  Round 6: their results are not reproducible while a weight-gradient kernel of another stream is resident, and
    beside MFMAs they are slow - the build passes -fno-slp-vectorize (and -fno-vectorize where the loop vectoriser made them) and nothing of
    the product may contain one."""
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "k.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
        if not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return 0
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
    return len(re.findall(r"\bv_pk_\w+_f32\b", dis))


def check_packed_fp32(objdir=None, verbose=True):
    """-> [(object, count)] for every built object whose device code contains packed fp32 instructions"""
    objdir = objdir or DEFAULT_OBJDIR
    bad = [(o, n) for o in sorted(os.listdir(objdir)) if o.endswith(".o") for n in [packed_fp32_instructions(os.path.join(objdir, o))] if n]
    if verbose:
        print(f"kernel_spills: packed fp32 instructions in {objdir}: {bad or 'none'}")
    return bad


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"\(.*$", "", re.sub(r"^void ", "", n)) for n in out]


def check_built_objects(objdir=None, verbose=True):
    """-> (list of (object, kernel, vgpr_spill, sgpr_spill, scratch bytes) for every kernel that spills, number of kernels looked at)."""
    objdir = objdir or DEFAULT_OBJDIR
    bad, total = [], 0
    for o in sorted(os.listdir(objdir)):
        if not o.endswith(".o"):
            continue
        ks = code_object_kernels(os.path.join(objdir, o))
        total += len(ks)
        for k, name in zip(ks, demangle([k["name"] for k in ks])):
            # (SGPR spills are not counted: they park scalars in VGPR lanes - v_writelane / v_readlane, no memory; a kernel fails the check
            # when vector registers go to scratch MEMORY: vgpr_spill_count or a private segment)
            if k.get("vgpr_spill", 0) or k.get("scratch", 0):
                bad.append((o, name, k.get("vgpr_spill", 0), k.get("sgpr_spill", 0), k.get("scratch", 0)))
    if verbose:
        print(f"kernel_spills: {total} kernels in {objdir}, {len(bad)} with spills / scratch")
        for b in bad:
            print("  %s: %s  vgpr_spill %d sgpr_spill %d scratch %d B" % b)
    return bad, total


if __name__ == "__main__":
    bad, total = check_built_objects(sys.argv[1] if len(sys.argv) > 1 else None)
    packed = check_packed_fp32(sys.argv[1] if len(sys.argv) > 1 else None)
    sys.exit(1 if bad or packed or total == 0 else 0)
