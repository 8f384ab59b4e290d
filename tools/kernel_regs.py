"""Register / LDS / occupancy report of every kernel in a .hip source (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.

    python tools/kernel_regs.py super_gradients_amd/csrc/conv.hip [filter-substring]
"""
import os
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "super_gradients_amd", "csrc"))
import build as product_build  # noqa: E402  (the product's per-file flags: the report must describe the code objects that ship)

base = os.path.basename(src)
extra = (["-ffp-contract=off"] if base in product_build.NO_CONTRACT else []) + product_build.EXTRA.get(base, [])
cmd = [product_build.HIPCC] + product_build.COMMON + extra + ["-Wno-pass-failed", "-Rpass-analysis=kernel-resource-usage", "-x", "hip", "-c", src, "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rec, rows = {}, []
for line in out.splitlines():
    m = re.search(r"remark: (?:\s*)([A-Za-z ]+?)(?: \[[^\]]*\])?: (.+?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        rec = {"name": v}
        rows.append(rec)
    else:
        rec[k] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*$", "", re.sub(r"^void ", "", name))
    if flt in name:
        print(f"{name:<70} vgpr {r.get('VGPRs', '?'):>4} agpr {r.get('AGPRs', '?'):>4} sgpr {r.get('SGPRs', '?'):>4} spill {r.get('VGPRs Spill', '?'):>3} "
              f"occ {r.get('Occupancy', r.get('Occupancy [waves/SIMD]', '?')):>2} lds {r.get('LDS Size', r.get('LDS Size [bytes/block]', '?')):>6}")
