"""Condense a rocprofv3 --kernel-trace CSV into one short line per dispatch (what fits gpurun_out's transfer limit and what the launch-shape
analysis needs): kernel (template arguments kept, parameter list dropped), grid in workgroups, workgroup size, LDS bytes, registers, queue,
start / end in ns relative to the first dispatch.

    python tools/trace_compact.py <dir with *kernel_trace.csv> <out.csv>
"""
import csv
import glob
import os
import re
import sys

csv.field_size_limit(1 << 30)


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n[:110]


def main():
    d, out = sys.argv[1:3]
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            wg = int(r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or 1) * int(r.get("Workgroup_Size_Y") or 1) * int(r.get("Workgroup_Size_Z") or 1)
            grid = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0) * int(r.get("Grid_Size_Y") or 1) * int(r.get("Grid_Size_Z") or 1)
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id") or r.get("Stream_Id") or "?", short(r["Kernel_Name"]),
                         grid // max(wg, 1), wg, r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""), r.get("SGPR_Count", "")))
    rows.sort()
    t0 = rows[0][0] if rows else 0
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["start_ns", "end_ns", "queue", "kernel", "workgroups", "wg_size", "lds", "vgpr", "agpr", "sgpr"])
        for s, e, q, n, g, wgs, lds, v, a, sg in rows:
            w.writerow([s - t0, e - t0, q, n, g, wgs, lds, v, a, sg])
    print(f"{len(rows)} dispatches -> {out} ({os.path.getsize(out) / 1e6:.1f} MB)")


if __name__ == "__main__":
    main()
