"""Condense rocprofv3 CSV output into the small text summaries that are committed under profiles/.

    python tools/prof_summary.py stats <dir>    per-kernel calls / total / average / share (from *kernel_stats.csv, or
                                                recomputed from *kernel_trace.csv) + register counts per kernel
    python tools/prof_summary.py pmc   <dir>    per-kernel sums of every collected counter (from *counter_collection.csv)
"""
import collections
import csv
import glob
import os
import re
import sys

csv.field_size_limit(1 << 30)


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)  # drop the argument list
    return name if len(name) <= 90 else name[:87] + "..."


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


def stats(d):
    rows = collections.OrderedDict()
    regs = {}
    traces = find(d, "*kernel_trace.csv")
    for f in traces:
        for r in csv.DictReader(open(f, newline="")):
            n = short(r["Kernel_Name"])
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            e = rows.setdefault(n, [0, 0, 1 << 62, 0])
            e[0] += 1
            e[1] += dur
            e[2] = min(e[2], dur)
            e[3] = max(e[3], dur)
            regs[n] = (r.get("VGPR_Count", "?"), r.get("Accum_VGPR_Count", "?"), r.get("SGPR_Count", "?"), r.get("LDS_Block_Size", "?"),
                       r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")))
    if not rows:
        for f in find(d, "*kernel_stats.csv"):
            for r in csv.DictReader(open(f, newline="")):
                rows[short(r["Name"])] = [int(r["Calls"]), int(r["TotalDurationNs"]), int(r["MinNs"]), int(r["MaxNs"])]
    tot = sum(e[1] for e in rows.values()) or 1
    print(f"# source: {len(traces)} kernel_trace csv(s) under {os.path.basename(d.rstrip('/'))}; total kernel time {tot / 1e6:.3f} ms over {sum(e[0] for e in rows.values())} dispatches")
    print(f"{'kernel':<92} {'calls':>7} {'total_ms':>10} {'avg_us':>9} {'min_us':>8} {'max_us':>9} {'share%':>7}  vgpr/agpr/sgpr/lds/wg")
    for n, e in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        rg = "/".join(str(x) for x in regs.get(n, ()))
        print(f"{n:<92} {e[0]:>7} {e[1] / 1e6:>10.3f} {e[1] / e[0] / 1e3:>9.2f} {e[2] / 1e3:>8.2f} {e[3] / 1e3:>9.2f} {100.0 * e[1] / tot:>7.2f}  {rg}")


def pmc(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    files = find(d, "*counter_collection.csv")
    for f in files:
        for r in csv.DictReader(open(f, newline="")):
            n = short(r["Kernel_Name"])
            acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (r.get("Dispatch_Id"), r.get("Process_Id"))
            if key not in calls[n] and r.get("End_Timestamp") and r.get("Start_Timestamp"):
                dur[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            calls[n].add(key)
    names = sorted({c for v in acc.values() for c in v})
    print(f"# source: {len(files)} counter_collection csv(s) under {os.path.basename(d.rstrip('/'))}; values are SUMS over all dispatches of the kernel (raw counter units; "
          "FETCH_SIZE/WRITE_SIZE in KiB as rocprofv3 reports them - see MI355X_MICROARCH.md for the gfx950 x2 read correction)")
    print(f"{'kernel':<92} {'calls':>7} {'dur_ms':>9} " + " ".join(f"{c:>22}" for c in names))
    tot = collections.defaultdict(float)
    for n, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].values())):
        print(f"{n:<92} {len(calls[n]):>7} {dur[n] / 1e6:>9.3f} " + " ".join(f"{v.get(c, 0.0):>22.1f}" for c in names))
        for c in names:
            tot[c] += v.get(c, 0.0)
    print(f"{'TOTAL':<92} {sum(len(s) for s in calls.values()):>7} {sum(dur.values()) / 1e6:>9.3f} " + " ".join(f"{tot[c]:>22.1f}" for c in names))


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2])
