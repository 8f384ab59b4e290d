"""Condense rocprofv3 CSV output into the small text summaries that are committed under profiles/.

    python tools/prof_summary.py stats <dir>    per-kernel calls / total / average / share (from *kernel_stats.csv, or
                                                recomputed from *kernel_trace.csv) + register counts per kernel
    python tools/prof_summary.py pmc   <dir>    per-kernel sums of every collected counter (from *counter_collection.csv)
    python tools/prof_summary.py timeline <dir> where the wall clock of the traced run goes (from *kernel_trace.csv): device idle time,
                                                time with one / several kernels resident, per-queue busy time and gap histogram, the
                                                largest idle gaps with their neighbours - answers "launch-gap bound or throughput bound?"
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
    if "GRBM_GUI_ACTIVE" in names and "SQ_VALU_MFMA_BUSY_CYCLES" in names:
        # ratios per kernel: GRBM_GUI_ACTIVE is summed over the 8 XCDs -> cycles of kernel time = GUI / 8; the SQ counters are summed over the
        # chip: 256 CUs (SQ_BUSY_CU_CYCLES, SQ_LDS_IDX_ACTIVE) x 4 SIMDs (SQ_VALU_MFMA_BUSY_CYCLES)
        print()
        print("# per-kernel ratios (the kernels of this pass, 10 largest by time): matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs); "
              "LDS busy = SQ_LDS_IDX_ACTIVE / (cycles x 256 CUs); CU busy = SQ_BUSY_CU_CYCLES / (cycles x 256); cycles = GRBM_GUI_ACTIVE / 8")
        print(f"{'kernel':<92} {'calls':>7} {'dur_ms':>9} {'clock_GHz':>10} {'mfma_busy':>10} {'lds_busy':>9} {'cu_busy':>8} {'lds_conflict':>13}")
        for n, v in sorted(acc.items(), key=lambda kv: -dur[kv[0]])[:14]:
            cyc = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
            if cyc <= 0 or dur[n] <= 0:
                continue
            print(f"{n:<92} {len(calls[n]):>7} {dur[n] / 1e6:>9.3f} {cyc / dur[n]:>10.3f} {v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (cyc * 1024):>10.3f} "
                  f"{v.get('SQ_LDS_IDX_ACTIVE', 0.0) / (cyc * 256):>9.3f} {v.get('SQ_BUSY_CU_CYCLES', 0.0) / (cyc * 256):>8.3f} "
                  f"{v.get('SQ_LDS_BANK_CONFLICT', 0.0) / max(v.get('SQ_LDS_IDX_ACTIVE', 0.0), 1.0):>13.4f}")


def hbm(d_fetch, d_write, top=36):
    """HBM rate per kernel from the FETCH_SIZE pass and the WRITE_SIZE pass of the same command: (FETCH_SIZE x 2 [gfx950: counts 64 B per
    128-B request, MI355X_MICROARCH.md] + WRITE_SIZE) KiB / the kernel's time in the FETCH pass."""
    def load(d, counter):
        val, dur, calls = collections.defaultdict(float), collections.defaultdict(float), collections.defaultdict(set)
        for f in find(d, "*counter_collection.csv"):
            for r in csv.DictReader(open(f, newline="")):
                if r["Counter_Name"] != counter:
                    continue
                n = short(r["Kernel_Name"])
                val[n] += float(r["Counter_Value"])
                key = (r.get("Dispatch_Id"), r.get("Process_Id"))
                if key not in calls[n]:
                    dur[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                calls[n].add(key)
        return val, dur, calls
    fv, fd, fc = load(d_fetch, "FETCH_SIZE")
    wv, _, _ = load(d_write, "WRITE_SIZE")
    print("# HBM bytes and rate per kernel: (FETCH_SIZE x 2 + WRITE_SIZE) KiB over the kernel's summed duration (two rocprofv3 --pmc passes of the same command)")
    print(f"{'kernel':<92} {'calls':>7} {'dur_ms':>9} {'read_GB':>9} {'write_GB':>9} {'TB/s':>7}")
    for n in sorted(fv, key=lambda k: -fd[k])[:top]:
        rd, wr = fv[n] * 2 * 1024 / 1e9, wv.get(n, 0.0) * 1024 / 1e9
        if fd[n] > 0:
            print(f"{n:<92} {len(fc[n]):>7} {fd[n] / 1e6:>9.3f} {rd:>9.2f} {wr:>9.2f} {(rd + wr) / (fd[n] / 1e9) / 1e3:>7.2f}")


def timeline(d, top=12, steps=0):
    """steps > 0: only the last `steps` whole train steps of the trace (from the end of one adamw_kernel launch to the end of the last one) -
    the process start (library load, first-touch allocation, the oracle check) otherwise dominates every figure"""
    ev = []
    for f in find(d, "*kernel_trace.csv"):
        for r in csv.DictReader(open(f, newline="")):
            q = r.get("Queue_Id") or r.get("Stream_Id") or "?"
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), str(q), short(r["Kernel_Name"])))
    if not ev:
        print("# no *kernel_trace.csv under", d)
        return
    ev.sort()
    if steps > 0:
        marks = [e for s_, e, _, n in ev if n.startswith("adamw_kernel")]
        if len(marks) > steps:
            lo, hi = marks[-steps - 1], marks[-1]
            ev = [e for e in ev if e[0] >= lo and e[1] <= hi]
            print(f"# window: the last {steps} train steps ({(hi - lo) / 1e6 / steps:.3f} ms per step)")
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    wall = t1 - t0
    # sweep line over start/end points: time with 0 / 1 / >= 2 kernels resident
    pts = sorted([(s, 1) for s, _, _, _ in ev] + [(e, -1) for _, e, _, _ in ev])
    depth, last, by_depth = 0, t0, collections.Counter()
    for t, dlt in pts:
        by_depth[min(depth, 2)] += t - last
        last, depth = t, depth + dlt
    print(f"# source: kernel_trace csv(s) under {os.path.basename(d.rstrip('/'))}: {len(ev)} dispatches over {wall / 1e6:.3f} ms of wall clock")
    print(f"device idle (no kernel resident) {by_depth[0] / 1e6:9.3f} ms  {100.0 * by_depth[0] / wall:5.1f} %")
    print(f"exactly one kernel resident      {by_depth[1] / 1e6:9.3f} ms  {100.0 * by_depth[1] / wall:5.1f} %")
    print(f"two or more kernels resident     {by_depth[2] / 1e6:9.3f} ms  {100.0 * by_depth[2] / wall:5.1f} %")
    print(f"sum of kernel durations          {sum(e - s for s, e, _, _ in ev) / 1e6:9.3f} ms")
    # per queue: busy time, gaps between consecutive dispatches of the queue
    edges = [2e3, 5e3, 10e3, 50e3]
    labels = ["<2us", "2-5us", "5-10us", "10-50us", ">50us"]
    print(f"\n{'queue':<12} {'kernels':>8} {'busy_ms':>9} {'gap_ms':>9}  gaps: " + " ".join(f"{l:>8}" for l in labels))
    byq = collections.defaultdict(list)
    for e in ev:
        byq[e[2]].append(e)
    for q, es in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e - s for s, e, _, _ in es)
        hist, gap_total, prev_end = [0] * 5, 0, None
        for s, e, _, _ in es:
            if prev_end is not None and s > prev_end:
                g = s - prev_end
                gap_total += g
                hist[sum(g >= x for x in edges)] += 1
            prev_end = e if prev_end is None else max(prev_end, e)
        print(f"{q:<12} {len(es):>8} {busy / 1e6:>9.3f} {gap_total / 1e6:>9.3f}        " + " ".join(f"{h:>8}" for h in hist))
    # largest device-idle gaps with their neighbours
    gaps, cur_end, cur_name = [], ev[0][1], ev[0][3]
    for s, e, _, n in ev[1:]:
        if s > cur_end:
            gaps.append((s - cur_end, cur_name, n))
        if e > cur_end:
            cur_end, cur_name = e, n
    print(f"\n{len(gaps)} device-idle gaps, {sum(g for g, _, _ in gaps) / 1e6:.3f} ms in total; the largest:")
    for g, a, b in sorted(gaps, key=lambda t: -t[0])[:top]:
        print(f"  {g / 1e3:9.1f} us  after {a[:60]:<60} before {b[:60]}")
    # which kernels sit next to the idle time (sum of the gap that FOLLOWS each kernel class)
    after = collections.Counter()
    for g, a, _ in gaps:
        after[a] += g
    print("\nidle time by the kernel it follows:")
    for n, g in after.most_common(top):
        print(f"  {g / 1e6:9.3f} ms  {n[:90]}")


def neighbours(d, pattern, steps=4):
    """Where do the dispatches of the kernels matching `pattern` sit: per queue, which kernel precedes / follows them on that queue, over the
    last `steps` train steps - to attribute anonymous runtime kernels (copyBuffer, fillBuffer) to the host code that issued them."""
    ev = []
    for f in find(d, "*kernel_trace.csv"):
        for r in csv.DictReader(open(f, newline="")):
            q = r.get("Queue_Id") or r.get("Stream_Id") or "?"
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), str(q), short(r["Kernel_Name"])))
    ev.sort()
    marks = [e for s_, e, _, n in ev if n.startswith("adamw_kernel")]
    if len(marks) > steps:
        lo, hi = marks[-steps - 1], marks[-1]
        ev = [e for e in ev if e[0] >= lo and e[1] <= hi]
    byq = collections.defaultdict(list)
    for e in ev:
        byq[e[2]].append(e)
    ctx = collections.Counter()
    for q, lst in byq.items():
        for i, e in enumerate(lst):
            if re.search(pattern, e[3]):
                prev = lst[i - 1][3] if i else "-"
                nxt = lst[i + 1][3] if i + 1 < len(lst) else "-"
                ctx[(q, prev[:60], nxt[:60])] += 1
    print(f"# dispatches matching {pattern!r} over the last {steps} steps, by (queue, previous kernel, next kernel)")
    for (q, a, b), n in ctx.most_common(40):
        print(f"{n / steps:>7.1f}/step  q{q}  after {a:<60}  before {b}")


if __name__ == "__main__":
    if sys.argv[1] == "neighbours":
        neighbours(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 4)
    elif sys.argv[1] == "hbm":
        hbm(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "timeline":
        timeline(sys.argv[2], steps=int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    else:
        {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2])
