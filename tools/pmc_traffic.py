"""Condense the FETCH_SIZE / WRITE_SIZE rocprofv3 passes into HBM bytes per igemm launch (profiles/igemm_traffic.json).

    python tools/pmc_traffic.py <fetch_pass_dir> <write_pass_dir> <out.json> [<pass dir with GRBM_GUI_ACTIVE> [<nms calls traced>]]
Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half the bytes of wide coalesced
streaming reads (TCC_EA0_RDREQ x 64 B for 128-B requests) -> x2; both counters are in KiB; WRITE_SIZE is taken as reported
(uncalibrated per the guide).  Per-launch = sum over all dispatches of the family / number of dispatches; families: forward / data gradient =
igemm_kernel + pconv_kernel, weight gradient = wgrad_kernel + wpatch_kernel (the SAME launches bench.py's `wgrad.launches_per_step` and
`wgrad.algorithmic_bytes_per_launch` count: round 5's file summed wgrad_kernel only - 27 of the 42 launches per step)."""
import csv
import glob
import json
import os
import sys

csv.field_size_limit(1 << 30)


def total(d, counter, match):
    s, n = 0.0, set()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            if match in r["Kernel_Name"] and r["Counter_Name"] == counter:
                s += float(r["Counter_Value"])
                n.add((r.get("Dispatch_Id"), r.get("Process_Id")))
    return s, len(n)


def clock(d):
    """Loaded shader clock of a pass that collected GRBM_GUI_ACTIVE: busy cycles of the (8) XCDs / kernel time, over the conv kernels."""
    cyc = ns = 0.0
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and any(m in r["Kernel_Name"] for m in ("igemm_kernel", "pconv_kernel", "wgrad_kernel", "wpatch_kernel")):
                cyc += float(r["Counter_Value"])
                ns += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return (cyc / 8.0 / ns) if ns else None  # GHz


def main():
    fd, wd, out = sys.argv[1:4]
    res = {}
    for name, matches in (("igemm", ("igemm_kernel", "pconv_kernel")), ("wgrad", ("wgrad_kernel", "wpatch_kernel")), ("nms", ("nms_",))):
        f = w = 0.0
        nf = nw = 0
        for m in matches:
            a, b = total(fd, "FETCH_SIZE", m)
            f, nf = f + a, nf + b
            a, b = total(wd, "WRITE_SIZE", m)
            w, nw = w + a, nw + b
        if nf == 0 and nw == 0:
            continue
        res[name] = {"fetch_KiB_reported": f, "write_KiB_reported": w, "dispatches": nf,
                     "bytes_per_launch": (2.0 * f * 1024 / max(nf, 1)) + (w * 1024 / max(nw, 1))}
    rec = {}
    if "igemm" in res:
        rec["bytes_per_launch"] = round(res["igemm"]["bytes_per_launch"])
    if "wgrad" in res:
        rec["wgrad_bytes_per_launch"] = round(res["wgrad"]["bytes_per_launch"])
    if "nms" in res:  # all kernels of the post-prediction call: bytes per CALL = sum over its launches / calls (argv[5] = number of calls traced)
        calls = int(sys.argv[5]) if len(sys.argv) > 5 else 1
        rec["nms_bytes_per_call"] = round((2.0 * res["nms"]["fetch_KiB_reported"] + res["nms"]["write_KiB_reported"]) * 1024 / calls)
    if len(sys.argv) > 4 and sys.argv[4] not in ("", "-"):
        ghz = clock(sys.argv[4])
        if ghz:
            rec["loaded_clock_ghz"] = round(ghz, 3)
    rec.update({"detail": res, "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `python bench.py --steps 1 --warmup 1` ({os.path.basename(fd.rstrip('/'))}, "
                                        f"{os.path.basename(wd.rstrip('/'))}); FETCH_SIZE x2 (gfx950), KiB -> bytes"})
    json.dump(rec, open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
