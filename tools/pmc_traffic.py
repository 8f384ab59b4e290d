"""Condense the FETCH_SIZE / WRITE_SIZE rocprofv3 passes into HBM bytes per igemm launch (profiles/igemm_traffic.json).

    python tools/pmc_traffic.py <fetch_pass_dir> <write_pass_dir> <out.json>
Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half the bytes of wide coalesced
streaming reads (TCC_EA0_RDREQ x 64 B for 128-B requests) -> x2; both counters are in KiB; WRITE_SIZE is taken as reported
(uncalibrated per the guide).  Per-launch = sum over all igemm_kernel dispatches / number of dispatches."""
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


def main():
    fd, wd, out = sys.argv[1:4]
    res = {}
    for name, match in (("igemm", "igemm_kernel"), ("wgrad", "wgrad_kernel")):
        f, nf = total(fd, "FETCH_SIZE", match)
        w, nw = total(wd, "WRITE_SIZE", match)
        res[name] = {"fetch_KiB_reported": f, "write_KiB_reported": w, "dispatches": nf,
                     "bytes_per_launch": (2.0 * f * 1024 / max(nf, 1)) + (w * 1024 / max(nw, 1))}
    json.dump({"bytes_per_launch": round(res["igemm"]["bytes_per_launch"]), "wgrad_bytes_per_launch": round(res["wgrad"]["bytes_per_launch"]),
               "detail": res, "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `python bench.py --steps 1 --warmup 1` ({os.path.basename(fd.rstrip('/'))}, "
                                        f"{os.path.basename(wd.rstrip('/'))}); FETCH_SIZE x2 (gfx950), KiB -> bytes"}, open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
