"""Host logic of the measurement tools that can only run their measurements on the GPU box: table construction of tools/conv_tune.py and
its round trip through the library's tuning-table loader, the timeline analysis of tools/prof_summary.py on a synthetic kernel trace."""
import csv
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_conv_tune_table_round_trip(backend, tmp_path):
    import conv_tune
    from super_gradients_amd._lib import lib, load_conv_tuning

    agg = {
        ("fwd", 32, 80, 80, 64, 64, 3, 1, 1): {(0, 0, 0): 1200.0, (64, 64, 7): 1000.0, (128, 64, 0): 1100.0},      # 17 % faster: in
        ("dgrad", 32, 80, 80, 64, 64, 3, 1, 1): {(0, 0, 0): 900.0, (64, 64, 7): 895.0},                             # < 2 %: stays heuristic
        ("wgrad", 32, 40, 40, 96, 96, 3, 1, 1): {(0, 0, 0): 1000.0, (96, 128, 2048): 800.0, (64, 64, 8192): float("inf")},
        ("fwd", 32, 20, 20, 192, 192, 3, 1, 1): {(0, 0, 0): 400.0, (0, 0, 7): 300.0},                               # heuristic tile, variant 7 (16-deep loop)
    }
    entries, meta = conv_tune.build_table(agg, 0.02)
    assert {(e["kind"], e["bm"], e["bn"], e["variant"]) for e in entries} == {("fwd", 64, 64, 7), ("wgrad", 96, 128, 2048), ("fwd", 0, 0, 7)}
    assert meta["ms_per_step_heuristic"] == 3.5 and meta["ms_per_step_table"] == 3.0
    f = tmp_path / "table.json"
    json.dump(dict(meta=meta, entries=entries), open(f, "w"))
    try:
        assert load_conv_tuning(str(f)) == 3 and lib().sgx_conv_tuning_size() == 3
    finally:
        load_conv_tuning([])


def test_prof_timeline_on_synthetic_trace(tmp_path, capsys):
    import prof_summary

    rows, t = [], 1_000_000
    for i in range(20):
        rows.append(dict(Queue_Id=1, Kernel_Name=f"void igemm_kernel<{i % 2}>(IgemmParams)", Start_Timestamp=t, End_Timestamp=t + 10_000))
        if i % 4 == 0:  # a concurrent side-stream kernel covering half of the main one
            rows.append(dict(Queue_Id=2, Kernel_Name="void wgrad_kernel<64>(WgradParams)", Start_Timestamp=t + 5_000, End_Timestamp=t + 10_000))
        t += 10_000 + 3_000  # 3 us idle between main-stream kernels
    d = tmp_path / "trace"
    d.mkdir()
    with open(d / "x_kernel_trace.csv", "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    prof_summary.timeline(str(d))
    out = capsys.readouterr().out
    assert "25 dispatches" in out
    import re

    def ms(prefix):
        line = [l for l in out.splitlines() if l.startswith(prefix)][0]
        return float(re.search(r"([0-9.]+) ms", line).group(1))

    assert abs(ms("device idle") - 19 * 3e-3) < 1e-6          # 19 gaps of 3 us
    assert abs(ms("two or more") - 5 * 5e-3) < 1e-6           # 5 overlaps of 5 us


def test_host_overhead_null_library_covers_the_abi(tmp_path):
    """tools/host_overhead.py: the generated no-op library must export every entry point of include/sgx_hip.h (else binding it fails), and a
    train step of a small detector must run against it - i.e. nothing on the step's host path reads a kernel result back."""
    import ctypes
    import subprocess

    import host_overhead
    from super_gradients_amd import _lib

    so = host_overhead.build_null_library()
    lib = ctypes.CDLL(so)
    assert all(hasattr(lib, name) for name in _lib.PROTOTYPES)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_overhead.py"), "--steps", "2"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "host time per step without kernels" in out.stdout, out.stderr[-2000:]


def test_prof_counter_ratios_and_hbm_rate_on_synthetic_passes(tmp_path, capsys):
    """tools/prof_summary.py `pmc` (per-kernel busy ratios) and `hbm` (rate per kernel from the FETCH_SIZE and WRITE_SIZE passes) on
    hand-made counter files: one kernel, two dispatches of 1 ms; 8 XCDs x 2.0e6 cycles; the matrix pipe busy on a quarter of the
    SIMD-cycles, the LDS on half of the CU-cycles; 1 GiB fetched (reported as 0.5 GiB: the gfx950 correction doubles it), 0.5 GiB written."""
    import prof_summary

    def write(d, counters):
        os.makedirs(d)
        with open(os.path.join(d, "x_counter_collection.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=["Dispatch_Id", "Process_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"])
            w.writeheader()
            for disp in (1, 2):
                for name, v in counters.items():
                    w.writerow(dict(Dispatch_Id=disp, Process_Id=7, Kernel_Name="void igemm_kernel<64, 64>(IgemmParams)", Counter_Name=name, Counter_Value=v,
                                    Start_Timestamp=disp * 10_000_000, End_Timestamp=disp * 10_000_000 + 1_000_000))

    cyc = 2.0e6
    write(str(tmp_path / "sq"), {"GRBM_GUI_ACTIVE": 8 * cyc, "SQ_VALU_MFMA_BUSY_CYCLES": 0.25 * cyc * 1024, "SQ_LDS_IDX_ACTIVE": 0.5 * cyc * 256,
                                 "SQ_BUSY_CU_CYCLES": 0.75 * cyc * 256, "SQ_LDS_BANK_CONFLICT": 0.0, "SQ_WAVE_CYCLES": 1.0})
    prof_summary.pmc(str(tmp_path / "sq"))
    out = capsys.readouterr().out
    row = [l for l in out.splitlines() if l.startswith("igemm_kernel<64, 64>") and "0.250" in l]
    assert row, out
    cols = row[0].split()
    assert cols[-4:] == ["0.250", "0.500", "0.750", "0.0000"] and cols[-5] == "2.000"  # busy ratios; cycles per ns of kernel time
    gib = 1 << 30
    write(str(tmp_path / "rd"), {"FETCH_SIZE": 0.25 * gib / 1024})   # KiB per dispatch, as rocprofv3 reports it (half of the bytes moved)
    write(str(tmp_path / "wr"), {"WRITE_SIZE": 0.25 * gib / 1024})
    prof_summary.hbm(str(tmp_path / "rd"), str(tmp_path / "wr"))
    out = capsys.readouterr().out
    row = [l for l in out.splitlines() if l.startswith("igemm_kernel<64, 64>")][0].split()
    # 2 dispatches: read 2 x 0.25 GiB x 2 = 1.074 GB, written 0.537 GB, over 2 ms -> 0.81 TB/s
    assert row[-3:] == ["1.07", "0.54", "0.81"], row


def test_no_kernel_of_the_built_product_spills():
    """Round 6 (review item 6): round 5 shipped instantiations with 1-50 spilled dwords (one of them ran six times per step) while DESIGN
    said "no spills".  The check reads the metadata notes of the gfx950 code objects the build left in csrc/_obj (what was linked into
    libsgx_hip.so): every kernel, vgpr_spill_count == 0 and no private segment."""
    import os
    import sys

    import pytest

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import kernel_spills

    if not os.path.isdir(kernel_spills.DEFAULT_OBJDIR) or not os.path.exists(os.path.join(kernel_spills.LLVM, "clang-offload-bundler")):
        pytest.skip("no built objects / no LLVM tools here (the GPU box runs the prebuilt library)")
    bad, total = kernel_spills.check_built_objects(verbose=False)
    assert total > 250, f"only {total} kernels found in {kernel_spills.DEFAULT_OBJDIR}"
    assert not bad, f"kernels spilling to scratch: {bad}"
    # (round 6, DESIGN.md 11.12: packed fp32 VALU instructions do not repeat their results beside a weight-gradient kernel of another stream)
    assert not kernel_spills.check_packed_fp32(verbose=False), "packed fp32 instructions in the built device code"
