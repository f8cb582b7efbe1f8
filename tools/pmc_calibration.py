#!/usr/bin/env python
"""Calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on this repo's access patterns (kernels: csrc/experiments/pmc_calib.hip).

    python tools/pmc_calibration.py run                 # launches the four kernels of known traffic (run THIS under rocprofv3 --pmc ...)
    python tools/pmc_calibration.py report FETCH_DIR WRITE_DIR   # reads the two counter_collection.csv files, prints counter / known bytes

Sizes: streaming 64 Mi float4 (1 GiB, past the 256 MiB Infinity Cache) and 16 Mi float4 (256 MiB); records: 2 000 000 x 48 B = 96 MB read /
written once each in a random permutation (the blend kernels gather SplatRec like this; the backward writes its entry records like this)."""
import csv
import ctypes
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("calib_stream_read", 0, 1 << 26, 16), ("calib_gather48", 1, 2_000_000, 48), ("calib_stream_write", 2, 1 << 26, 16),
         ("calib_record_write48", 3, 2_000_000, 48)]


def run():
    import torch
    lib = ctypes.CDLL(os.path.join(ROOT, "gs_icp_slam_amd", "libgsicp_experiments.so"))
    lib.gsicp_exp_pmc_calib.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.gsicp_exp_pmc_calib.restype = ctypes.c_int
    dev = torch.device("cuda", 0)
    sink = torch.zeros(4, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for name, which, n, width in CASES:
        buf = torch.rand(n * width // 4, device=dev)
        perm = torch.randperm(n, device=dev).to(torch.int32) if which in (1, 3) else None
        torch.cuda.synchronize()
        assert lib.gsicp_exp_pmc_calib(which, n, ctypes.c_void_p(perm.data_ptr()) if perm is not None else None, ctypes.c_void_p(buf.data_ptr()),
                                       ctypes.c_void_p(sink.data_ptr()), 5, stream) == 0
        torch.cuda.synchronize()
        del buf, perm
    print("ran", [c[0] for c in CASES])


def means(d):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = next((c[0] for c in CASES if c[0] in row["Kernel_Name"]), None)
            if k is None:
                continue
            a = acc.setdefault((k, row["Counter_Name"]), [0.0, 0])
            a[0] += float(row["Counter_Value"]); a[1] += 1
    return {k: s / n for k, (s, n) in acc.items()}


def report(fetch_dir, write_dir):
    f, w = means(fetch_dir), means(write_dir)
    out = {"what": "rocprofv3 FETCH_SIZE / WRITE_SIZE (KiB per launch x 1024) over the KNOWN bytes each kernel must move; 5 launches each", "cases": {}}
    for name, which, n, width in CASES:
        known = n * width + (4 * n if which in (1, 3) else 0)
        fs, ws = f.get((name, "FETCH_SIZE")), w.get((name, "WRITE_SIZE"))
        out["cases"][name] = {"known_bytes": known, "FETCH_SIZE_bytes": None if fs is None else fs * 1024, "WRITE_SIZE_bytes": None if ws is None else ws * 1024,
                              "fetch_over_known": None if fs is None else round(fs * 1024 / known, 4),
                              "write_over_known": None if ws is None else round(ws * 1024 / known, 4)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "report":
        report(sys.argv[2], sys.argv[3])
    else:
        run()
