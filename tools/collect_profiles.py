#!/usr/bin/env python
"""Turn gpurun_out/<tag>/ (written by tools/capture_profiles.sh on the GPU box) into the tracked files profiles/<tag>_*.

  <tag>_bench.json, _bench_tum.json, _bench_eager.json, _bench_under_rocprof.json   the bench lines
  <tag>_rocprofv3_kernel_stats.csv    rocprofv3 --kernel-trace --stats, kernel names shortened
  <tag>_rocprofv3_pmc_hbm_traffic.csv, <tag>_pmc_traffic.json   FETCH_SIZE / WRITE_SIZE (KiB -> bytes) per launch
  <tag>_rocprofv3_pmc_sq.csv          SQ_* counters per launch (mean)
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE_OF = {"blend_backward_strip_kernel": "blend_backward", "blend_backward_tile_kernel": "blend_backward", "blend_forward_strip_kernel": "blend_forward",
            "entry_sum_kernel": "entry_grad_sum", "preprocess_backward_kernel": "preprocess_backward",
            "preprocess_kernel": "preprocess", "tile_scan_lpt_kernel": "tile_scan_lpt"}


def short(name):
    name = name.strip().strip('"').replace("(anonymous namespace)::", "").replace("gsicp::", "")
    name = re.sub(r"^void\s+", "", name)
    depth, cut = 0, len(name)
    for i, ch in enumerate(name):          # drop the argument list: first "(" outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    name = name[:cut].strip()
    if "<" in name and not name.startswith("tile_sort_kernel"):
        head, tail = name.split("<", 1)
        if head.startswith("at::") or head.startswith("rocprim"):
            m = re.search(r"(\w+Functor|\w+_functor|\w+Op)\b", tail)
            name = head + ("<" + m.group(1) + ">" if m else "")
        else:
            name = head
    return name.replace(",", ";")


def find(d, pattern):
    hits = sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True))
    return hits[0] if hits else None


def last_json_line(path):
    if not os.path.exists(path):
        return None
    txt = open(path).read()
    for line in reversed(txt.splitlines()):
        if line.startswith("{") and line.rstrip().endswith("}"):
            try:
                return json.loads(line)
            except ValueError:
                break
    try:
        return json.loads(txt)          # an indented (multi-line) document
    except ValueError:
        return None


def pmc_means(d):
    """kernel -> counter -> mean value per launch"""
    f = find(d, "*counter_collection.csv")
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    if not f:
        return {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            a = acc[short(row["Kernel_Name"])][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return {k: {c: s / n for c, (s, n) in v.items()} for k, v in acc.items()}


def main():
    src_tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    tag = sys.argv[2] if len(sys.argv) > 2 else src_tag          # collect_profiles.py <gpurun_out subdir> [<profiles prefix>]
    src = os.path.join(ROOT, "gpurun_out", src_tag)
    dst = os.environ.get("GSICP_PROFILES_DST") or os.path.join(ROOT, "profiles")     # on the GPU box: a directory under gpurun_out/ (only that comes back)
    os.makedirs(dst, exist_ok=True)
    for name in ("bench", "bench_tum", "bench_basin", "bench_eager", "bench_under_rocprof", "reference_run_replica", "reference_run_tum_shaped",
                 "reference_run_unlimit400", "reference_run_limit30_300", "reference_run_tum_layout60", "bench_gpus2_gloo_one_gpu",
                 "mfma_cov_experiment", "bench_mapper_only", "bench_tracker_only", "bench_force_collectives", "rccl_graph_probe",
                 "bench_pair_survey", "bench_pair_basin", "bench_driver_cmd", "tracker_vs_map", "reference_run_fused_unlimit400", "reference_run_fused_limit30_300",
                 "reference_run_unlimit1500", "reference_run_fused_unlimit1500", "bench_trained_leg", "fused_pacing_sweep", "fused_pacing_sweep_v2",
                 "bench_mapper_only_emit_walk", "bench_mapper_only_legacy_backward", "bench_mapper_only_inkernel_bump"):
        if name == "bench_trained_leg" and not os.path.exists(os.path.join(src, name + ".json")):
            continue
        j = last_json_line(os.path.join(src, name + ".json"))
        if j is not None:
            json.dump(j, open(os.path.join(dst, f"{tag}_{name}.json"), "w"), indent=1)
            if "value" in j:
                print(name, j["value"], j["unit"], j["ms_per_step"], "ms/step")
    for name in ("pytest_gpu_final.log", "slam_demo.txt", "reference_call_trace.json", "reference_call_trace_fused.json", "tracker_latency_survey.txt", "tracker_latency_map300k.txt",
                 "map_quality_curve.json", "scale_coverage.json", "pmc_calibration.json"):
        if os.path.exists(os.path.join(src, name)):
            import shutil
            shutil.copy(os.path.join(src, name), os.path.join(dst, f"{tag}_{name}"))
    for sub, suffix in (("kt", ""), ("kt_mapper", "_mapper_only"), ("kt_tracker", "_tracker_only"), ("kt_trained", "_trained_map"),
                        ("kt_mapper_emit_walk", "_mapper_only_emit_walk"), ("kt_mapper_legacy", "_mapper_only_legacy_backward"),
                        ("kt_mapper_inkernel_bump", "_mapper_only_inkernel_bump")):
        ks = find(os.path.join(src, sub), "*kernel_stats.csv")
        if ks:
            rows = list(csv.DictReader(open(ks)))
            with open(os.path.join(dst, f"{tag}_rocprofv3_kernel_stats{suffix}.csv"), "w") as fh:
                fh.write("kernel,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
                for r in rows:
                    fh.write(",".join([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"],
                                       r["MaxNs"], r["StdDev"]]) + "\n")
            print("kernel stats" + suffix + ":", len(rows), "kernels")
    reports = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "parity_report", "*.json")))
    if reports:     # written by tests/test_raster_gpu.py (one file per scene / resolution / depth rule)
        json.dump({os.path.basename(f)[:-5]: json.load(open(f)) for f in reports}, open(os.path.join(dst, f"{tag}_parity_report.json"), "w"), indent=1)
        print("parity report:", len(reports), "cases")
    fetch, write = pmc_means(os.path.join(src, "fetch")), pmc_means(os.path.join(src, "write"))
    if fetch or write:
        traffic = {}
        with open(os.path.join(dst, f"{tag}_rocprofv3_pmc_hbm_traffic.csv"), "w") as fh:
            fh.write("kernel,fetch_bytes_per_launch,write_bytes_per_launch\n")
            for k in sorted(set(fetch) | set(write)):
                fb = int(fetch.get(k, {}).get("FETCH_SIZE", 0.0) * 1024)
                wb = int(write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024)
                fh.write(f"{k},{fb},{wb}\n")
                if k in STAGE_OF:
                    traffic[STAGE_OF[k]] = {"fetch_bytes": fb, "write_bytes": wb, "kernel": k}
        json.dump(traffic, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
        print("traffic:", {k: (v["fetch_bytes"] + v["write_bytes"]) // 1000000 for k, v in traffic.items()}, "MB")
    for a_, b_, suffix in (("sq", "sq2", ""), ("sq_trained", "sq2_trained", "_trained_map")):
        sq = pmc_means(os.path.join(src, a_))
        for k_, v_ in pmc_means(os.path.join(src, b_)).items():      # second counter pass (tools/capture_profiles.sh)
            sq.setdefault(k_, {}).update(v_)
        if sq:
            cols = sorted({c for v in sq.values() for c in v})
            with open(os.path.join(dst, f"{tag}_rocprofv3_pmc_sq{suffix}.csv"), "w") as fh:
                fh.write("kernel," + ",".join(cols) + "\n")
                for k in sorted(sq):
                    fh.write(k + "," + ",".join(f"{sq[k].get(c, 0.0):.1f}" for c in cols) + "\n")
            print("sq counters" + suffix + ":", len(sq), "kernels")


def refresh_bench_lines(tag):
    """bench.py reads profiles/<tag>_pmc_traffic.json and _pmc_sq.csv for `roofline.traffic` and the issue-floor note; the bench
    lines captured in the same gpurun call were produced BEFORE this collection, so restate both fields from the fresh counters."""
    dst = os.environ.get("GSICP_PROFILES_DST") or os.path.join(ROOT, "profiles")
    tpath, sqpath = os.path.join(dst, f"{tag}_pmc_traffic.json"), os.path.join(dst, f"{tag}_rocprofv3_pmc_sq.csv")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    sq = {r["kernel"]: r for r in csv.DictReader(open(sqpath))} if os.path.exists(sqpath) else {}
    for name in ("bench", "bench_basin", "bench_eager", "bench_under_rocprof"):
        path = os.path.join(dst, f"{tag}_{name}.json")
        if not os.path.exists(path):
            continue
        j = json.load(open(path))
        rf = j.get("roofline") or {}
        if "blend_backward" in traffic:     # `traffic` itself stays what the run measured (bench.py's own child passes); this is the capture's figure
            rf["traffic_last_capture"] = int(traffic["blend_backward"]["fetch_bytes"] + traffic["blend_backward"]["write_bytes"])
            rf["traffic_source"] = f"profiles/{tag}_pmc_traffic.json (rocprofv3 FETCH_SIZE + WRITE_SIZE passes of the bench step in the same capture)"
        row = sq.get("blend_backward_tile_kernel")
        if row and float(row.get("SQ_INSTS_VALU", 0) or 0) > 0:
            valu = float(row["SQ_INSTS_VALU"])
            rf["note"] = ("working set (~60 MB) sits in the 256 MiB Infinity Cache: the HBM fraction is a formality; the kernel runs ~80 % VALU-busy and tracks the per-entry dependent chain (DESIGN 3.3)"
                          f"; SQ_INSTS_VALU = {valu / 1e6:.1f} M wave-instructions x 4 cycles / 1024 SIMDs / 2.4 GHz = "
                          f"{valu * 4.0 / 1024.0 / 2.4e3:.0f} us issue floor vs kernel_us (profiles/{tag}_rocprofv3_pmc_sq.csv)")
        j["roofline"] = rf
        json.dump(j, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
    refresh_bench_lines(sys.argv[2] if len(sys.argv) > 2 else (sys.argv[1] if len(sys.argv) > 1 else "r01"))
