#!/usr/bin/env python3
"""profiles/rNN/pmc_summary.json from a tools/pmc_bench.sh output directory: per-kernel counter sums, kernel stats, the
bench lines of the profiled runs, and the hash of the kernel sources they were taken on (bench.py replays
roofline.traffic from this file only while that hash matches the build).
    python tools/make_pmc_summary.py gpurun_out/r02b profiles/r02"""
import collections, csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "trace_kernel" in k:
            acc["bvh" if "bvh" in k else "flat"][r["Counter_Name"]] += float(r["Counter_Value"])
stats = {}
for name, key in (("stats", "flat"), ("bvh_stats", "bvh")):
    for f in glob.glob(os.path.join(src, name, "*kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            if "trace_kernel" in r["Name"]:
                stats[key] = {"kernel": r["Name"].split("(")[0], "calls": int(r["Calls"]), "average_ms": float(r["AverageNs"]) / 1e6,
                              "min_ms": float(r["MinNs"]) / 1e6, "max_ms": float(r["MaxNs"]) / 1e6}
        os.makedirs(os.path.join(dst, name), exist_ok=True)
        subprocess.run(["cp", f, os.path.join(dst, name, os.path.basename(f))], check=True)
lines = {}
for name in ("stats", "bvh_stats"):
    p = os.path.join(src, name + "_bench_line.json")
    if os.path.exists(p) and os.path.getsize(p):
        lines[name] = json.loads(open(p).read().strip().splitlines()[-1])
        json.dump(lines[name], open(os.path.join(dst, name + "_bench_line.json"), "w"))
flat = acc["flat"]
out = {
    "commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip(),
    "kernel_sources_sha256": bench.kernel_sources_sha256(),
    "command": "bash tools/pmc_bench.sh <dir>  (one rocprofv3 run per counter group; python bench.py --no-cpu-baseline --no-also [--traversal bvh])",
    "FETCH_SIZE_KiB": flat.get("FETCH_SIZE"), "WRITE_SIZE_KiB": flat.get("WRITE_SIZE"),
    "kernel_stats": stats, "counters": {k: dict(v) for k, v in acc.items()},
}
for k, v in acc.items():
    if v.get("SQ_INSTS_VALU"):
        out.setdefault("derived", {})[k] = {
            "active_lanes_per_valu_instruction": v["SQ_THREAD_CYCLES_VALU"] / v["SQ_INSTS_VALU"],
            "wait_any_share_of_wave_cycles": v.get("SQ_WAIT_ANY", 0) / v["SQ_WAVE_CYCLES"] if v.get("SQ_WAVE_CYCLES") else None,
            "wait_inst_any_share_of_wave_cycles": v.get("SQ_WAIT_INST_ANY", 0) / v["SQ_WAVE_CYCLES"] if v.get("SQ_WAVE_CYCLES") else None,
        }
json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    d = os.path.join(dst, os.path.basename(os.path.dirname(f)))
    os.makedirs(d, exist_ok=True)
    subprocess.run(["cp", f, d], check=True)
print(json.dumps(out["derived"], indent=1), out["FETCH_SIZE_KiB"], out["WRITE_SIZE_KiB"], stats)
