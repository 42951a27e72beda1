#!/usr/bin/env python3
"""profiles/rNN/pmc_summary.json from ONE tools/pmc_bench.sh output directory: per-kernel counters, kernel stats, the bench
lines of the profiled runs, and the hash of the kernel sources they were taken on (bench.py replays roofline.traffic from this
file only while that hash matches the build).

Rules (round-3 review / advisor):
* a counter collected in several passes is NOT summed: `counters[kernel][name]` holds the value of ONE pass and
  `counters_by_pass[pass][kernel][name]` every pass's own value (GRBM_GUI_ACTIVE rides in more than one counter group: its sum
  would halve every busy fraction normalised by it);
* nothing is carried over from an earlier summary: every figure in the file was measured on `kernel_sources_sha256`.  FETCH_SIZE of
  the flat-list kernel is bimodal from launch to launch (chunk-sum lines written piecemeal are sometimes evicted before they are
  complete), so pmc_bench.sh takes it several times; `FETCH_SIZE_KiB` is the LARGEST of this run's passes (what bench.py replays),
  `FETCH_SIZE_KiB_passes` lists them all.
    python tools/make_pmc_summary.py gpurun_out/r04pmc profiles/r04"""
import collections, csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
by_pass = {}
for f in sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)):
    name = os.path.relpath(f, src).split(os.sep)[0]
    acc = by_pass.setdefault(name, collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "trace_kernel" in k:
            acc["bvh" if "bvh" in k else "flat"][r["Counter_Name"]] += float(r["Counter_Value"])  # (summed over the launch's dispatch rows / XCDs)
counters = collections.defaultdict(dict)
for name in sorted(by_pass):
    for kern, cs in by_pass[name].items():
        for c, v in cs.items():
            counters[kern].setdefault(c, v)  # first pass that has it; never a sum over passes
stats = {}
for name, key in (("stats", "flat"), ("bvh_stats", "bvh")):
    for f in glob.glob(os.path.join(src, name, "*kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            if "trace_kernel" in r["Name"]:
                stats[key] = {"kernel": r["Name"].split("(")[0], "calls": int(r["Calls"]), "average_ms": float(r["AverageNs"]) / 1e6,
                              "min_ms": float(r["MinNs"]) / 1e6, "max_ms": float(r["MaxNs"]) / 1e6}
        os.makedirs(os.path.join(dst, name), exist_ok=True)
        subprocess.run(["cp", f, os.path.join(dst, name, os.path.basename(f))], check=True)
for name in ("stats", "bvh_stats"):
    p = os.path.join(src, name + "_bench_line.json")
    if os.path.exists(p) and os.path.getsize(p):
        json.dump(json.loads(open(p).read().strip().splitlines()[-1]), open(os.path.join(dst, name + "_bench_line.json"), "w"))
fetch_passes = {n: by_pass[n]["flat"]["FETCH_SIZE"] for n in sorted(by_pass) if "FETCH_SIZE" in by_pass[n].get("flat", {})}
write_passes = {n: by_pass[n]["flat"]["WRITE_SIZE"] for n in sorted(by_pass) if "WRITE_SIZE" in by_pass[n].get("flat", {})}
out = {
    "commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip(),
    "kernel_sources_sha256": bench.kernel_sources_sha256(),
    "command": "bash tools/pmc_bench.sh <dir>  (one rocprofv3 run per counter group; python bench.py --no-cpu-baseline --no-also [--traversal bvh])",
    "FETCH_SIZE_KiB": max(fetch_passes.values()) if fetch_passes else None,
    "WRITE_SIZE_KiB": max(write_passes.values()) if write_passes else None,
    "FETCH_SIZE_KiB_passes": fetch_passes, "WRITE_SIZE_KiB_passes": write_passes,
    "kernel_stats": stats, "counters": {k: dict(v) for k, v in counters.items()},
    "counters_by_pass": {n: {k: dict(v) for k, v in acc.items()} for n, acc in sorted(by_pass.items())},
    "note": "every value measured on kernel_sources_sha256 (nothing carried over); a counter present in several passes is listed per pass, "
            "`counters` holds one pass's value, never their sum",
}
for k, v in counters.items():
    if v.get("SQ_INSTS_VALU"):
        out.setdefault("derived", {})[k] = {
            "active_lanes_per_valu_instruction": v["SQ_THREAD_CYCLES_VALU"] / v["SQ_INSTS_VALU"],
            "wait_any_share_of_wave_cycles": v.get("SQ_WAIT_ANY", 0) / v["SQ_WAVE_CYCLES"] if v.get("SQ_WAVE_CYCLES") else None,
            "wait_inst_any_share_of_wave_cycles": v.get("SQ_WAIT_INST_ANY", 0) / v["SQ_WAVE_CYCLES"] if v.get("SQ_WAVE_CYCLES") else None,
        }
json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    d = os.path.join(dst, os.path.relpath(f, src).split(os.sep)[0])
    os.makedirs(d, exist_ok=True)
    subprocess.run(["cp", f, d], check=True)
print(json.dumps(out.get("derived"), indent=1), fetch_passes, write_passes, stats)
