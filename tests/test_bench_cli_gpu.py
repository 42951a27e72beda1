"""The two entry points a user (and the driver) runs: `bench.py` prints one JSON line with the agreed keys, and
the `rayz <img_w> [out.ppm]` CLI renders the reference's scene to a P3 PPM (src/rayz.zig:12-43)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract(gpu):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--width", "96", "--spp", "4", "--grid", "4",
                        "--steps", "2", "--warmup", "1", "--cpu-seconds", "0.5"], capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, RAYZ_BENCH_MULTI_ALSO_AT_N1="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(r.stdout.strip().splitlines()) == 1  # ONE line on stdout: library chatter goes to stderr
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "Msamples/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and rf["frac"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["unit"] == "Msamples/s" and cb["value"] > 0 and cb["sample"]
    assert d["value"] > 0 and abs(d["value"] - 96 * 54 * 4 * 2 / (d["ms_per_step"] * 2e-3) / 1e6) < 1e-6 * d["value"] + 1e-9
    assert d["also"]["bvh_traversal"]["value"] > 0 and d["also"]["bvh_traversal"]["roofline"]["frac"] > 0
    # (the f64 fidelity mode's filters — reject tests, box walk — run in f32: its kernels are priced against the FP32 peak)
    assert d["also"]["f64_flat_list"]["value"] > 0 and d["also"]["f64_bvh_traversal"]["roofline"]["bound"] == "valu_fp32"
    m = d["also"]["c_abi_multi_device_entry"]  # rayz_hip_multi_render on one device: RCCL really ran
    assert m["value"] > 0 and m["n_devices"] == 1 and m["rccl_version"] > 0 and m["identical_to_device_path"] is True
    assert m["gather_ms"] > 0 and len(m["per_device_kernel_ms"]) == 1
    # the f64 fidelity mode's entries claim no utilisation fraction (its f64 work is not counted), only the filter's share
    assert d["also"]["f64_bvh_traversal"]["roofline"]["frac"] is None and d["also"]["f64_bvh_traversal"]["roofline"]["filter_frac"] > 0
    # every BASELINE config rides along: 2 (flat + BVH), 4 (one GPU's 1/8 share, BVH + flat at reduced spp), 5 (mesh, BVH)
    for k in ("config2_flat_list", "config2_bvh", "config4_one_gpu_share_bvh", "config4_one_gpu_share_flat_list", "config5_triangle_mesh_bvh"):
        e = d["also"][k]
        assert e["value"] > 0 and e["kernel_ms"] > 0 and e["segments_per_sample"] >= 1 and e["roofline"]["frac"] > 0 and e["workload"], k
    assert d["also"]["config4_one_gpu_share_bvh"]["spp"] == 16 and "(row // 8) % 8 == 0" in d["also"]["config4_one_gpu_share_bvh"]["workload"]
    pr = d["per_rank"]
    assert pr["kernel_ms"]["min"] <= pr["kernel_ms"]["mean"] <= pr["kernel_ms"]["max"] and len(pr["kernel_ms"]["all"]) == 1
    assert pr["gather_ms"]["max"] > 0 and d["roofline"]["kernel_ms"] == pr["kernel_ms"]["max"]
    rfm = d["roofline"]["reference_formulation"]
    assert rfm["frac"] is None or rfm["frac"] <= 1
    # the frame's fingerprint: the same image through the flat list (headline), the BVH, and the code path of the N > 1 BVH block
    assert len(d["frame_sha256"]) == 64 and d["also"]["bvh_traversal"]["frame_sha256"] == d["frame_sha256"]
    mb = d["also"]["multi_gpu_bvh_block_at_n1"]
    assert mb["frame_sha256"] == d["frame_sha256"] and mb["value"] > 0 and mb["launches"]["timed"] == 5
    assert len(mb["per_rank"]["kernel_ms"]["all"]) == 1 and mb["per_rank"]["gather_ms"]["max"] > 0
    # every `also` frame is timed over several launches
    for k, e in d["also"].items():
        if "launches" in e and "kernel_ms" in e["launches"]:
            L = e["launches"]
            assert L["timed"] >= 2 and L["kernel_ms"]["min"] <= L["kernel_ms"]["mean"] <= L["kernel_ms"]["max"], k
    hb = d["roofline"]["hbm"]
    assert hb["algorithmic_bytes"] < hb["workspace_bytes"] * 10 and hb["chunk_sums_per_pixel"] >= 1


@pytest.mark.parametrize("world", [2, 3])
def test_bench_ranks_on_one_gpu_print_the_same_frame_hash(gpu, world):
    """bench.py's N > 1 path — row deal, gather, per-rank table, frame hash, the N > 1 BVH block — with two and three ranks (an uneven deal: 7 row
    tiles) under torch.distributed.run.  The box has ONE GPU, so the test hook puts every rank on it and runs the collectives over gloo: the line
    says `test_hook`, its numbers mean nothing, but its `frame_sha256` must equal the one-rank run's (the image does not depend on N),
    and the BVH block's as well."""
    base = [os.path.join(ROOT, "bench.py"), "--width", "96", "--spp", "4", "--grid", "4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, *base, "--gpus", "1", "--no-also"], capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    two = subprocess.run([sys.executable, *base, "--gpus", str(world)], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, RAYZ_BENCH_TEST_SHARED_GPU="1"))
    assert two.returncode == 0, two.stderr[-3000:]
    lines = [l for l in two.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 prints, the others stay silent
    d2 = json.loads(lines[0])
    assert d2["n_gpus"] == world and "test_hook" in d2 and d2["config"]["collective_backend"] == "gloo"
    assert d2["frame_sha256"] == d1["frame_sha256"] and len(d1["frame_sha256"]) == 64
    pr = d2["per_rank"]
    assert len(pr["kernel_ms"]["all"]) == world and min(pr["kernel_ms"]["all"]) > 0 and len(pr["segments"]) == world
    assert abs(sum(pr["segments"]) - d1["config"]["segments_per_sample"] * 96 * 54 * 4) < 0.5  # the shards' segments add up to the frame's
    b = d2["also"]["bvh_traversal"]
    assert b["frame_sha256"] == d1["frame_sha256"] and b["value"] > 0 and len(b["per_rank"]["kernel_ms"]["all"]) == world


def test_cli_renders_reference_scene(gpu, tmp_path):
    exe = os.path.join(ROOT, "rayz_amd", "host", "rayz")
    out = tmp_path / "out.ppm"
    env = dict(os.environ, RAYZ_SEED="7", RAYZ_SPP="3", RAYZ_BOUNCES="6")
    r = subprocess.run([exe, "96", str(out)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "Finished render (" in r.stderr and "rps and" in r.stderr and "us per ray" in r.stderr  # src/rayz.zig:30-34
    toks = out.read_text().split()
    assert toks[:4] == ["P3", "96", "54", "255"]
    px = np.array(toks[4:], dtype=np.int64).reshape(54, 96, 3)
    assert px.min() >= 0 and px.max() <= 255 and px[:5].mean() > px[-5:].mean()  # sky on top
    # same seed, same image; stdout form when no file is given
    r2 = subprocess.run([exe, "96"], capture_output=True, text=True, env=env, timeout=600)
    assert r2.returncode == 0 and r2.stdout == out.read_text()
