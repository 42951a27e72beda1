#!/usr/bin/env python3
"""bench.py — Msamples/s of the render hot path on BASELINE.json's headline configuration.

Workload (configs[2], the one `metric` is quoted on): the `randomBouncing` generator with its grid widened
to a,b in [-50,50) (~10k spheres, flat hit list), 1920x1080, 1024 samples per pixel, 50 bounces, f32 reject
test + f64 candidate roots.  One "step" = one full frame (2.12 G pixel-samples), inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU; the frame's rows are dealt to ranks in interleaved tiles of 8 rows (tile k -> rank k % N, the
library's one default, include/rayz_hip.h; no data-path collective while tracing), then ONE RCCL all_gather of the f32
framebuffer tiles per step ("strong" scaling: the frame is fixed, per-GPU work shrinks with N).  Rank 0 prints one JSON
line.  The line carries `frame_sha256` of the gathered frame: the image does not depend on N, so the N = 1 line and every
N > 1 line of the same command must carry the same hash.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_VALU_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
# FP64 vector peak: half the FP32 figure (64-bit FMA issues every 4 cycles per wave and SIMD).  Not in the local guide;
# measured with tools/ubench/valu_peak.hip on the whole chip (profiles/r02/valu_peak.jsonl): v_fma_f64 75.1 TFLOP/s,
# v_pk_fma_f32 140.2, v_fma_f32 121.7 under the clock the chip holds (95.5 % / 89 % / 77 % of the nominal figures).
PEAK_VALU_F64_TFLOPS = 78.6  # (measured 75.1, tools/ubench/valu_peak; no kernel is priced against it since the filters went f32)
PEAK_HBM_GBPS = 8000.0
PMC_ROUND = "r04"  # profiles/<round>/pmc_summary.json: the rocprofv3 PMC passes roofline.traffic is replayed from
FLOP_PER_BOX_TEST = 24        # trace_kernel_bvh: 6 fma (12 flop) + 6 min + 6 max on host-padded boxes — no slack fma since round 3; the compare not counted
FLOP_PER_TEST_MOVING = 24     # centre-at-time 6 + offset 3 + half_b 5 + c 7 + disc 3
FLOP_PER_TEST_STATIC = 18
# what trace_kernel executes per reject test: p1 (2 FMA) + p2 (3 FMA) + r² − p1² − p2² (2 FMA);
# +1 FMA for a y-velocity; +5 for a general one
EXEC_FLOP_STATIC, EXEC_FLOP_MOVY, EXEC_FLOP_MOVG = 14, 16, 24
FLOP_PER_TRI_TEST = 43        # tri_filter: two cross products (9 each), three dot products (5 each), s (3), 2 mul, add + fma, 2 min


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--grid", type=int, default=50, help="half-width of the sphere grid (50 -> ~10k spheres)")
    ap.add_argument("--bounces", type=int, default=50)
    ap.add_argument("--scene-seed", type=int, default=42)
    ap.add_argument("--render-seed", type=int, default=1)
    ap.add_argument("--traversal", choices=["linear", "bvh"], default="linear")
    ap.add_argument("--precision", choices=["f32", "f64"], default="f32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the extra frames reported beside the headline (BVH traversal; f64 fidelity mode; BASELINE configs 2, 4, 5)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    return ap.parse_args()


def kernel_sources_sha256() -> str:
    """Hash of the files the trace kernels are compiled from (+ the compile flags): ties a committed profile to a build."""
    import hashlib

    from rayz_amd import _build

    h = hashlib.sha256()
    for f in ("rayz_amd/csrc/rayz_device.hpp", "rayz_amd/csrc/rayz_hip.hip", "rayz_amd/csrc/bvh_build.hpp", "include/rayz_hip.h"):
        h.update(open(os.path.join(ROOT, f), "rb").read())
    h.update(" ".join(_build.HIPFLAGS).encode())
    return h.hexdigest()


def cpu_baseline(t, target_s: float):
    """Oracle mode A (the reference as written: f64, BVH, one xoshiro stream, ONE thread) on a bounded sample
    of the same workload: every 36th row of the 1920x1080 frame at reduced spp.  Baseline only.  Two more numbers
    ride along: the same port scanning the flat list (algorithm-matched with the GPU kernel), and the
    BVH port on all host cores (rows dealt to threads, one stream each)."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import binding as oracle

    scene, cam, p = t.scene_desc(), t.camera_desc(), t.params()
    p.precision, p.tmin = 1, 1e-10  # the reference's own numbers, src/renderer.zig:107
    rows = list(range(18, p.height, 36))

    def run(rows_, spp_, rng_, linear=False):
        q = type(p).from_buffer_copy(p)
        q.samples_per_px = spp_
        n, s = 0, 0
        for r in rows_:
            _, st = oracle.render_a(scene, cam, q, rng_, row_begin=r, row_end=r + 1, linear=linear)
            n += st.primary_rays
            s += st.segments
        return n, s

    rng = t.rng_state().copy()
    spp = 1
    samples, secs, segs = 0, 0.0, 0
    while True:
        t0 = time.perf_counter()
        n, s = run(rows, spp, rng)
        dt = time.perf_counter() - t0
        samples, secs, segs = n, dt, s
        if dt >= target_s / 2 or spp >= 256:
            break
        spp = max(spp * 2, int(spp * min(8.0, 0.8 * target_s / max(dt, 1e-3))))
    one = samples / secs / 1e6
    out = {
        "value": one, "unit": "Msamples/s", "cores": 1, "kind": "port",
        "sample": f"oracle mode A (f64, BVH, sequential xoshiro256++), g++ -O3 -mavx2 -mfma, rows 18::36 of the "
                  f"{p.width}x{p.height} frame at {spp} spp = {samples} samples in {secs:.1f} s "
                  f"({segs / max(samples, 1):.2f} segments/sample)",
    }
    # the same port on every host core: rows dealt round-robin, one independent stream per thread (ctypes drops the GIL)
    threads = max(1, min(os.cpu_count() or 1, 32))
    states = []
    for k in range(threads):
        st = t.rng_state().copy()
        st[0] ^= 0x9E3779B97F4A7C15 * (k + 1) & 0xFFFFFFFFFFFFFFFF
        states.append(st)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(lambda k: run(rows[k::threads], spp * 4 if threads >= 4 else spp, states[k]), range(threads)))
    dt = time.perf_counter() - t0
    out["all_cores"] = {"value": sum(n for n, _ in parts) / dt / 1e6, "unit": "Msamples/s", "cores": threads,
                        "sample": f"same port, {threads} threads, {sum(n for n, _ in parts)} samples in {dt:.1f} s"}
    # algorithm-matched: the port scanning the flat list instead of walking the BVH (1 thread, a few rows at 1 spp)
    few = rows[:: max(1, len(rows) // 3)][:3]
    t0 = time.perf_counter()
    n, s = run(few, 1, t.rng_state().copy(), linear=True)
    dt = time.perf_counter() - t0
    out["linear_scan"] = {"value": n / dt / 1e6, "unit": "Msamples/s", "cores": 1,
                          "sample": f"mode A over the flat hit list, rows {few} at 1 spp = {n} samples in {dt:.1f} s"}
    return out


def visible_gpu_count():
    """GPUs this process could use, WITHOUT initialising any GPU runtime (the launcher's parent must stay GPU-free: a
    process that has touched the GPU may not start the ranks): KFD topology nodes with SIMDs, cut down by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when set.  None when the topology cannot be read
    (then --gpus is trusted and every rank checks its own device)."""
    import glob

    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    n = 0
    for f in nodes:
        try:
            props = dict(l.split(None, 1) for l in open(f).read().splitlines() if " " in l)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        except (OSError, ValueError):
            return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(args) -> None:
    """`python bench.py --gpus N` without a launcher: start N ranks under torch.distributed.run and exit with its
    code.  Runs BEFORE anything touches the GPU in this process — the parent imports neither torch nor the library, it
    counts devices from sysfs and waits.  Never prints a line for fewer GPUs than were asked for."""
    import socket
    import subprocess

    have = visible_gpu_count()
    if os.environ.get("RAYZ_BENCH_TEST_SHARED_GPU") == "1":
        have = None  # (test hook, see run(): the ranks share device 0 over gloo; the line says so)
    if have is not None and have < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible; refusing to measure fewer")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    print("bench.py: WORLD_SIZE unset, launching " + " ".join(cmd), file=sys.stderr, flush=True)
    raise SystemExit(subprocess.run(cmd).returncode)


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args)  # does not return (the ranks inherit this process's stdout)
    # stdout carries exactly ONE line, the JSON record: everything else this process (or a library it loads — RCCL prints
    # a version banner on stdout at its first communicator) writes to fd 1 goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        run(args, json_fd)
    finally:
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        os.close(json_fd)


def run(args, json_fd):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:  # never report n_gpus different from what was asked for
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")

    import numpy as np
    import torch
    import torch.distributed as dist

    from rayz_amd import capi, render, tracer

    # RAYZ_BENCH_TEST_SHARED_GPU=1: the test suite's hook for boxes with ONE GPU — every rank uses device 0 and the collectives run
    # over gloo, so that the whole N > 1 code path of this file (sharding, gather, per-rank table, frame hash, the BVH block)
    # executes before a real N-GPU node runs it.  The line it prints says so (`test_hook`) and is no measurement of N GPUs.
    shared_gpu = os.environ.get("RAYZ_BENCH_TEST_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) are visible")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        backend = dist.get_backend()
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")
        # one rank per physical GPU: every rank must sit on a different device
        ident = [None] * world
        dist.all_gather_object(ident, (os.uname().nodename, str(torch.cuda.get_device_properties(local_rank).uuid)))
        if len(set(ident)) != world and not shared_gpu:
            raise SystemExit(f"bench.py: ranks share a GPU: {ident}")
    render.init(local_rank)

    # ---- workload: same generator and seeds on every rank (the scene is replicated, 320 KB) ----
    t = tracer.randomBouncing(args.width, -args.grid, args.grid, seed=args.scene_seed)
    t.samples_per_px = args.spp
    t.max_bounces = args.bounces
    t.set_gpu(render_seed=args.render_seed,
              precision=capi.PRECISION_F64 if args.precision == "f64" else capi.PRECISION_F32,
              traversal=capi.TRAVERSAL_BVH if args.traversal == "bvh" else capi.TRAVERSAL_LINEAR)
    scene, cam, p = t.scene_desc(), t.camera_desc(), t.params()
    from rayz_amd import dist as rdist

    p = rdist.shard_params(p, rank, world)
    H, W = p.height, p.width
    dtype = torch.float64 if args.precision == "f64" else torch.float32
    dev = torch.device("cuda", local_rank)
    fg = rdist.FrameGather(H, W, world, rank, dev, dtype=dtype)  # this rank's row tiles + the gathered frame
    dscene = render.DeviceScene(scene)
    stream = torch.cuda.current_stream().cuda_stream

    kernel_ms, seg_total, gather_ms = [], [], []
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]  # on torch's current stream = the render's stream

    def step(record: bool):
        dscene.render_into(cam, p, fg.tile.data_ptr(), stream)
        ev[0].record()
        fg.gather()  # N > 1: one RCCL all_gather of the f32 row tiles + un-interleave; N = 1: a copy
        ev[1].record()
        if record:  # per-step kernel time from the library's HIP events on this stream (forces a sync)
            st = dscene.sync()
            kernel_ms.append(st.kernel_ms)
            seg_total.append(st.segments)
            ev[1].synchronize()
            gather_ms.append(ev[0].elapsed_time(ev[1]))  # this rank: its tile done -> frame assembled (incl. waiting for the slowest rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        segs = torch.tensor([float(np.mean(seg_total))], dtype=torch.float64, device=dev)
        dist.all_reduce(segs, op=dist.ReduceOp.SUM)
        frame_segments = float(segs.item())
        # per-rank figures, so that load imbalance between the row shards can be read off the line: every rank's mean trace-kernel
        # time and mean gather time (its own tile ready -> frame assembled: transfer + waiting for the slowest rank)
        per_rank = rdist.rank_stats([np.mean(kernel_ms), np.mean(gather_ms), np.mean(seg_total)], dev, world)
        kernel_ms_avg = float(per_rank[:, 0].max())
    else:
        frame_segments = float(np.mean(seg_total))
        kernel_ms_avg = float(np.mean(kernel_ms))
        per_rank = rdist.rank_stats([kernel_ms_avg, np.mean(gather_ms), frame_segments], dev, 1)
    # outside the timed region: the gathered frame must be a usable image (every rank holds all of it)
    if not bool(torch.isfinite(fg.frame).all().item()) or bool((fg.frame < 0).any().item()):
        raise SystemExit("bench.py: the rendered frame holds non-finite or negative radiance")

    # The frame's fingerprint, outside the timed region: sha256 over the gathered f32 (or f64) frame, row-major RGB.  The image is
    # independent of the shard count by construction (RNG streams keyed by global pixel, chunk schedule by the full frame), so
    # the N = 1 line (BENCH) and every N > 1 line (SCALE) of the same command must carry the SAME hash — a multi-GPU run proves itself.
    frame_hash = rdist.frame_sha256

    frame_sha256 = frame_hash(fg.frame) if rank == 0 else None

    # N > 1: the same frame through the BVH traversal on every rank's shard (the kernel RAYZ_TRAVERSAL_AUTO gives callers; its
    # share is ~60 ms per GPU at N = 8, so the per-step host sync and the gather show here, not on the 1.1 s flat-list share).
    # Bit-identical to the flat list, so its gathered frame must hash to the headline's.  Same barrier + max-over-ranks timing.
    also_multi = None
    # (RAYZ_BENCH_MULTI_ALSO_AT_N1=1: the test suite's hook — the development pool has one GPU per box, so the block below runs at
    #  N = 1 there, collectives skipped, to keep this code from meeting its first execution on the driver's scaling run)
    multi_also = world > 1 or os.environ.get("RAYZ_BENCH_MULTI_ALSO_AT_N1") == "1"
    if multi_also and args.traversal == "linear" and args.precision == "f32" and not args.no_also:
        t.set_gpu(render_seed=args.render_seed, traversal=capi.TRAVERSAL_BVH, precision=capi.PRECISION_F32, tmin=1e-3)
        pb = rdist.shard_params(t.params(), rank, world)
        kb, gb, sb = [], [], []

        def bvh_step(record):
            dscene.render_into(cam, pb, fg.tile.data_ptr(), stream)
            ev[0].record()
            fg.gather()
            ev[1].record()
            if record:
                stb = dscene.sync()
                kb.append(stb.kernel_ms)
                sb.append(stb.segments)
                ev[1].synchronize()
                gb.append(ev[0].elapsed_time(ev[1]))

        bvh_step(False)
        barrier()
        n_l = 5
        t1 = time.perf_counter()
        for _ in range(n_l):
            bvh_step(True)
        barrier()
        el = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        prb = rdist.rank_stats([np.mean(kb), np.mean(gb), np.mean(sb)], dev, world)
        if rank == 0:
            also_multi = {"bvh_traversal": {
                "value": H * W * args.spp * n_l / float(el.item()) / 1e6, "unit": "Msamples/s", "spp": args.spp,
                "ms_per_step": float(el.item()) / n_l * 1e3, "launches": {"timed": n_l},
                "per_rank": rdist.per_rank_block(prb), "frame_sha256": frame_hash(fg.frame),
                "note": "every rank's shard through trace_kernel_bvh + the same all_gather; barrier + max over ranks around the timed launches"}}
        t.set_gpu(render_seed=args.render_seed, traversal=capi.TRAVERSAL_LINEAR, precision=capi.PRECISION_F32, tmin=1e-3)

    # N = 1 only, after the timed region: the same frame through the reference's own accelerator (BVH traversal), and
    # the f64 fidelity mode (the reference's scalar type and tmin, src/vec.zig:4-8, src/renderer.zig:107) through both —
    # reported beside the headline, which stays the f32 flat hit list BASELINE.json names.  All of them at the headline's own
    # samples per pixel.
    also = None
    if world == 1 and args.traversal == "linear" and args.precision == "f32" and not args.no_also:
        also, frames = {}, {}

        def velocity_classes(sd_):
            ns = sum(1 for i in range(sd_.n_spheres) if all(sd_.spheres[i].velocity[k] == 0 for k in range(3)))
            ny = sum(1 for i in range(sd_.n_spheres)
                     if sd_.spheres[i].velocity[1] != 0 and sd_.spheres[i].velocity[0] == 0 and sd_.spheres[i].velocity[2] == 0)
            return ns, ny, sd_.n_spheres - ns - ny

        def extra(name, tr, ds, traversal, precision, spp, shard=(0, 1), workload=None, keep_frame=False):
            """One more frame beside the headline: one untimed launch (uploads, sizes the workspace), then timed launches — at
            least 3, up to 8 or ~2 s of GPU time for the short frames (an 80 ms frame is not one sample).
            `value` / `ms_per_step` / `kernel_ms` are the MEAN over the timed launches; `launches` holds min / mean / max."""
            tr.samples_per_px = spp
            tr.set_gpu(render_seed=args.render_seed, traversal=traversal, precision=precision,
                       tmin=1e-10 if precision == capi.PRECISION_F64 else 1e-3)
            sde, came = tr.scene_desc(), tr.camera_desc()
            pe = rdist.shard_params(tr.params(), shard[0], shard[1])
            rows = render.shard_rows(pe)
            f64 = precision == capi.PRECISION_F64
            buf = torch.empty((rows, pe.width, 3), dtype=torch.float64 if f64 else torch.float32, device=dev)
            ds.render_into(came, pe, buf.data_ptr(), stream)
            ds.sync()
            walls, kms = [], []
            while True:
                t1 = time.perf_counter()
                ds.render_into(came, pe, buf.data_ptr(), stream)
                st = ds.sync()
                walls.append(time.perf_counter() - t1)
                kms.append(st.kernel_ms)
                n_l, spent = len(walls), sum(walls)
                if n_l >= 3 and (spent > 2.0 or n_l >= 8):
                    break
            dt = float(np.mean(walls))
            if keep_frame:
                frames[name] = buf.cpu().numpy()
            bvh = traversal == capi.TRAVERSAL_BVH
            ns, ny, ng = velocity_classes(sde)
            # the box walk and the reject tests run in f32 in BOTH precisions (they only filter; the f64
            # quadratic of the f64 ray decides): what is priced here is that filter arithmetic, against the FP32 vector peak
            if bvh:
                fl = st.node_tests * FLOP_PER_BOX_TEST + st.sphere_tests * FLOP_PER_TEST_MOVING
            else:
                fl = st.segments * (ns * EXEC_FLOP_STATIC + ny * EXEC_FLOP_MOVY + ng * EXEC_FLOP_MOVG + sde.n_triangles * FLOP_PER_TRI_TEST)
            kernel_ms_mean = float(np.mean(kms))
            ach = fl / (kernel_ms_mean * 1e-3) / 1e12
            n_samples = rows * pe.width * spp
            rec = {"value": n_samples / dt / 1e6, "unit": "Msamples/s", "spp": spp, "ms_per_step": dt * 1e3,
                   "kernel_ms": kernel_ms_mean, "segments_per_sample": st.segments / st.primary_rays,
                   "launches": {"timed": len(walls), "kernel_ms": {"min": float(min(kms)), "mean": kernel_ms_mean, "max": float(max(kms))},
                                "value": {"min": n_samples / max(walls) / 1e6, "mean": n_samples / dt / 1e6, "max": n_samples / min(walls) / 1e6}},
                   "roofline": {"bound": "valu_fp32", "achieved": ach, "peak": PEAK_VALU_F32_TFLOPS, "unit": "TFLOP/s",
                                "frac": ach / PEAK_VALU_F32_TFLOPS,
                                "kernel": ("trace_kernel_bvh" if bvh else "trace_kernel") + ("<double>" if f64 else "<float>")}}
            if workload:
                rec["workload"] = workload
            if f64:
                # the f64 kernels also issue f64 VALU (camera ray, unit(d), hit record, scatter, the candidates' roots) that this
                # count leaves out: the figure is the FILTER's share of the FP32 peak, not a utilisation — `frac` is withheld
                rec["roofline"]["filter_frac"] = rec["roofline"].pop("frac")
                rec["roofline"]["frac"] = None
                rec["roofline"]["note"] = ("filter flops (f32 box / reject tests) only; the f64 narrow phase, hit records and shading are "
                                           "not counted, so no utilisation fraction is claimed for the f64 fidelity mode")
            elif bvh:
                rec["roofline"]["note"] = (f"per-lane tree walk at ~31 of 64 lanes per instruction, bound by its dependent chains, the vector-memory address pipe and the "
                                           f"issue slots together rather than by flops (DESIGN.md 6, profiles/): priced "
                                           f"against the vector peak with {FLOP_PER_BOX_TEST} flop per box test + {FLOP_PER_TEST_MOVING} per leaf test")
            if bvh:
                rec["node_tests_per_segment"] = st.node_tests / max(st.segments, 1)
                rec["sphere_tests_per_segment"] = st.sphere_tests / max(st.segments, 1)
            also[name] = rec

        F32, F64, LIN, BVH = capi.PRECISION_F32, capi.PRECISION_F64, capi.TRAVERSAL_LINEAR, capi.TRAVERSAL_BVH
        extra("bvh_traversal", t, dscene, BVH, F32, args.spp, keep_frame=True)
        also["bvh_traversal"]["frame_sha256"] = frame_hash(torch.from_numpy(frames["bvh_traversal"]))  # must equal the headline's
        # the reference's own precision at the metric's FULL size (round 3 measured these at spp / 16 and spp / 4)
        extra("f64_flat_list", t, dscene, LIN, F64, args.spp)
        extra("f64_bvh_traversal", t, dscene, BVH, F64, args.spp)
        t.samples_per_px = args.spp
        # the same frame through the C ABI's one-call multi-device entry (rayz_hip_multi_render: per-device scene, RCCL
        # gather of the row tiles, un-interleave, copy to HOST memory) on this one device: what a Zig / C caller of the
        # drop-in gets, PCIe included
        t.set_gpu(traversal=capi.TRAVERSAL_BVH, precision=capi.PRECISION_F32, tmin=1e-3)
        ms = render.MultiScene(scene, [local_rank])
        pm = t.params()
        ms.render(cam, pm)
        dtms = []
        for _ in range(3):
            t1 = time.perf_counter()
            frame_m, stm = ms.render(cam, pm)
            dtms.append(time.perf_counter() - t1)
        dtm = float(np.mean(dtms))
        gms, fms = ms.timing()
        also["c_abi_multi_device_entry"] = {
            "value": H * W * args.spp / dtm / 1e6, "unit": "Msamples/s", "ms_per_step": dtm * 1e3, "kernel_ms": stm.kernel_ms,
            "gather_ms": gms, "gather_and_copy_out_ms": fms, "per_device_kernel_ms": [d.kernel_ms for d in ms.device_stats()],
            "devices": 1, "traversal": "bvh", "output": "host memory (PCIe-inclusive)", **ms.info(),
            "launches": {"timed": len(dtms), "ms_per_step": {"min": min(dtms) * 1e3, "mean": dtm * 1e3, "max": max(dtms) * 1e3}},
            "identical_to_device_path": bool(np.array_equal(frame_m, frames["bvh_traversal"]))}
        ms.close()
        t.set_gpu(traversal=capi.TRAVERSAL_LINEAR, precision=capi.PRECISION_F32, tmin=1e-3)
        t.samples_per_px = args.spp
        frames.clear()

        # ---- the other BASELINE.json configs (the headline above is configs[2]); sizes follow the command line, so the
        #      default run measures them at BASELINE's own sizes: config 2 = the reference's shipped scene at spp/4;
        #      config 4 = twice the width, 4x the samples, ONE GPU's 1/8 row share (the flat list at a stated reduced spp:
        #      its rate does not depend on spp and the full count would take 18 s per frame); config 5 = the 224x224-quad mesh
        #      at spp/2 (build-defined primitive).  Each gets its own device scene; launches as in extra().
        full = args.width >= 1920 and args.grid >= 50
        c2 = tracer.randomBouncing(args.width, seed=args.scene_seed)
        c2.max_bounces = args.bounces
        ds2 = render.DeviceScene(c2.scene_desc())
        spp2 = max(1, args.spp // 4)
        w2 = f"randomBouncing as shipped (src/rayz.zig:45-168: {c2.info().n_spheres} spheres), {c2.info().width}x{c2.info().height}, {spp2} spp"
        extra("config2_flat_list", c2, ds2, LIN, F32, spp2, workload=w2)
        extra("config2_bvh", c2, ds2, BVH, F32, spp2, workload=w2)
        ds2.close()
        c4 = tracer.randomBouncing(2 * args.width, -args.grid, args.grid, seed=args.scene_seed)
        c4.max_bounces = args.bounces
        ds4 = render.DeviceScene(c4.scene_desc())
        w4 = f"{c4.info().n_spheres} spheres, {c4.info().width}x{c4.info().height}, 8-row tiles (row // 8) % 8 == 0 (one GPU's share of an 8-GPU frame)"
        extra("config4_one_gpu_share_bvh", c4, ds4, BVH, F32, 4 * args.spp, shard=(0, 8), workload=w4 + f", {4 * args.spp} spp")
        spp4f = max(1, args.spp // 8)
        extra("config4_one_gpu_share_flat_list", c4, ds4, LIN, F32, spp4f, shard=(0, 8),
              workload=w4 + f", {spp4f} spp (REDUCED from {4 * args.spp}: the rate is spp-invariant; the full count is 32x this frame time)")
        ds4.close()
        c5 = tracer.triangleMesh(args.width, 224 if full else 12, seed=1)
        c5.max_bounces = args.bounces
        ds5 = render.DeviceScene(c5.scene_desc())
        spp5 = max(1, args.spp // 2)
        extra("config5_triangle_mesh_bvh", c5, ds5, BVH, F32, spp5,
              workload=f"{c5.info().n_triangles} triangles + {c5.info().n_spheres} spheres (build-defined primitive, DESIGN.md 4.7), "
                       f"{c5.info().width}x{c5.info().height}, {spp5} spp")
        ds5.close()

    if rank == 0:
        samples_per_step = H * W * args.spp
        info = t.info()
        sd = scene
        n_moving = sum(1 for i in range(sd.n_spheres) if any(sd.spheres[i].velocity[k] != 0 for k in range(3)))
        n_static = sd.n_spheres - n_moving
        # compulsory HBM bytes of a frame = the framebuffer write + the scene fetched once per XCD; the
        # chunk-sum workspace (written by the trace kernel, read back by resolve_kernel) is design traffic, listed apart
        import ctypes

        n_chunks = capi.load().rayz_hip_chunk_schedule(ctypes.byref(p), None, 0)
        compulsory_bytes = (H * W / world) * 12 + 8 * (n_static * 16 + n_moving * 32)
        workspace_bytes = (H * W / world) * n_chunks * 16 * 2
        if args.traversal == "linear":
            # roofline of the dominant kernel (trace_kernel): algorithmic flops of the reject test per launch.
            # With N GPUs each launch covers 1/N of the frame; rank-mean flops over the slowest rank's time.
            flops = frame_segments / world * (n_static * FLOP_PER_TEST_STATIC + n_moving * FLOP_PER_TEST_MOVING)
            n_movy = sum(1 for i in range(sd.n_spheres)
                         if sd.spheres[i].velocity[1] != 0 and sd.spheres[i].velocity[0] == 0 and sd.spheres[i].velocity[2] == 0)
            executed_flops = frame_segments / world * (n_static * EXEC_FLOP_STATIC + n_movy * EXEC_FLOP_MOVY +
                                                       (n_moving - n_movy) * EXEC_FLOP_MOVG)
            kernel = "trace_kernel<%s>" % ("double" if args.precision == "f64" else "float")
            note = ("VALU-bound, not HBM/MFMA (SURVEY.md 8d): the scene (320 KB) is L2 / scalar-cache resident. achieved/frac "
                    "count the flops of the kernel's own reject test (7 / 8 / 12 FMA = 14 / 16 / 24 flop per static / "
                    "y-moving / moving sphere, DESIGN.md 4.3) x segments; `reference_formulation` prices the same frame "
                    "with SURVEY 8d's figure for the reference's quadratic (18 / 24 flop per test), which this kernel "
                    "does not execute - that fraction can exceed 1.  The kernel is power-limited: ~1.35 kW, shader clock "
                    "~2.2 GHz instead of the 2.4 GHz the peak assumes (tools/clk_probe.sh, profiles/r02/clk_probe.log), so its "
                    "rate follows the clock a given box sustains (8.8-9.3 s per frame over eight boxes)")
        else:
            st_last = dscene.sync()
            flops = (st_last.node_tests * FLOP_PER_BOX_TEST + st_last.sphere_tests * FLOP_PER_TEST_MOVING)
            executed_flops = flops
            kernel = "trace_kernel_bvh<%s>" % ("double" if args.precision == "f64" else "float")
            note = ("per-lane tree walk at ~31 of 64 lanes per vector instruction: bound by its dependent chains, the vector-memory "
                    "address pipe (TA busy ~84 %) and the issue slots together, not by flops (DESIGN.md 6, profiles/r03, r04); priced "
                    f"against the same FP32 vector peak: algorithmic flops = {FLOP_PER_BOX_TEST} x box tests + {FLOP_PER_TEST_MOVING} x leaf tests")
        peak = PEAK_VALU_F32_TFLOPS  # reject tests and box walk are f32 arithmetic in both precisions
        reference_achieved = flops / (kernel_ms_avg * 1e-3) / 1e12   # SURVEY 8d accounting
        achieved = executed_flops / (kernel_ms_avg * 1e-3) / 1e12    # what the kernel executes
        # HBM bytes per launch (FETCH_SIZE doubled — gfx950 wide-read correction, MI355X_MICROARCH.md — plus WRITE_SIZE)
        # come from rocprofv3 PMC passes of this very command, which cannot run inside this process: they are REPLAYED
        # from profiles/, and only when the kernel sources hash to what the passes were taken on; otherwise null.
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, "profiles", PMC_ROUND, "pmc_summary.json")
        if (os.path.exists(pmc) and world == 1 and args.traversal == "linear" and args.precision == "f32"
                and (args.width, args.spp, args.grid) == (1920, 1024, 50)):
            try:
                z = json.load(open(pmc))
                if z.get("kernel_sources_sha256") == kernel_sources_sha256():
                    traffic = (2 * z["FETCH_SIZE_KiB"] + z["WRITE_SIZE_KiB"]) * 1024
                    traffic_source = {"replayed_from": f"profiles/{PMC_ROUND}/pmc_summary.json", "commit": z.get("commit"),
                                      "kernel_sources_sha256": z.get("kernel_sources_sha256")}
            except Exception:
                traffic = None
        out = {
            "metric": "Msamples/sec, 10k-sphere 1920x1080x1024spp (%s)" % ("flat hit list" if args.traversal == "linear" else "BVH traversal"),
            "value": samples_per_step * args.steps / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "frame_sha256": frame_sha256,
            **({"test_hook": "RAYZ_BENCH_TEST_SHARED_GPU: all ranks on ONE GPU, collectives over gloo — exercises the N > 1 code path, measures nothing"}
               if shared_gpu else {}),
            "dtype": "f32" if args.precision == "f32" else "f64",
            "data": "synthetic",
            "config": {
                "workload": f"randomBouncing grid [-{args.grid},{args.grid}) = {info.n_spheres} spheres "
                            f"({n_static} static, {n_moving} moving), {W}x{H}, {args.spp} spp, {args.bounces} bounces, "
                            f"{args.traversal} traversal, scene seed {args.scene_seed}, render seed {args.render_seed}",
                "parallelism": (f"rows dealt in interleaved 8-row tiles x{world} + one {'RCCL' if backend == 'nccl' else backend} all_gather per frame"
                                if world > 1 else "1 GPU"),
                "collective_backend": backend, "collective_world_size": world if world > 1 else None,
                "segments_per_sample": frame_segments / samples_per_step,
                "arithmetic": ("f32 path state and reject test, f64 candidate roots (DESIGN.md 4.3), tmin 1e-3"
                               if args.precision == "f32" else "f64 path state, roots, hit records and shading (the reference's scalar type), tmin 1e-10; "
                               "the reject tests and the box walk — which only filter — in f32 (DESIGN.md 4.3 / 4.8)"),
            },
            "roofline": {
                "bound": "valu_fp32", "achieved": achieved, "peak": peak,
                "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                "kernel": kernel, "kernel_ms": kernel_ms_avg, "note": note,
                # SURVEY 8d's price of the reference's quadratic, which this kernel does not execute: as a fraction of the peak it
                # is only reported while it stays a fraction (it exceeds 1 on the headline frame: the work is simply not done)
                "reference_formulation": {"achieved": reference_achieved, "unit": "TFLOP/s",
                                          "frac": reference_achieved / peak if reference_achieved <= peak else None},
                "traffic_source": traffic_source,
                "hbm": {"algorithmic_bytes": compulsory_bytes, "achieved": compulsory_bytes / (kernel_ms_avg * 1e-3) / 1e9,
                        "workspace_bytes": workspace_bytes, "chunk_sums_per_pixel": n_chunks,
                        "profiled_bytes": traffic,
                        "profiled_over_algorithmic": None if traffic is None else traffic / compulsory_bytes,
                        "profiled": None if traffic is None else traffic / (kernel_ms_avg * 1e-3) / 1e9,
                        "peak": PEAK_HBM_GBPS, "unit": "GB/s"},
            },
        }
        if args.traversal == "linear":
            # SURVEY 8d "logical scan bytes": every segment reads every scan record once per LANE in the reference's
            # formulation; the kernel reads it once per WAVE through the scalar cache (records: 16 B static, 20 B
            # y-moving, 32 B moving) - served by K$ / L2, never HBM
            rec = n_static * 16 + n_movy * 20 + (n_moving - n_movy) * 32
            lane_bytes = frame_segments / world * rec
            out["roofline"]["logical_scan"] = {
                "per_lane_bytes": lane_bytes, "per_lane_TBps": lane_bytes / (kernel_ms_avg * 1e-3) / 1e12,
                "scalar_cache_bytes": lane_bytes / 64, "scalar_cache_TBps": lane_bytes / 64 / (kernel_ms_avg * 1e-3) / 1e12,
            }
        # every rank's own figures (N > 1: load imbalance between the row shards and the cost of the gather show here)
        out["per_rank"] = rdist.per_rank_block(per_rank)
        if also:
            out["also"] = also
        if also_multi:
            out.setdefault("also", {}).update({("bvh_traversal" if world > 1 else "multi_gpu_bvh_block_at_n1"): also_multi["bvh_traversal"]})
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(t, args.cpu_seconds)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
