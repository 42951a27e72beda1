#!/usr/bin/env python3
"""One-off: the random-scene parity fuzz of tests/test_fuzz_gpu.py over many more seeds (GPU vs oracle, bit for bit,
flat list and BVH, f32 and f64).  usage: fuzz_more.py <first_seed> <count> [axis]   (axis: the axis-aligned generator)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import binding as oracle
from rayz_amd import capi, render
from test_fuzz_gpu import random_scene, axis_scene

render.init(0)
if 'nodes16' in sys.argv:  # force the 32-byte node records (16-bit plane indices) however small the tree
    sys.argv.remove('nodes16'); render.debug_set(capi.DEBUG_BVH_NODES, 2); oracle.load()
first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, first + count):
    t = (axis_scene if len(sys.argv) > 3 and sys.argv[3] == 'axis' else random_scene)(seed)
    for trav in (capi.TRAVERSAL_LINEAR, capi.TRAVERSAL_BVH):
        for prec in (capi.PRECISION_F32, capi.PRECISION_F64):
            t.set_gpu(traversal=trav, precision=prec)
            scene, cam, p = t.scene_desc(), t.camera_desc(), t.params()
            got, gst = render.render_host(scene, cam, p)
            want, ost = oracle.render_b(scene, cam, p)
            same = np.array_equal(got, want, equal_nan=True) and gst.segments == ost.segments
            if not same:
                bad += 1
                d = np.abs(got.astype(np.float64) - want)
                print(f"MISMATCH seed {seed} traversal {trav} precision {prec}: {int((d > 0).sum())} values differ, max {np.nanmax(d):.3e}, "
                      f"segments {gst.segments} vs {ost.segments}", flush=True)
    if (seed - first) % 50 == 49:
        print(f"... {seed - first + 1} scenes, {bad} mismatches", flush=True)
print(f"done: {count} scenes x 4 variants, {bad} mismatches")
