"""Builders of rayz_hip_kat records (include/rayz_hip.h: RayzKatOp) shared by the CPU and GPU known-answer tests."""
import numpy as np

from rayz_amd import capi

S_IN, S_OUT = capi.KAT_IN_STRIDE, capi.KAT_OUT_STRIDE


def f32r(x):
    """Round to f32-representable values, so that an f32 and an f64 evaluation start from the same numbers."""
    return np.asarray(x, dtype=np.float64).astype(np.float32).astype(np.float64)


def blank(n):
    return np.zeros((n, S_IN))


def unit(v):
    v = np.asarray(v, dtype=np.float64)
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def uniforms(rng, shape):
    """Uniforms of the form k / 2^24: exactly what the kernel's f32 stream can produce, exact in f64 too."""
    return rng.integers(0, 1 << 24, size=shape).astype(np.float64) / float(1 << 24)


# ---- the reference's own vectors ----------------------------------------------------------------------------
def refract_reference():  # src/material.zig:213-223
    rec = blank(1)
    rec[0, 0:3] = unit([-0.3125, -0.3125, -1.0])
    rec[0, 3:6] = [-0.558127, -0.558127, 0.613994]
    rec[0, 6] = 1.0 / 1.5
    return rec, np.array([0.144881, 0.144881, -0.978784])


def get_ray_reference(camera_desc, no_rng=False):  # src/renderer.zig:129-149 (focus sqrt(12), defocus 0: SURVEY.md §4)
    """no_rng: n_u = -1, i.e. getRay(px, py, null) exactly as the reference's test calls it; otherwise n_u = 0 (every draw 0.5)."""
    rec = blank(2)
    c = camera_desc
    for i, (px, py) in enumerate([(0, 0), (112, 199)]):
        rec[i, 0:18] = np.concatenate([np.array(x) for x in (c.look_from, c.px_du, c.px_dv, c.px_origin, c.defocus_u,
                                                               c.defocus_v)])
        rec[i, 18], rec[i, 19], rec[i, 20], rec[i, 21] = c.defocus, px, py, (-1 if no_rng else 0)  # 0: no list, every draw is 0.5 = no jitter
    want = np.array([[-0.935834, 0.815856, -7.75169], [-0.998817, -4.18732, -2.8115]])
    return rec, want


def box_hit_reference():  # src/hit.zig:247-279 ("bbox hit", "bbox hit 2")
    rows = [
        ([0] * 3, [1] * 3, [-1] * 3, [1] * 3, 0, 10, 1),
        ([0] * 3, [1] * 3, [-1] * 3, [-1] * 3, 0, 10, 0),
        ([0] * 3, [1] * 3, [-1] * 3, [0.5] * 3, 0, 10, 1),
        ([-1000, -2000, -1000], [1000, 2, 1000], [13, 2, 3], [-9.6, -1.5, -2.3], 0, 10, 1),
    ]
    rec = blank(len(rows))
    for i, (lo, hi, o, d, tmin, tmax, _) in enumerate(rows):
        rec[i, 0:3], rec[i, 3:6], rec[i, 6:9], rec[i, 9:12], rec[i, 12], rec[i, 13] = lo, hi, o, d, tmin, tmax
    return rec, np.array([r[-1] for r in rows], dtype=np.float64)


# ---- random records ---------------------------------------------------------------------------------------------
def random_sphere_hits(rng, n, big=False):
    """(ray, sphere) pairs: rays aimed near the sphere so that about half hit; some spheres move; some rays start
    inside.  `big` adds the r = 1000 ground-sphere regime (grazing rays from points on its surface)."""
    rec = blank(n)
    c = rng.uniform(-10, 10, (n, 3))
    r = rng.uniform(0.05, 2.0, n)
    v = np.where(rng.random((n, 1)) < 0.4, rng.uniform(-0.5, 0.5, (n, 3)) * [[0, 1, 0]], 0.0)
    gen = rng.random(n) < 0.1
    v = np.where(gen[:, None], rng.uniform(-0.5, 0.5, (n, 3)), v)
    if big:
        c[: n // 4] = [0, -1000, 0]
        r[: n // 4] = 1000.0
        v[: n // 4] = 0
    o = c + unit(rng.normal(size=(n, 3))) * (r * rng.uniform(0.3, 8.0, n))[:, None]
    if big:
        k = n // 4
        o[:k] = c[:k] + unit(rng.normal(size=(k, 3)) + [0, 3, 0]) * (r[:k] * (1 + rng.uniform(0, 1e-3, k)))[:, None]
    time = uniforms(rng, n)
    aim = c + v * time[:, None] + unit(rng.normal(size=(n, 3))) * (r * rng.uniform(0, 2.0, n))[:, None]
    d = (aim - o) * rng.uniform(0.2, 5.0, (n, 1))
    rec[:, 0:3], rec[:, 3:6], rec[:, 6] = f32r(c), f32r(v), f32r(r)
    rec[:, 7:10], rec[:, 10:13], rec[:, 13] = f32r(o), f32r(d), time
    rec[:, 14], rec[:, 15] = 1e-3, np.inf
    return rec


def random_scatters(rng, n, n_u=30):
    rec = blank(n)
    kind = rng.integers(0, 3, n)
    method = np.where(kind == 0, rng.integers(0, 3, n), 2)
    param = np.where(kind == 1, np.where(rng.random(n) < 0.3, 0.0, rng.uniform(0, 1.5, n)), rng.uniform(1.1, 2.4, n))
    nrm = unit(rng.normal(size=(n, 3)))
    front = rng.random(n) < 0.7
    d = unit(rng.normal(size=(n, 3)))
    dn = (d * nrm).sum(1, keepdims=True)
    d = np.where(dn > 0, d - 2 * dn * nrm, d)  # Hit.init: the stored normal always opposes the ray
    d = d * rng.uniform(0.3, 4.0, (n, 1))
    pt = rng.uniform(-20, 20, (n, 3))
    rec[:, 0], rec[:, 1], rec[:, 2] = kind, method, f32r(param)
    rec[:, 3:6] = f32r(pt - d)
    rec[:, 6:9], rec[:, 9:12], rec[:, 12:15] = f32r(d), f32r(pt), f32r(nrm)
    rec[:, 15], rec[:, 16] = front, n_u
    rec[:, 17:17 + n_u] = uniforms(rng, (n, n_u))
    return rec


def random_get_rays(rng, n, camera_desc, n_u=12):
    rec = blank(n)
    c = camera_desc
    rec[:, 0:18] = f32r(np.concatenate([np.array(x) for x in (c.look_from, c.px_du, c.px_dv, c.px_origin, c.defocus_u,
                                                               c.defocus_v)]))[None, :]
    rec[:, 18] = c.defocus
    rec[:, 19], rec[:, 20], rec[:, 21] = rng.integers(0, 1920, n), rng.integers(0, 1080, n), n_u
    rec[:, 22:22 + n_u] = uniforms(rng, (n, n_u))
    return rec


def random_refracts(rng, n):
    """Non-TIR only (eta * sin <= 0.98): the reference's bare sqrt is NaN beyond, the kernel clamps."""
    rec = blank(n)
    nrm = unit(rng.normal(size=(n, 3)))
    t = unit(np.cross(nrm, rng.normal(size=(n, 3))))
    eta = np.where(rng.random(n) < 0.5, 1 / 1.5, 1.5) * rng.uniform(0.8, 1.2, n)
    smax = np.minimum(0.98 / eta, 0.999)
    sin = rng.uniform(0, 1, n) * smax
    ud = -np.sqrt(1 - sin ** 2)[:, None] * nrm + sin[:, None] * t
    rec[:, 0:3], rec[:, 3:6], rec[:, 6] = unit(f32r(ud)), unit(f32r(nrm)), f32r(eta)
    return rec


def random_boxes(rng, n):
    rec = blank(n)
    lo = rng.uniform(-10, 10, (n, 3))
    hi = lo + rng.uniform(0.01, 6, (n, 3))
    o = rng.uniform(-15, 15, (n, 3))
    aim = lo + (hi - lo) * rng.uniform(-0.5, 1.5, (n, 3))
    d = (aim - o) * rng.uniform(0.1, 3, (n, 1))
    rec[:, 0:3], rec[:, 3:6], rec[:, 6:9], rec[:, 9:12] = f32r(lo), f32r(hi), f32r(o), f32r(d)
    rec[:, 12], rec[:, 13] = 1e-3, np.where(rng.random(n) < 0.5, np.inf, rng.uniform(0.1, 3, n))
    rec[:, 26] = np.arange(n) % 2  # either node record format (include/rayz_hip.h: RAYZ_KAT_BOX_HIT); no draw: the sets after this one stay what they were
    return rec


def axis_parallel_boxes(rng, n):
    """Rays with one or two direction components EXACTLY zero (±0) against boxes on either side of the origin and
    across it: the case where a reciprocal of ±inf turns one slab distance into −inf and the other into NaN.  The
    origin never lies on a box plane (the reference's own test is 0/0 there).  Returns (records, geometric truth,
    rows whose ray runs parallel to a slab it is OUTSIDE of by at least 0.1: never a hit)."""
    rec = blank(n)
    lo = f32r(rng.uniform(-10, 6, (n, 3)))
    hi = f32r(lo + rng.uniform(0.5, 8, (n, 3)))
    inside = rng.random((n, 3)) < 0.7
    o = np.where(inside, lo + (hi - lo) * rng.uniform(0.05, 0.95, (n, 3)),
                 np.where(rng.random((n, 3)) < 0.5, lo - rng.uniform(0.1, 5, (n, 3)), hi + rng.uniform(0.1, 5, (n, 3))))
    o = f32r(o)
    aim = lo + (hi - lo) * rng.uniform(-0.3, 1.3, (n, 3))
    d = f32r((aim - o) * rng.uniform(0.1, 3, (n, 1)))
    nz = rng.integers(1, 3, n)  # how many components are zero
    for i in range(n):
        ax = rng.permutation(3)[: nz[i]]
        d[i, ax] = rng.choice([0.0, -0.0], len(ax))
    rec[:, 0:3], rec[:, 3:6], rec[:, 6:9], rec[:, 9:12] = lo, hi, o, d
    rec[:, 12], rec[:, 13] = 1e-3, np.inf
    rec[:, 26] = np.arange(n) % 2  # either node record format
    # geometric truth: a parallel axis is satisfied iff o lies strictly inside the slab; the others by exact intervals
    with np.errstate(divide="ignore", invalid="ignore"):
        a, b = (lo - o) / d, (hi - o) / d
    par = d == 0
    ok_par = ((o > lo) & (o < hi)) | ~par
    t_in = np.where(par, -np.inf, np.minimum(a, b)).max(1)
    t_out = np.where(par, np.inf, np.maximum(a, b)).min(1)
    truth = ok_par.all(1) & (t_out > np.maximum(t_in, 1e-3))
    return rec, truth, ~ok_par.all(1)


def random_checkers(rng, n):
    rec = blank(n)
    rec[:, 0:3] = f32r(rng.uniform(-50, 50, (n, 3)))
    rec[:, 3] = f32r(rng.choice([0.32, 0.5, 0.11, 2.0, 0.7], n))
    return rec


def random_triangles(rng, n):
    rec = blank(n)
    v0 = rng.uniform(-5, 5, (n, 3))
    v1 = v0 + rng.uniform(-2, 2, (n, 3))
    v2 = v0 + rng.uniform(-2, 2, (n, 3))
    w = rng.dirichlet([1, 1, 1], n) * rng.uniform(0.2, 1.8, (n, 1))
    aim = v0 * w[:, :1] + v1 * w[:, 1:2] + v2 * w[:, 2:3] + (1 - w.sum(1, keepdims=True)) * v0
    o = rng.uniform(-10, 10, (n, 3))
    d = (aim - o) * rng.uniform(0.2, 3, (n, 1))
    rec[:, 0:3], rec[:, 3:6], rec[:, 6:9], rec[:, 9:12], rec[:, 12:15] = f32r(v0), f32r(v1), f32r(v2), f32r(o), f32r(d)
    rec[:, 15], rec[:, 16] = 1e-3, np.inf
    return rec


def random_scan_blocks(rng, n):
    """Blocks of four spheres of the flat list's scan streams (RAYZ_KAT_SCAN_DISCS): small spheres on a plane as
    randomBouncing makes them, a quarter of the blocks with the r = 1000 ground sphere in slot 0, rays aimed near one
    sphere of the block so that about a third of the tests are candidates; half the blocks y-moving."""
    rec = blank(n)
    cx, cz = rng.uniform(-50, 50, (n, 4)), rng.uniform(-50, 50, (n, 4))
    r = rng.uniform(0.15, 1.0, (n, 4))
    cy = r.copy()
    big = rng.random(n) < 0.25
    cx[big, 0], cy[big, 0], cz[big, 0], r[big, 0] = 0.0, -1000.0, 0.0, 1000.0
    cls = (rng.random(n) < 0.5).astype(np.float64)
    cls[big] = 0.0
    vy = rng.uniform(0, 0.5, (n, 4)) * cls[:, None]
    o = np.stack([rng.uniform(-20, 20, n), rng.uniform(0.2, 6, n), rng.uniform(-20, 20, n)], 1)
    k = rng.integers(0, 4, n)
    tgt = np.stack([cx[np.arange(n), k], cy[np.arange(n), k], cz[np.arange(n), k]], 1)
    aim = tgt + unit(rng.normal(size=(n, 3))) * (r[np.arange(n), k] * rng.uniform(0, 3.0, n))[:, None]
    d = (aim - o) * rng.uniform(0.2, 3.0, (n, 1))
    rec[:, 0:4], rec[:, 4:8], rec[:, 8:12], rec[:, 12:16], rec[:, 16:20] = f32r(cx), f32r(cy), f32r(cz), f32r(r), f32r(vy)
    rec[:, 20:23], rec[:, 23:26], rec[:, 26], rec[:, 27] = f32r(o), f32r(d), uniforms(rng, n), cls
    return rec
