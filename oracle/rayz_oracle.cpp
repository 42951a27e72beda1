// rayz_oracle.cpp — CPU restatement of jlucier/rayz's `Tracer.render()` path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under rayz_amd/ may include, link, load or call this file;
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, as the checker.
//
// Parity status: the reference is Zig and cannot be built in this image (no zig toolchain), its
// seed comes from getrandom and is never printed (src/renderer.zig:55-59), and none of its own
// tests covers hitInner, scatter, bounceRay or a rendered image.  The functions the
// reference DOES test are pinned by tests/test_oracle_kat.py against those vectors (refract
// src/material.zig:213-223, get ray src/renderer.zig:129-149, sphere bbox src/geom.zig:69-84,
// AABB src/hit.zig:237-279, V3 src/vec.zig:169-215, utils src/utils.zig:15-32).  The rendered image
// and the RNG stream (Zig std, not under /root/reference) are "parity unpinned" against the real
// reference; mode A below is a line-by-line restatement instead.
//
// Two modes:
//   mode A — the reference as written: f64, one sequential xoshiro256++ stream (Zig DefaultPrng),
//            recursive BVH build + traversal, recursive bounceRay, tmin as given (1e-10 there).
//            No FMA contraction (built with -ffp-contract=off).  It is the CPU baseline.
//   mode B — the arithmetic the HIP kernel is specified to perform: real = f32 or
//            f64, one PCG32 stream per (pixel, sample), iterative bounce loop, flat hit list in
//            device order (static spheres, then moving ones), explicit fma in stated places; the
//            per-sphere reject test runs in `real`, an accepted candidate's roots are computed in
//            f64 from the pool's f64 sphere (no f32 self-intersection on the r=1000 ground).
//            The HIP path must reproduce mode B bit for bit.
//
// All file:line citations are into /root/reference (jlucier/rayz @ 2025-07-25).

#include "../include/rayz_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint64_t u64;
typedef uint32_t u32;

namespace {

// ------------------------------------------------------------------------------------------------
// RNGs
// ------------------------------------------------------------------------------------------------

// Zig std.Random.SplitMix64 (Vigna's splitmix64; third-party to the reference, see header).
struct SplitMix64 {
    u64 s;
    u64 next() {
        s += 0x9e3779b97f4a7c15ull;
        u64 z = s;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
};

static inline u64 rotl64(u64 x, int k) { return (x << k) | (x >> (64 - k)); }

// Zig std.Random.DefaultPrng = Xoshiro256 (xoshiro256++), seeded through SplitMix64.
// Call sites: src/renderer.zig:22,55-59.
struct Xoshiro256pp {
    u64 s[4];
    void seed(u64 v) {
        SplitMix64 g{v};
        for (int i = 0; i < 4; ++i) s[i] = g.next();
    }
    u64 next() {
        const u64 r = rotl64(s[0] + s[3], 23) + s[0];
        const u64 t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl64(s[3], 45);
        return r;
    }
    // Zig 0.13/0.14 std.Random.float(f64): 52 mantissa bits from one u64, exponent from its
    // leading-zero count (restated from memory of Zig std; source not in the container).
    double float64() {
        const u64 r = next();
        unsigned lz = r ? (unsigned)__builtin_clzll(r) : 64u;
        if (lz >= 12) {
            lz = 12;
            for (;;) {
                const u64 a = next();
                const unsigned alz = a ? (unsigned)__builtin_clzll(a) : 64u;
                lz += alz;
                if (alz != 64) break;
                if (lz >= 1022) {
                    lz = 1022;
                    break;
                }
            }
        }
        const u64 mant = r & 0xFFFFFFFFFFFFFull;
        const u64 expo = (u64)(1022 - lz) << 52;
        const u64 bits = expo | mant;
        double d;
        std::memcpy(&d, &bits, 8);
        return d;
    }
};

// PCG32 (O'Neill, XSH-RR 64/32), one stream per path.  DESIGN.md §4.1.
static inline u64 mix64(u64 z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
struct Pcg32 {
    u64 state, inc;
    void srandom(u64 initstate, u64 initseq) {
        state = 0;
        inc = (initseq << 1) | 1u;
        next();
        state += initstate;
        next();
    }
    u32 next() {
        const u64 old = state;
        state = old * 6364136223846793005ull + inc;
        const u32 xorshifted = (u32)(((old >> 18u) ^ old) >> 27u);
        const u32 rot = (u32)(old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
    }
    // key of path (pixel_index, sample): DESIGN.md §4.1
    void seed_path(u64 seed, u64 path_id) {
        const u64 a = mix64(seed + (path_id + 1) * 0x9e3779b97f4a7c15ull);
        const u64 b = mix64(a + 0x9e3779b97f4a7c15ull);
        srandom(a, b);
    }
};

// ------------------------------------------------------------------------------------------------
// Mode A — the reference as written (f64)
// ------------------------------------------------------------------------------------------------
namespace A {

// src/vec.zig:4-157
struct V3 {
    double x = 0, y = 0, z = 0;
    static V3 of(double v) { return {v, v, v}; }
    double at(int ax) const { return ax == 0 ? x : (ax == 1 ? y : z); }      // :26-33
    V3 add(V3 o) const { return {x + o.x, y + o.y, z + o.z}; }               // :47-53
    V3 sub(V3 o) const { return {x - o.x, y - o.y, z - o.z}; }               // :55-61
    V3 mul(double v) const { return {x * v, y * v, z * v}; }                 // :63-65
    V3 div(double v) const { return mul(1 / v); }                            // :67-69
    double dot(V3 o) const { return x * o.x + y * o.y + z * o.z; }           // :95-97
    double mag() const { return std::sqrt(dot(*this)); }                     // :71-73
    V3 unit() const { return div(mag()); }                                   // :75-77
    V3 cross(V3 o) const {                                                   // :99-105
        return {y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x};
    }
    bool nearZero() const {                                                  // :107-110
        const double tol = 1e-8;
        return std::fabs(x) <= tol && std::fabs(y) <= tol && std::fabs(z) <= tol;
    }
    bool close(V3 o) const { return sub(o).nearZero(); }                     // :112-114
    V3 vmul(V3 o) const { return {x * o.x, y * o.y, z * o.z}; }              // :118-124
    V3 vdiv(V3 o) const { return {x / o.x, y / o.y, z / o.z}; }              // :126-132
    V3 vmin(V3 o) const { return {std::fmin(x, o.x), std::fmin(y, o.y), std::fmin(z, o.z)}; } // :134-140
    V3 vmax(V3 o) const { return {std::fmax(x, o.x), std::fmax(y, o.y), std::fmax(z, o.z)}; } // :142-148
    int amax() const {                                                       // :150-156
        if (x > y) return x > z ? 0 : 2;
        return y > z ? 1 : 2;
    }
    V3 sqrt() const {                                                        // :87-93
        return {x > 0 ? std::sqrt(x) : 0, y > 0 ? std::sqrt(y) : 0, z > 0 ? std::sqrt(z) : 0};
    }
    V3 clamp(double lo, double hi) const {                                   // :79-85, src/utils.zig:3-13
        auto c = [&](double v) {
            const double m = v > lo ? v : lo;
            return m < hi ? m : hi;
        };
        return {c(x), c(y), c(z)};
    }
};
static V3 v3(const double* p) { return {p[0], p[1], p[2]}; }
// The reference's functions take a `std.Random`; here the generator type is a template parameter so that the
// known-answer entry can hand the SAME functions a list of uniforms instead of the xoshiro stream.
struct ListRng {
    const double* u;
    u32 n, i;
    double float64() {
        const double v = i < n ? u[i] : 0.5;
        ++i;
        return v;
    }
};
template <class G> static V3 v3random(G& r, double lo, double hi) {          // src/vec.zig:9-16
    const double scale = hi - lo;
    V3 v;
    v.x = r.float64() * scale + lo;
    v.y = r.float64() * scale + lo;
    v.z = r.float64() * scale + lo;
    return v;
}

struct Ray {                                                                 // src/vec.zig:159-167
    V3 origin, dir;
    double time = 0;
    V3 at(double t) const { return origin.add(dir.mul(t)); }
};

struct AABB {                                                                // src/hit.zig:44-99
    V3 low = V3::of(std::numeric_limits<double>::infinity());
    V3 high = V3::of(-std::numeric_limits<double>::infinity());
    static AABB init(V3 a, V3 b) { return {a.vmin(b), a.vmax(b)}; }
    static AABB enclose(const AABB& a, const AABB& b) { return {a.low.vmin(b.low), a.high.vmax(b.high)}; }
    int longestAxis() const { return high.sub(low).amax(); }
    bool hit(const Ray& ray, double tmin, double tmax) const {               // :70-98
        const V3 t0s = low.sub(ray.origin).vdiv(ray.dir);
        const V3 t1s = high.sub(ray.origin).vdiv(ray.dir);
        double t0 = tmin, t1 = tmax;
        for (int ax = 0; ax < 3; ++ax) {
            const double v0 = t0s.at(ax), v1 = t1s.at(ax);
            if (v0 < v1) {
                t0 = std::fmax(v0, t0);
                t1 = std::fmin(v1, t1);
            } else {
                t0 = std::fmax(v1, t0);
                t1 = std::fmin(v0, t1);
            }
        }
        return t1 > t0;
    }
};

struct Hit {                                                                 // src/hit.zig:16-42
    V3 point, normal;
    double t = 0;
    bool front_face = false;
    u32 material = 0;
    bool valid = false;
    static Hit init(const Ray& ray, V3 point, V3 normal, double t, u32 material) {
        Hit h;
        h.front_face = normal.dot(ray.dir) < 0;
        h.point = point;
        h.normal = h.front_face ? normal : normal.mul(-1);
        h.t = t;
        h.material = material;
        h.valid = true;
        return h;
    }
};

struct Sphere {                                                              // src/geom.zig:11-67
    Ray center;
    double radius;
    u32 material;
    AABB boundingBox() const {                                               // :24-31
        const V3 rad = V3::of(radius);
        const V3 o1 = center.origin, o2 = center.at(1);
        return AABB::enclose(AABB::init(o1.sub(rad), o1.add(rad)), AABB::init(o2.sub(rad), o2.add(rad)));
    }
    Hit hitInner(const Ray& ray, double tmin, double tmax) const {           // :38-66
        const V3 origin_now = center.at(ray.time);
        const V3 offset = origin_now.sub(ray.origin);
        const double a = ray.dir.dot(ray.dir);
        const double half_b = ray.dir.dot(offset);
        const double c = offset.dot(offset) - radius * radius;
        const double disc = half_b * half_b - a * c;
        if (disc < 0) return Hit{};
        const double rt = std::sqrt(disc);
        const double t1 = (half_b - rt) / a, t2 = (half_b + rt) / a;
        double t;
        if (t1 >= tmin && t1 <= tmax) t = t1;
        else if (t2 >= tmin && t2 <= tmax) t = t2;
        else return Hit{};
        const V3 point = ray.at(t);
        const V3 n = point.sub(origin_now).unit();
        return Hit::init(ray, point, n, t, material);
    }
};

// BUILD-DEFINED triangle hittable (the reference has spheres only): Möller–Trumbore, two-sided, nearest root in
// [tmin, tmax], geometric normal handed to Hit.init.  Its box is padded by 1e-4 per side: AABB.hit's strict
// `t1 > t0` (src/hit.zig:97) would never hit the flat box of an axis-aligned triangle.
struct Triangle {
    V3 v0, v1, v2;
    u32 material;
    AABB boundingBox() const {
        const V3 pad = V3::of(1e-4);
        return AABB{v0.vmin(v1).vmin(v2).sub(pad), v0.vmax(v1).vmax(v2).add(pad)};
    }
    Hit hitInner(const Ray& ray, double tmin, double tmax) const {
        const V3 e1 = v1.sub(v0), e2 = v2.sub(v0);
        const V3 p = ray.dir.cross(e2);
        const double det = e1.dot(p);
        if (det == 0) return Hit{};
        const V3 s = ray.origin.sub(v0);
        const double u = s.dot(p) / det;
        const V3 q = s.cross(e1);
        const double v = ray.dir.dot(q) / det;
        if (u < 0 || v < 0 || u + v > 1) return Hit{};
        const double t = e2.dot(q) / det;
        if (!(t >= tmin && t <= tmax)) return Hit{};
        return Hit::init(ray, ray.at(t), e1.cross(e2).unit(), t, material);
    }
};

struct Hittable {                                                            // src/hit.zig:8-12
    AABB bbox;
    u32 sphere; // hittable index: spheres first, then triangles (index - n_spheres)
};

struct Counters {
    u64 segments = 0, sphere_tests = 0, node_tests = 0;
};

struct Scene {
    std::vector<Sphere> spheres;
    std::vector<Triangle> triangles;
    Hit hitPrim(u32 i, const Ray& ray, double tmin, double tmax) const {
        return i < spheres.size() ? spheres[i].hitInner(ray, tmin, tmax)
                                  : triangles[i - spheres.size()].hitInner(ray, tmin, tmax);
    }
    std::vector<RayzMaterial> materials;
    std::vector<RayzTexture> textures;
    std::vector<Hittable> hittables;
};

struct BVH {                                                                 // src/hit.zig:101-217
    struct Node {
        AABB bbox;
        size_t starti = 0, endi = 0;
        int left = -1, right = -1;
    };
    std::vector<Node> nodes;
    int build(std::vector<Hittable>& h, size_t si, size_t ei) {              // :130-161
        const int me = (int)nodes.size();
        nodes.push_back(Node{});
        const size_t nobjs = ei - si;
        AABB bb;
        for (size_t i = si; i < ei; ++i) bb = AABB::enclose(bb, h[i].bbox);
        nodes[me].bbox = bb;
        if (nobjs <= 2) {
            nodes[me].starti = si;
            nodes[me].endi = ei;
        } else {
            const int ax = bb.longestAxis();
            // std.mem.sort is a stable sort (Zig std block sort)
            std::stable_sort(h.begin() + si, h.begin() + ei, [ax](const Hittable& a, const Hittable& b) {
                return a.bbox.low.at(ax) < b.bbox.low.at(ax);
            });
            const size_t mid = nobjs / 2 + si;
            const int l = build(h, si, mid);
            const int r = build(h, mid, ei);
            nodes[me].left = l;
            nodes[me].right = r;
        }
        return me;
    }
    Hit findHit(int ni, const Scene& sc, const Ray& ray, double tmin, double tmax, Counters& c) const { // :181-216
        const Node& n = nodes[ni];
        c.node_tests++;
        if (!n.bbox.hit(ray, tmin, tmax)) return Hit{};
        Hit maybe;
        if (n.left >= 0) {
            maybe = findHit(n.left, sc, ray, tmin, tmax, c);
            const double maxt = maybe.valid ? maybe.t : tmax;
            const Hit nh = findHit(n.right, sc, ray, tmin, maxt, c);
            if (nh.valid) maybe = nh;
            return maybe;
        }
        for (size_t i = n.starti; i < n.endi; ++i) {
            const double maxt = maybe.valid ? maybe.t : tmax;
            c.sphere_tests++;
            const Hit nh = sc.hitPrim(sc.hittables[i].sphere, ray, tmin, maxt);
            if (nh.valid) maybe = nh;
        }
        return maybe;
    }
};

// src/material.zig:32-36: which of a checker's two textures covers `point`
static int64_t checkerParity(V3 point, double scale) {
    const int64_t x = (int64_t)std::floor(point.x / scale);
    const int64_t y = (int64_t)std::floor(point.y / scale);
    const int64_t z = (int64_t)std::floor(point.z / scale);
    const int64_t s = x + y + z;
    return ((s % 2) + 2) % 2; // @mod: floored
}
// src/material.zig:19-51
static V3 textureValue(const Scene& sc, u32 idx, V3 point) {
    const RayzTexture& t = sc.textures[idx];
    if (t.kind == RAYZ_TEX_SOLID) return v3(t.color);
    return textureValue(sc, checkerParity(point, t.scale) == 0 ? t.even : t.odd, point);
}
// src/renderer.zig:124-125 (not a lerp)
static V3 background(V3 dir) {
    const double t = 0.5 * (dir.unit().y + 1.0);
    return V3::of(1).mul(1.0 - t).add(V3{0.5, 0.7, 1.0}).mul(t);
}

template <class G> static V3 randomInUnitSphere(G& r) {                      // src/material.zig:196-202
    for (;;) {
        const V3 v = v3random(r, -1, 1);
        if (v.mag() <= 1) return v;
    }
}
template <class G> static V3 randomUnit(G& r) { return randomInUnitSphere(r).unit(); } // :204-206
template <class G> static V3 randomInHemisphere(G& r, V3 n) {                // :208-211
    const V3 v = randomInUnitSphere(r);
    return v.dot(n) > 0 ? v : v.mul(-1);
}
static double reflectance(double cos, double ri) {                           // :179-183
    double r0 = (1 - ri) / (1 + ri);
    r0 *= r0;
    return r0 + (1 - r0) * std::pow(1 - cos, 5);
}
static V3 reflect(const Ray& ray, const Hit& hit) {                          // :185-187
    return ray.dir.sub(hit.normal.mul(2 * ray.dir.dot(hit.normal)));
}
static V3 refract(V3 unit_dir, V3 norm, double eta) {                        // :189-194
    const double cos_theta = unit_dir.mul(-1).dot(norm);
    const V3 perp = norm.mul(cos_theta).add(unit_dir).mul(eta);
    const V3 par = norm.mul(-std::sqrt(1 - perp.dot(perp)));
    return perp.add(par);
}

struct Scatter {
    bool ok = false;
    Ray ray;
    V3 att;
};

template <class G>
static Scatter scatter(const Scene& sc, const RayzMaterial& m, G& rng, const Ray& ray, const Hit& hit) {
    Scatter s;
    if (m.kind == RAYZ_MAT_DIFFUSE) {                                        // src/material.zig:77-101
        V3 target;
        if (m.method == RAYZ_DIFFUSE_UNIT_SPHERE) target = hit.point.add(hit.normal).add(randomInUnitSphere(rng));
        else if (m.method == RAYZ_DIFFUSE_UNIT_SPHERE_SURFACE) target = hit.point.add(hit.normal).add(randomUnit(rng));
        else target = hit.point.add(randomInHemisphere(rng, hit.normal));
        if (target.nearZero()) target = hit.normal;
        s.ok = true;
        s.ray.origin = hit.point;
        s.ray.dir = target.sub(hit.point);
        s.ray.time = ray.time;
        s.att = textureValue(sc, m.texture, hit.point);
    } else if (m.kind == RAYZ_MAT_METALLIC) {                                // :108-131
        V3 d = reflect(ray, hit).unit();
        if (m.param > 0) d = d.add(randomUnit(rng).mul(std::fmin(m.param, 1.0)));
        if (d.dot(hit.normal) <= 0) return s;
        s.ok = true;
        s.ray.origin = hit.point;
        s.ray.dir = d;
        s.ray.time = ray.time;
        s.att = textureValue(sc, m.texture, hit.point);
    } else {                                                                 // :137-159
        const double eta = hit.front_face ? 1 / m.param : m.param;
        const V3 ud = ray.dir.unit();
        const double cos_theta = ud.mul(-1).dot(hit.normal);
        const double sin_theta = std::sqrt(1 - cos_theta * cos_theta);
        V3 dir;
        if (eta * sin_theta > 1.0 || reflectance(cos_theta, eta) > rng.float64()) dir = reflect(ray, hit);
        else dir = refract(ud, hit.normal, eta);
        s.ok = true;
        s.ray.origin = hit.point;
        s.ray.dir = dir;
        s.ray.time = ray.time;
        s.att = V3::of(1);
    }
    return s;
}

struct Camera {                                                              // src/camera.zig:8-91
    V3 look_from, px_du, px_dv, px_origin, defocus_u, defocus_v;
    bool defocus = false;
    static Camera init(double vfov, double focus_dist, double defocus_angle, V3 look_from, V3 look_at, V3 vup,
                       size_t img_h, size_t img_w) {                         // :18-57
        const double DEG = M_PI / 180.0;
        const double fh = (double)img_h, fw = (double)img_w;
        const double vp_h = 2 * std::tan(vfov * DEG / 2.0) * focus_dist;
        const double vp_w = vp_h * fw / fh;
        const V3 w = look_from.sub(look_at).unit();
        const V3 u = vup.cross(w).unit();
        const V3 v = w.cross(u);
        const V3 vp_u = u.mul(vp_w), vp_v = v.mul(-vp_h);
        const V3 px_du = vp_u.div(fw), px_dv = vp_v.div(fh);
        const double dr = std::tan(defocus_angle * DEG / 2) * focus_dist;
        const V3 vp_origin =
            look_from.sub(w.mul(focus_dist)).sub(vp_u.div(2)).sub(vp_v.div(2)).add(px_du.add(px_dv).mul(0.5));
        Camera c;
        c.look_from = look_from;
        c.px_du = px_du;
        c.px_dv = px_dv;
        c.px_origin = vp_origin;
        c.defocus_u = u.mul(dr);
        c.defocus_v = v.mul(dr);
        c.defocus = defocus_angle > 0;
        return c;
    }
    template <class G> V3 randomInDefocus(G& r) const {                      // :79-90
        if (!defocus) return V3{};
        for (;;) {
            V3 v;
            v.x = r.float64() * 2 - 1;
            v.y = r.float64() * 2 - 1;
            v.z = 0;
            if (v.dot(v) <= 1) return defocus_u.mul(v.x).add(defocus_v.mul(v.y));
        }
    }
    template <class G> Ray getRay(size_t px, size_t py, G* r) const {        // :59-77
        double x = (double)px, y = (double)py;
        V3 origin = look_from;
        if (r) {
            x += r->float64() - 0.5;
            y += r->float64() - 0.5;
            origin = origin.add(randomInDefocus(*r));
        }
        Ray ray;
        ray.dir = px_du.mul(x).add(px_dv.mul(y)).add(px_origin).sub(origin);
        ray.origin = origin;
        ray.time = r ? r->float64() : 0;
        return ray;
    }
};

static Camera cameraFrom(const RayzCameraDesc& d) {
    Camera c;
    c.look_from = v3(d.look_from);
    c.px_du = v3(d.px_du);
    c.px_dv = v3(d.px_dv);
    c.px_origin = v3(d.px_origin);
    c.defocus_u = v3(d.defocus_u);
    c.defocus_v = v3(d.defocus_v);
    c.defocus = d.defocus != 0;
    return c;
}
static void cameraTo(const Camera& c, RayzCameraDesc* d) {
    auto put = [](double* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; };
    put(d->look_from, c.look_from);
    put(d->px_du, c.px_du);
    put(d->px_dv, c.px_dv);
    put(d->px_origin, c.px_origin);
    put(d->defocus_u, c.defocus_u);
    put(d->defocus_v, c.defocus_v);
    d->defocus = c.defocus ? 1u : 0u;
    d->_pad = 0;
}

struct Tracer {                                                              // src/renderer.zig:18-126
    Scene sc;
    BVH bvh;
    Camera cam;
    Xoshiro256pp rng;
    double tmin = 1e-10;
    bool linear = false;
    Counters cnt;

    Hit findHit(const Ray& ray) {
        cnt.segments++;
        const double inf = std::numeric_limits<double>::infinity();
        if (sc.hittables.empty()) return Hit{};
        if (!linear) return bvh.findHit(0, sc, ray, tmin, inf, cnt);
        // flat hit list: same acceptance rule as a BVH leaf (src/hit.zig:208-214) over all hittables
        Hit maybe;
        for (size_t i = 0; i < sc.hittables.size(); ++i) {
            const double maxt = maybe.valid ? maybe.t : inf;
            cnt.sphere_tests++;
            const Hit nh = sc.hitPrim(sc.hittables[i].sphere, ray, tmin, maxt);
            if (nh.valid) maybe = nh;
        }
        return maybe;
    }
    V3 bounceRay(const Ray& ray, size_t depth) {                             // :103-126
        if (depth == 0) return V3{};
        const Hit hit = findHit(ray);
        if (hit.valid) {
            V3 ret;
            const RayzMaterial& m = sc.materials[hit.material];
            const Scatter s = scatter(sc, m, rng, ray, hit);
            if (s.ok) ret = bounceRay(s.ray, depth - 1).vmul(s.att);
            return ret;
        }
        return background(ray.dir);
    }
};

} // namespace A

// ------------------------------------------------------------------------------------------------
// Mode B — the kernel arithmetic, real = float | double
// ------------------------------------------------------------------------------------------------
namespace B {

template <class R> struct V {
    R x, y, z;
};
template <class R> static inline R fm(R a, R b, R c) { return std::fma(a, b, c); }
template <class R> static inline R dot3(V<R> a, V<R> b) { return fm(a.z, b.z, fm(a.y, b.y, a.x * b.x)); }
template <class R> static inline V<R> scale(V<R> a, R s) { return {a.x * s, a.y * s, a.z * s}; }
template <class R> static inline V<R> neg(V<R> a) { return {-a.x, -a.y, -a.z}; }
template <class R> static inline V<R> unit(V<R> a) {
    const R m = std::sqrt(dot3(a, a));
    const R inv = R(1) / m;
    return scale(a, inv);
}

template <class R> struct Rng {
    Pcg32 g;
    R uniform();
};
template <> inline float Rng<float>::uniform() { return (float)(g.next() >> 8) * 0x1p-24f; }
template <> inline double Rng<double>::uniform() { return (double)g.next() * 0x1p-32; }

template <class R> struct Tex {
    u32 kind, even, odd;
    R scale;
    V<R> color;
};
template <class R> struct Mat {
    u32 kind, texture, method;
    R param;     // fuzz | ior
    R inv_param; // 1/ior
};
template <class R> struct Sph {
    V<R> c;     // broad phase, narrowed to R
    double radius;
    V<R> v;
    V<float> cf, vf; // the reject test's copy: f32 for both precisions (the test only filters, §4.3)
    float r2f;       //   .. the PADDED square of the conservative filter: (r + E)² rounded up, padRadius2Scan()
    double c64[3], v64[3], r2_64; // narrow phase: the pool's own f64 values
    u32 mat;
    u32 pool; // index in MemPool.spheres
};
template <class R> struct Tri { // build-defined triangle: v0, e1 = v1 - v0, e2 = v2 - v0 (subtracted in f64, narrowed)
    V<R> v0, e1, e2;
    u32 mat;
};
template <class R> struct SceneB {
    std::vector<Tri<R>> tri; // hittable index = n_spheres + position
    std::vector<Sph<R>> sph; // scan order (any: the result does not depend on it); here static, then moving
    std::vector<u32> by_pool; // pool index → position in sph
    u32 n_static = 0;
    // BVH traversal: the mode-A tree (src/hit.zig:130-161) in depth-first pre-order with skip links; boxes
    // narrowed outward to R
    struct Node {
        float lo[3], hi[3]; // f32 for both precisions, rounded outward: the box test only culls
        u32 skip, first, count;
    };
    std::vector<Node> nodes;
    std::vector<u32> leaf_order; // pool indices in leaf order
    std::vector<Mat<R>> mats;
    std::vector<Tex<R>> texs;
};
template <class R> struct CamB {
    V<R> from, du, dv, pxo, defu, defv;
    bool defocus;
};

template <class R> static V<R> narrow3(const double* p) { return {(R)p[0], (R)p[1], (R)p[2]}; }

template <class R> static R roundUp(double v);
static double norm3(const double* v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
// Conservative reject filter: r_pad = r + E, E = 32·u·(|c| + |v| + r + S); u = unit roundoff of f32,
// S = a bound on |o| of every ray.  Restated here, not shared with the product.
// The filter runs in f32 for both precisions.  For R = double the ray reaches it narrowed to f32 (origin,
// unit direction, time: ≤ u·S + u·(|c| + S) + u·|v| more on the line's distance to the centre): pad 40u instead of 32u.
template <class R> static float padRadius2Scan(const RayzSphere& q, double S) {
    const double u = (double)std::numeric_limits<float>::epsilon() / 2;
    const double E = (sizeof(R) == 4 ? 32.0 : 40.0) * u * (norm3(q.center) + norm3(q.velocity) + std::fabs(q.radius) + S);
    const double rp = std::fabs(q.radius) + E;
    return roundUp<float>(rp * rp);
}
static double originBound(const RayzSceneDesc& d, const RayzCameraDesc* c) {
    double S = 0;
    for (u32 i = 0; i < d.n_spheres; ++i)
        S = std::max(S, norm3(d.spheres[i].center) + norm3(d.spheres[i].velocity) + std::fabs(d.spheres[i].radius));
    for (u32 i = 0; i < d.n_triangles; ++i)
        S = std::max({S, norm3(d.triangles[i].v0), norm3(d.triangles[i].v1), norm3(d.triangles[i].v2)});
    S *= 1.0 + 1e-3;
    if (c) S = std::max(S, norm3(c->look_from) + norm3(c->defocus_u) + norm3(c->defocus_v));
    return S;
}

template <class R> static SceneB<R> buildScene(const RayzSceneDesc& d, double S) {
    SceneB<R> s;
    for (int pass = 0; pass < 2; ++pass) {
        for (u32 i = 0; i < d.n_spheres; ++i) {
            const RayzSphere& q = d.spheres[i];
            const bool moving = q.velocity[0] != 0 || q.velocity[1] != 0 || q.velocity[2] != 0;
            if ((pass == 1) != moving) continue;
            Sph<R> o;
            o.c = narrow3<R>(q.center);
            o.v = narrow3<R>(q.velocity);
            o.cf = narrow3<float>(q.center);
            o.vf = narrow3<float>(q.velocity);
            o.r2f = padRadius2Scan<R>(q, S);
            o.radius = q.radius;
            for (int k = 0; k < 3; ++k) o.c64[k] = q.center[k], o.v64[k] = q.velocity[k];
            o.r2_64 = q.radius * q.radius;
            o.mat = q.material;
            o.pool = i;
            s.sph.push_back(o);
        }
        if (pass == 0) s.n_static = (u32)s.sph.size();
    }
    s.by_pool.resize(s.sph.size());
    for (u32 k = 0; k < s.sph.size(); ++k) s.by_pool[s.sph[k].pool] = k;
    for (u32 i = 0; i < d.n_triangles; ++i) {
        const RayzTriangle& q = d.triangles[i];
        Tri<R> o;
        o.v0 = narrow3<R>(q.v0);
        for (int pass = 0; pass < 2; ++pass) {
            const double* w = pass ? q.v2 : q.v1;
            const V<R> e{(R)(w[0] - q.v0[0]), (R)(w[1] - q.v0[1]), (R)(w[2] - q.v0[2])};
            (pass ? o.e2 : o.e1) = e;
        }
        o.mat = q.material;
        s.tri.push_back(o);
    }
    for (u32 i = 0; i < d.n_materials; ++i) {
        const RayzMaterial& m = d.materials[i];
        Mat<R> o;
        o.kind = m.kind;
        o.texture = m.texture;
        o.method = m.method;
        o.param = (R)m.param;
        o.inv_param = R(1) / o.param;
        s.mats.push_back(o);
    }
    for (u32 i = 0; i < d.n_textures; ++i) {
        const RayzTexture& t = d.textures[i];
        Tex<R> o;
        o.kind = t.kind;
        o.even = t.even;
        o.odd = t.odd;
        o.scale = (R)t.scale;
        o.color = narrow3<R>(t.color);
        s.texs.push_back(o);
    }
    return s;
}

template <class R> static R roundDown(double v) {
    R r = (R)v;
    if ((double)r > v) r = std::nextafter(r, -std::numeric_limits<R>::infinity());
    return r;
}
template <class R> static R roundUp(double v) {
    R r = (R)v;
    if ((double)r < v) r = std::nextafter(r, std::numeric_limits<R>::infinity());
    return r;
}

// padding of the BVH's boxes as mode B's slab test sees them (boxHit below carries no slack of its own)
static constexpr double kBoxPadUlps = 16.0;
static inline double boxPad(double S, double B) { return kBoxPadUlps * (std::numeric_limits<float>::epsilon() / 2.0) * std::max(S, B); }
// index one past the subtree of node ni (pre-order layout)
static u32 subtreeEnd(const A::BVH& t, int ni) {
    const A::BVH::Node& n = t.nodes[ni];
    return n.left < 0 ? (u32)ni + 1 : subtreeEnd(t, n.right);
}

template <class R> static void buildBvh(const RayzSceneDesc& d, SceneB<R>& s, double S) { // S: bound on every ray origin
    if (d.n_spheres + d.n_triangles == 0) return;
    std::vector<A::Hittable> hs;
    for (u32 i = 0; i < d.n_spheres; ++i) {
        A::Sphere q;
        q.center.origin = A::v3(d.spheres[i].center);
        q.center.dir = A::v3(d.spheres[i].velocity);
        q.radius = d.spheres[i].radius;
        hs.push_back({q.boundingBox(), i});
    }
    for (u32 i = 0; i < d.n_triangles; ++i) {
        A::Triangle t;
        t.v0 = A::v3(d.triangles[i].v0), t.v1 = A::v3(d.triangles[i].v1), t.v2 = A::v3(d.triangles[i].v2);
        hs.push_back({t.boundingBox(), d.n_spheres + i});
    }
    A::BVH t;
    t.build(hs, 0, hs.size());
    double B = 0;
    for (const A::BVH::Node& n : t.nodes)
        for (int k = 0; k < 3; ++k) B = std::max({B, std::fabs(n.bbox.low.at(k)), std::fabs(n.bbox.high.at(k))});
    const double pad = boxPad(S, B); // the slab test is bare: the boxes carry its slack (boxHit)
    for (size_t i = 0; i < t.nodes.size(); ++i) {
        const A::BVH::Node& n = t.nodes[i];
        typename SceneB<R>::Node o;
        for (int k = 0; k < 3; ++k) o.lo[k] = roundDown<float>(n.bbox.low.at(k) - pad), o.hi[k] = roundUp<float>(n.bbox.high.at(k) + pad);
        o.skip = subtreeEnd(t, (int)i);
        o.first = n.left < 0 ? (u32)n.starti : 0;
        o.count = n.left < 0 ? (u32)(n.endi - n.starti) : 0;
        s.nodes.push_back(o);
    }
    for (const A::Hittable& h : hs) s.leaf_order.push_back(h.sphere);
}

template <class R> static V<R> cross3(V<R> a, V<R> b) {
    return {fm(a.y, b.z, -(a.z * b.y)), fm(a.z, b.x, -(a.x * b.z)), fm(a.x, b.y, -(a.y * b.x))};
}

template <class R> static CamB<R> buildCamera(const RayzCameraDesc& d) {
    CamB<R> c;
    c.from = narrow3<R>(d.look_from);
    c.du = narrow3<R>(d.px_du);
    c.dv = narrow3<R>(d.px_dv);
    c.pxo = narrow3<R>(d.px_origin);
    c.defu = narrow3<R>(d.defocus_u);
    c.defv = narrow3<R>(d.defocus_v);
    c.defocus = d.defocus != 0;
    return c;
}

static const int kMaxRejectionTries = 64;
static const int kMaxTextureDepth = 8;

// ---- the pieces of the kernel arithmetic, one function each.  tracePath() below strings them together
// exactly as the trace kernels do; the known-answer entry (rayz_oracle_kat_b) calls them one at a time, as
// rayz_hip_kat does with their device twins. -----------------------------------------------------------------------
template <class R> struct ListRng { // draws from a list (0.5 beyond its end): known-answer entry only
    const double* u;
    u32 n, i;
    R uniform() {
        const R v = i < n ? (R)u[i] : R(0.5);
        ++i;
        return v;
    }
};

template <class R, class G> static V<R> randomInUnitSphere(G& g) {           // src/material.zig:196-202
    V<R> v{0, 0, 0};
    for (int i = 0; i < kMaxRejectionTries; ++i) {
        v.x = fm(g.uniform(), R(2), R(-1));                                  // V3.random x,y,z order, src/vec.zig:9-16
        v.y = fm(g.uniform(), R(2), R(-1));
        v.z = fm(g.uniform(), R(2), R(-1));
        if (std::sqrt(dot3(v, v)) <= R(1)) break;
    }
    return v;
}

template <class R> static u32 checkerParity(V<R> p, R scale) {               // src/material.zig:32-36
    const R lim = R(1073741824.0);
    auto cell = [&](R c) {
        R f = std::floor(c / scale);
        f = f < -lim ? -lim : f;
        f = f > lim ? lim : f;
        return (int32_t)f;
    };
    const u32 s = (u32)cell(p.x) + (u32)cell(p.y) + (u32)cell(p.z);
    return s & 1u;
}
template <class R> static V<R> textureValue(const SceneB<R>& sc, u32 idx, V<R> p) { // src/material.zig:19-51
    for (int depth = 0; depth < kMaxTextureDepth; ++depth) {
        const Tex<R>& t = sc.texs[idx];
        if (t.kind == RAYZ_TEX_SOLID) return t.color;
        idx = checkerParity<R>(p, t.scale) == 0 ? t.even : t.odd;
    }
    return V<R>{0, 0, 0};
}

template <class R> static V<R> background(V<R> ud) {                         // src/renderer.zig:124-125
    const R t = R(0.5) * (ud.y + R(1));
    const R w = R(1) - t;
    return {(w + R(0.5)) * t, (w + R(0.7)) * t, (w + R(1.0)) * t};
}

// camera ray, src/camera.zig:59-90 (draw order: jitter x, jitter y, lens tries, time)
template <class R, class G>
static void cameraRay(const CamB<R>& cam, G& g, u32 px, u32 py, V<R>& o, V<R>& d, R& time) {
    const R x = (R)px + (g.uniform() - R(0.5));
    const R y = (R)py + (g.uniform() - R(0.5));
    o = cam.from;
    if (cam.defocus) {                                                       // :79-90
        R vx = 0, vy = 0;
        for (int i = 0; i < kMaxRejectionTries; ++i) {
            vx = fm(g.uniform(), R(2), R(-1));
            vy = fm(g.uniform(), R(2), R(-1));
            if (fm(vy, vy, vx * vx) <= R(1)) break;
        }
        o.x = cam.from.x + fm(cam.defv.x, vy, cam.defu.x * vx);
        o.y = cam.from.y + fm(cam.defv.y, vy, cam.defu.y * vx);
        o.z = cam.from.z + fm(cam.defv.z, vy, cam.defu.z * vx);
    }
    d.x = (fm(cam.dv.x, y, cam.du.x * x) + cam.pxo.x) - o.x;
    d.y = (fm(cam.dv.y, y, cam.du.y * x) + cam.pxo.y) - o.y;
    d.z = (fm(cam.dv.z, y, cam.du.z * x) + cam.pxo.z) - o.z;
    time = g.uniform();
}

// Reject-test basis: e1 ⟂ ud in the xz-plane, e2 = ud × e1.  The squared distance from the
// line to a centre c is p1² + p2² with p1 = c·e1 + k1, p2 = c·e2 + k2; e1.y = 0, so a y-velocity never enters p1.
template <class R> struct Basis {
    R e1x, e1z, e2x, e2y, e2z, k1, k2;
};
template <class R> static Basis<R> makeBasis(V<R> ud, V<R> o) {
    Basis<R> b;
    const R h2 = fm(ud.z, ud.z, ud.x * ud.x);
    b.e1x = R(1), b.e1z = R(0);
    if (h2 > R(1e-30)) {
        const R ih = R(1) / std::sqrt(h2);
        b.e1x = ud.z * ih;
        b.e1z = -(ud.x * ih);
    }
    b.e2x = ud.y * b.e1z, b.e2y = fm(ud.z, b.e1x, -(ud.x * b.e1z)), b.e2z = -(ud.y * b.e1x);
    b.k1 = -fm(o.z, b.e1z, o.x * b.e1x);
    b.k2 = -fm(o.z, b.e2z, fm(o.y, b.e2y, o.x * b.e2x));
    return b;
}
// r_pad² − p1² − p2² in R: ≥ 0 makes the sphere a candidate.  `r2` is the PADDED square (padRadius2), which makes
// the filter conservative.  A zero velocity component is skipped (fm(0, x, p) = p: same bits as the device's
// unconditional form).
template <class R> static R sphereFilter(const Basis<R>& b, R time, V<R> c, V<R> v, R r2) {
    R p1 = fm(c.z, b.e1z, fm(c.x, b.e1x, b.k1));
    if (v.x != R(0)) p1 = fm(v.x, time * b.e1x, p1);
    if (v.z != R(0)) p1 = fm(v.z, time * b.e1z, p1);
    R p2 = fm(c.z, b.e2z, fm(c.y, b.e2y, fm(c.x, b.e2x, b.k2)));
    if (v.x != R(0)) p2 = fm(v.x, time * b.e2x, p2);
    if (v.y != R(0)) p2 = fm(v.y, time * b.e2y, p2);
    if (v.z != R(0)) p2 = fm(v.z, time * b.e2z, p2);
    return fm(-p1, p1, fm(-p2, p2, r2));
}
// narrow phase: the reference's quadratic (src/geom.zig:40-58) in f64 on the f64 sphere, for the ray as the kernel
// holds it; the chosen root is rounded to R before the comparisons.  Returns the f64 discriminant.
template <class R>
static double narrowRoots(const double* c64, const double* v64, double r2_64, V<R> o, V<R> d, R time, R tmin, int pool, R& tbest,
                          int& ibest) {
    const double dx = d.x, dy = d.y, dz = d.z, tm = time;
    const double a2 = std::fma(dz, dz, std::fma(dy, dy, dx * dx));
    const double inv_a2 = 1.0 / a2;
    double qx = c64[0] - (double)o.x, qy = c64[1] - (double)o.y, qz = c64[2] - (double)o.z;
    qx = std::fma(v64[0], tm, qx);
    qy = std::fma(v64[1], tm, qy);
    qz = std::fma(v64[2], tm, qz);
    const double hb2 = std::fma(dz, qz, std::fma(dy, qy, dx * qx));
    const double cc2 = std::fma(qz, qz, std::fma(qy, qy, std::fma(qx, qx, -r2_64)));
    const double disc2 = std::fma(-a2, cc2, hb2 * hb2);
    if (!(disc2 >= 0.0)) return disc2;
    const double rt = std::sqrt(disc2);
    const R t1 = (R)((hb2 - rt) * inv_a2), t2 = (R)((hb2 + rt) * inv_a2);
    const R t = t1 >= tmin ? t1 : t2;
    if (t >= tmin && (t < tbest || (t == tbest && pool > ibest))) {
        tbest = t;
        ibest = pool;
    }
    return disc2;
}
// hit record, src/geom.zig:63-65 + src/hit.zig:25-41
template <class R> static void sphereHitRecord(V<R> c, V<R> v, V<R> o, V<R> d, R time, R t, V<R>& pt, V<R>& nrm) {
    pt = {fm(d.x, t, o.x), fm(d.y, t, o.y), fm(d.z, t, o.z)};
    const V<R> cn{fm(v.x, time, c.x), fm(v.y, time, c.y), fm(v.z, time, c.z)};
    nrm = unit(V<R>{pt.x - cn.x, pt.y - cn.y, pt.z - cn.z});
}
template <class R> static bool faceForward(V<R> d, V<R>& nrm) {              // Hit.init, src/hit.zig:33-36
    const bool front = dot3(nrm, d) < R(0);
    if (!front) nrm = neg(nrm);
    return front;
}
template <class R> static R reflectance(R cosv, R eta) {                     // src/material.zig:179-183, pow(x,5) → x²·x²·x
    R r0 = (R(1) - eta) / (R(1) + eta);
    r0 = r0 * r0;
    const R xx = R(1) - cosv;
    const R x2 = xx * xx;
    const R x5 = (x2 * x2) * xx;
    return fm(R(1) - r0, x5, r0);
}
template <class R> static V<R> reflect(V<R> d, V<R> nrm) {                   // :185-187, unnormalised d
    const R k = R(2) * dot3(d, nrm);
    return {fm(-k, nrm.x, d.x), fm(-k, nrm.y, d.y), fm(-k, nrm.z, d.z)};
}
template <class R> static V<R> refract(V<R> ud, V<R> nrm, R cosv, R eta) {   // :189-194
    const V<R> perp{fm(nrm.x, cosv, ud.x) * eta, fm(nrm.y, cosv, ud.y) * eta, fm(nrm.z, cosv, ud.z) * eta};
    // clamped at 0: see DESIGN.md §4.5 (f32 rounds 1 − |perp|² below 0 near the critical angle)
    const R sp = -std::sqrt(std::fmax(R(1) - dot3(perp, perp), R(0)));
    return {fm(nrm.x, sp, perp.x), fm(nrm.y, sp, perp.y), fm(nrm.z, sp, perp.z)};
}
// `Material.scatter` without the texture lookup: false = absorbed
template <class R, class G>
static bool scatterDir(u32 kind, u32 method, R param, R inv_param, G& g, V<R> d, V<R> ud, V<R> pt, V<R> nrm, bool front,
                       V<R>& nd) {
    if (kind == RAYZ_MAT_DIFFUSE) {                                          // src/material.zig:77-101
        V<R> target;
        V<R> r = randomInUnitSphere<R>(g);
        if (method == RAYZ_DIFFUSE_HEMISPHERE) {
            if (!(dot3(r, nrm) > R(0))) r = neg(r);                          // :208-211
            target = {pt.x + r.x, pt.y + r.y, pt.z + r.z};
        } else {
            if (method == RAYZ_DIFFUSE_UNIT_SPHERE_SURFACE) r = unit(r);
            target = {(pt.x + nrm.x) + r.x, (pt.y + nrm.y) + r.y, (pt.z + nrm.z) + r.z};
        }
        const R tol = (R)1e-8;
        if (std::fabs(target.x) <= tol && std::fabs(target.y) <= tol && std::fabs(target.z) <= tol)
            target = nrm;                                                    // :85-86 (tests the POINT; preserved)
        nd = {target.x - pt.x, target.y - pt.y, target.z - pt.z};
    } else if (kind == RAYZ_MAT_METALLIC) {                                  // :108-131
        V<R> r = unit(reflect<R>(d, nrm));
        if (param > R(0)) {
            const V<R> ru = unit(randomInUnitSphere<R>(g));
            const R f = param < R(1) ? param : R(1);
            r = {fm(ru.x, f, r.x), fm(ru.y, f, r.y), fm(ru.z, f, r.z)};
        }
        if (dot3(r, nrm) <= R(0)) return false;                              // absorbed → black
        nd = r;
    } else {                                                                 // :137-159
        const R eta = front ? inv_param : param;
        const R cosv = -dot3(ud, nrm);
        const R sinv = std::sqrt(fm(-cosv, cosv, R(1)));
        bool refl = eta * sinv > R(1);
        if (!refl) refl = reflectance<R>(cosv, eta) > g.uniform();           // the draw only when not TIR, :145
        nd = refl ? reflect<R>(d, nrm) : refract<R>(ud, nrm, cosv, eta);
    }
    return true;
}
// build-defined triangle: sign-free barycentric test in R, then t = (e2·q) / det
template <class R> static R triFilter(V<R> v0, V<R> e1, V<R> e2, V<R> o, V<R> d) {
    const V<R> pv = cross3(d, e2);
    const R det = dot3(e1, pv);
    const V<R> sv{o.x - v0.x, o.y - v0.y, o.z - v0.z};
    const R su = dot3(sv, pv) * det;
    const V<R> qv = cross3(sv, e1);
    const R svv = dot3(d, qv) * det;
    const R w = fm(det, det, -(su + svv));
    return std::fmin(std::fmin(su, svv), w);
}
template <class R> static void triAccept(R filt, V<R> v0, V<R> e1, V<R> e2, V<R> o, V<R> d, R tmin, int prim, R& tbest, int& ibest) {
    if (!(filt >= R(0))) return;
    const V<R> pv = cross3(d, e2);
    const R det = dot3(e1, pv);
    if (det == R(0)) return;
    const V<R> sv{o.x - v0.x, o.y - v0.y, o.z - v0.z};
    const V<R> qv = cross3(sv, e1);
    const R t = dot3(e2, qv) / det;
    if (t >= tmin && (t < tbest || (t == tbest && prim > ibest))) {
        tbest = t;
        ibest = prim;
    }
}
// slab test, src/hit.zig:70-98, with 1/d and −o/d hoisted (one fma per plane), IN F32 FOR BOTH PRECISIONS, bare (t1 ≥ t0):
// it never culls a box the f64 narrow phase would hit because the BOXES are padded — by E = 16·u·max(S, B) per side
// (boxPad; u = 2^-24, S ≥ |o| of every ray, B = largest box coordinate), which covers the rounding of 1/d, of −o·inv and of
// the fma.  Reciprocals are held to ±2^64: a zero direction component must not reach the test as ±inf —
// one plane of a box that straddles 0 then gives −inf, the other NaN, and max(−inf, NaN) = −inf culls a box the ray lies
// inside.
template <class R> struct SlabRay {
    V<float> inv, noi;
};
template <class R> static SlabRay<R> slabRay(V<R> o, V<R> d) {
    SlabRay<R> s;
    auto clampf = [](float v, float lim) { return v > lim ? lim : (v < -lim ? -lim : v); };
    auto inv = [&](R dk) {
        float f = (float)dk;
        if (sizeof(R) == 8) f = clampf(f, 0x1p100f); // beyond f32's range the reciprocal would be 0
        return clampf(1.0f / f, 0x1p64f);
    };
    s.inv = {inv(d.x), inv(d.y), inv(d.z)};
    s.noi = {(float)(-(o.x * (R)s.inv.x)), (float)(-(o.y * (R)s.inv.y)), (float)(-(o.z * (R)s.inv.z))};
    return s;
}
// tmin rounded DOWN to f32, tbest rounded UP; lo / hi: the PADDED box, rounded outward to f32
template <class R> static bool boxHit(const float* lo, const float* hi, const SlabRay<R>& s, R tmin, R tbest, float& t0) {
    const float tmin32 = roundDown<float>((double)tmin), tb32 = roundUp<float>((double)tbest);
    const float ax = fm(lo[0], s.inv.x, s.noi.x), bx = fm(hi[0], s.inv.x, s.noi.x);
    const float ay = fm(lo[1], s.inv.y, s.noi.y), by = fm(hi[1], s.inv.y, s.noi.y);
    const float az = fm(lo[2], s.inv.z, s.noi.z), bz = fm(hi[2], s.inv.z, s.noi.z);
    t0 = std::fmax(std::fmax(std::fmin(ax, bx), std::fmin(ay, by)), std::fmax(std::fmin(az, bz), tmin32));
    const float t1 = std::fmin(std::fmin(std::fmax(ax, bx), std::fmax(ay, by)), std::fmin(std::fmax(az, bz), tb32));
    return t1 >= t0;
}

// RAYZ_TRAVERSAL_AUTO (include/rayz_hip.h): flat list up to RAYZ_AUTO_BVH_MIN hittables, BVH above
static inline bool useBvh(const RayzRenderParams& p, u32 n_hittables) {
    return p.traversal == RAYZ_TRAVERSAL_BVH || (p.traversal == RAYZ_TRAVERSAL_AUTO && n_hittables > RAYZ_AUTO_BVH_MIN);
}

template <class R> struct PathResult {
    V<R> L;
    u32 segments;
    u64 node_tests, sphere_tests;
};
// What the filter audit counts over a flat-list render (tests/test_filter_conservative.py): for every (segment,
// sphere) pair the f64 discriminant the narrow phase would compute, against the R filter's verdict.
struct FilterAudit {
    u64 pairs = 0, candidates = 0, f64_hits = 0, false_negatives = 0, unpadded_false_negatives = 0;
};

template <class R>
static PathResult<R> tracePath(const SceneB<R>& sc, const CamB<R>& cam, const RayzRenderParams& p, u32 px, u32 py,
                               u32 s, FilterAudit* audit = nullptr) {
    const R tmin = (R)p.tmin;
    Rng<R> g;
    const u64 pixel_index = (u64)py * p.width + px;
    g.g.seed_path(p.seed, pixel_index * p.samples_per_px + s);

    V<R> o, d;
    R time;
    cameraRay<R>(cam, g, px, py, o, d, time);

    V<R> thr{1, 1, 1};
    PathResult<R> res{{0, 0, 0}, 0, 0, 0};
    const R inf = std::numeric_limits<R>::infinity();

    for (u32 seg = 0; seg < p.max_bounces; ++seg) {                          // src/renderer.zig:103-126, iterative
        res.segments++;
        // --- nearest hit, src/geom.zig:38-66 per sphere ---
        // Reject test in R with the UNIT direction; candidates go to the narrow phase in f64.  Ties in t go to the
        // larger pool index — what the reference's "t ≤ maxt, later wins" gives over its flat hittable list
        // (src/hit.zig:208-214) — so the result does not depend on the order in which the spheres are examined.
        const V<R> ud = unit(d);
        R tbest = inf;
        int ibest = -1;
        // the reject test: in f32 for both precisions, on the ray narrowed to f32 (flat-list scan and BVH leaves alike)
        const Basis<float> basisf = makeBasis<float>(V<float>{(float)ud.x, (float)ud.y, (float)ud.z},
                                                     V<float>{(float)o.x, (float)o.y, (float)o.z});
        const float timef = (float)time;
        auto testSphere = [&](const Sph<R>& q) {
            const bool cand = sphereFilter<float>(basisf, timef, q.cf, q.vf, q.r2f) >= 0.0f;
            if (audit) {
                R tb = inf;
                int ib = -1;
                const double disc2 = narrowRoots<R>(q.c64, q.v64, q.r2_64, o, d, time, tmin, (int)q.pool, tb, ib);
                const float plain = (float)q.radius * (float)q.radius; // what the filter used before it was made conservative
                audit->pairs++;
                audit->candidates += cand;
                audit->f64_hits += disc2 >= 0.0;
                audit->false_negatives += disc2 >= 0.0 && !cand;
                audit->unpadded_false_negatives += disc2 >= 0.0 && !(sphereFilter<float>(basisf, timef, q.cf, q.vf, plain) >= 0.0f);
            }
            if (!cand) return;
            narrowRoots<R>(q.c64, q.v64, q.r2_64, o, d, time, tmin, (int)q.pool, tbest, ibest);
        };
        const u32 n_sph = (u32)sc.sph.size();
        auto testTriangle = [&](const Tri<R>& q, u32 prim) {
            triAccept<R>(triFilter<R>(q.v0, q.e1, q.e2, o, d), q.v0, q.e1, q.e2, o, d, tmin, (int)prim, tbest, ibest);
        };
        if (useBvh(p, (u32)(sc.sph.size() + sc.tri.size()))) {
            // src/hit.zig:181-216 as a skip-link walk
            const SlabRay<R> slab = slabRay<R>(o, d);
            const u32 nn = (u32)sc.nodes.size();
            u32 idx = 0;
            while (idx < nn) {
                const typename SceneB<R>::Node& nd = sc.nodes[idx];
                res.node_tests++;
                float t0;
                u32 next = nd.skip;
                if (boxHit<R>(nd.lo, nd.hi, slab, tmin, tbest, t0)) {
                    if (nd.count == 0) next = idx + 1;
                    for (u32 k = 0; k < nd.count; ++k) {
                        res.sphere_tests++;
                        const u32 prim = sc.leaf_order[nd.first + k];
                        if (prim < n_sph) testSphere(sc.sph[sc.by_pool[prim]]);
                        else testTriangle(sc.tri[prim - n_sph], prim);
                    }
                }
                idx = next;
            }
        } else {
            res.sphere_tests += n_sph + (u32)sc.tri.size();
            for (u32 i = 0; i < n_sph; ++i) testSphere(sc.sph[i]);
            for (u32 i = 0; i < (u32)sc.tri.size(); ++i) testTriangle(sc.tri[i], n_sph + i);
        }
        if (ibest < 0) {                                                     // miss, src/renderer.zig:124-125
            const V<R> col = background<R>(ud);
            res.L = {thr.x * col.x, thr.y * col.y, thr.z * col.z};
            return res;
        }
        V<R> pt, nrm;
        u32 mat_idx;
        if ((u32)ibest < n_sph) {
            const Sph<R>& q = sc.sph[sc.by_pool[ibest]];
            sphereHitRecord<R>(q.c, q.v, o, d, time, tbest, pt, nrm);
            mat_idx = q.mat;
        } else {
            const Tri<R>& q = sc.tri[(u32)ibest - n_sph];
            pt = {fm(d.x, tbest, o.x), fm(d.y, tbest, o.y), fm(d.z, tbest, o.z)};
            nrm = unit(cross3(q.e1, q.e2));
            mat_idx = q.mat;
        }
        const bool front = faceForward<R>(d, nrm);

        const Mat<R>& m = sc.mats[mat_idx];
        V<R> nd;
        if (!scatterDir<R>(m.kind, m.method, m.param, m.inv_param, g, d, ud, pt, nrm, front, nd)) return res; // absorbed
        const V<R> att = m.kind == RAYZ_MAT_DIELECTRIC ? V<R>{1, 1, 1} : textureValue(sc, m.texture, pt);
        thr = {thr.x * att.x, thr.y * att.y, thr.z * att.z};
        o = pt;
        d = nd;
    }
    return res; // depth exhausted → black, src/renderer.zig:104-105
}

// Chunk schedule, restated: which consecutive samples of a pixel are summed by one work item.
// Largest chunk of the automatic schedule: pixels · spp / 2^24 held to [64, 256], a power of two, at most
// spp / 2 — no work item larger than 1/8 of a lane's share when the frame is dealt to 8 GPUs of 2^18 lanes.
static u32 autoChunk(u64 pixels, u32 spp) {
    u64 share = pixels >= (1ull << 32) ? 256 : (pixels * spp) >> 24;
    if (share < 64) share = 64;
    if (share > 256) share = 256;
    u32 c = 1;
    while (2 * (u64)c <= share) c *= 2;
    u32 half = 1;
    while (2 * (u64)half <= spp / 2) half *= 2;
    return c < half ? c : half;
}
static std::vector<u32> chunkSchedule(const RayzRenderParams& p) {
    std::vector<u32> st{0};
    const u32 spp = p.samples_per_px;
    if (p.chunk_spp != 0 || (u64)p.width * p.height < (1ull << 19) || spp < 64) {
        const u32 c = p.chunk_spp ? p.chunk_spp : 16;
        for (u64 s0 = c; s0 < spp; s0 += c) st.push_back((u32)s0);
        st.push_back(spp);
        return st;
    }
    auto pow2floor = [](u32 v) {
        u32 r = 1;
        while (2 * (u64)r <= v) r *= 2;
        return r;
    };
    u32 C = autoChunk((u64)p.width * p.height, spp);
    u32 at = 0, rem = spp;
    while (rem >= 2 * C) {
        at += C, rem -= C;
        st.push_back(at);
    }
    while (rem > 16) {
        u32 c = pow2floor(rem / 2);
        if (c < 16) c = 16;
        at += c, rem -= c;
        st.push_back(at);
    }
    if (rem) st.push_back(at + rem);
    return st;
}

static inline u32 shardRows(const RayzRenderParams& p, std::vector<u32>* rows) {
    const u32 tr = p.tile_rows ? p.tile_rows : 8;
    const u32 sc = p.shard_count ? p.shard_count : 1;
    u32 n = 0;
    for (u32 r = 0; r < p.height; ++r)
        if ((r / tr) % sc == p.shard_index) {
            if (rows) rows->push_back(r);
            ++n;
        }
    return n;
}

template <class R>
static int render(const RayzSceneDesc* sd, const RayzCameraDesc* cd, const RayzRenderParams* pp,
                  const u32* pixel_list, u32 n_list, R* out, RayzRenderStats* stats, int threads) {
    if (!sd || !cd || !pp || !out) return RAYZ_ERR_BAD_ARG;
    const RayzRenderParams p = *pp;
    if (!p.width || !p.height || !p.samples_per_px) return RAYZ_ERR_BAD_ARG;
    {   // the library refuses schedules of 2^20 chunks per pixel or more, whatever chunk_spp is (0 = automatic included)
        const u64 spp = p.samples_per_px;
        const bool uniform = p.chunk_spp != 0 || (u64)p.width * p.height < (1ull << 19) || spp < 64;
        const u64 c = p.chunk_spp ? p.chunk_spp : 16;
        u64 n = (spp + c - 1) / c;
        if (!uniform) { // the automatic schedule's length, exactly: full chunks of C while 2C remain, then the halving tail
            const u64 C = autoChunk((u64)p.width * p.height, (u32)spp);
            n = 0;
            u64 rem = spp;
            if (rem >= 2 * C) n = (rem - 2 * C) / C + 1, rem -= n * C;
            while (rem > 16) {
                u64 h = 1;
                while (h <= rem / 4) h *= 2;
                rem -= h < 16 ? 16 : h, ++n;
            }
            n += rem ? 1 : 0;
        }
        if (n >= (1ull << 20)) return RAYZ_ERR_BAD_ARG;
    }
    SceneB<R> sc = buildScene<R>(*sd, originBound(*sd, cd));
    if (useBvh(*pp, sd->n_spheres + sd->n_triangles)) buildBvh<R>(*sd, sc, originBound(*sd, cd));
    const CamB<R> cam = buildCamera<R>(*cd);
    const std::vector<u32> chunks = chunkSchedule(p);
    std::vector<u32> pixels; // global pixel indices, in output order
    if (pixel_list) pixels.assign(pixel_list, pixel_list + n_list);
    else {
        std::vector<u32> rows;
        shardRows(p, &rows);
        for (u32 r : rows)
            for (u32 i = 0; i < p.width; ++i) pixels.push_back(r * p.width + i);
    }
    u64 segs = 0, ntests = 0, stests = 0;
    const long np = (long)pixels.size();
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : segs, ntests, stests)
#endif
    for (long k = 0; k < np; ++k) {
        const u32 px = pixels[k] % p.width, py = pixels[k] / p.width;
        V<R> pix{0, 0, 0};
        for (size_t c = 0; c + 1 < chunks.size(); ++c) {
            const u32 s0 = chunks[c], s1 = chunks[c + 1];
            V<R> acc{0, 0, 0};
            for (u32 s = s0; s < s1; ++s) {
                const PathResult<R> r = tracePath<R>(sc, cam, p, px, py, s);
                acc = {acc.x + r.L.x, acc.y + r.L.y, acc.z + r.L.z};
                segs += r.segments;
                ntests += r.node_tests;
                stests += r.sphere_tests;
            }
            pix = {pix.x + acc.x, pix.y + acc.y, pix.z + acc.z};
        }
        const R inv = R(1) / (R)p.samples_per_px;                            // acc.div(spp), src/renderer.zig:94-95
        out[3 * k + 0] = pix.x * inv;
        out[3 * k + 1] = pix.y * inv;
        out[3 * k + 2] = pix.z * inv;
    }
    (void)threads;
    if (stats) {
        stats->primary_rays = (u64)np * p.samples_per_px;
        stats->segments = segs;
        stats->sphere_tests = stests;
        stats->node_tests = ntests;
        stats->kernel_ms = 0;
    }
    return RAYZ_OK;
}

} // namespace B

static A::Scene sceneA(const RayzSceneDesc& d) {
    A::Scene sc;
    for (u32 i = 0; i < d.n_spheres; ++i) {
        A::Sphere s;
        s.center.origin = A::v3(d.spheres[i].center);
        s.center.dir = A::v3(d.spheres[i].velocity);
        s.radius = d.spheres[i].radius;
        s.material = d.spheres[i].material;
        sc.spheres.push_back(s);
    }
    sc.materials.assign(d.materials, d.materials + d.n_materials);
    sc.textures.assign(d.textures, d.textures + d.n_textures);
    for (u32 i = 0; i < d.n_triangles; ++i) {
        A::Triangle t;
        t.v0 = A::v3(d.triangles[i].v0), t.v1 = A::v3(d.triangles[i].v1), t.v2 = A::v3(d.triangles[i].v2);
        t.material = d.triangles[i].material;
        sc.triangles.push_back(t);
    }
    for (u32 i = 0; i < d.n_spheres; ++i)                                    // src/ecs.zig:43-51
        sc.hittables.push_back({sc.spheres[i].boundingBox(), i});
    for (u32 i = 0; i < d.n_triangles; ++i) sc.hittables.push_back({sc.triangles[i].boundingBox(), d.n_spheres + i});
    return sc;
}

} // namespace

// ------------------------------------------------------------------------------------------------
// C entry points (loaded with ctypes by tests/ and bench.py only)
// ------------------------------------------------------------------------------------------------
extern "C" {

// ---- mode B ----
int rayz_oracle_render_b_f32(const RayzSceneDesc* s, const RayzCameraDesc* c, const RayzRenderParams* p,
                             const uint32_t* pixel_list, uint32_t n_list, float* out, RayzRenderStats* st,
                             int threads) {
    return B::render<float>(s, c, p, pixel_list, n_list, out, st, threads);
}
int rayz_oracle_render_b_f64(const RayzSceneDesc* s, const RayzCameraDesc* c, const RayzRenderParams* p,
                             const uint32_t* pixel_list, uint32_t n_list, double* out, RayzRenderStats* st,
                             int threads) {
    return B::render<double>(s, c, p, pixel_list, n_list, out, st, threads);
}
uint32_t rayz_oracle_shard_rows(const RayzRenderParams* p) { return B::shardRows(*p, nullptr); }
uint32_t rayz_oracle_chunk_schedule(const RayzRenderParams* p, uint32_t* starts, uint32_t capacity) {
    const std::vector<u32> v = B::chunkSchedule(*p);
    for (size_t i = 0; starts && i < v.size() && i < capacity; ++i) starts[i] = v[i];
    return (uint32_t)v.size() - 1;
}

// ---- known answers: the pieces of mode B (kernel arithmetic) and of mode A (the reference as written) on the record
// formats of rayz_hip_kat (include/rayz_hip.h: RayzKatOp, RAYZ_KAT_IN_STRIDE / RAYZ_KAT_OUT_STRIDE) ---------------------
} // extern "C"
namespace {
template <class R> static void katB(uint32_t op, const double* a, double* r) {
    using namespace B;
    auto v3 = [&](int k) { return V<R>{(R)a[k], (R)a[k + 1], (R)a[k + 2]}; };
    auto put3 = [&](int k, V<R> v) { r[k] = (double)v.x, r[k + 1] = (double)v.y, r[k + 2] = (double)v.z; };
    switch (op) {
    case RAYZ_KAT_REFRACT: {
        const V<R> ud = v3(0), nrm = v3(3);
        put3(0, refract<R>(ud, nrm, -dot3(ud, nrm), (R)a[6]));
        break;
    }
    case RAYZ_KAT_REFLECTANCE: r[0] = (double)reflectance<R>((R)a[0], (R)a[1]); break;
    case RAYZ_KAT_GET_RAY: {
        CamB<R> cam;
        cam.from = v3(0), cam.du = v3(3), cam.dv = v3(6), cam.pxo = v3(9), cam.defu = v3(12), cam.defv = v3(15);
        cam.defocus = a[18] != 0.0;
        ListRng<R> g{a + 22, a[21] < 0.0 ? 0u : (u32)a[21], 0u};
        V<R> o, d;
        R time;
        if (a[21] < 0.0) { // n_u = -1: getRay(px, py, null) in mode B's operation order (no draw, lens centre, time 0)
            const R x = (R)(u32)a[19], y = (R)(u32)a[20];
            o = cam.from;
            d = {(fm(cam.dv.x, y, cam.du.x * x) + cam.pxo.x) - o.x, (fm(cam.dv.y, y, cam.du.y * x) + cam.pxo.y) - o.y,
                 (fm(cam.dv.z, y, cam.du.z * x) + cam.pxo.z) - o.z};
            time = R(0);
        } else
            cameraRay<R>(cam, g, (u32)a[19], (u32)a[20], o, d, time);
        put3(0, o);
        put3(3, d);
        r[6] = (double)time;
        r[7] = (double)g.i;
        break;
    }
    case RAYZ_KAT_BOX_HIT: {
        // the box test of the DEVICE walk, restated: the box as 16-bit plane indices on a grid over it (lower planes the
        // largest index at or below plane − E, upper ones the smallest at or above plane + E, E = 16u·(max(S, B) + X), S = this
        // ray's origin, B = this box, X = the grid's extent), each plane distance fm(float(index), cell·inv, fm(glo, inv, −o·inv)),
        // hit iff t1 ≥ t0.  (Mode B's own walk tests padded f32 boxes — boxHit above; the image depends on neither.)
        // a[26] != 0: the other record format — the box padded by E = 16u·max(S, B) and rounded outward to f32 (as mode B's
        // own walk holds its boxes)
        const SlabRay<R> slab = slabRay<R>(v3(6), v3(9));
        double Bk = 0;
        for (int k = 0; k < 6; ++k) Bk = std::max(Bk, std::fabs(a[k]));
        if (a[26] != 0.0) {
            const double pad = boxPad(norm3(a + 6), Bk);
            const float lo[3] = {roundDown<float>(a[0] - pad), roundDown<float>(a[1] - pad), roundDown<float>(a[2] - pad)};
            const float hi[3] = {roundUp<float>(a[3] + pad), roundUp<float>(a[4] + pad), roundUp<float>(a[5] + pad)};
            float t0;
            r[0] = boxHit<R>(lo, hi, slab, (R)a[12], (R)a[13], t0) ? 1.0 : 0.0;
            r[1] = (double)t0;
            break;
        }
        const double u = std::numeric_limits<float>::epsilon() / 2.0;
        double pad = kBoxPadUlps * u * (std::max(norm3(a + 6), Bk) + 2.0 * Bk);
        float glo[3], cell[3];
        double extent = 0;
        for (int k = 0; k < 3; ++k) { // the grid over [lo − 2 pad, hi + 2 pad]: 65,535 cells, the cell size rounded up until the last plane reaches
            const double ga = a[k] - 2.0 * pad, gb = a[3 + k] + 2.0 * pad;
            float f = roundDown<float>(ga);
            float c = (float)((gb - (double)f) / 65535.0);
            if (!(c > 0.0f)) c = std::numeric_limits<float>::min();
            while ((double)f + 65535.0 * (double)c < gb) c = std::nextafter(c, std::numeric_limits<float>::infinity());
            glo[k] = f, cell[k] = c;
            extent = std::max(extent, 65535.0 * (double)c);
        }
        pad = kBoxPadUlps * u * (std::max(norm3(a + 6), Bk) + extent);
        auto plane = [&](int k, double i) { return (double)glo[k] + i * (double)cell[k]; };
        float qlo[3], qhi[3];
        for (int k = 0; k < 3; ++k) {
            double i = std::min(65535.0, std::max(0.0, std::floor((a[k] - pad - (double)glo[k]) / (double)cell[k])));
            while (i > 0 && plane(k, i) > a[k] - pad) i -= 1;
            while (i < 65535 && plane(k, i + 1) <= a[k] - pad) i += 1;
            qlo[k] = (float)i;
            double j = std::min(65535.0, std::max(0.0, std::ceil((a[3 + k] + pad - (double)glo[k]) / (double)cell[k])));
            while (j < 65535 && plane(k, j) < a[3 + k] + pad) j += 1;
            while (j > 0 && plane(k, j - 1) >= a[3 + k] + pad) j -= 1;
            qhi[k] = (float)j;
        }
        const float inv[3] = {slab.inv.x, slab.inv.y, slab.inv.z}, noi[3] = {slab.noi.x, slab.noi.y, slab.noi.z};
        float tn[3], tf[3];
        for (int k = 0; k < 3; ++k) {
            const float qa = cell[k] * inv[k], qb = fm(glo[k], inv[k], noi[k]);
            const float ta = fm(qlo[k], qa, qb), tb = fm(qhi[k], qa, qb);
            tn[k] = std::fmin(ta, tb), tf[k] = std::fmax(ta, tb);
        }
        const float tmin32 = roundDown<float>((double)(R)a[12]), tb32 = roundUp<float>((double)(R)a[13]);
        const float t0 = std::fmax(std::fmax(tn[0], tn[1]), std::fmax(tn[2], tmin32));
        const float t1 = std::fmin(std::fmin(tf[0], tf[1]), std::fmin(tf[2], tb32));
        r[0] = t1 >= t0 ? 1.0 : 0.0;
        r[1] = (double)t0;
        break;
    }
    case RAYZ_KAT_SPHERE_HIT: {
        RayzSphere q{};
        for (int k = 0; k < 3; ++k) q.center[k] = a[k], q.velocity[k] = a[3 + k];
        q.radius = a[6];
        const V<R> o = v3(7), d = v3(10);
        const R time = (R)a[13], tmin = (R)a[14];
        const double S = std::max(norm3(a + 7), norm3(q.center) + norm3(q.velocity) + std::fabs(q.radius));
        const V<R> c = v3(0), v = v3(3);
        const V<R> udk = unit(d); // the reject test as the kernels run it: f32 for both precisions, the ray narrowed to f32
        const Basis<float> b = makeBasis<float>(V<float>{(float)udk.x, (float)udk.y, (float)udk.z}, V<float>{(float)o.x, (float)o.y, (float)o.z});
        const bool cand = sphereFilter<float>(b, (float)time, V<float>{(float)c.x, (float)c.y, (float)c.z},
                                              V<float>{(float)v.x, (float)v.y, (float)v.z}, padRadius2Scan<R>(q, S)) >= 0.0f;
        r[9] = cand ? 1.0 : 0.0;
        R tbest = (R)a[15];
        int ibest = -1;
        if (cand) narrowRoots<R>(q.center, q.velocity, q.radius * q.radius, o, d, time, tmin, 1, tbest, ibest);
        if (ibest >= 0) {
            V<R> pt, nrm;
            sphereHitRecord<R>(c, v, o, d, time, tbest, pt, nrm);
            const bool front = faceForward<R>(d, nrm);
            r[0] = 1.0, r[1] = (double)tbest;
            put3(2, pt);
            put3(5, nrm);
            r[8] = front ? 1.0 : 0.0;
        }
        break;
    }
    case RAYZ_KAT_SCATTER: {
        const R param = (R)a[2];
        const V<R> d = v3(6);
        ListRng<R> g{a + 17, (u32)a[16], 0u};
        V<R> nd{0, 0, 0};
        const bool ok = scatterDir<R>((u32)a[0], (u32)a[1], param, R(1) / param, g, d, unit(d), v3(9), v3(12), a[15] != 0.0, nd);
        r[0] = ok ? 1.0 : 0.0;
        put3(1, nd);
        r[4] = (double)g.i;
        break;
    }
    case RAYZ_KAT_CHECKER: r[0] = (double)checkerParity<R>(v3(0), (R)a[3]); break;
    case RAYZ_KAT_SCAN_DISCS: { // the flat list's reject test on a block of 4 spheres: sphereFilter, the ONE form mode B has
        const V<R> o = v3(20), d = v3(23);
        const V<R> udk = unit(d);
        const Basis<float> b = makeBasis<float>(V<float>{(float)udk.x, (float)udk.y, (float)udk.z}, V<float>{(float)o.x, (float)o.y, (float)o.z});
        const float ft = (float)(R)a[26];
        double S = norm3(a + 20);
        RayzSphere q[4] = {};
        for (int k = 0; k < 4; ++k) {
            q[k].center[0] = a[k], q[k].center[1] = a[4 + k], q[k].center[2] = a[8 + k];
            q[k].radius = a[12 + k];
            q[k].velocity[1] = a[27] != 0.0 ? a[16 + k] : 0.0;
            S = std::max(S, norm3(q[k].center) + norm3(q[k].velocity) + std::fabs(q[k].radius));
        }
        for (int k = 0; k < 4; ++k) {
            const V<float> c{(float)a[k], (float)a[4 + k], (float)a[8 + k]}, v{0.0f, (float)q[k].velocity[1], 0.0f};
            r[k] = r[4 + k] = (double)sphereFilter<float>(b, ft, c, v, padRadius2Scan<R>(q[k], S));
        }
        break;
    }
    case RAYZ_KAT_BACKGROUND: put3(0, background<R>(unit(v3(0)))); break;
    case RAYZ_KAT_TRIANGLE_HIT: {
        const V<R> v0 = v3(0), o = v3(9), d = v3(12);
        const V<R> e1{(R)(a[3] - a[0]), (R)(a[4] - a[1]), (R)(a[5] - a[2])}, e2{(R)(a[6] - a[0]), (R)(a[7] - a[1]), (R)(a[8] - a[2])};
        const R f = triFilter<R>(v0, e1, e2, o, d);
        R tbest = (R)a[16];
        int ibest = -1;
        triAccept<R>(f, v0, e1, e2, o, d, (R)a[15], 1, tbest, ibest);
        r[0] = ibest >= 0 ? 1.0 : 0.0;
        r[1] = ibest >= 0 ? (double)tbest : 0.0;
        r[2] = f >= R(0) ? 1.0 : 0.0;
        break;
    }
    default: break;
    }
}

// The same records through the reference's own functions (mode A, f64, literal operation order).
static void katA(uint32_t op, const double* a, double* r) {
    using namespace A;
    auto put3 = [&](int k, V3 v) { r[k] = v.x, r[k + 1] = v.y, r[k + 2] = v.z; };
    switch (op) {
    case RAYZ_KAT_REFRACT: put3(0, refract(v3(a), v3(a + 3), a[6])); break;
    case RAYZ_KAT_REFLECTANCE: r[0] = reflectance(a[0], a[1]); break;
    case RAYZ_KAT_GET_RAY: {
        Camera c;
        c.look_from = v3(a), c.px_du = v3(a + 3), c.px_dv = v3(a + 6), c.px_origin = v3(a + 9);
        c.defocus_u = v3(a + 12), c.defocus_v = v3(a + 15);
        c.defocus = a[18] != 0.0;
        A::ListRng g{a + 22, a[21] < 0.0 ? 0u : (u32)a[21], 0u};
        const Ray ray = a[21] < 0.0 ? c.getRay<A::ListRng>((size_t)a[19], (size_t)a[20], nullptr) // n_u = -1: the reference's rng == null
                                    : c.getRay((size_t)a[19], (size_t)a[20], &g);
        put3(0, ray.origin);
        put3(3, ray.dir);
        r[6] = ray.time;
        r[7] = (double)g.i;
        break;
    }
    case RAYZ_KAT_BOX_HIT: {
        Ray ray;
        ray.origin = v3(a + 6), ray.dir = v3(a + 9);
        r[0] = AABB{v3(a), v3(a + 3)}.hit(ray, a[12], a[13]) ? 1.0 : 0.0;
        break;
    }
    case RAYZ_KAT_SPHERE_HIT: {
        Sphere q;
        q.center.origin = v3(a), q.center.dir = v3(a + 3);
        q.radius = a[6];
        q.material = 0;
        Ray ray;
        ray.origin = v3(a + 7), ray.dir = v3(a + 10), ray.time = a[13];
        const Hit h = q.hitInner(ray, a[14], a[15]);
        if (h.valid) {
            r[0] = 1.0, r[1] = h.t;
            put3(2, h.point);
            put3(5, h.normal);
            r[8] = h.front_face ? 1.0 : 0.0;
        }
        r[9] = r[0];
        break;
    }
    case RAYZ_KAT_SCATTER: {
        Scene sc;
        RayzTexture white{};
        white.kind = RAYZ_TEX_SOLID;
        white.color[0] = white.color[1] = white.color[2] = 1.0;
        sc.textures.push_back(white);
        RayzMaterial m{};
        m.kind = (u32)a[0], m.method = (u32)a[1], m.param = a[2], m.texture = 0;
        Ray ray;
        ray.origin = v3(a + 3), ray.dir = v3(a + 6);
        Hit h;
        h.point = v3(a + 9), h.normal = v3(a + 12), h.front_face = a[15] != 0.0, h.valid = true;
        A::ListRng g{a + 17, (u32)a[16], 0u};
        const Scatter sct = scatter(sc, m, g, ray, h);
        r[0] = sct.ok ? 1.0 : 0.0;
        if (sct.ok) put3(1, sct.ray.dir);
        r[4] = (double)g.i;
        break;
    }
    case RAYZ_KAT_CHECKER: r[0] = (double)checkerParity(v3(a), a[3]); break;
    case RAYZ_KAT_SCAN_DISCS: { // the reference's own discriminant (src/geom.zig:40-50) for the four spheres: sign only
        const V3 o = v3(a + 20), d = v3(a + 23);
        for (int k = 0; k < 4; ++k) {
            const V3 c{a[k], a[4 + k] + (a[27] != 0.0 ? a[16 + k] * a[26] : 0.0), a[8 + k]};
            const V3 oc = c.sub(o);
            const double aa = d.dot(d), hb = d.dot(oc), cc = oc.dot(oc) - a[12 + k] * a[12 + k];
            r[k] = r[4 + k] = hb * hb - aa * cc;
        }
        break;
    }
    case RAYZ_KAT_BACKGROUND: put3(0, background(v3(a))); break;
    case RAYZ_KAT_TRIANGLE_HIT: {
        Triangle t{v3(a), v3(a + 3), v3(a + 6), 0};
        Ray ray;
        ray.origin = v3(a + 9), ray.dir = v3(a + 12);
        const Hit h = t.hitInner(ray, a[15], a[16]);
        r[0] = h.valid ? 1.0 : 0.0;
        r[1] = h.valid ? h.t : 0.0;
        r[2] = r[0];
        break;
    }
    default: break;
    }
}
} // namespace
extern "C" {

int rayz_oracle_kat_b(uint32_t op, uint32_t precision, const double* in, uint32_t n, double* out) {
    if (op > RAYZ_KAT_SCAN_DISCS || precision > RAYZ_PRECISION_F64 || (n && (!in || !out))) return RAYZ_ERR_BAD_ARG;
    for (uint32_t i = 0; i < n; ++i) {
        double* r = out + (size_t)i * RAYZ_KAT_OUT_STRIDE;
        std::fill(r, r + RAYZ_KAT_OUT_STRIDE, 0.0);
        if (precision == RAYZ_PRECISION_F32) katB<float>(op, in + (size_t)i * RAYZ_KAT_IN_STRIDE, r);
        else katB<double>(op, in + (size_t)i * RAYZ_KAT_IN_STRIDE, r);
    }
    return RAYZ_OK;
}
int rayz_oracle_kat_a(uint32_t op, const double* in, uint32_t n, double* out) {
    if (op > RAYZ_KAT_SCAN_DISCS || (n && (!in || !out))) return RAYZ_ERR_BAD_ARG;
    for (uint32_t i = 0; i < n; ++i) {
        double* r = out + (size_t)i * RAYZ_KAT_OUT_STRIDE;
        std::fill(r, r + RAYZ_KAT_OUT_STRIDE, 0.0);
        katA(op, in + (size_t)i * RAYZ_KAT_IN_STRIDE, r);
    }
    return RAYZ_OK;
}

// Flat-list replay of the listed pixels that checks the reject filter against the f64 discriminant for EVERY
// (segment, sphere) pair: out[0..4] = pairs, candidates, f64 hits, false negatives, false negatives of the unpadded
// filter (what the kernel used before the filter was made conservative).
int rayz_oracle_filter_audit(const RayzSceneDesc* sd, const RayzCameraDesc* cd, const RayzRenderParams* pp, uint32_t precision,
                             const uint32_t* pixel_list, uint32_t n_list, uint64_t* out) {
    if (!sd || !cd || !pp || !out || (n_list && !pixel_list)) return RAYZ_ERR_BAD_ARG;
    RayzRenderParams p = *pp;
    p.traversal = RAYZ_TRAVERSAL_LINEAR;
    B::FilterAudit au;
    auto run = [&](auto tag) {
        typedef decltype(tag) R;
        const B::SceneB<R> sc = B::buildScene<R>(*sd, B::originBound(*sd, cd));
        const B::CamB<R> cam = B::buildCamera<R>(*cd);
        for (uint32_t k = 0; k < n_list; ++k)
            for (uint32_t s = 0; s < p.samples_per_px; ++s)
                B::tracePath<R>(sc, cam, p, pixel_list[k] % p.width, pixel_list[k] / p.width, s, &au);
    };
    if (precision == RAYZ_PRECISION_F32) run(float{});
    else run(double{});
    out[0] = au.pairs, out[1] = au.candidates, out[2] = au.f64_hits, out[3] = au.false_negatives, out[4] = au.unpadded_false_negatives;
    return RAYZ_OK;
}

// ---- mode A: `Tracer.render` over rows [row_begin,row_end) with ONE sequential stream ----
// rng_state: 4 u64 in/out (the Tracer's DefaultPrng, continued from scene generation, src/rayz.zig:109).
// out: (row_end-row_begin)*width*3 doubles; sumsq (optional) receives per-pixel Σ L² per channel.
int rayz_oracle_render_a(const RayzSceneDesc* sd, const RayzCameraDesc* cd, const RayzRenderParams* pp,
                         uint32_t row_begin, uint32_t row_end, uint64_t* rng_state, int linear, double* out,
                         double* sumsq, RayzRenderStats* stats) {
    if (!sd || !cd || !pp || !out || !rng_state || row_end > pp->height || row_begin > row_end)
        return RAYZ_ERR_BAD_ARG;
    A::Tracer t;
    t.sc = sceneA(*sd);
    t.cam = A::cameraFrom(*cd);
    std::memcpy(t.rng.s, rng_state, 32);
    t.tmin = pp->tmin;
    t.linear = linear != 0;
    if (!t.linear && !t.sc.hittables.empty()) t.bvh.build(t.sc.hittables, 0, t.sc.hittables.size()); // src/renderer.zig:76-78
    u64 rays = 0;
    for (u32 j = row_begin; j < row_end; ++j) {                              // src/renderer.zig:80-97
        for (u32 i = 0; i < pp->width; ++i) {
            A::V3 acc, sq;
            for (u32 r = 0; r < pp->samples_per_px; ++r) {
                const A::Ray ray = t.cam.getRay(i, j, &t.rng);
                rays++;
                const A::V3 L = t.bounceRay(ray, pp->max_bounces);
                acc = acc.add(L);
                sq = sq.add(L.vmul(L));
            }
            const A::V3 px = acc.div((double)pp->samples_per_px);
            const size_t k = ((size_t)(j - row_begin) * pp->width + i) * 3;
            out[k] = px.x;
            out[k + 1] = px.y;
            out[k + 2] = px.z;
            if (sumsq) {
                sumsq[k] = sq.x;
                sumsq[k + 1] = sq.y;
                sumsq[k + 2] = sq.z;
            }
        }
    }
    std::memcpy(rng_state, t.rng.s, 32);
    if (stats) {
        stats->primary_rays = rays;
        stats->segments = t.cnt.segments;
        stats->sphere_tests = t.cnt.sphere_tests;
        stats->node_tests = t.cnt.node_tests;
        stats->kernel_ms = 0;
    }
    return RAYZ_OK;
}

// ---- the BVH `render()` builds (src/renderer.zig:76-78), flattened in pre-order with skip links ----
// Arrays may be NULL; returns the node count.  boxes: 6 doubles per node (low, high).
uint32_t rayz_oracle_bvh_flat(const RayzSceneDesc* sd, double* boxes, uint32_t* skip, uint32_t* first, uint32_t* count,
                              uint32_t* order) {
    if (!sd || sd->n_spheres == 0) return 0;
    A::Scene sc = sceneA(*sd);
    A::BVH t;
    t.build(sc.hittables, 0, sc.hittables.size());
    for (size_t i = 0; i < t.nodes.size(); ++i) {
        const A::BVH::Node& n = t.nodes[i];
        if (boxes)
            for (int k = 0; k < 3; ++k) boxes[6 * i + k] = n.bbox.low.at(k), boxes[6 * i + 3 + k] = n.bbox.high.at(k);
        if (skip) skip[i] = B::subtreeEnd(t, (int)i);
        if (first) first[i] = n.left < 0 ? (u32)n.starti : 0;
        if (count) count[i] = n.left < 0 ? (u32)(n.endi - n.starti) : 0;
    }
    if (order)
        for (size_t i = 0; i < sc.hittables.size(); ++i) order[i] = sc.hittables[i].sphere;
    return (uint32_t)t.nodes.size();
}

// ---- pieces the reference's own tests pin ----
void rayz_oracle_camera_init(double vfov, double focus_dist, double defocus_angle, const double* look_from,
                             const double* look_at, const double* vup, uint32_t img_h, uint32_t img_w,
                             RayzCameraDesc* out) {
    A::cameraTo(A::Camera::init(vfov, focus_dist, defocus_angle, A::v3(look_from), A::v3(look_at), A::v3(vup), img_h,
                                img_w),
                out);
}
void rayz_oracle_get_ray_norng(const RayzCameraDesc* c, uint32_t px, uint32_t py, double* origin, double* dir) {
    const A::Ray r = A::cameraFrom(*c).getRay<Xoshiro256pp>(px, py, nullptr);
    origin[0] = r.origin.x, origin[1] = r.origin.y, origin[2] = r.origin.z;
    dir[0] = r.dir.x, dir[1] = r.dir.y, dir[2] = r.dir.z;
}
void rayz_oracle_refract(const double* unit_dir, const double* n, double eta, double* out) {
    const A::V3 r = A::refract(A::v3(unit_dir), A::v3(n), eta);
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
double rayz_oracle_reflectance(double cos, double ri) { return A::reflectance(cos, ri); }
void rayz_oracle_sphere_bbox(const RayzSphere* s, double* low, double* high) {
    A::Sphere q;
    q.center.origin = A::v3(s->center);
    q.center.dir = A::v3(s->velocity);
    q.radius = s->radius;
    q.material = s->material;
    const A::AABB b = q.boundingBox();
    low[0] = b.low.x, low[1] = b.low.y, low[2] = b.low.z;
    high[0] = b.high.x, high[1] = b.high.y, high[2] = b.high.z;
}
void rayz_oracle_aabb_enclose(const double* alo, const double* ahi, const double* blo, const double* bhi, double* lo,
                              double* hi) {
    const A::AABB r = A::AABB::enclose(A::AABB::init(A::v3(alo), A::v3(ahi)), A::AABB::init(A::v3(blo), A::v3(bhi)));
    lo[0] = r.low.x, lo[1] = r.low.y, lo[2] = r.low.z;
    hi[0] = r.high.x, hi[1] = r.high.y, hi[2] = r.high.z;
}
int rayz_oracle_aabb_hit(const double* a, const double* b, const double* origin, const double* dir, double tmin,
                         double tmax) {
    A::Ray r;
    r.origin = A::v3(origin);
    r.dir = A::v3(dir);
    return A::AABB::init(A::v3(a), A::v3(b)).hit(r, tmin, tmax) ? 1 : 0;
}
// op: 0 add, 1 sub, 2 mul(b[0]), 3 unit, 4 cross, 5 vmul, 6 sqrt, 7 clamp(b[0],b[1]), 8 div(b[0])
void rayz_oracle_v3_op(int op, const double* a, const double* b, double* out) {
    const A::V3 x = A::v3(a), y = b ? A::v3(b) : A::V3{};
    A::V3 r;
    switch (op) {
    case 0: r = x.add(y); break;
    case 1: r = x.sub(y); break;
    case 2: r = x.mul(y.x); break;
    case 3: r = x.unit(); break;
    case 4: r = x.cross(y); break;
    case 5: r = x.vmul(y); break;
    case 6: r = x.sqrt(); break;
    case 7: r = x.clamp(y.x, y.y); break;
    case 8: r = x.div(y.x); break;
    default: break;
    }
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
double rayz_oracle_v3_dot(const double* a, const double* b) { return A::v3(a).dot(A::v3(b)); }
double rayz_oracle_v3_mag(const double* a) { return A::v3(a).mag(); }
int rayz_oracle_v3_amax(const double* a) { return A::v3(a).amax(); }

// `Image.writePPM`'s per-pixel transform, src/image.zig:35-38
void rayz_oracle_ppm_u8(const double* rgb, uint8_t* out) {
    const A::V3 c = A::v3(rgb).sqrt().clamp(0, 1);
    out[0] = (uint8_t)(c.x * 255);
    out[1] = (uint8_t)(c.y * 255);
    out[2] = (uint8_t)(c.z * 255);
}
// whole file, src/image.zig:29-41
int rayz_oracle_write_ppm(const char* path, const double* rgb, uint32_t w, uint32_t h) {
    FILE* f = std::fopen(path, "w");
    if (!f) return -1;
    std::fprintf(f, "P3\n%u %u\n%d\n", w, h, 255);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        uint8_t c[3];
        rayz_oracle_ppm_u8(rgb + 3 * i, c);
        std::fprintf(f, "%u %u %u\n", c[0], c[1], c[2]);
    }
    std::fclose(f);
    return 0;
}

// ---- RNG streams ----
void rayz_oracle_splitmix64(uint64_t seed, uint32_t n, uint64_t* out) {
    SplitMix64 g{seed};
    for (u32 i = 0; i < n; ++i) out[i] = g.next();
}
void rayz_oracle_xoshiro_seed(uint64_t seed, uint64_t* state) {
    Xoshiro256pp g;
    g.seed(seed);
    std::memcpy(state, g.s, 32);
}
void rayz_oracle_xoshiro_u64(uint64_t* state, uint32_t n, uint64_t* out) {
    Xoshiro256pp g;
    std::memcpy(g.s, state, 32);
    for (u32 i = 0; i < n; ++i) out[i] = g.next();
    std::memcpy(state, g.s, 32);
}
void rayz_oracle_xoshiro_f64(uint64_t* state, uint32_t n, double* out) {
    Xoshiro256pp g;
    std::memcpy(g.s, state, 32);
    for (u32 i = 0; i < n; ++i) out[i] = g.float64();
    std::memcpy(state, g.s, 32);
}
void rayz_oracle_pcg32(uint64_t initstate, uint64_t initseq, uint32_t n, uint32_t* out) {
    Pcg32 g;
    g.srandom(initstate, initseq);
    for (u32 i = 0; i < n; ++i) out[i] = g.next();
}
void rayz_oracle_pcg32_path(uint64_t seed, uint64_t path_id, uint32_t n, uint32_t* out) {
    Pcg32 g;
    g.seed_path(seed, path_id);
    for (u32 i = 0; i < n; ++i) out[i] = g.next();
}

// ---- scene `randomBouncing`, src/rayz.zig:45-168, grid a,b ∈ [lo,hi) (reference: -11, 11) ----
// Draws from the Tracer's stream (rng_state in/out, src/rayz.zig:109).  counts = {spheres, materials, textures}.
int rayz_oracle_random_bouncing(uint64_t* rng_state, int lo, int hi, RayzSphere* sph, uint32_t cap_s,
                                RayzMaterial* mat, uint32_t cap_m, RayzTexture* tex, uint32_t cap_t,
                                uint32_t* counts) {
    Xoshiro256pp g;
    std::memcpy(g.s, rng_state, 32);
    u32 ns = 0, nm = 0, nt = 0;
    auto addTex = [&](u32 kind, double scale, u32 even, u32 odd, double r, double gg, double b) -> int {
        if (nt >= cap_t) return -1;
        RayzTexture t{};
        t.kind = kind, t.even = even, t.odd = odd, t.scale = scale;
        t.color[0] = r, t.color[1] = gg, t.color[2] = b;
        tex[nt] = t;
        return (int)nt++;
    };
    auto addMat = [&](u32 kind, u32 texture, double param) -> int {
        if (nm >= cap_m) return -1;
        RayzMaterial m{};
        m.kind = kind, m.texture = texture, m.method = RAYZ_DIFFUSE_HEMISPHERE, m.param = param;
        mat[nm] = m;
        return (int)nm++;
    };
    auto addSph = [&](double cx, double cy, double cz, double vy, double r, u32 m) -> int {
        if (ns >= cap_s) return -1;
        RayzSphere s{};
        s.center[0] = cx, s.center[1] = cy, s.center[2] = cz;
        s.velocity[1] = vy;
        s.radius = r, s.material = m;
        sph[ns] = s;
        return (int)ns++;
    };
    // ground: Zig evaluates the nested `try` arguments innermost-first: even, odd, checker, material, sphere
    int e = addTex(RAYZ_TEX_SOLID, 0, 0, 0, 0.2, 0.3, 0.1);                  // :64-68
    int o = addTex(RAYZ_TEX_SOLID, 0, 0, 0, 0.9, 0.9, 0.9);                  // :69-71
    if (e < 0 || o < 0) return -1;
    int ck = addTex(RAYZ_TEX_CHECKER, 0.32, (u32)e, (u32)o, 0, 0, 0);        // :62-72
    if (ck < 0) return -1;
    int m = addMat(RAYZ_MAT_DIFFUSE, (u32)ck, 0);
    if (m < 0 || addSph(0, -1000, 0, 0, 1000, (u32)m) < 0) return -1;        // :58-74
    m = addMat(RAYZ_MAT_DIELECTRIC, 0, 1.5);                                 // :77-83
    if (m < 0 || addSph(0, 1, 0, 0, 1.0, (u32)m) < 0) return -1;
    int t = addTex(RAYZ_TEX_SOLID, 0, 0, 0, 0.4, 0.2, 0.1);                  // :84-92
    if (t < 0) return -1;
    m = addMat(RAYZ_MAT_DIFFUSE, (u32)t, 0);
    if (m < 0 || addSph(-4, 1, 0, 0, 1.0, (u32)m) < 0) return -1;
    t = addTex(RAYZ_TEX_SOLID, 0, 0, 0, 0.7, 0.6, 0.5);                      // :93-106
    if (t < 0) return -1;
    m = addMat(RAYZ_MAT_METALLIC, (u32)t, 0);
    if (m < 0 || addSph(4, 1, 0, 0, 1.0, (u32)m) < 0) return -1;
    for (int a = lo; a < hi; ++a) {                                          // :110-166
        for (int b = lo; b < hi; ++b) {
            const double rand_mat = g.float64();
            A::V3 center{(double)a + 0.9 * g.float64(), 0.2, (double)b + 0.9 * g.float64()};
            if (center.sub(A::V3{4, 0.2, 0}).mag() <= 0.9) continue;
            double vy = 0;
            int mh;
            if (rand_mat < 0.8) {
                const A::V3 c1 = A::v3random(g, 0, 1.0);
                const A::V3 c2 = A::v3random(g, 0, 1.0);
                const A::V3 col = c1.vmul(c2);
                t = addTex(RAYZ_TEX_SOLID, 0, 0, 0, col.x, col.y, col.z);
                if (t < 0) return -1;
                mh = addMat(RAYZ_MAT_DIFFUSE, (u32)t, 0);
                vy = A::V3{0, 1, 0}.mul(g.float64() * 0.5).y;
            } else if (rand_mat < 0.95) {
                const double fuzz = g.float64() * 0.5;                       // field order: fuzz, then texture
                const A::V3 col = A::v3random(g, 0.5, 1.0);
                t = addTex(RAYZ_TEX_SOLID, 0, 0, 0, col.x, col.y, col.z);
                if (t < 0) return -1;
                mh = addMat(RAYZ_MAT_METALLIC, (u32)t, fuzz);
            } else {
                mh = addMat(RAYZ_MAT_DIELECTRIC, 0, 1.5);
            }
            if (mh < 0 || addSph(center.x, center.y, center.z, vy, 0.2, (u32)mh) < 0) return -1;
        }
    }
    std::memcpy(rng_state, g.s, 32);
    counts[0] = ns, counts[1] = nm, counts[2] = nt;
    return RAYZ_OK;
}

} // extern "C"
