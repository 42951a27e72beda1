// rayz_device.hpp — gfx950 device code of the render path (included by rayz_hip.hip only).
//
// One work-item per pixel-sample path.  A wave keeps 64 paths in flight and re-fills finished lanes
// from a global work queue (wave64 ballot + mbcnt prefix ranks, one atomic per wave), so the sphere
// scan — ≥95 % of the time on a flat hit list — always runs with a full EXEC mask.  The scan is
// wave-uniform over the sphere index: sphere records come through the SCALAR path (s_load_dwordx4/x8
// from a constant-address-space pointer) and feed the VALU as SGPR operands, which costs no VGPRs, no
// LDS bandwidth and no 64× replicated vector loads.  Arithmetic is the "mode B" specification of
// DESIGN.md §4; every statement below has its twin in oracle/rayz_oracle.cpp namespace B, and the two
// must agree bit for bit (build with -ffp-contract=off: FMAs appear only where written).
//
// Reference semantics restated here (file:line into jlucier/rayz):
//   camera ray      src/camera.zig:59-90      sphere test   src/geom.zig:38-66
//   hit record      src/hit.zig:25-41         scatter       src/material.zig:73-160,179-211
//   textures        src/material.zig:19-51    bounce loop   src/renderer.zig:103-126 (recursion → loop)
//   pixel mean      src/renderer.zig:86-95
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rayz_dev {

#define RAYZ_CONSTANT __attribute__((address_space(4)))
#define RAYZ_GLOBAL __attribute__((address_space(1)))

typedef float f4 __attribute__((ext_vector_type(4)));
typedef double d4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int kMaxRejectionTries = 64;
constexpr int kMaxTextureDepth = 8;
#ifndef RAYZ_GROUP
#define RAYZ_GROUP 4
#endif
// spheres per scan group (and per SoA block of a stream): 4 — the two ping-pong SGPR sets of a y-moving group are
// 2 x 5 x G registers, and more than ~40 SGPRs spill inside the loop.  (The scan streams are f32 for both precisions;
// group_size<double> = 2 belonged to the f64 streams the f64 kernel scanned until round 2.)
template <class R> constexpr int group_size() { return sizeof(R) == 8 ? RAYZ_GROUP / 2 : RAYZ_GROUP; }
constexpr int kStaticGroup = RAYZ_GROUP; // A stream is padded to a whole
constexpr int kMovYGroup = RAYZ_GROUP;   //   number of group PAIRS plus two spare groups, so that the prefetch of the
constexpr int kMovGGroup = 2;   //   next group never leaves the array.
constexpr int kTriGroup = 2;

template <class R> struct VecOf;
template <> struct VecOf<float> { typedef f4 type; typedef f2 pair; };
template <> struct VecOf<double> { typedef d4 type; typedef d2 pair; };

// ---- device-resident scene (HBM layout) ------------------------------------------
// Scan streams, one per velocity class, each padded with never-hit records (r² = -inf):
//   static  v = 0            stat: blocks of G = 4 spheres, SoA inside a block:
//                                  cx[G] cy[G] cz[G] r²[G]
//   mov-Y   v = (0, vy, 0)   movy: blocks of G: cx[G] cy[G] cz[G] r²[G] vy[G]
//                            (what randomBouncing makes)
//   mov-G   any other v      movg[2i] = {cx, cy, cz, r²}, movg[2i+1] = {vx, vy, vz, 0}
// The block layout puts the same field of two neighbouring spheres in one aligned SGPR pair, which is what a
// v_pk_fma_f32 takes as a single scalar operand.  A "slot" numbers the records stat | movy | movg
// in that order.  All three are f32 for both precisions (the reject test only filters); the f64
// copies feed the narrow phase.
template <class R> struct DevScene {
    typedef typename VecOf<R>::type r4;
    const float* stat;       // [4 * ns_pad + spare block]   (the scan streams are f32 for both precisions: the reject test
                             //                               only filters, the narrow phase decides — DESIGN.md §4.3)
    const float* movy;       // [5 * ny_pad + spare block]
    const f4* movg;
    const d4* slot64;        // [2 * slots] the pool's own f64 values per slot: {cx, cy, cz, r²}, {vx, vy, vz, 0}
    const uint32_t* slot_pool; // [slots] pool index of each slot
    const r4* sph_pool;      // [2 * n_spheres] by POOL index: {cx, cy, cz, r²}, {vx, vy, vz, bits(material)}
    const r4* mat;           // [n_mat] {bits(kind | method << 8), bits(texture), param, 1/param}
    const r4* tex;           // [2 * n_tex] {bits(kind), bits(even), bits(odd), scale}, {r, g, b, 0}
    uint32_t ns_pad, ny_pad, ng_pad, n_spheres;
    // build-defined triangles (hittable index = n_spheres + i): {v0, bits(material)}, {e1 = v1 - v0, 0}, {e2 = v2 - v0, 0};
    // pool order, padded with degenerate records (e1 = e2 = 0: det = 0, never accepted) like the sphere streams
    const r4* tri;           // [3 * (nt_pad + kTriGroup)]
    uint32_t nt_pad, n_triangles;
    // BVH traversal (RAYZ_TRAVERSAL_BVH): the reference's tree in depth-first pre-order, DESIGN.md §6
    const f4* bvh_nodes;     // per INNER node, its two children's boxes for BOTH precisions (the box test only culls, §4.8), in one
                             // of two formats (bvh_quantized; the host picks by tree size, rayz_hip.hip: use_quantized_nodes):
                             //  * f32 planes, 64 B = four 16-B loads:
                             //      {L.lo.xyz, bits(L.ref)}, {L.hi.xyz, 0}, {R.lo.xyz, bits(R.ref)}, {R.hi.xyz, 0}
                             //  * 16-bit plane indices on a scene-wide grid (plane = bvh_glo[k] + index · bvh_cell[k]; lower
                             //    planes rounded down, upper ones up), 32 B = two loads:
                             //      {L.lo.x | L.hi.x << 16, L.lo.y | L.hi.y << 16, L.lo.z | L.hi.z << 16, bits(L.ref)}, {R ...}
                             //    Half the bytes: the walk is bound by the vector-memory address pipe, which works per
                             //    instruction and per byte requested (profiles/r03/lds_top), and twice as many records fit
                             //    the LDS top — at the price of 12 conversions per step: it pays for trees much larger than
                             //    the LDS top (config 5: +7.7 %), not for those the top mostly covers (config 3: −2.7 %)
                             //   ref = the child's record's byte offset (index << 6 or << 5), or
                             //         kBvhLeafFlag | first << 4 | type1 << 3 | type0 << 2 | count
    float bvh_glo[3], bvh_cell[3]; // the grid of the plane indices (f32 planes: glo = 0, cell = 1 — a plane IS its "index")
    const r4* bvh_leaf;      // [stride * slots] leaf order.  sphere: {c, r²}, {v, bits(hittable)};
                             //                  triangle: {v0, bits(hittable)}, {e1, 0}, {e2, 0}
    const d4* bvh_sph64;     // [2 * slots] leaf order, sphere slots only: {c, r²}, {v, 0}
    uint32_t bvh_n_nodes, bvh_leaf_stride; // bvh_n_nodes = number of inner-node records (0: nothing in the tree)
    // oversized hittables kept out of the tree (bvh_build.hpp), tested once per segment before the walk: up to 4 leaf
    // descriptors (first << 4 | type1 << 3 | type0 << 2 | count) naming slots after the tree's own in bvh_leaf / bvh_sph64
    uint32_t bvh_n_big_leaves, bvh_big[4];
    // the first bvh_top / 64 inner-node records (the top levels of the tree, numbered breadth-first) are copied to LDS by
    // every workgroup: a quarter of all box steps then read their node in ~100 cycles instead of a global round trip.
    // In BYTES, as inner references are (index << 6).
    uint32_t bvh_top;
};

template <class R> struct DevCamera {
    R from[3], du[3], dv[3], pxo[3], defu[3], defv[3];
    uint32_t defocus, _pad;
};

template <class R> struct TraceArgs {
    DevScene<R> sc;
    DevCamera<R> cam;
    typename VecOf<R>::type* partial; // [total_items] chunk sums
    unsigned long long* counters;     // [0] work-queue head, [1] segments, [2] node tests, [3] sphere tests (BVH)
    unsigned long long seed;
    R tmin;
    uint32_t width, height, spp, max_bounces;
    const uint32_t* chunk_start; // [chunks_per_px + 1] first sample of every chunk of a pixel (the chunk schedule)
    uint32_t chunks_per_px;
    uint32_t chunk_uniform, chunk_n_uniform; // the table's uniform prefix: chunk k < chunk_n_uniform covers samples k·chunk_uniform .. (k+1)·chunk_uniform
    uint32_t tile_rows, shard_index, shard_count, shard_pixels;
    uint32_t tiled_pixels;   // the first this-many local pixels are dealt to the queue as 8x8 tiles (place_item)
    uint32_t total_items;
    uint32_t bvh_keep;       // BVH kernel: keep_active | keep_stepping << 8 (see trace_kernel_bvh)
    uint32_t bvh_top_words;   // BVH kernel: u32s of LDS taken by the copy of the tree's top (the per-lane stacks follow)
    uint32_t bvh_big_words;   // BVH kernel: where (in u32s of LDS) the oversized hittables' records are kept, after the stacks
    uint32_t queue_grab;     // work items a wave reserves per atomic on the queue head (kQueueGrab; scheduling only)
#ifdef RAYZ_EXPERIMENTS // (kept out of the product kernels' argument block: every field costs scalar registers in all of them)
    uint32_t x_words;        // exchange kernel (trace_kernel_bvhx): where (in u32s of LDS) its exchange area starts
    uint32_t x_slots;        //   ray slots per walker wave
    uint32_t x_cfg;          //   scheduling: exchange when this many lanes finished | shader's minimum batch << 8 | its patience << 16 | priorities << 24
#endif
};

// ---- small helpers -----------------------------------------------------------------------------
__device__ __forceinline__ float fm(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fm(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float mx(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double mx(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ float sq(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ double sq(double x) { return __builtin_sqrt(x); }
__device__ __forceinline__ float fl(float x) { return __builtin_floorf(x); }
__device__ __forceinline__ double fl(double x) { return __builtin_floor(x); }
__device__ __forceinline__ float ab(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ double ab(double x) { return __builtin_fabs(x); }
__device__ __forceinline__ uint32_t bits(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t bits(double x) { return (uint32_t)__builtin_bit_cast(uint64_t, x); }

// The acceptance rule of every narrow phase: a root in range that is nearer, or as near with the larger hittable
// index.  Written with & and | on purpose: `&&` / `||` make the compiler branch per term (five s_and_saveexec / s_cbranch_execz
// pairs around two moves in the candidates' phase); this is four compares, three scalar mask operations and two selects.
#ifndef RAYZ_BRANCHY_ACCEPT
template <class R> __device__ __forceinline__ void accept_root(R t, int index, R tmin, R& tbest, int& ibest) {
    const bool take = (t >= tmin) & ((t < tbest) | ((t == tbest) & (index > ibest)));
    tbest = take ? t : tbest;
    ibest = take ? index : ibest;
}
#else
template <class R> __device__ __forceinline__ void accept_root(R t, int index, R tmin, R& tbest, int& ibest) {
    if (t >= tmin && (t < tbest || (t == tbest && index > ibest))) {
        tbest = t;
        ibest = index;
    }
}
#endif

template <class R> struct Bits; // a u32 carried in the bits of an R
template <> struct Bits<float> {
    static __host__ __device__ __forceinline__ float from(uint32_t u) { return __builtin_bit_cast(float, u); }
};
template <> struct Bits<double> {
    static __host__ __device__ __forceinline__ double from(uint32_t u) { return __builtin_bit_cast(double, (uint64_t)u); }
};

template <class R> struct V {
    R x, y, z;
};
template <class R> __device__ __forceinline__ R dot3(V<R> a, V<R> b) { return fm(a.z, b.z, fm(a.y, b.y, a.x * b.x)); }
template <class R> __device__ __forceinline__ V<R> neg(V<R> a) { return {-a.x, -a.y, -a.z}; }
template <class R> __device__ __forceinline__ V<R> unit(V<R> a) {
    const R m = sq(dot3(a, a));
    const R inv = R(1) / m;
    return {a.x * inv, a.y * inv, a.z * inv};
}

// ---- PCG32 (XSH-RR 64/32), one stream per path: DESIGN.md §4.1 ---------------------------------
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
struct Pcg32 {
    unsigned long long state, inc;
    __device__ __forceinline__ uint32_t next() {
        const unsigned long long old = state;
        state = old * 6364136223846793005ull + inc;
        const uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        const uint32_t rot = (uint32_t)(old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
    }
    __device__ __forceinline__ void seed_path(unsigned long long seed, unsigned long long path_id) {
        const unsigned long long a = mix64(seed + (path_id + 1) * 0x9e3779b97f4a7c15ull);
        const unsigned long long b = mix64(a + 0x9e3779b97f4a7c15ull);
        state = 0;
        inc = (b << 1) | 1u;
        next();
        state += a;
        next();
    }
};
template <class R> __device__ __forceinline__ R uniform(Pcg32& g);
template <> __device__ __forceinline__ float uniform<float>(Pcg32& g) { return (float)(g.next() >> 8) * 0x1p-24f; }
template <> __device__ __forceinline__ double uniform<double>(Pcg32& g) { return (double)g.next() * 0x1p-32; }
// Known-answer entry (rayz_hip_kat) only: draws come from a caller-supplied list, so that the device functions can
// be evaluated on the same uniforms as the reference restatement (0.5 once the list is used up).
struct ListRng {
    const double* u;
    uint32_t n, i;
};
template <class R> __device__ __forceinline__ R uniform(ListRng& g) {
    const R v = g.i < g.n ? (R)g.u[g.i] : R(0.5);
    g.i++;
    return v;
}

template <class R, class G> __device__ __forceinline__ V<R> random_in_unit_sphere(G& g) { // src/material.zig:196-202
    V<R> v{0, 0, 0};
    for (int i = 0; i < kMaxRejectionTries; ++i) {
        v.x = fm(uniform<R>(g), R(2), R(-1));
        v.y = fm(uniform<R>(g), R(2), R(-1));
        v.z = fm(uniform<R>(g), R(2), R(-1));
        // `v.mag() <= 1` (src/material.zig:199) without the square root: for a correctly rounded sqrt,
        // sqrt(x) <= 1  <=>  x <= nextafter(1): sqrt(1 + ulp) = 1 + ulp/2 - ulp²/8 lies below the midpoint and rounds to 1,
        // sqrt(1 + 2 ulp) rounds to 1 + ulp.  Same decision, bit for bit, as the oracle's literal form.
        if (dot3(v, v) <= (sizeof(R) == 4 ? R(0x1.000002p0f) : R(0x1.0000000000001p0))) break;
    }
    return v;
}

// Cell parity of a checker at point p: floor(p_k / scale) summed as integers, floored mod 2 (src/material.zig:32-36;
// the reference sums i64s, here each floor is clamped to ±2^30 and summed as int32 — same parity inside that range).
template <class R> __device__ __forceinline__ uint32_t checker_parity(V<R> p, R scale) {
    const R lim = R(1073741824.0);
    R fx = fl(p.x / scale), fy = fl(p.y / scale), fz = fl(p.z / scale);
    fx = fx < -lim ? -lim : fx;
    fx = fx > lim ? lim : fx;
    fy = fy < -lim ? -lim : fy;
    fy = fy > lim ? lim : fy;
    fz = fz < -lim ? -lim : fz;
    fz = fz > lim ? lim : fz;
    const uint32_t s = (uint32_t)(int32_t)fx + (uint32_t)(int32_t)fy + (uint32_t)(int32_t)fz;
    return s & 1u;
}
template <class R, class TexPtr>
__device__ __forceinline__ V<R> texture_value(TexPtr tex, uint32_t idx, V<R> p) { // src/material.zig:19-51
    typedef typename VecOf<R>::type r4;
    for (int depth = 0; depth < kMaxTextureDepth; ++depth) { // rayz_hip_scene_create refuses deeper chains and cycles
        const r4 h = tex[2 * idx];
        if (bits(h.x) == 1u) { // RAYZ_TEX_SOLID
            const r4 c = tex[2 * idx + 1];
            return {c.x, c.y, c.z};
        }
        idx = checker_parity<R>(p, h.w) == 0u ? bits(h.y) : bits(h.z);
    }
    return {R(0), R(0), R(0)};
}

// ---- narrow phase: one candidate that passed the reject test, src/geom.zig:40-61 ------------------
// The reference's quadratic in f64 on the pool's f64 sphere, for the ray as the kernel holds it; the
// chosen root is rounded to R.  Ties in t go to the larger pool index (the reference's "t ≤ maxt, later
// wins" over its flat list, src/hit.zig:208-214), which makes the result independent of the order in which
// candidates are examined — so the flat-list scan only PARKS candidate slots (≤ 4 per ray) and evaluates
// them afterwards for all lanes together, instead of interrupting the scan ~30 times per segment for one or two
// lanes each.
template <class R>
__device__ __forceinline__ void narrow_eval(const d4 c, const d4 v, int pool, V<R> o, V<R> d, R time, double inv_a2, R tmin,
                                            R& tbest, int& ibest) {
    const double dx = d.x, dy = d.y, dz = d.z, tm = time;
    const double qx = fm(v.x, tm, c.x - (double)o.x), qy = fm(v.y, tm, c.y - (double)o.y), qz = fm(v.z, tm, c.z - (double)o.z);
    const double a2 = fm(dz, dz, fm(dy, dy, dx * dx));
    const double hb2 = fm(dz, qz, fm(dy, qy, dx * qx));
    const double cc2 = fm(qz, qz, fm(qy, qy, fm(qx, qx, -c.w)));
    const double disc2 = fm(-a2, cc2, hb2 * hb2);
    if (disc2 >= 0.0) {
        const double rt = __builtin_sqrt(disc2);
        const R t1 = (R)((hb2 - rt) * inv_a2), t2 = (R)((hb2 + rt) * inv_a2);
        const R t = t1 >= tmin ? t1 : t2;
        accept_root<R>(t, pool, tmin, tbest, ibest);
    }
}

// ---- reject test -------------------------------------------------------------------------
// Per ray: an orthonormal pair (e1, e2) perpendicular to the unit direction ud, with e1 in the xz-plane
// (e1.y = 0), e2 = ud × e1, and k = −o·e.  The squared distance from the ray's line to a centre c is
// p1² + p2² with p1 = c·e1 + k1 (2 FMAs; a y-velocity never enters it) and p2 = c·e2 + k2 (3 FMAs), so the test
// r² − p1² − p2² ≥ 0 costs 7 VALU per static sphere, 8 with a y-velocity, 12 with a general one — against
// 10 / 11 / 13 for the textbook (ud·oc)² − (|oc|² − r²), and needs no c − o.
template <class R> struct RayBasis {
    R e1x, e1z, e2x, e2y, e2z, k1, k2;
};
template <class R> __device__ __forceinline__ RayBasis<R> make_basis(V<R> ud, V<R> o) {
    RayBasis<R> b;
    const R h2 = fm(ud.z, ud.z, ud.x * ud.x);
    b.e1x = R(1);
    b.e1z = R(0);
    if (h2 > R(1e-30)) { // below that (|ud.x|, |ud.z| < 1e-15) the x axis is perpendicular to ud far inside the slack
        const R ih = R(1) / sq(h2);
        b.e1x = ud.z * ih;
        b.e1z = -(ud.x * ih);
    }
    b.e2x = ud.y * b.e1z;
    b.e2y = fm(ud.z, b.e1x, -(ud.x * b.e1z));
    b.e2z = -(ud.y * b.e1x);
    b.k1 = -fm(o.z, b.e1z, o.x * b.e1x);
    b.k2 = -fm(o.z, b.e2z, fm(o.y, b.e2y, o.x * b.e2x));
    return b;
}
template <class R> __device__ __forceinline__ R basis_p1(const RayBasis<R>& b, R cx, R cz) {
    return fm(cz, b.e1z, fm(cx, b.e1x, b.k1));
}
template <class R> __device__ __forceinline__ R basis_p2(const RayBasis<R>& b, R cx, R cy, R cz) {
    return fm(cz, b.e2z, fm(cy, b.e2y, fm(cx, b.e2x, b.k2)));
}
template <class R> __device__ __forceinline__ R basis_disc(R p1, R p2, R r2) { return fm(-p1, p1, fm(-p2, p2, r2)); }
// The filter is CONSERVATIVE: `r2` is not r² but (r + E)², E = 32·u·(|c| + |v| + r + S) added on the host
// (pad_radius2 in rayz_hip.hip; u = unit roundoff of R, S = bound on every ray origin).  E covers the rounding of
// unit(d), of the basis, of k, p1, p2 and of this expression (derivation: DESIGN.md §4.3), so a sphere whose f64
// discriminant is ≥ 0 always reaches the narrow phase; the narrow phase decides.
template <class R> __device__ __forceinline__ bool sphere_candidate(R p1, R p2, R r2_padded) {
    return basis_disc<R>(p1, p2, r2_padded) >= R(0);
}

template <class R, int N> __device__ __forceinline__ R max_of(const R (&v)[N]) {
    R m = v[0];
#pragma unroll
    for (int k = 1; k < N; ++k) m = mx(m, v[k]);
    return m;
}

template <class R> __device__ __forceinline__ V<R> cross3(V<R> a, V<R> b) {
    return {fm(a.y, b.z, -(a.z * b.y)), fm(a.z, b.x, -(a.x * b.z)), fm(a.x, b.y, -(a.y * b.x))};
}
__device__ __forceinline__ float mn(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ double mn(double a, double b) { return __builtin_fmin(a, b); }

// ---- build-defined triangle (Möller–Trumbore in R) ----------------------------------
// Filter value: ≥ 0 iff the barycentrics pass, written without a division or a sign branch:
// su = (s·p)·det, sv = (d·q)·det, w = det² − (su + sv); candidate iff min(su, sv, w) ≥ 0.  There is no f64 phase
// behind it: for triangles this value IS the decision (in R, identically in the oracle), not a pre-filter.
template <class R> __device__ __forceinline__ R tri_filter(V<R> v0, V<R> e1, V<R> e2, V<R> o, V<R> d) {
    const V<R> pv = cross3(d, e2);
    const R det = dot3(e1, pv);
    const V<R> sv{o.x - v0.x, o.y - v0.y, o.z - v0.z};
    const R su = dot3(sv, pv) * det;
    const V<R> qv = cross3(sv, e1);
    const R svv = dot3(d, qv) * det;
    const R w = fm(det, det, -(su + svv));
    return mn(mn(su, svv), w);
}
// Candidate: t = (e2·q) / det, then the same acceptance rule as spheres (nearest, ties to the larger index).
template <class R>
__device__ __forceinline__ void tri_accept(R filt, V<R> v0, V<R> e1, V<R> e2, V<R> o, V<R> d, R tmin, int prim, R& tbest,
                                           int& ibest) {
    if (filt >= R(0)) {
        const V<R> pv = cross3(d, e2);
        const R det = dot3(e1, pv);
        if (det != R(0)) {
            const V<R> sv{o.x - v0.x, o.y - v0.y, o.z - v0.z};
            const V<R> qv = cross3(sv, e1);
            const R t = dot3(e2, qv) / det;
            accept_root<R>(t, prim, tmin, tbest, ibest);
        }
    }
}

// One scan group of a velocity class, held in SGPRs: load() issues the scalar loads, test() runs the
// reject test of its spheres against 64 rays and sends candidates to the narrow phase.
template <class R, int CLS> struct ScanGroup;

// The reject tests of a block run two spheres per instruction: each stage is ONE packed FMA (v_pk_fma_f32 for
// R = float) whose scalar operand is an SGPR PAIR — the same field of two neighbouring spheres.  Measured on
// gfx950 (tools/ubench): a VALU instruction that reads a different SGPR each time issues at ≈2.75 cycles, not 2,
// so the 7 scalar reads of a test bound the scalar-FMA form at ≈21 ticks per wave-test; the packed form needs
// 3.5 pair reads per test and runs at 14.7.  Every half of a packed FMA is an ordinary IEEE FMA: results are
// bit-identical to the scalar form (and to the oracle).
template <class R> struct ScanGroup<R, 0> { // static
    typedef typename VecOf<R>::pair pr;
    static constexpr int G = group_size<R>(), H = G / 2;
    pr cx[H], cy[H], cz[H], r2[H];
    template <class SC> static __device__ __forceinline__ const RAYZ_CONSTANT R* stream(const SC& sc) { return (const RAYZ_CONSTANT R*)sc.stat; }
    __device__ __forceinline__ void load(const RAYZ_CONSTANT R* base, int i) {
        const RAYZ_CONSTANT pr* p = (const RAYZ_CONSTANT pr*)(base + 4 * i);
#pragma unroll
        for (int q = 0; q < H; ++q) cx[q] = p[q], cy[q] = p[H + q], cz[q] = p[2 * H + q], r2[q] = p[3 * H + q];
    }
    __device__ __forceinline__ void touch() const { asm volatile("" ::"s"(cx[0])); }
    __device__ __forceinline__ void opaque() {
#pragma unroll
        for (int q = 0; q < H; ++q) asm volatile("" : "+s"(cx[q]), "+s"(cy[q]), "+s"(cz[q]), "+s"(r2[q]));
    }
    __device__ __forceinline__ void discs(R (&out)[G], const RayBasis<R>& b, R) const {
        const pr E1x{b.e1x, b.e1x}, E1z{b.e1z, b.e1z}, E2x{b.e2x, b.e2x}, E2y{b.e2y, b.e2y}, E2z{b.e2z, b.e2z},
            K1{b.k1, b.k1}, K2{b.k2, b.k2};
#pragma unroll
        for (int q = 0; q < H; ++q) {
            pr p1 = __builtin_elementwise_fma(cx[q], E1x, K1);
            pr p2 = __builtin_elementwise_fma(cx[q], E2x, K2);
            p1 = __builtin_elementwise_fma(cz[q], E1z, p1);
            p2 = __builtin_elementwise_fma(cy[q], E2y, p2);
            p2 = __builtin_elementwise_fma(cz[q], E2z, p2);
            const pr d = __builtin_elementwise_fma(-p1, p1, __builtin_elementwise_fma(-p2, p2, r2[q]));
            out[2 * q] = d.x;
            out[2 * q + 1] = d.y;
        }
    }
    template <class SC> static __device__ __forceinline__ int slot0(const SC&) { return 0; }
};
template <class R> struct ScanGroup<R, 1> { // mov-Y
    typedef typename VecOf<R>::pair pr;
    static constexpr int G = group_size<R>(), H = G / 2;
    pr cx[H], cy[H], cz[H], r2[H], vy[H];
    template <class SC> static __device__ __forceinline__ const RAYZ_CONSTANT R* stream(const SC& sc) { return (const RAYZ_CONSTANT R*)sc.movy; }
    __device__ __forceinline__ void load(const RAYZ_CONSTANT R* base, int i) {
        const RAYZ_CONSTANT pr* p = (const RAYZ_CONSTANT pr*)(base + 5 * i);
#pragma unroll
        for (int q = 0; q < H; ++q)
            cx[q] = p[q], cy[q] = p[H + q], cz[q] = p[2 * H + q], r2[q] = p[3 * H + q], vy[q] = p[4 * H + q];
    }
    __device__ __forceinline__ void touch() const { asm volatile("" ::"s"(cx[0]), "s"(vy[0])); }
    __device__ __forceinline__ void opaque() {
#pragma unroll
        for (int q = 0; q < H; ++q) asm volatile("" : "+s"(cx[q]), "+s"(cy[q]), "+s"(cz[q]), "+s"(r2[q]), "+s"(vy[q]));
    }
    __device__ __forceinline__ void discs(R (&out)[G], const RayBasis<R>& b, R time) const {
        const R t2y = time * b.e2y;
        const pr E1x{b.e1x, b.e1x}, E1z{b.e1z, b.e1z}, E2x{b.e2x, b.e2x}, E2y{b.e2y, b.e2y}, E2z{b.e2z, b.e2z},
            K1{b.k1, b.k1}, K2{b.k2, b.k2}, T2y{t2y, t2y};
#pragma unroll
        for (int q = 0; q < H; ++q) {
            pr p1 = __builtin_elementwise_fma(cx[q], E1x, K1);
            pr p2 = __builtin_elementwise_fma(cx[q], E2x, K2);
            p1 = __builtin_elementwise_fma(cz[q], E1z, p1);
            p2 = __builtin_elementwise_fma(cy[q], E2y, p2);
            p2 = __builtin_elementwise_fma(cz[q], E2z, p2);
            p2 = __builtin_elementwise_fma(vy[q], T2y, p2);
            const pr d = __builtin_elementwise_fma(-p1, p1, __builtin_elementwise_fma(-p2, p2, r2[q]));
            out[2 * q] = d.x;
            out[2 * q + 1] = d.y;
        }
    }
    template <class SC> static __device__ __forceinline__ int slot0(const SC& sc) { return (int)sc.ns_pad; }
};
template <class R> struct ScanGroup<R, 2> { // mov-G
    typedef typename VecOf<R>::type r4;
    static constexpr int G = kMovGGroup;
    r4 c[G], v[G];
    template <class SC> static __device__ __forceinline__ const RAYZ_CONSTANT R* stream(const SC& sc) { return (const RAYZ_CONSTANT R*)sc.movg; }
    __device__ __forceinline__ void load(const RAYZ_CONSTANT R* base, int i) {
        const RAYZ_CONSTANT r4* p = (const RAYZ_CONSTANT r4*)base + 2 * i;
#pragma unroll
        for (int k = 0; k < G; ++k) c[k] = p[2 * k], v[k] = p[2 * k + 1];
    }
    __device__ __forceinline__ void touch() const { asm volatile("" ::"s"(c[0].x)); }
    __device__ __forceinline__ void opaque() {
#pragma unroll
        for (int k = 0; k < G; ++k)
            asm volatile("" : "+s"(c[k].x), "+s"(c[k].y), "+s"(c[k].z), "+s"(c[k].w), "+s"(v[k].x), "+s"(v[k].y), "+s"(v[k].z));
    }
    __device__ __forceinline__ R disc(int k, const RayBasis<R>& b, R time) const {
        const R p1 = fm(v[k].z, time * b.e1z, fm(v[k].x, time * b.e1x, basis_p1<R>(b, c[k].x, c[k].z)));
        const R p2 = fm(v[k].z, time * b.e2z,
                        fm(v[k].y, time * b.e2y, fm(v[k].x, time * b.e2x, basis_p2<R>(b, c[k].x, c[k].y, c[k].z))));
        return basis_disc<R>(p1, p2, c[k].w);
    }
    __device__ __forceinline__ void discs(R (&out)[G], const RayBasis<R>& b, R time) const {
#pragma unroll
        for (int k = 0; k < G; ++k) out[k] = disc(k, b, time);
    }
    template <class SC> static __device__ __forceinline__ int slot0(const SC& sc) { return (int)(sc.ns_pad + sc.ny_pad); }
};

// What the scan needs of one ray (one of the NR rays a lane carries).
template <class R> struct ScanRay {
    V<R> o, d;
    R time;
    RayBasis<float> basis; // the reject test runs in f32 for both precisions (FilterR): built from ud, o narrowed to f32
    float ftime;
    double inv_a2; // 1 / (d·d) in f64, for the narrow phase
    R tbest;
    int ibest;
    uint32_t cand[4]; // parked candidate slots (spheres whose line-distance test passed), evaluated by narrow_flush
    uint32_t ncand;
};

// Reject tests of one group for the lane's NR rays; the running maximum feeds the pair's single branch.
template <class R, int CLS, int NR>
__device__ __forceinline__ void group_discs(const ScanGroup<float, CLS>& g, ScanRay<R> (&ray)[NR],
                                            float (&disc)[NR][ScanGroup<float, CLS>::G], float& m, bool first) {
    constexpr int G = ScanGroup<float, CLS>::G;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        g.discs(disc[r], ray[r].basis, ray[r].ftime);
#pragma unroll
        for (int k = 0; k < G; ++k) m = (first && r == 0 && k == 0) ? disc[0][0] : mx(m, disc[r][k]);
    }
}
// Evaluate every parked candidate of the wave's lanes (round j: lanes with more than j parked slots).
template <class R, int NR> __device__ __forceinline__ void narrow_flush(const DevScene<R>& sc, ScanRay<R> (&ray)[NR], R tmin) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (__ballot(ray[r].ncand > (uint32_t)j) == 0ull) break;
            if (ray[r].ncand > (uint32_t)j) {
                const uint32_t slot = ray[r].cand[j];
                narrow_eval<R>(sc.slot64[2 * slot], sc.slot64[2 * slot + 1], (int)sc.slot_pool[slot], ray[r].o, ray[r].d,
                               ray[r].time, ray[r].inv_a2, tmin, ray[r].tbest, ray[r].ibest);
            }
        }
        ray[r].ncand = 0;
    }
}
// Park one group's candidates (slots first .. first+G-1); a lane whose list is full forces a flush first.
template <class R, int CLS, int NR>
__device__ __forceinline__ void group_collect(const DevScene<R>& sc, int first, ScanRay<R> (&ray)[NR],
                                              const float (&disc)[NR][ScanGroup<float, CLS>::G], R tmin) {
    constexpr int G = ScanGroup<float, CLS>::G;
#pragma unroll
    for (int k = 0; k < G; ++k)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const bool want = disc[r][k] >= 0.0f;
            // usually ONE of the group's spheres made the wave take this path: the others cost a compare and a scalar
            // branch instead of the whole insertion (a small scene takes this path in almost every group: measured +2.5 %
            // on config 2; parking whole group pairs in LDS and re-testing them per lane afterwards was 13 % SLOWER)
            if (__ballot(want) == 0ull) continue;
            if (__ballot(want && ray[r].ncand == 4u) != 0ull) narrow_flush<R, NR>(sc, ray, tmin);
            if (want) {
                const uint32_t slot = (uint32_t)(first + k), n = ray[r].ncand;
                ray[r].cand[0] = n == 0u ? slot : ray[r].cand[0];
                ray[r].cand[1] = n == 1u ? slot : ray[r].cand[1];
                ray[r].cand[2] = n == 2u ? slot : ray[r].cand[2];
                ray[r].cand[3] = n == 3u ? slot : ray[r].cand[3];
                ray[r].ncand = n + 1u;
            }
        }
}

// One velocity class.  n is a multiple of 2·G and the stream carries two spare groups.  Per iteration: wait for
// the two groups loaded during the previous iteration, test group a and immediately reload its SGPR set with the
// group after next, the same for b, then ONE reject branch for the 2·G tests.  (Scalar loads return out of
// order, so a wave can only wait for all of them — lgkmcnt(0) — hence the explicit order; the sched_barriers
// keep hipcc from sinking the loads.  The branch costs ≈10 cycles of a wave's time: once per 8 tests, not 4.)
template <class R, int CLS, int NR>
__device__ __forceinline__ void scan_class(const DevScene<R>& sc, int n, ScanRay<R> (&ray)[NR], R tmin) {
    constexpr int G = ScanGroup<float, CLS>::G;
    if (n == 0) return;
    // the stream's base as a value of its own: read out of the kernel arguments it is one lane of a 16-register block,
    // and a spilled block comes back whole (the f64 kernel paid 20 v_readlane per iteration for this one pointer)
    const RAYZ_CONSTANT float* base = ScanGroup<float, CLS>::stream(sc);
    if constexpr (sizeof(R) == 8) asm volatile("" : "+s"(base));
    ScanGroup<float, CLS> a, b;
    a.load(base, 0);
    b.load(base, G);
    for (int i = 0; i < n; i += 2 * G) {
        float da[NR][G], db[NR][G];
        float m = -1.0f;
#ifdef RAYZ_DEBUG_NOFEED // timing experiment only: never reload (wrong results); values kept opaque to the compiler
        a.opaque();
        group_discs<R, CLS, NR>(a, ray, da, m, true);
        b.opaque();
        group_discs<R, CLS, NR>(b, ray, db, m, false);
#else
        a.touch();
        group_discs<R, CLS, NR>(a, ray, da, m, true);
        a.load(base, i + 2 * G);
        __builtin_amdgcn_sched_barrier(0);
        group_discs<R, CLS, NR>(b, ray, db, m, false);
        b.load(base, i + 3 * G);
        __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef RAYZ_DEBUG_NONARROW // timing experiment only (wrong results)
        if (m >= 1e30f) {
#else
        if (m >= 0.0f) { // any lane, any ray, any of the 2·G spheres: rare
#endif
            const int slot0 = ScanGroup<float, CLS>::slot0(sc);
            group_collect<R, CLS, NR>(sc, slot0 + i, ray, da, tmin);
            group_collect<R, CLS, NR>(sc, slot0 + i + G, ray, db, tmin);
        }
    }
}

// Triangle stream of the flat list: same ping-pong scalar prefetch, groups of kTriGroup.
template <class R> struct TriGroup {
    typedef typename VecOf<R>::type r4;
    r4 a[kTriGroup], b[kTriGroup], c[kTriGroup];
    __device__ __forceinline__ void load(const RAYZ_CONSTANT R* base, int i) {
        const RAYZ_CONSTANT r4* p = (const RAYZ_CONSTANT r4*)base + 3 * i;
#pragma unroll
        for (int k = 0; k < kTriGroup; ++k) a[k] = p[3 * k], b[k] = p[3 * k + 1], c[k] = p[3 * k + 2];
    }
    __device__ __forceinline__ void touch() const { asm volatile("" ::"s"(a[0].x)); }
    template <int NR>
    __device__ __forceinline__ void test(const DevScene<R>& sc, int i, ScanRay<R> (&ray)[NR], R tmin) const {
        R f[NR][kTriGroup];
        R m = R(-1);
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int k = 0; k < kTriGroup; ++k) {
                f[r][k] = tri_filter<R>(V<R>{a[k].x, a[k].y, a[k].z}, V<R>{b[k].x, b[k].y, b[k].z},
                                        V<R>{c[k].x, c[k].y, c[k].z}, ray[r].o, ray[r].d);
                m = (r == 0 && k == 0) ? f[0][0] : mx(m, f[r][k]);
            }
        if (m >= R(0)) {
#pragma unroll
            for (int k = 0; k < kTriGroup; ++k)
#pragma unroll
                for (int r = 0; r < NR; ++r)
                    tri_accept<R>(f[r][k], V<R>{a[k].x, a[k].y, a[k].z}, V<R>{b[k].x, b[k].y, b[k].z},
                                  V<R>{c[k].x, c[k].y, c[k].z}, ray[r].o, ray[r].d, tmin, (int)sc.n_spheres + i + k,
                                  ray[r].tbest, ray[r].ibest);
        }
    }
};
template <class R, int NR>
__device__ __forceinline__ void scan_triangles(const DevScene<R>& sc, ScanRay<R> (&ray)[NR], R tmin) {
    const int n = (int)sc.nt_pad;
    if (n == 0) return;
    const RAYZ_CONSTANT R* base = (const RAYZ_CONSTANT R*)sc.tri;
    if constexpr (sizeof(R) == 8) asm volatile("" : "+s"(base));
    TriGroup<R> a, b;
    a.load(base, 0);
    for (int i = 0; i < n; i += 2 * kTriGroup) {
        a.touch();
        b.load(base, i + kTriGroup);
        __builtin_amdgcn_sched_barrier(0);
        a.template test<NR>(sc, i, ray, tmin);
        b.touch();
        a.load(base, i + 2 * kTriGroup);
        __builtin_amdgcn_sched_barrier(0);
        b.template test<NR>(sc, i + kTriGroup, ray, tmin);
    }
}

// ---- the flat-list scan: nearest hit of each of the lane's NR rays over every hittable -------------------
// Wave-uniform in the sphere index: records arrive by scalar loads and feed the VALU as SGPR operands.  What
// bounds the loop was established by elimination (tools/ubench/): not the sphere feed (a build
// that never reloads runs in the same time), not occupancy, not instruction-level parallelism — but the rate at
// which VALU instructions can read DIFFERENT scalar registers (≈2.75 cycles each), plus ≈10 cycles per reject
// branch.  Hence two spheres per packed FMA (ScanGroup::discs) and one branch per 8 tests (scan_class).
// NR = rays per lane; the product instantiates NR = 1 (two rays per lane halve the scalar loads per test and
// were measured no faster).
template <class R, int NR>
__device__ __forceinline__ void scan_begin(ScanRay<R>& ray, V<R> o, V<R> d, V<R> ud, R time) {
    ray.o = o;
    ray.d = d;
    ray.time = time;
    ray.basis = make_basis<float>(V<float>{(float)ud.x, (float)ud.y, (float)ud.z}, V<float>{(float)o.x, (float)o.y, (float)o.z});
    ray.ftime = (float)time;
    const double ddx = d.x, ddy = d.y, ddz = d.z;
    ray.inv_a2 = 1.0 / fm(ddz, ddz, fm(ddy, ddy, ddx * ddx));
    ray.tbest = (R)__builtin_inff();
    ray.ibest = -1;
    ray.cand[0] = ray.cand[1] = ray.cand[2] = ray.cand[3] = 0u;
    ray.ncand = 0u;
}
template <class R, int NR> __device__ __forceinline__ void scan_spheres(const DevScene<R>& sc, ScanRay<R> (&ray)[NR], R tmin) {
    scan_class<R, 0, NR>(sc, (int)sc.ns_pad, ray, tmin);
    scan_class<R, 1, NR>(sc, (int)sc.ny_pad, ray, tmin);
    scan_class<R, 2, NR>(sc, (int)sc.ng_pad, ray, tmin);
    narrow_flush<R, NR>(sc, ray, tmin);
    scan_triangles<R, NR>(sc, ray, tmin);
}

// ---- pieces of the shading step; the trace kernels use them through shade(), the known-answer entry calls them
// directly (the same instructions either way) ------------------------------------------------------------------------
// Background of a ray that hits nothing: ((1−t)·(1,1,1) + (0.5,0.7,1.0))·t, t = ½(unit(d).y + 1) — not a lerp,
// src/renderer.zig:124-125.
template <class R> __device__ __forceinline__ V<R> background(V<R> ud) {
    const R t = R(0.5) * (ud.y + R(1));
    const R w = R(1) - t;
    return {(w + R(0.5)) * t, (w + R(0.7)) * t, (w + R(1.0)) * t};
}
// Hit point, outward normal of a (moving) sphere and the front-face flip: src/geom.zig:63-65, src/hit.zig:25-41.
template <class R>
__device__ __forceinline__ void sphere_hit_record(typename VecOf<R>::type c, typename VecOf<R>::type v, V<R> o, V<R> d, R time,
                                                  R t, V<R>& pt, V<R>& nrm) {
    pt = {fm(d.x, t, o.x), fm(d.y, t, o.y), fm(d.z, t, o.z)};
    const V<R> cn{fm(v.x, time, c.x), fm(v.y, time, c.y), fm(v.z, time, c.z)};
    nrm = unit(V<R>{pt.x - cn.x, pt.y - cn.y, pt.z - cn.z});
}
template <class R> __device__ __forceinline__ bool face_forward(V<R> d, V<R>& nrm) { // Hit.init, src/hit.zig:33-36
    const bool front = dot3(nrm, d) < R(0);
    if (!front) nrm = neg(nrm);
    return front;
}
// Schlick's reflectance with pow(x, 5) as multiplies: src/material.zig:179-183.
template <class R> __device__ __forceinline__ R reflectance(R cosv, R eta) {
    R r0 = (R(1) - eta) / (R(1) + eta);
    r0 = r0 * r0;
    const R xx = R(1) - cosv;
    const R x2 = xx * xx;
    const R x5 = (x2 * x2) * xx;
    return fm(R(1) - r0, x5, r0);
}
template <class R> __device__ __forceinline__ V<R> reflect(V<R> d, V<R> nrm) { // src/material.zig:185-187
    const R k = R(2) * dot3(d, nrm);
    return {fm(-k, nrm.x, d.x), fm(-k, nrm.y, d.y), fm(-k, nrm.z, d.z)};
}
// src/material.zig:189-194 with cos = −ud·n handed in (the caller has it).  Exact arithmetic has 1 − |perp|² ≥ 0
// here (eta·sin ≤ 1); in f32 it rounds below 0 about once per 1e9 samples and the reference's bare sqrt (:192)
// would make the pixel NaN: clamped at 0.
template <class R> __device__ __forceinline__ V<R> refract(V<R> ud, V<R> nrm, R cosv, R eta) {
    const V<R> perp{fm(nrm.x, cosv, ud.x) * eta, fm(nrm.y, cosv, ud.y) * eta, fm(nrm.z, cosv, ud.z) * eta};
    const R sp = -sq(mx(R(1) - dot3(perp, perp), R(0)));
    return {fm(nrm.x, sp, perp.x), fm(nrm.y, sp, perp.y), fm(nrm.z, sp, perp.z)};
}
// `Material.scatter` (src/material.zig:73-160) without the texture lookup: the scattered direction, or false when
// a metal absorbs the ray.  `d` is the incoming direction as the ray holds it (unnormalised), `ud` = unit(d).
template <class R, class G>
__device__ __forceinline__ bool scatter_dir(uint32_t kind, uint32_t method, R param, R inv_param, G& g, V<R> d, V<R> ud, V<R> pt,
                                            V<R> nrm, bool front, V<R>& nd) {
    // Diffuse and fuzzy-metal lanes both start from a point in the unit sphere: drawn here, ONCE for the wave's lanes of
    // either kind (each lane consumes exactly the draws it would inside its own branch) — two rejection loops in two
    // divergent branches cost the wave twice its slowest lane.
    V<R> r{0, 0, 0};
    if (kind == 0u || (kind == 1u && param > R(0))) r = random_in_unit_sphere<R>(g);
    if (kind == 0u) { // diffuse, :77-101
        V<R> target;
        if (method == 2u) { // HEMISPHERE, :208-211
            if (!(dot3(r, nrm) > R(0))) r = neg(r);
            target = {pt.x + r.x, pt.y + r.y, pt.z + r.z};
        } else {
            if (method == 1u) r = unit(r); // UNIT_SPHERE_SURFACE
            target = {(pt.x + nrm.x) + r.x, (pt.y + nrm.y) + r.y, (pt.z + nrm.z) + r.z};
        }
        const R tol = (R)1e-8;
        if (ab(target.x) <= tol && ab(target.y) <= tol && ab(target.z) <= tol) target = nrm; // :85-86
        nd = {target.x - pt.x, target.y - pt.y, target.z - pt.z};
    } else if (kind == 1u) { // metallic, :108-131
        V<R> m = unit(reflect<R>(d, nrm));
        if (param > R(0)) {
            const V<R> ru = unit(r);
            const R f = param < R(1) ? param : R(1);
            m = {fm(ru.x, f, m.x), fm(ru.y, f, m.y), fm(ru.z, f, m.z)};
        }
        if (dot3(m, nrm) <= R(0)) return false; // absorbed
        nd = m;
    } else { // dielectric, :137-159
        const R eta = front ? inv_param : param;
        const R cosv = -dot3(ud, nrm);
        const R sinv = sq(fm(-cosv, cosv, R(1)));
        bool refl = eta * sinv > R(1);
        if (!refl) refl = reflectance<R>(cosv, eta) > uniform<R>(g); // the draw happens only when not TIR, :145
        nd = refl ? reflect<R>(d, nrm) : refract<R>(ud, nrm, cosv, eta);
    }
    return true;
}

// ---- shading of one segment: returns false when the path ends ------------------------------------
// On a miss adds thr ⊙ background to acc (src/renderer.zig:124-125); on absorption adds nothing.
// `ibest` is a POOL index; `ud` = unit(d) as the scan used it.
template <class R>
__device__ __forceinline__ bool shade(const DevScene<R>& sc, Pcg32& g, V<R>& o, V<R>& d, V<R> ud, R time, R tbest,
                                      int ibest, V<R>& thr, V<R>& acc) {
    typedef typename VecOf<R>::type r4;
    // The pass's three table pointers, read ONCE here: left to itself the compiler re-loads each kernel argument where it is
    // used (scalar registers are short), a scalar load + wait in front of every dependent table fetch of the pass.
    const r4* pool_gen = sc.sph_pool;
    const r4* mat_gen = sc.mat;
    const r4* tex_gen = sc.tex;
    asm volatile("" : "+s"(pool_gen), "+s"(mat_gen), "+s"(tex_gen));
    // (.. and named GLOBAL again: behind the empty asm the compiler no longer knows the address space and emits flat_load,
    //  which counts on both wait counters and resolves its aperture per lane)
    const RAYZ_GLOBAL r4* pool_p = (const RAYZ_GLOBAL r4*)pool_gen;
    const RAYZ_GLOBAL r4* mat_p = (const RAYZ_GLOBAL r4*)mat_gen;
    const RAYZ_GLOBAL r4* tex_p = (const RAYZ_GLOBAL r4*)tex_gen;
    if (ibest < 0) {
        const V<R> col = background<R>(ud);
        acc.x = acc.x + thr.x * col.x;
        acc.y = acc.y + thr.y * col.y;
        acc.z = acc.z + thr.z * col.z;
        return false;
    }
    // hit record: src/geom.zig:63-65, src/hit.zig:25-41 (spheres); geometric normal for triangles
    V<R> pt, nrm;
    uint32_t mat_idx;
    if ((uint32_t)ibest < sc.n_spheres) {
        const r4 q = pool_p[2 * ibest], w4 = pool_p[2 * ibest + 1];
        sphere_hit_record<R>(q, w4, o, d, time, tbest, pt, nrm);
        mat_idx = bits(w4.w);
    } else {
        const uint32_t ti = (uint32_t)ibest - sc.n_spheres;
        const r4 a = sc.tri[3 * ti], b = sc.tri[3 * ti + 1], c = sc.tri[3 * ti + 2];
        pt = {fm(d.x, tbest, o.x), fm(d.y, tbest, o.y), fm(d.z, tbest, o.z)};
        nrm = unit(cross3(V<R>{b.x, b.y, b.z}, V<R>{c.x, c.y, c.z}));
        mat_idx = bits(a.w);
    }
    const bool front = face_forward<R>(d, nrm);

    const r4 m = mat_p[mat_idx];
    const uint32_t kind = bits(m.x) & 0xffu, method = (bits(m.x) >> 8) & 0xffu, texture = bits(m.y);
    V<R> nd;
    if (!scatter_dir<R>(kind, method, m.z, m.w, g, d, ud, pt, nrm, front, nd)) return false;
    const V<R> att = kind == 2u ? V<R>{R(1), R(1), R(1)} : texture_value<R>(tex_p, texture, pt);
    thr = {thr.x * att.x, thr.y * att.y, thr.z * att.z};
    o = pt;
    d = nd;
    return true;
}

// ---- camera ray of path (px, py, s): src/camera.zig:59-90 ---------------------------------------
template <class R, class G>
__device__ __forceinline__ void camera_ray(const DevCamera<R>& cam, G& g, uint32_t px, uint32_t py, V<R>& o, V<R>& d,
                                           R& time) {
    const R x = (R)px + (uniform<R>(g) - R(0.5));
    const R y = (R)py + (uniform<R>(g) - R(0.5));
    o = {cam.from[0], cam.from[1], cam.from[2]};
    if (cam.defocus) {
        R vx = 0, vy = 0;
        for (int i = 0; i < kMaxRejectionTries; ++i) {
            vx = fm(uniform<R>(g), R(2), R(-1));
            vy = fm(uniform<R>(g), R(2), R(-1));
            if (fm(vy, vy, vx * vx) <= R(1)) break;
        }
        o.x = cam.from[0] + fm(cam.defv[0], vy, cam.defu[0] * vx);
        o.y = cam.from[1] + fm(cam.defv[1], vy, cam.defu[1] * vx);
        o.z = cam.from[2] + fm(cam.defv[2], vy, cam.defu[2] * vx);
    }
    d.x = (fm(cam.dv[0], y, cam.du[0] * x) + cam.pxo[0]) - o.x;
    d.y = (fm(cam.dv[1], y, cam.du[1] * x) + cam.pxo[1]) - o.y;
    d.z = (fm(cam.dv[2], y, cam.du[2] * x) + cam.pxo[2]) - o.z;
    time = uniform<R>(g);
}

// `getRay(px, py, null)` (src/camera.zig:59-77 with rng == null: the pixel's centre, the lens centre, time 0) — the form the
// reference's own "get ray" test calls (src/renderer.zig:129-149).  No path ever takes it (the kernels always draw); it exists for
// the known-answer entry, in the same operation order as camera_ray above.
template <class R> __device__ __forceinline__ void camera_ray_no_rng(const DevCamera<R>& cam, uint32_t px, uint32_t py, V<R>& o, V<R>& d, R& time) {
    const R x = (R)px, y = (R)py;
    o = {cam.from[0], cam.from[1], cam.from[2]};
    d.x = (fm(cam.dv[0], y, cam.du[0] * x) + cam.pxo[0]) - o.x;
    d.y = (fm(cam.dv[1], y, cam.du[1] * x) + cam.pxo[1]) - o.y;
    d.z = (fm(cam.dv[2], y, cam.du[2] * x) + cam.pxo[2]) - o.z;
    time = R(0);
}

// ---- per-path state of one of a lane's rays, and the steps every trace kernel shares ---------------------
template <class R> struct PathState {
    Pcg32 g;
    V<R> o, d, thr, acc;
    R time;
    uint32_t item, px, py, s_cur, s_end, seg;
    bool has_item, alive;
};
template <class R> __device__ __forceinline__ void path_init(PathState<R>& p) {
    p.g = Pcg32{0, 1};
    p.o = {R(0), R(0), R(0)};
    p.d = {R(0), R(0), R(1)};
    p.thr = {R(1), R(1), R(1)};
    p.acc = {R(0), R(0), R(0)};
    p.time = R(0);
    p.item = p.px = p.py = p.s_cur = p.s_end = p.seg = 0;
    p.has_item = p.alive = false;
}
// Samples of chunk k (the chunk schedule).  Every schedule starts with a run of equal chunks — all of it for a
// uniform one, all but the halving tail for the automatic one — and the host hands over that run's size and length: a lane
// that pops an item of the run computes its bounds; only the tail's few items read the table (two dependent loads in the
// refill of a pass whenever some lane pops — with 32-sample chunks that is every second pass).
template <class R> __device__ __forceinline__ void chunk_bounds(const TraceArgs<R>& A, uint32_t k, uint32_t& s0, uint32_t& s1) {
    if (k < A.chunk_n_uniform) {
        s0 = k * A.chunk_uniform;
        s1 = s0 + A.chunk_uniform;
    } else {
        s0 = A.chunk_start[k];
        s1 = A.chunk_start[k + 1];
    }
}

// ---- the work queue, as a wave sees it ------------------------------------------------------------------------
// Items come off one global counter.  A wave does not pay an atomic per refill (one word takes ≈88 atomics per µs on this
// chip: with the short items of a small scene the queue head, not the tracing, set the pace — 100 spheres ran at 3.8
// instead of 5.7 Gsamples/s at 64 spp): it reserves kQueueGrab consecutive items at a time and hands them to its lanes
// as they fall idle (ballot → prefix rank), touching the counter again only when the reserve runs out.  All fields are
// wave-uniform.  Which lane traces an item never affects the image.
// Which pixel and chunk a queue entry is: entry e = k · shard_pixels + i covers chunk k of the i-th pixel dealt.  Pixels are
// dealt in 8x8 TILES of the shard's local rows (the first A.tiled_pixels of them: whole tiles only, the rest row by row), so
// that the 64 items a wave grabs together are neighbours in both directions: their primary rays walk the same part of the
// tree (+1 … 2.6 % through the BVH, profiles/r03/tree/tile8.log).  The sums are stored by ROW-MAJOR pixel as before: `item`
// becomes k · shard_pixels + (local row · width + column), what resolve_kernel reads.  Returns k.
// (kTiled = false — the flat-list kernel, whose lanes all scan the same list: rows as they come.)
template <class R, bool kTiled> __device__ __forceinline__ uint32_t place_item(const TraceArgs<R>& A, uint32_t& item, uint32_t& px, uint32_t& py) {
    const uint32_t k = item / A.shard_pixels;
    uint32_t lp = item - k * A.shard_pixels;
    if (kTiled && lp < A.tiled_pixels) {
        const uint32_t tile = lp >> 6, w8 = A.width >> 3, trow = tile / w8, tcol = tile - trow * w8;
        lp = (trow * 8u + ((lp >> 3) & 7u)) * A.width + tcol * 8u + (lp & 7u);
        item = k * A.shard_pixels + lp;
    }
    const uint32_t lr = lp / A.width;
    px = lp - lr * A.width;
    const uint32_t tl = lr / A.tile_rows, within = lr - tl * A.tile_rows;
    py = (tl * A.shard_count + A.shard_index) * A.tile_rows + within;
    return k;
}

struct WaveQueue {
    uint32_t base = 0, count = 0; // the wave's reserve: items base .. base+count-1
    bool drained = false;         // the last reservation reached the end of the queue
};
constexpr uint32_t kQueueGrab = 64;
template <class R> __device__ __forceinline__ bool queue_empty(const WaveQueue& wq, const TraceArgs<R>& A) {
    return wq.drained && (wq.count == 0u || wq.base >= A.total_items);
}
// Every lane calls this; lanes with `need` get the next items (true + `item`) while the queue lasts.
template <class R>
__device__ __forceinline__ bool queue_pop(const TraceArgs<R>& A, WaveQueue& wq, uint32_t lane, bool need, uint32_t& item) {
    const unsigned long long need_mask = __ballot(need);
    if (need_mask == 0ull) return false;
    const uint32_t n_need = (uint32_t)__popcll(need_mask);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need_mask, 0u));
    unsigned long long mine;
    if (wq.count >= n_need) {
        mine = (unsigned long long)wq.base + rank;
        wq.base += n_need;
        wq.count -= n_need;
    } else { // the reserve is short: hand out what is left of it, then the head of a new reservation
        const uint32_t old_base = wq.base, old_count = wq.count, more = n_need - old_count;
        const uint32_t grab = more > A.queue_grab ? more : A.queue_grab;
        const int leader = __ffsll((long long)need_mask) - 1;
        unsigned long long nb = 0;
        if ((int)lane == leader) nb = atomicAdd(&A.counters[0], (unsigned long long)grab);
        nb = __shfl(nb, leader);
        mine = rank < old_count ? (unsigned long long)old_base + rank : nb + (rank - old_count);
        wq.base = (uint32_t)(nb + more); // (the host keeps total_items + every wave's last overshoot below 2^32)
        wq.count = grab - more;
        if (nb + grab >= (unsigned long long)A.total_items) wq.drained = true;
    }
    item = (uint32_t)mine;
    return need && mine < (unsigned long long)A.total_items;
}

// Retire a finished chunk, pop a new work item for an idle slot, start the slot's next path.
template <class R>
__device__ __forceinline__ void path_refill(PathState<R>& p, const TraceArgs<R>& A, uint32_t lane, WaveQueue& wq) {
    typedef typename VecOf<R>::type r4;
    if (!p.alive && p.has_item && p.s_cur == p.s_end) {
        A.partial[p.item] = r4{p.acc.x, p.acc.y, p.acc.z, R(0)};
        p.has_item = false;
    }
    {
        uint32_t got_item = 0;
        if (queue_pop<R>(A, wq, lane, !p.alive && !p.has_item && !queue_empty<R>(wq, A), got_item)) {
            p.item = got_item;
            p.has_item = true;
            const uint32_t k = place_item<R, false>(A, p.item, p.px, p.py);
            chunk_bounds<R>(A, k, p.s_cur, p.s_end);
            p.acc = {R(0), R(0), R(0)};
        }
    }
    if (!p.alive && p.has_item) { // start the next path of this slot's chunk
        const unsigned long long pixel_index = (unsigned long long)p.py * A.width + p.px;
        p.g.seed_path(A.seed, pixel_index * A.spp + p.s_cur);
        camera_ray<R>(A.cam, p.g, p.px, p.py, p.o, p.d, p.time);
        p.thr = {R(1), R(1), R(1)};
        p.seg = 0;
        p.s_cur++;
        p.alive = true;
    }
}

// ---- the persistent trace kernel (flat hit list) ----------------------------------------------------------
// Work item = (pixel of this shard, chunk k of the pixel's chunk schedule: samples chunk_start[k] .. chunk_start[k+1]).
// The schedule puts the big chunks first and ends in small ones (host: chunk_schedule), so the queue's tail is made
// of short items.  Items are numbered chunk-major so
// that the 64 lanes of a wave start on 64 neighbouring pixels.  Each of a lane's NR slots owns one item at a
// time, runs its paths one after the other, adds their radiance in sample order, and stores the chunk sum to
// partial[item]; resolve_kernel adds the chunk sums of a pixel in chunk order.  The summation tree is therefore
// fixed by the chunk schedule alone — a function of (width, height, spp, chunk_spp) — not by the launch, the grid
// size, NR or the number of GPUs.
#ifndef RAYZ_FLAT_WAVES_F64
#define RAYZ_FLAT_WAVES_F64 4
#endif
template <class R, int NR> constexpr int flat_waves() { return sizeof(R) == 8 ? RAYZ_FLAT_WAVES_F64 : NR == 1 ? 4 : 3; }
template <class R, int NR> __global__ __launch_bounds__(256, (flat_waves<R, NR>())) void trace_kernel(const TraceArgs<R> A) {
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    PathState<R> p[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) path_init<R>(p[r]);
    uint32_t nseg = 0;
    WaveQueue wq; // wave-uniform
#ifdef RAYZ_FLAT_PROFILE // measurement build only: wave time per phase (refill, ray setup, scan, narrow flush, shade)
    unsigned long long ft[5] = {0, 0, 0, 0, 0}, ft0 = __builtin_amdgcn_s_memtime(), fiters = 0;
#define RAYZ_FPROF(k) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ft[k] += now_ - ft0; ft0 = now_; }
#else
#define RAYZ_FPROF(k)
#endif

    for (;;) {
        // ---- retire finished chunks, refill idle slots ----
        bool any = false;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            path_refill<R>(p[r], A, lane, wq);
            any = any || p[r].alive;
        }
        if (__ballot(any) == 0ull) break; // queue drained and every slot idle: the wave is done

        RAYZ_FPROF(0)
        // ---- nearest hit (full EXEC; idle tail slots recompute their last ray, results unused) ----
        ScanRay<R> ray[NR];
        V<R> ud[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            ud[r] = unit(p[r].d);
            scan_begin<R, NR>(ray[r], p[r].o, p[r].d, ud[r], p[r].time);
        }
        RAYZ_FPROF(1)
#ifdef RAYZ_FLAT_PROFILE
        fiters++;
        scan_class<R, 0, NR>(A.sc, (int)A.sc.ns_pad, ray, A.tmin);
        scan_class<R, 1, NR>(A.sc, (int)A.sc.ny_pad, ray, A.tmin);
        scan_class<R, 2, NR>(A.sc, (int)A.sc.ng_pad, ray, A.tmin);
        RAYZ_FPROF(2)
        narrow_flush<R, NR>(A.sc, ray, A.tmin);
        scan_triangles<R, NR>(A.sc, ray, A.tmin);
        RAYZ_FPROF(3)
#else
        scan_spheres<R, NR>(A.sc, ray, A.tmin);
#endif

        // ---- shade ----
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (p[r].alive) {
                nseg++;
                p[r].seg++;
                bool cont = shade<R>(A.sc, p[r].g, p[r].o, p[r].d, ud[r], p[r].time, ray[r].tbest, ray[r].ibest, p[r].thr,
                                     p[r].acc);
                if (p[r].seg >= A.max_bounces) cont = false; // depth exhausted → black, src/renderer.zig:104-105
                p[r].alive = cont;
            }
        RAYZ_FPROF(4)
    }
#ifdef RAYZ_FLAT_PROFILE
    if (lane == 0) {
        for (int k = 0; k < 5; ++k) atomicAdd(&A.counters[4 + k], ft[k]);
        atomicAdd(&A.counters[9], fiters);
    }
#endif
    // ---- counters: one atomic per wave ----
    unsigned long long tot = nseg;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
    if (lane == 0) atomicAdd(&A.counters[1], tot);
}

// ---- BVH traversal (src/hit.zig:181-216) ---------------------------------------------------------------
// The reference recurses left-then-right with a shrinking tmax.  Here each lane walks the same tree with a small
// stack (LDS) and visits the NEARER child first; the nearest hit and its tie rule do not depend on visiting
// order, so the result is the reference's.  Per-lane state of one query:
// Threads per workgroup of the one-path BVH kernel = stride of its LDS stacks.  1,024 = ONE workgroup per CU (16 waves, 4
// per SIMD as before), so that ONE copy of the tree's top serves the whole CU and can take all the LDS the stacks leave:
// ≈1,300 inner-node records for config 3 instead of 256 per 256-thread workgroup.  The walk's shared bottleneck is the
// vector-memory address pipe (TA_TA_BUSY 84 %: every step gathers 64 B per lane, 4 divergent dwordx4 loads); a node read
// from LDS does not go through it.  config 3: 42 % of the lane-steps are served by a 256-record top, 55 % by 1,280
// (profiles/r03/lds_top/): 3,554 instead of 3,366 Msamples/s; same speed as 256-thread workgroups at equal top size.
#ifndef RAYZ_BVH_WG
#define RAYZ_BVH_WG 1024
#endif
constexpr uint32_t kBvhWg = RAYZ_BVH_WG;
#ifdef RAYZ_EXPERIMENTS
constexpr uint32_t kBvh2Wg = 256; // (the retired two-path kernel needs 168 VGPRs: 256-thread workgroups, 3 per CU)
#endif
// LDS a BVH workgroup may ask for: hipFuncSetAttribute(MaxDynamicSharedMemorySize) refuses requests near the CU's 160 KB
// (151,552 B accepted, 155,648 B refused when probed in round 2; rounds 3-4 ran every BVH render with 153,600 B), so the top of
// the tree is sized for 150 KB in all.  A stack that refuses that is not fatal: render_impl retries with a shorter prefix
// of the top, 8 KB at a time (tests/test_bvh.py exercises the retry through RAYZ_DEBUG_LDS_PAD).
constexpr size_t kBvhLdsBudget = 150 * 1024;
// .. of which this much, after the stacks, holds the oversized hittables' records (the filter record's three words in R and
// the f64 sphere record, per entry; at most 4 descriptors of 2): the per-segment set-up reads them from LDS, not through
// two dependent trips to the L2
constexpr uint32_t kBvhBigEntries = 8;
template <class R> constexpr uint32_t bvh_big_entry_bytes() { return 2u * (uint32_t)sizeof(d4) + 4u * (uint32_t)sizeof(typename VecOf<R>::type); } // (f64 record first: 32-byte aligned)
constexpr size_t kBvhBigLdsBytes = kBvhBigEntries * (2 * 32 + 4 * 32);
constexpr int kBvhStackDepth = 32;            // ≥ tree depth: the halving tree's ceil(log2(n / 2)) + 1 (n ≤ 2^27) + the SAH build's
                                              // 4 extra levels (bvh_build.hpp); (32 + 3) rows of 4 KB still fit kBvhLdsBudget
constexpr uint32_t kBvhDone = 0x7fffffffu;     // cursor: nothing left to visit (positive: not a leaf reference)
constexpr uint32_t kBvhLeafFlag = 0x80000000u; // child reference / stack entry is a leaf descriptor, not an inner index

// Per-lane choices under a wave mask held in scalar registers (bit i = lane i): one vector instruction each.
__device__ __forceinline__ uint32_t mask_select(unsigned long long m, uint32_t if_set, uint32_t if_clear) {
    uint32_t r;
    asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
    return r;
}
__device__ __forceinline__ uint32_t mask_add(uint32_t x, unsigned long long m) { // x + (the lane's bit of m)
    uint32_t r;
    unsigned long long carry;
    asm volatile("v_addc_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(r), "=s"(carry) : "v"(x), "s"(m));
    return r;
}
__device__ __forceinline__ uint32_t mask_sub(uint32_t x, unsigned long long m) { // x − (the lane's bit of m)
    uint32_t r;
    unsigned long long borrow;
    asm volatile("v_subb_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(r), "=s"(borrow) : "v"(x), "s"(m));
    return r;
}

// max / min of a computed value x and a bound the caller knows is not NaN: the bare instruction.
__device__ __forceinline__ float max_bound(float x, float bound) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(bound));
    return r;
}
// .. the same with a WAVE-UNIFORM bound read straight from a scalar register (as a "v" operand the loop pays a v_mov per step)
__device__ __forceinline__ float max_bound_uniform(float x, float bound) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "s"(bound), "v"(x));
    return r;
}
__device__ __forceinline__ float min_bound(float x, float bound) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(bound));
    return r;
}

template <class R> struct LeafBasis;
template <> struct LeafBasis<float> {
    RayBasis<float> b;
    __device__ __forceinline__ void make(V<float> ud, V<float> o) { b = make_basis<float>(ud, o); }
    __device__ __forceinline__ RayBasis<float> get(V<float>, V<float>) const { return b; }
};
template <> struct LeafBasis<double> {
    __device__ __forceinline__ void make(V<double>, V<double>) {}
    __device__ __forceinline__ RayBasis<float> get(V<double> ud, V<double> o) const {
        return make_basis<float>(V<float>{(float)ud.x, (float)ud.y, (float)ud.z}, V<float>{(float)o.x, (float)o.y, (float)o.z});
    }
};
template <class R> struct BvhQuery {
    // the slab test runs in f32 whatever R is: it only culls, and stays conservative under the conversion (§4.8)
    V<float> qa;    // cell_k / d_k per component (1 / d_k capped, bvh_begin): a slab distance is ONE fma on the plane's
    V<float> qb;    // 16-bit grid index i, t = fm(float(i), qa, qb), with qb = (glo_k − o_k) / d_k
    float tb32;     // tbest as the box steps see it: rounded UP to f32 (refreshed before every run of box steps)
    double inv_a2;  // 1 / (d·d) in f64 for the narrow phase
    R tbest;
    int ibest;      // hittable index
    uint32_t cur;   // what the lane holds: an inner node to visit (< kBvhDone), a parked leaf (kBvhLeafFlag | descriptor),
                    // or kBvhDone — then the stack is empty as well and the walk is complete
    uint32_t sp;    // index of the stack's top entry; entry 0 is a kBvhDone sentinel, so popping an empty stack ends the walk
    uint32_t top;   // the stack's top entry, stack[sp], read AHEAD: every step (and bvh_pop) ends by loading the entry that
                    // will be the top at the next pop, so that a pop takes a register instead of waiting for an LDS round
                    // trip between the box tests and the next node fetch (the walk is latency-bound per wave: 49 % of a
                    // wave's cycles are waits, profiles/r02)
    // the reject test's orthonormal pair, made once per segment by the set-up instead of once per leaf
    // phase (a square root and a division, ≈35 vector instructions, ≈3 times per segment at ≈26 lanes).  R = float only: the
    // f64 kernel sits at 128 VGPRs and keeps making it at the leaves (LeafBasis<double> is empty).
    LeafBasis<R> lb;
};

// The f32 slab test carries NO slack of its own (round 3): the boxes the device holds are padded on the host by
// E = kBoxPadUlps·u·max(S, B) per side (u = 2^-24; S = bound on every ray origin, B = largest box coordinate) before
// they are rounded outward to f32 — see bvh_box_hit for why that makes the test conservative.
constexpr double kBoxPadUlps = 16.0;
// a tree gets 32-byte records of 16-bit plane indices instead of 64-byte records of f32 planes when it has more than this
// many times the inner nodes the LDS top would hold as f32 planes (DevScene::bvh_nodes; measured in profiles/r03/lds_top)
constexpr size_t kQuantizeAboveTops = 12;
__device__ __forceinline__ float round_up_f32(float v) { return v; }
__device__ __forceinline__ float round_up_f32(double v) { return __double2float_ru(v); }
__device__ __forceinline__ float round_down_f32(float v) { return v; }
__device__ __forceinline__ float round_down_f32(double v) { return __double2float_rd(v); }
// 1 / d_k, held to ±K = ±2^64.  A direction component of 0 (or so small that its reciprocal overflows) must not reach
// the slab test as ±inf: fm(plane, ±inf, −o·(±inf)) is −inf for one plane of a box that straddles 0 and NaN for the
// other, and max(−inf, NaN) = −inf then culls a box the ray lies inside.  (Such directions are not exotic: a diffuse
// scatter at |p_k| = 50 returns exactly 0 in one component about once in 10^5 bounces, when the offset is absorbed by the
// rounding of p_k + r_k.)  With the clamp the ray is treated as one whose component is 1 / K: finite distances of the
// right sign (the host's padding of the boxes covers their rounding: bvh_box_hit).
constexpr float kInvCap = 0x1p64f;
__device__ __forceinline__ float capped_inverse(float dk) { // (one v_med3_f32 behind the division)
    return __builtin_amdgcn_fmed3f(1.0f / dk, -kInvCap, kInvCap);
}
// (kGrid = false — a kernel whose records hold f32 planes: the grid is origin 0, cell 1, so qa = inv and qb = noi exactly;
//  the six operations and the grid's kernel arguments are skipped)
template <class R, bool kGrid = true, class SC>
__device__ __forceinline__ void bvh_begin(BvhQuery<R>& q, const SC& sc, V<R> o, V<R> d, V<R> ud, uint32_t n_inner) {
    q.lb.make(ud, o);
    V<float> inv;
    if constexpr (sizeof(R) == 8) { // d_k narrowed to f32 first; a component beyond f32's range would make inv 0: held to ±2^100
        inv = {capped_inverse(__builtin_amdgcn_fmed3f((float)d.x, -0x1p100f, 0x1p100f)),
               capped_inverse(__builtin_amdgcn_fmed3f((float)d.y, -0x1p100f, 0x1p100f)),
               capped_inverse(__builtin_amdgcn_fmed3f((float)d.z, -0x1p100f, 0x1p100f))};
    } else {
        inv = {capped_inverse(d.x), capped_inverse(d.y), capped_inverse(d.z)};
    }
    // −o·inv with the origin at full precision, rounded once; then the grid folded in: a plane with index i lies at
    // glo + i·cell, so its distance (glo + i·cell − o)·inv is fm(i, cell·inv, fm(glo, inv, −o·inv))
    const V<float> noi{(float)(-(o.x * (R)inv.x)), (float)(-(o.y * (R)inv.y)), (float)(-(o.z * (R)inv.z))};
    if constexpr (kGrid) {
        q.qa = {sc.bvh_cell[0] * inv.x, sc.bvh_cell[1] * inv.y, sc.bvh_cell[2] * inv.z};
        q.qb = {fm(sc.bvh_glo[0], inv.x, noi.x), fm(sc.bvh_glo[1], inv.y, noi.y), fm(sc.bvh_glo[2], inv.z, noi.z)};
    } else {
        q.qa = inv;
        q.qb = noi;
    }
    const double ddx = d.x, ddy = d.y, ddz = d.z;
    q.inv_a2 = 1.0 / fm(ddz, ddz, fm(ddy, ddy, ddx * ddx));
    q.tbest = (R)__builtin_inff();
    q.tb32 = __builtin_inff();
    q.ibest = -1;
    q.cur = n_inner ? 0u : kBvhDone;
    q.sp = 0;
    q.top = kBvhDone; // = stack[0], the sentinel
}

// Slab test (AABB.hit, src/hit.zig:70-98) of one child box held as three words of 16-bit grid indices (lower | upper << 16
// per axis): each plane distance is one conversion + one fma, t = fm(float(index), qa_k, qb_k), and the test is bare:
// t1 ≥ t0.  It is CONSERVATIVE — rounding never culls a box the f64 narrow phase would hit — because the BOX carries the
// slack, not the comparison.  With inv = (1/d)(1+e1), noi = −o·inv(1+e2), qa = cell·inv(1+e4), qb = (glo·inv + noi)(1+e5)
// and the fma's own rounding e3, the distance computed for index i is the EXACT distance of the true ray to a plane p' with
//     p' − o = (1+e1)(1+e3)·[i·cell(1+e4) + (glo − o − o·e2)(1+e5)]
//     |p' − p| ≤ (|e1|+|e3|)|p − o| + |e4|·i·cell + |e5||glo − o| + |e2||o|   (+ second order),   p = glo + i·cell
// every |e| ≤ u = 2^-24 except |e1| ≤ 2u for R = double (d_k narrowed to f32, then divided): |p' − p| ≤ 4u(B + S) + u·X + u(B + S)
// + u·S < 7u·(max(S, B) + X) for box coordinates |p|, |glo| ≤ B, origins |o| ≤ S and a grid of extent X.  The host picks
// each lower index as the LARGEST whose plane lies at or below the true plane − E, each upper one as the SMALLEST at or above
// the true plane + E, E = 16u·(max(S, B) + X) (rayz_hip.hip: quantize_box), so each computed slab interval contains the true
// box's exact one, for either sign of d_k.  A direction component of 0 reaches here as ±1/K (bvh_begin): huge finite
// distances whose SIGN is right as long as the origin is not within rounding (< E) of the plane — and a true ray that
// runs parallel to a slab is inside it only if it is strictly between the TRUE planes, i.e. at least E inside the held ones.
// `tmin` is the caller's tmin rounded DOWN to f32, q.tb32 tbest rounded UP.  Returns the entry distance through `t0`.
template <class R, bool kUniformTmin>
__device__ __forceinline__ bool bvh_box_hit_planes(V<float> lo, V<float> hi, const BvhQuery<R>& q, float tmin, float& t0);
__device__ __forceinline__ float plane_lo(uint32_t w) { // float(w & 0xffff): one v_cvt_f32_u32 with a 16-bit source select
    float r;
    asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(r) : "v"(w));
    return r;
}
__device__ __forceinline__ float plane_hi(uint32_t w) { // float(w >> 16)
    float r;
    asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r) : "v"(w));
    return r;
}
template <class R, bool kUniformTmin = false>
__device__ __forceinline__ bool bvh_box_hit(uint32_t wx, uint32_t wy, uint32_t wz, const BvhQuery<R>& q, float tmin, float& t0) {
    return bvh_box_hit_planes<R, kUniformTmin>(V<float>{plane_lo(wx), plane_lo(wy), plane_lo(wz)},
                                               V<float>{plane_hi(wx), plane_hi(wy), plane_hi(wz)}, q, tmin, t0);
}
// .. of a box held as f32 planes (glo = 0, cell = 1: qa = 1/d, qb = −o/d, and a plane is its own "index")
template <class R, bool kUniformTmin>
__device__ __forceinline__ bool bvh_box_hit_planes(V<float> lo, V<float> hi, const BvhQuery<R>& q, float tmin, float& t0) {
    const float ax = fm(lo.x, q.qa.x, q.qb.x), bx = fm(hi.x, q.qa.x, q.qb.x);
    const float ay = fm(lo.y, q.qa.y, q.qb.y), by = fm(hi.y, q.qa.y, q.qb.y);
    const float az = fm(lo.z, q.qa.z, q.qb.z), bz = fm(hi.z, q.qa.z, q.qb.z);
    // (tmin and tbest are never NaN: max_bound / min_bound spare the canonicalising copy fmax / fmin would put in front of
    // every use — two vector instructions per step)
    // (kUniformTmin: the trace kernels' tmin is the launch's, the same for every lane; the known-answer kernel's is per record)
    t0 = mx(mx(mn(ax, bx), mn(ay, by)), kUniformTmin ? max_bound_uniform(mn(az, bz), tmin) : max_bound(mn(az, bz), tmin));
    float tb;
    if constexpr (sizeof(R) == 4) tb = q.tbest; else tb = q.tb32; // (f32: tbest itself — no second register)
    const float t1 = mn(mn(mx(ax, bx), mx(ay, by)), min_bound(mx(az, bz), tb));
    return t1 >= t0;
}

// The node array's base as scalar registers of its own (the step's global loads take it as their SGPR base: read straight
// out of the kernel-argument block under scalar-register pressure, hipcc hands the inline assembly a VGPR pair).
__device__ __forceinline__ const f4* scalar_base(const f4* p) {
    const unsigned long long nb = (unsigned long long)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)nb), hi = __builtin_amdgcn_readfirstlane((uint32_t)(nb >> 32));
    return (const f4*)(uintptr_t)((unsigned long long)lo | ((unsigned long long)hi << 32));
}
// Phase N — one step of a lane that holds an inner node: fetch the node's record, slab-test both children, push the
// farther hit child (inner index or flagged leaf descriptor alike) and take the nearer one; with nothing hit, take the
// top of the stack instead.  Whatever the lane then holds says what it does next: an inner node → another step, a leaf →
// parked for phase L, kBvhDone (the sentinel under the stack) → walk complete.  Branch-free: the push lands above the
// top when there is nothing to push, the pop is a read every stepping lane makes.  `stack` is this lane's column of
// the workgroup's LDS stack (entry s at stack[s * 256]).
template <class R, uint32_t WG, bool QUANT>
__device__ __forceinline__ void bvh_node_step(const DevScene<R>& sc, const f4* nodes_base, BvhQuery<R>& q, float tmin, uint32_t* stack
#ifdef RAYZ_BVH_PROFILE
                                              , unsigned long long& g_fetch_ticks
#endif
) {
#ifdef RAYZ_BVH_PROFILE
    const unsigned long long tl0 = __builtin_amdgcn_s_memtime();
#endif
    // Both homes of a node — the LDS copy of the tree's top (at LDS address 0), global memory for the rest — are read
    // from the SAME 32-bit offset (the reference itself): the lanes of either kind take turns under exec, into the same
    // registers.  Two vector instructions (a compare, a shift) instead of the nine a flat-address select costs.
    // (an inner reference IS its record's byte offset, index << 6 or << 5: the host keeps the node count below 2^25;
    //  sc.bvh_top is a byte count likewise — no shift, no second register)
    const unsigned long long in_top = __ballot(q.cur < sc.bvh_top);
    const uint32_t off = q.cur;
    unsigned long long saved;
    float tl, tr;
    bool hl, hr;
    uint32_t l, r; // each child's reference: an inner record's offset, or kBvhLeafFlag | leaf descriptor
    if constexpr (QUANT) {
        u4 nl, nr; // the two children: {x, y, z plane words, reference}
        asm volatile("s_mov_b64 %[sv], exec\n\t"
                     "s_and_b64 exec, %[sv], %[mt]\n\t" // (SCC = some lane: an empty turn is skipped — the memory
                     "s_cbranch_scc0 1f\n\t"            //  pipes would still spend their cycles on it)
                     "ds_read_b128 %[n0], %[off]\n\t"
                     "ds_read_b128 %[n1], %[off] offset:16\n"
                     "1:\n\t"
                     "s_andn2_b64 exec, %[sv], %[mt]\n\t"
                     "s_cbranch_scc0 2f\n\t"
                     "global_load_dwordx4 %[n0], %[off], %[base]\n\t"
                     "global_load_dwordx4 %[n1], %[off], %[base] offset:16\n"
                     "2:\n\t"
                     "s_mov_b64 exec, %[sv]\n\t"
                     "s_waitcnt vmcnt(1) lgkmcnt(1)" // the LEFT child's load (and the stack top read ahead before it)
                     : [n0] "=&v"(nl), [n1] "=&v"(nr), [sv] "=&s"(saved)
                     : [off] "v"(off), [mt] "s"(in_top), [base] "s"(nodes_base)
                     : "memory", "scc");
        hl = bvh_box_hit<R, true>(nl.x, nl.y, nl.z, q, tmin, tl);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(nr) : "v"(tl) : "memory"); // (see the f32 form below)
        hr = bvh_box_hit<R, true>(nr.x, nr.y, nr.z, q, tmin, tr);
        l = nl.w, r = nr.w;
    } else {
        f4 llo, lhi, rlo, rhi;
        asm volatile("s_mov_b64 %[sv], exec\n\t"
                     "s_and_b64 exec, %[sv], %[mt]\n\t"
                     "s_cbranch_scc0 1f\n\t"
                     "ds_read_b128 %[n0], %[off]\n\t"
                     "ds_read_b128 %[n1], %[off] offset:16\n\t"
                     "ds_read_b128 %[n2], %[off] offset:32\n\t"
                     "ds_read_b128 %[n3], %[off] offset:48\n"
                     "1:\n\t"
                     "s_andn2_b64 exec, %[sv], %[mt]\n\t"
                     "s_cbranch_scc0 2f\n\t"
                     "global_load_dwordx4 %[n0], %[off], %[base]\n\t"
                     "global_load_dwordx4 %[n1], %[off], %[base] offset:16\n\t"
                     "global_load_dwordx4 %[n2], %[off], %[base] offset:32\n\t"
                     "global_load_dwordx4 %[n3], %[off], %[base] offset:48\n"
                     "2:\n\t"
                     "s_mov_b64 exec, %[sv]\n\t"
                     "s_waitcnt vmcnt(2) lgkmcnt(2)" // the LEFT child's two loads (and the stack top read ahead before them)
                     : [n0] "=&v"(llo), [n1] "=&v"(lhi), [n2] "=&v"(rlo), [n3] "=&v"(rhi), [sv] "=&s"(saved)
                     : [off] "v"(off), [mt] "s"(in_top), [base] "s"(nodes_base)
                     : "memory", "scc");
        hl = bvh_box_hit_planes<R, true>(V<float>{llo.x, llo.y, llo.z}, V<float>{lhi.x, lhi.y, lhi.z}, q, tmin, tl);
        // .. and the left box is tested while the right child's loads are still on their way (they return in order; the
        // empty dependence on tl keeps the left test above this wait, the "+v" keep the compiler's hands off the registers
        // in flight): +1.7 % config 3, profiles/r03/bvh_step/split_wait.log
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(rlo), "+v"(rhi) : "v"(tl) : "memory");
        hr = bvh_box_hit_planes<R, true>(V<float>{rlo.x, rlo.y, rlo.z}, V<float>{rhi.x, rhi.y, rhi.z}, q, tmin, tr);
        l = bits(llo.w), r = bits(rlo.w);
    }
#ifdef RAYZ_BVH_PROFILE // time from issuing the node fetch to having it (the wave's own view), accumulated in g_prof_fetch
    g_fetch_ticks += __builtin_amdgcn_s_memtime() - tl0;
#endif
    // Which child was hit decides everything below, so the decisions live as WAVE MASKS in scalar registers (the boolean
    // algebra runs on the scalar unit, which has slack; the vector unit, which has none, spends one v_cndmask per choice):
    const unsigned long long ml = __ballot(hl), mr = __ballot(hr), mlt = __ballot(tr < tl);
    const unsigned long long swap = mr & (~ml | mlt), both = ml & mr, none = ~(ml | mr);
    // Nearer child first (a child that was not hit may ride along as `near` or `far`: `none` and `both` decide what is used)
    const uint32_t near = mask_select(swap, r, l);
    uint32_t far = mask_select(swap, l, r);
    q.cur = mask_select(none, q.top, near); // nothing hit => nothing pushed: the entry read ahead IS the top
    // The push goes AFTER that select in program order (the empty asm makes `far` look computed from q.cur): the compiler
    // guards the select's read of q.top — the previous step's read-ahead, long since back behind the node fetch — with
    // s_waitcnt lgkmcnt(0), and a store issued before it would put an LDS round trip into every step's dependent chain.
    asm volatile("" : "+v"(far) : "v"(q.cur));
    stack[WG * (q.sp + 1u)] = far; // lands above the top unless both were hit
    const uint32_t sp = mask_add(q.sp, both);
    q.sp = mask_sub(sp, none); // popping the sentinel leaves sp at −1: the lane holds kBvhDone and steps no more until
                               // bvh_begin (its read-ahead below lands in the guard row under the stack)
    q.top = stack[WG * q.sp]; // read ahead for the NEXT pop — after the store above (LDS keeps a wave's order), consumed
                               // one step later, behind that step's node fetch: its latency is off the critical path
}

// The next entry of a lane that is done with its parked leaf.
template <class R, uint32_t WG> __device__ __forceinline__ void bvh_pop(BvhQuery<R>& q, const uint32_t* stack) {
    q.cur = q.top;
    q.sp -= 1u;
    q.top = stack[WG * q.sp];
}

// The reject test of a leaf entry (general velocity form of DESIGN.md §4.3, f32 for both precisions): the ONE place it is
// written — the BVH kernels' leaves, the oversized hittables and the known-answer entry all call it.
__device__ __forceinline__ bool leaf_reject_test(const RayBasis<float>& b, float ft, V<float> c, V<float> v, float r2_padded) {
    const float p1 = fm(v.z, ft * b.e1z, fm(v.x, ft * b.e1x, basis_p1<float>(b, c.x, c.z)));
    const float p2 = fm(v.z, ft * b.e2z, fm(v.y, ft * b.e2y, fm(v.x, ft * b.e2x, basis_p2<float>(b, c.x, c.y, c.z))));
    return sphere_candidate<float>(p1, p2, r2_padded);
}

// Phase L — entry k of a parked leaf: a triangle is decided here (R arithmetic only); a sphere gets the reject
// test in R and, if its line meets the sphere, is parked as a candidate (slot + 1) for phase C.
// (a sphere's pool index rides in its record's last word, v.w: phase C takes it from there — see bvh_leaf_pair)
template <class R>
__device__ __forceinline__ uint32_t bvh_leaf_eval(const DevScene<R>& sc, BvhQuery<R>& q, uint32_t leaf, uint32_t k, typename VecOf<R>::type c,
                                                  typename VecOf<R>::type v, V<R> o, V<R> d, V<R> ud, R time, R tmin,
                                                  const typename VecOf<R>::type* third_word = nullptr) {
    typedef typename VecOf<R>::type r4;
    const uint32_t slot = (leaf >> 4) + k;
    if ((leaf >> (2u + k)) & 1u) { // triangle
        const r4 e2 = third_word ? *third_word : sc.bvh_leaf[(size_t)sc.bvh_leaf_stride * slot + 2];
        const V<R> v0{c.x, c.y, c.z}, e1{v.x, v.y, v.z}, ee2{e2.x, e2.y, e2.z};
        tri_accept<R>(tri_filter<R>(v0, e1, ee2, o, d), v0, e1, ee2, o, d, tmin, (int)bits(c.w), q.tbest, q.ibest);
        return 0u;
    }
    // the flat list's reject test: in f32 for both precisions, on the ray narrowed to f32 (hipcc shares the basis between
    // both entries of a leaf); c.w = the f32 padded square
    const RayBasis<float> b = q.lb.get(ud, o);
    return leaf_reject_test(b, (float)time, V<float>{(float)c.x, (float)c.y, (float)c.z}, V<float>{(float)v.x, (float)v.y, (float)v.z}, (float)c.w)
               ? slot + 1u : 0u;
}
template <class R>
__device__ __forceinline__ uint32_t bvh_leaf_entry(const DevScene<R>& sc, BvhQuery<R>& q, uint32_t leaf, uint32_t k, V<R> o,
                                                   V<R> d, V<R> ud, R time, R tmin) {
    typedef typename VecOf<R>::type r4;
    const r4* rec = sc.bvh_leaf + (size_t)sc.bvh_leaf_stride * ((leaf >> 4) + k);
    return bvh_leaf_eval<R>(sc, q, leaf, k, rec[0], rec[1], o, d, ud, time, tmin);
}
// Both entries of a parked leaf (phase L): their records are FETCHED together — entry 1's loads do not wait for entry 0's
// test (two dependent memory round trips per leaf phase otherwise); a leaf of one repeats entry 0's address, its second
// result is dropped.  R = float only: the f64 kernel has no eight registers to spare and tests one entry after the other.
// The candidates' POOL indices (the tie rule's and the hit record's key) come back too: they ride in the records just
// fetched (v.w), and phase C would otherwise fetch them again — after its roots, when a root is accepted: the compiler
// sinks that load into the branch, a second memory round trip inside the phase.
template <class R>
__device__ __forceinline__ void bvh_leaf_pair(const DevScene<R>& sc, BvhQuery<R>& q, uint32_t leaf, V<R> o, V<R> d, V<R> ud, R time, R tmin,
                                              uint32_t& cand0, uint32_t& cand1, int& pool0, int& pool1) {
    typedef typename VecOf<R>::type r4;
    const bool two = (leaf & 3u) > 1u;
    if constexpr (sizeof(R) == 4) {
        const r4* rec = sc.bvh_leaf + (size_t)sc.bvh_leaf_stride * (leaf >> 4);
        const r4* rec1 = rec + (two ? sc.bvh_leaf_stride : 0u);
        const r4 c0 = rec[0], v0 = rec[1], c1 = rec1[0], v1 = rec1[1];
        cand0 = bvh_leaf_eval<R>(sc, q, leaf, 0u, c0, v0, o, d, ud, time, tmin);
        pool0 = (int)bits(v0.w);
        if (two) cand1 = bvh_leaf_eval<R>(sc, q, leaf, 1u, c1, v1, o, d, ud, time, tmin);
        pool1 = (int)bits(v1.w);
    } else {
        const r4* rec = sc.bvh_leaf + (size_t)sc.bvh_leaf_stride * (leaf >> 4);
        const r4 c0 = rec[0], v0 = rec[1];
        cand0 = bvh_leaf_eval<R>(sc, q, leaf, 0u, c0, v0, o, d, ud, time, tmin);
        pool0 = (int)bits(v0.w);
        if (two) {
            const r4 c1 = rec[sc.bvh_leaf_stride], v1 = rec[sc.bvh_leaf_stride + 1u];
            cand1 = bvh_leaf_eval<R>(sc, q, leaf, 1u, c1, v1, o, d, ud, time, tmin);
            pool1 = (int)bits(v1.w);
        }
    }
}

// Phase C — the f64 quadratic of a parked sphere candidate (same arithmetic as narrow_phase()).
template <class R>
__device__ __forceinline__ void bvh_candidate(const DevScene<R>& sc, BvhQuery<R>& q, uint32_t slot, int pool, V<R> o, V<R> d, R time,
                                              R tmin);
template <class R>
__device__ __forceinline__ void bvh_candidate(const DevScene<R>& sc, BvhQuery<R>& q, uint32_t slot, V<R> o, V<R> d, R time,
                                              R tmin) {
    bvh_candidate<R>(sc, q, slot, (int)bits(sc.bvh_leaf[(size_t)sc.bvh_leaf_stride * slot + 1].w), o, d, time, tmin);
}
template <class R>
__device__ __forceinline__ void bvh_candidate_eval(BvhQuery<R>& q, const d4 c2, const d4 v2, int pool, V<R> o, V<R> d, R time, R tmin);
template <class R>
__device__ __forceinline__ void bvh_candidate(const DevScene<R>& sc, BvhQuery<R>& q, uint32_t slot, int pool, V<R> o, V<R> d, R time,
                                              R tmin) {
    bvh_candidate_eval<R>(q, sc.bvh_sph64[2 * slot], sc.bvh_sph64[2 * slot + 1], pool, o, d, time, tmin);
}
template <class R>
__device__ __forceinline__ void bvh_candidate_eval(BvhQuery<R>& q, const d4 c2, const d4 v2, int pool, V<R> o, V<R> d, R time, R tmin) {
    const double dx = d.x, dy = d.y, dz = d.z, tm = time;
    const double qx = fm(v2.x, tm, c2.x - (double)o.x), qy = fm(v2.y, tm, c2.y - (double)o.y),
                 qz = fm(v2.z, tm, c2.z - (double)o.z);
    const double a2 = fm(dz, dz, fm(dy, dy, dx * dx));
    const double hb2 = fm(dz, qz, fm(dy, qy, dx * qx));
    const double cc2 = fm(qz, qz, fm(qy, qy, fm(qx, qx, -c2.w)));
    const double disc2 = fm(-a2, cc2, hb2 * hb2);
    if (disc2 >= 0.0) {
        const double rt = __builtin_sqrt(disc2);
        const R t1r = (R)((hb2 - rt) * q.inv_a2), t2r = (R)((hb2 + rt) * q.inv_a2);
        const R t = t1r >= tmin ? t1r : t2r;
        accept_root<R>(t, pool, tmin, q.tbest, q.ibest);
    }
}

// ---- persistent trace kernel, BVH traversal ------------------------------------------------------
// Same work items, queue and summation tree as trace_kernel.  Traversal is per lane, so the wave's lanes want
// different code at different times; each ROUND therefore runs three convergent phases instead of one divergent
// step: (N) box steps until most lanes are parked at a leaf, (L) the parked leaves' reject tests together,
// (C) the parked sphere candidates' f64 roots together.  The nearest hit (with its tie rule) does not depend on
// the order in which hittables are examined, so this scheduling changes no result — only how much pruning the
// shrinking tbest achieves.  When too few lanes still have nodes to visit, the finished lanes are shaded and
// refilled (ray regeneration) while the others keep their query state and resume.
#ifndef RAYZ_STAT_SPILL
#define RAYZ_STAT_SPILL 0x7fffffffu // tests build with a small value to exercise the spill
#endif
constexpr int kBvhKeepActive = 20;   // rounds continue while at least this many lanes still walk
constexpr int kBvhKeepStepping = 18; // phase N continues while at least this many lanes can take a box step
                                     // (defaults; TraceArgs::bvh_keep carries the values in use)

// Waves per SIMD the register allocation aims at: 4 (128 VGPRs, nothing spilled).  At 5 (96 VGPRs) the kernel spills 54
// VGPRs into the box-step loop and is 8 % slower; the box steps are issue-bound, not latency-bound (profiles/r02), so the
// fifth wave buys nothing.
#ifndef RAYZ_BVH_WAVES
#define RAYZ_BVH_WAVES 4
#endif
#ifndef RAYZ_BVH_WAVES_F64 // the f64 kernel fits 128 VGPRs too since its box walk and its reject tests run in f32 (it needed
#define RAYZ_BVH_WAVES_F64 4 // 151 and ran 3 waves per SIMD before: +13 % from the fourth)
#endif
template <class R> constexpr int bvh_waves() { return sizeof(R) == 8 ? RAYZ_BVH_WAVES_F64 : RAYZ_BVH_WAVES; }
template <class R, bool QUANT> __global__ __launch_bounds__(RAYZ_BVH_WG, (bvh_waves<R>() * 256 >= RAYZ_BVH_WG ? bvh_waves<R>() * 256 / RAYZ_BVH_WG : 1))
void trace_kernel_bvh(const TraceArgs<R> A) {
    typedef typename VecOf<R>::type r4;
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint32_t n_nodes = A.sc.bvh_n_nodes;
    const int keep_active = (int)(A.bvh_keep & 0xffu), keep_stepping = (int)((A.bvh_keep >> 8) & 0xffu);
    const float tmin32 = round_down_f32(A.tmin); // the box steps' tmin
    Pcg32 g{0, 1};
    V<R> o{0, 0, 0}, d{0, 0, 1}, ud{0, 0, 1}, thr{1, 1, 1}, acc{0, 0, 0};
    BvhQuery<R> q;
    q.qa = {1.0f, 1.0f, 1.0f};
    q.qb = {0.0f, 0.0f, 0.0f};
    q.tb32 = 0.0f;
    q.inv_a2 = 1.0;
    q.tbest = R(0);
    q.ibest = -1;
    q.cur = kBvhDone;
    q.sp = 0;
    q.top = kBvhDone;
    q.lb.make(ud, o);
    // dynamic shared memory, from LDS address 0 (the kernel has no static LDS): the top of the tree, copied once per
    // workgroup (A.bvh_top_words u32s; a node's LDS address is its index << 6 for f32), then the per-lane traversal stacks,
    // sized by the launch from the tree's depth: entry s of this lane at stack[256 * s] — conflict-free for any mix of s
    extern __shared__ uint32_t lds_words[];
    // the node fetch addresses the LDS copy by index << 6 from LDS address 0: refuse to run (loudly: the host turns the
    // flag into RAYZ_ERR_STATE) should a build ever place anything in front of the dynamic segment
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)lds_words != 0u) {
        if (threadIdx.x == 0) A.counters[31] = 1ull;
        return;
    }
    f4* top = (f4*)lds_words;
    uint32_t* stack = lds_words + A.bvh_top_words + kBvhWg + threadIdx.x; // (one guard row under entry 0: BvhQuery::top)
    for (uint32_t k = threadIdx.x; k < A.sc.bvh_top / 16u; k += kBvhWg) top[k] = A.sc.bvh_nodes[k];
    const f4* nodes_base = scalar_base(A.sc.bvh_nodes);
    stack[0] = kBvhDone; // the sentinel under every lane's stack
    // the oversized hittables' records, entry e = 2·(descriptor) + (entry of it): {filter record: 3 words in R, f64 sphere: 2}
    unsigned char* big_lds = (unsigned char*)(lds_words + A.bvh_big_words);
    if (threadIdx.x < 2u * A.sc.bvh_n_big_leaves) {
        const uint32_t desc = A.sc.bvh_big[threadIdx.x >> 1], j = threadIdx.x & 1u;
        if (j < (desc & 3u)) {
            const uint32_t slot = (desc >> 4) + j;
            d4* dst64 = (d4*)(big_lds + threadIdx.x * bvh_big_entry_bytes<R>());
            dst64[0] = A.sc.bvh_sph64[2 * slot];
            dst64[1] = A.sc.bvh_sph64[2 * slot + 1];
            r4* dst = (r4*)(dst64 + 2);
            const r4* rec = A.sc.bvh_leaf + (size_t)A.sc.bvh_leaf_stride * slot;
            dst[0] = rec[0];
            dst[1] = rec[1];
            dst[2] = A.sc.bvh_leaf_stride > 2u ? rec[2] : rec[1];
        }
    }
    __syncthreads();
    R time = 0;
    uint32_t item = 0, px = 0, py = 0, s_cur = 0, s_end = 0, seg = 0, nseg = 0, sphere_tests = 0;
    uint32_t node_tests = 0; // counted per WAVE (two per stepping lane, off the step's own lane count): scalar arithmetic
    bool has_item = false, alive = false, fresh = false;
    WaveQueue wq; // wave-uniform
#ifdef RAYZ_BVH_PROFILE
    unsigned long long pt[5] = {0, 0, 0, 0, 0}, pl[7] = {0, 0, 0, 0, 0, 0, 0}, px3[3] = {0, 0, 0}, pt0 = __builtin_amdgcn_s_memtime(), fetch_ticks = 0;
#define RAYZ_PROF_T(k) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); pt[k] += now_ - pt0; pt0 = now_; }
#define RAYZ_PROF_L(k, n) { pl[k] += (unsigned long long)(n); pl[k + 1] += 1; }
#else
#define RAYZ_PROF_T(k)
#define RAYZ_PROF_L(k, n)
#endif

    for (;;) {
        // ---- retire finished chunks, refill idle lanes (wave-aggregated queue pop) ----
        if (!alive && has_item && s_cur == s_end) {
            A.partial[item] = r4{acc.x, acc.y, acc.z, R(0)};
            has_item = false;
        }
        {
            const bool popping = __ballot(!alive && !has_item && !queue_empty<R>(wq, A)) != 0ull;
            uint32_t got_item = 0;
            if (queue_pop<R>(A, wq, lane, !alive && !has_item && !queue_empty<R>(wq, A), got_item)) {
                item = got_item;
                has_item = true;
                const uint32_t k = place_item<R, true>(A, item, px, py);
                chunk_bounds<R>(A, k, s_cur, s_end);
                acc = {R(0), R(0), R(0)};
            }
            // the per-lane u32 statistics would wrap after ≈30 minutes inside one launch: spill them when half full
            if (popping && node_tests > RAYZ_STAT_SPILL) {
                if (lane == 0) atomicAdd(&A.counters[2], (unsigned long long)node_tests);
                atomicAdd(&A.counters[3], (unsigned long long)sphere_tests);
                atomicAdd(&A.counters[1], (unsigned long long)nseg);
                node_tests = sphere_tests = nseg = 0;
            }
        }
        if (!alive && has_item) {
            const unsigned long long pixel_index = (unsigned long long)py * A.width + px;
            g.seed_path(A.seed, pixel_index * A.spp + s_cur);
            camera_ray<R>(A.cam, g, px, py, o, d, time);
            thr = {R(1), R(1), R(1)};
            seg = 0;
            s_cur++;
            alive = true;
            fresh = true;
        }
        if (__ballot(alive) == 0ull) break;
        // ---- every segment that starts here (camera rays above, scattered rays of the last shading pass): one place
        //      for the per-segment set-up (unit direction, slab constants) ----
        if (fresh) {
            ud = unit(d);
            bvh_begin<R, QUANT>(q, A.sc, o, d, ud, n_nodes);
        }
        // .. then the oversized hittables kept out of the tree: the walk starts with their tbest and culls behind it
        if (A.sc.bvh_n_big_leaves != 0u && __ballot(fresh) != 0ull) {
            if (fresh) {
                for (uint32_t k = 0; k < A.sc.bvh_n_big_leaves; ++k) { // wave-uniform trip count
                    const uint32_t desc = A.sc.bvh_big[k];
                    sphere_tests += desc & 3u;
                    for (uint32_t j = 0; j < (desc & 3u); ++j) { // (from the LDS copy: the same address in every lane)
                        const d4* rec64 = (const d4*)(big_lds + (2u * k + j) * bvh_big_entry_bytes<R>());
                        const r4* rec = (const r4*)(rec64 + 2);
                        const r4 c = rec[0], v = rec[1], w = rec[2];
                        if (bvh_leaf_eval<R>(A.sc, q, desc, j, c, v, o, d, ud, time, A.tmin, &w) != 0u)
                            bvh_candidate_eval<R>(q, rec64[0], rec64[1], (int)bits(v.w), o, d, time, A.tmin);
                    }
                }
            }
        }
        fresh = false;
        RAYZ_PROF_T(0)

        // ---- rounds of (N) box steps, (L) leaf tests, (C) candidate roots ----
        const int n_alive = __popcll(__ballot(alive));
        for (;;) {
            if constexpr (sizeof(R) == 8) q.tb32 = round_up_f32(q.tbest); // (tbest moves in phases L and C and in the set-up
                                                                          //  above, never in N)
            // phase N: lanes holding an inner node step; lanes holding a leaf wait.  It goes on while at least keep_stepping
            // lanes can step — or any, if nobody waits at a leaf (ONE wave-uniform flag, computed where the lane count is
            // known: the loop has a single exit test)
            bool can_step = q.cur < kBvhDone;
            int n_can = __popcll(__ballot(can_step));
            // (n_can != 0 first: with nobody able to step the loop must end whatever the threshold is — a threshold of 0 would
            //  otherwise spin for ever with no lane stepping)
            bool run = n_can != 0 && (n_can >= keep_stepping || __ballot((int32_t)q.cur < 0) == 0ull);
            // (wave priorities: the rounds' dependent chains — box steps above leaf and root phases — ahead of the throughput
            //  work of the shading pass and the refill, which other waves' issue slots serve as well late as early:
            //  +0.9 % config 3, +1.7 % config 5, profiles/r03/bvh_step/setprio.log)
            __builtin_amdgcn_s_setprio(3);
            while (run) {
                RAYZ_PROF_L(0, n_can)
#ifdef RAYZ_BVH_PROFILE
                px3[0] += __popcll(__ballot((int32_t)q.cur < 0));
                px3[1] += __popcll(__ballot(alive && q.cur == kBvhDone));
                px3[2] += __popcll(__ballot(!alive));
                if (can_step) bvh_node_step<R, kBvhWg, QUANT>(A.sc, nodes_base, q, tmin32, stack, fetch_ticks);
#else
                if (can_step) bvh_node_step<R, kBvhWg, QUANT>(A.sc, nodes_base, q, tmin32, stack);
                // .. and a second step for the lanes that still hold an inner node, without a new wave-level decision (lane
                // counts against thresholds are scalar work with a taken branch at its end: every other step is enough —
                // +1 % on configs 3 / 5, +2.6 % on config 2, profiles/r03/bvh_step/unroll.log; three or four lose it again)
                {
                    const bool again = q.cur < kBvhDone;
                    node_tests += 2u * (uint32_t)__popcll(__ballot(again));
                    if (again) bvh_node_step<R, kBvhWg, QUANT>(A.sc, nodes_base, q, tmin32, stack);
                }
#endif
                node_tests += 2u * (uint32_t)n_can;
                can_step = q.cur < kBvhDone;
                n_can = __popcll(__ballot(can_step));
                run = n_can != 0 && (n_can >= keep_stepping || __ballot((int32_t)q.cur < 0) == 0ull);
            }
            __builtin_amdgcn_s_setprio(2);
            RAYZ_PROF_T(1)
            const bool parked = (int32_t)q.cur < 0;
            if (__ballot(parked) == 0ull) break; // nobody parked: every walking lane ran out of nodes
            RAYZ_PROF_L(2, __popcll(__ballot(parked)))
            uint32_t cand0 = 0, cand1 = 0;
            int pool0 = 0, pool1 = 0;
            if (parked) { // phase L
                const uint32_t leaf = q.cur & ~kBvhLeafFlag;
                sphere_tests += leaf & 3u;
                bvh_pop<R, kBvhWg>(q, stack); // first: its LDS read-ahead travels while the records are fetched and tested
                bvh_leaf_pair<R>(A.sc, q, leaf, o, d, ud, time, A.tmin, cand0, cand1, pool0, pool1);
            }
            RAYZ_PROF_T(2)
            if (__ballot((cand0 | cand1) != 0u) != 0ull) { // phase C
                RAYZ_PROF_L(4, __popcll(__ballot((cand0 | cand1) != 0u)))
                // a lane's only candidate goes into the first pass whichever entry it came from: the second pass runs
                // only when some lane has two (the nearest hit does not depend on the order)
                const uint32_t c0 = cand0 != 0u ? cand0 : cand1, c1 = cand0 != 0u ? cand1 : 0u;
                const int p0 = cand0 != 0u ? pool0 : pool1;
                if (c0 != 0u) bvh_candidate<R>(A.sc, q, c0 - 1u, p0, o, d, time, A.tmin);
                if (__ballot(c1 != 0u) != 0ull) {
                    if (c1 != 0u) bvh_candidate<R>(A.sc, q, c1 - 1u, pool1, o, d, time, A.tmin);
                }
            }
            RAYZ_PROF_T(3)
            const int n_walking = __popcll(__ballot(q.cur != kBvhDone));
            if (n_walking == 0) break;
            if (n_walking < keep_active && n_walking < n_alive) break; // finished lanes wait: go shade / refill them
        }
        __builtin_amdgcn_s_setprio(0);

        // ---- shade lanes whose query is complete ----
        RAYZ_PROF_T(1)
#ifdef RAYZ_BVH_PROFILE
        pl[6] += __ballot(alive && q.cur == kBvhDone) != 0ull ? 1ull : 0ull; // shade passes
#endif
        if (alive && q.cur == kBvhDone) {
            nseg++;
            seg++;
            bool cont = shade<R>(A.sc, g, o, d, ud, time, q.tbest, q.ibest, thr, acc);
            if (seg >= A.max_bounces) cont = false;
            alive = cont;
            fresh = cont;
            // until its set-up runs at the top of the loop the lane must not look finished: no node, empty stack, but
            // `fresh` keeps it out of the next shading pass (the set-up always comes first)
        }
        RAYZ_PROF_T(4)
    }
#ifdef RAYZ_BVH_PROFILE
    RAYZ_PROF_T(4)
    if (lane == 0) {
        for (int k = 0; k < 5; ++k) atomicAdd(&A.counters[4 + k], pt[k]);
        for (int k = 0; k < 7; ++k) atomicAdd(&A.counters[9 + k], pl[k]);
        for (int k = 0; k < 3; ++k) atomicAdd(&A.counters[16 + k], px3[k]);
        atomicAdd(&A.counters[19], fetch_ticks);
    }
#endif
    unsigned long long t0 = nseg, t1 = node_tests, t2 = sphere_tests;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        t0 += __shfl_xor(t0, off);
        t2 += __shfl_xor(t2, off);
    }
    if (lane == 0) {
        atomicAdd(&A.counters[1], t0);
        atomicAdd(&A.counters[2], t1);
        atomicAdd(&A.counters[3], t2);
    }
}

#ifdef RAYZ_EXPERIMENTS
// ---- persistent trace kernel, BVH traversal, TWO paths per lane -------------------------------------------------
// RETIRED EXPERIMENT (round 3: bit-identical, 19 % slower — DESIGN.md §6): compiled only with -DRAYZ_EXPERIMENTS (tools/bvh2_bench.py
// builds its own library with it); the product library does not contain it.
// trace_kernel_bvh above issues its box steps for ~39 of 64 lanes (profiles/r02): a lane whose walk is complete sits idle
// until enough lanes have finished to make the long shading pass worth running (≈14 lanes on average), and the pass
// itself then runs for the ~45 lanes that are ready.  Here every lane owns TWO path contexts.  One is held by the lane's
// walker (ray + traversal state, in the registers the box step works on); the other is PARKED: either waiting for the
// service pass (shade → retire / pop / start the next path → per-segment set-up) or READY with a ray whose set-up is done.
// A lane whose walk completes swaps — a couple of dozen register moves, no arithmetic — and walks on; the service pass
// runs when most lanes have a parked context that needs it, so it runs on (nearly) full batches, and nobody waits for it
// while its other path still walks.  Same work items, queue, per-path arithmetic and summation tree as the other kernels:
// images are identical bit for bit; only which lane traces which item, and when, differs.
template <class R> struct PathCtx { // what a path carries between its segments, besides its ray
    Pcg32 g;
    V<R> thr, acc;
    uint32_t item, px, py, s_cur, s_end, seg;
    bool has_item;
};
template <class T> __device__ __forceinline__ T pick(bool s, T if_set, T if_clear) { return s ? if_set : if_clear; }
template <class R> __device__ __forceinline__ V<R> pick(bool s, V<R> a, V<R> b) { return {s ? a.x : b.x, s ? a.y : b.y, s ? a.z : b.z}; }
template <class R> __device__ __forceinline__ PathCtx<R> ctx_pick(bool s, const PathCtx<R>& c1, const PathCtx<R>& c0) {
    PathCtx<R> c;
    c.g.state = pick(s, c1.g.state, c0.g.state);
    c.g.inc = pick(s, c1.g.inc, c0.g.inc);
    c.thr = pick<R>(s, c1.thr, c0.thr);
    c.acc = pick<R>(s, c1.acc, c0.acc);
    c.item = pick(s, c1.item, c0.item);
    c.px = pick(s, c1.px, c0.px);
    c.py = pick(s, c1.py, c0.py);
    c.s_cur = pick(s, c1.s_cur, c0.s_cur);
    c.s_end = pick(s, c1.s_end, c0.s_end);
    c.seg = pick(s, c1.seg, c0.seg);
    c.has_item = pick(s, c1.has_item, c0.has_item);
    return c;
}
template <class R> __device__ __forceinline__ void ctx_init(PathCtx<R>& c) {
    c.g = Pcg32{0, 1};
    c.thr = {R(1), R(1), R(1)};
    c.acc = {R(0), R(0), R(0)};
    c.item = c.px = c.py = c.s_cur = c.s_end = c.seg = 0;
    c.has_item = false;
}
// state of a lane's parked context
constexpr uint32_t kParkIdle = 0;  // no path in flight: retire the finished chunk / pop an item / start the next path
constexpr uint32_t kParkDone = 1;  // its walk is complete: shade, then as above
constexpr uint32_t kParkReady = 2; // holds a ray with its set-up done: the walker can take it
constexpr uint32_t kParkDead = 3;  // no item and the queue has run dry: nothing left to do for this context
// scheduling thresholds (defaults; TraceArgs::bvh_keep carries the values in use — they change no result)
constexpr int kBvh2Service = 40; // run the service pass when this many lanes have a parked context waiting for it ..
constexpr int kBvh2Blocked = 10; // .. or when this many lanes can do nothing else (their walker is idle, too)
constexpr int kBvh2Swap = 6;     // run the swap when this many lanes have an idle walker and a ready ray
#ifndef RAYZ_BVH2_WAVES
#define RAYZ_BVH2_WAVES 3 // both contexts live in registers: 168 VGPRs
#endif
template <class R, bool QUANT> __global__ __launch_bounds__(256, RAYZ_BVH2_WAVES) void trace_kernel_bvh2(const TraceArgs<R> A) {
    typedef typename VecOf<R>::type r4;
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint32_t n_nodes = A.sc.bvh_n_nodes;
    const int t_service = (int)(A.bvh_keep & 0xffu), t_blocked = (int)((A.bvh_keep >> 8) & 0xffu),
              t_swap = (int)((A.bvh_keep >> 16) & 0xffu), keep_stepping = (int)((A.bvh_keep >> 24) & 0xffu);
    const float tmin32 = round_down_f32(A.tmin);
    extern __shared__ uint32_t lds_words[];
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)lds_words != 0u) { // see trace_kernel_bvh
        if (threadIdx.x == 0) A.counters[31] = 1ull;
        return;
    }
    f4* top = (f4*)lds_words;
    uint32_t* stack = lds_words + A.bvh_top_words + 256u + threadIdx.x; // (one guard row under entry 0: BvhQuery::top)
    for (uint32_t k = threadIdx.x; k < A.sc.bvh_top / 16u; k += 256u) top[k] = A.sc.bvh_nodes[k];
    stack[0] = kBvhDone;
    __syncthreads();

    const f4* nodes_base = scalar_base(A.sc.bvh_nodes);
    PathCtx<R> c0, c1; // the lane's two path contexts; the walker's is c[wsel], the parked one c[wsel ^ 1]
    ctx_init<R>(c0);
    ctx_init<R>(c1);
    // the walker
    V<R> o{0, 0, 0}, d{0, 0, 1}, ud{0, 0, 1};
    R time = 0;
    BvhQuery<R> q;
    q.qa = {1.0f, 1.0f, 1.0f};
    q.qb = {0.0f, 0.0f, 0.0f};
    q.tb32 = 0.0f;
    q.inv_a2 = 1.0;
    q.tbest = R(0);
    q.ibest = -1;
    q.cur = kBvhDone;
    q.sp = 0;
    q.top = kBvhDone;
    q.lb.make(ud, o);
    bool w_has = false; // the walker holds a context (walking while q.cur != kBvhDone, complete after)
    bool wsel = false;
    // the parked context's ray: complete (kParkDone: o, d, ud, time, tbest, ibest) or ready (kParkReady: everything)
    V<R> po{0, 0, 0}, pd{0, 0, 1}, pud{0, 0, 1};
    R ptime = 0;
    BvhQuery<R> pq = q;
    uint32_t p_state = kParkIdle;
    uint32_t nseg = 0, sphere_tests = 0, node_tests = 0;
    WaveQueue wq;
#ifdef RAYZ_BVH_PROFILE
    unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, pl[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, px3[3] = {0, 0, 0}, pt0 = __builtin_amdgcn_s_memtime();
#define RAYZ_PROF2_T(k) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); pt[k] += now_ - pt0; pt0 = now_; }
#define RAYZ_PROF2_L(k, n) { pl[k] += (unsigned long long)(n); pl[k + 1] += 1; }
#else
#define RAYZ_PROF2_T(k)
#define RAYZ_PROF2_L(k, n)
#endif

    for (;;) {
        // ---- service pass: parked contexts that wait for it (wave-uniform decision) ----
        const bool want_service = p_state == kParkIdle || p_state == kParkDone;
        const unsigned long long m_service = __ballot(want_service), m_walking = __ballot(q.cur != kBvhDone);
        const int n_service = __popcll(m_service), n_walking = __popcll(m_walking), n_blocked = __popcll(m_service & ~m_walking);
        bool progressed = false;
        if (n_service != 0 && (n_service >= t_service || n_blocked >= t_blocked || n_walking == 0)) {
            progressed = true;
            RAYZ_PROF2_L(0, n_service)
            PathCtx<R> c = ctx_pick<R>(!wsel, c1, c0); // the parked context: c[wsel ^ 1]
            bool alive = false;
            if (p_state == kParkDone) { // shade the completed segment
                nseg++;
                c.seg++;
                bool cont = shade<R>(A.sc, c.g, po, pd, pud, ptime, pq.tbest, pq.ibest, c.thr, c.acc);
                if (c.seg >= A.max_bounces) cont = false;
                alive = cont;
            }
            if (want_service && !alive && c.has_item && c.s_cur == c.s_end) { // retire the finished chunk
                A.partial[c.item] = r4{c.acc.x, c.acc.y, c.acc.z, R(0)};
                c.has_item = false;
            }
            {
                const bool need = want_service && !alive && !c.has_item && !queue_empty<R>(wq, A);
                const bool popping = __ballot(need) != 0ull;
                uint32_t got_item = 0;
                if (queue_pop<R>(A, wq, lane, need, got_item)) {
                    c.item = got_item;
                    c.has_item = true;
                    const uint32_t k = place_item<R, true>(A, c.item, c.px, c.py);
                    chunk_bounds<R>(A, k, c.s_cur, c.s_end);
                    c.acc = {R(0), R(0), R(0)};
                }
                if (popping && node_tests > RAYZ_STAT_SPILL) { // (see trace_kernel_bvh)
                    if (lane == 0) atomicAdd(&A.counters[2], (unsigned long long)node_tests);
                    atomicAdd(&A.counters[3], (unsigned long long)sphere_tests);
                    atomicAdd(&A.counters[1], (unsigned long long)nseg);
                    node_tests = sphere_tests = nseg = 0;
                }
            }
            bool fresh = alive;
            if (want_service && !alive && c.has_item) { // start the next path of the chunk
                const unsigned long long pixel_index = (unsigned long long)c.py * A.width + c.px;
                c.g.seed_path(A.seed, pixel_index * A.spp + c.s_cur);
                camera_ray<R>(A.cam, c.g, c.px, c.py, po, pd, ptime);
                c.thr = {R(1), R(1), R(1)};
                c.seg = 0;
                c.s_cur++;
                fresh = true;
            }
            // the per-segment set-up of every ray made above (scattered or camera), then the oversized hittables kept
            // out of the tree: the walk starts with their tbest
            if (fresh) {
                pud = unit(pd);
                bvh_begin<R, QUANT>(pq, A.sc, po, pd, pud, n_nodes);
            }
            if (A.sc.bvh_n_big_leaves != 0u && __ballot(fresh) != 0ull) {
                if (fresh) {
                    for (uint32_t k = 0; k < A.sc.bvh_n_big_leaves; ++k) {
                        const uint32_t desc = A.sc.bvh_big[k];
                        sphere_tests += desc & 3u;
                        const uint32_t b0 = bvh_leaf_entry<R>(A.sc, pq, desc, 0u, po, pd, pud, ptime, A.tmin);
                        const uint32_t b1 = (desc & 3u) > 1u ? bvh_leaf_entry<R>(A.sc, pq, desc, 1u, po, pd, pud, ptime, A.tmin) : 0u;
                        if (b0 != 0u) bvh_candidate<R>(A.sc, pq, b0 - 1u, po, pd, ptime, A.tmin);
                        if (b1 != 0u) bvh_candidate<R>(A.sc, pq, b1 - 1u, po, pd, ptime, A.tmin);
                    }
                }
            }
            if (want_service) {
                p_state = fresh ? kParkReady : kParkDead;
                if (wsel) c0 = c; else c1 = c;
            }
            RAYZ_PROF2_T(0)
        }
        // ---- swap: an idle walker takes the parked ray; the segment it completed is parked for the service pass ----
        {
            const bool w_idle = q.cur == kBvhDone;
            const bool can_swap = w_idle && (p_state == kParkReady || (w_has && p_state == kParkDead));
            const unsigned long long m_swap = __ballot(can_swap), m_walk2 = __ballot(!w_idle);
            const int n_swap = __popcll(m_swap);
            if (n_swap != 0 && (n_swap >= t_swap || progressed || m_walk2 == 0ull)) {
                progressed = true;
                RAYZ_PROF2_L(2, n_swap)
                if (can_swap) {
                    const bool take = p_state == kParkReady, give = w_has;
                    const V<R> to = o, td = d, tud = ud;
                    const R tt = time, ttb = q.tbest;
                    const int tib = q.ibest;
                    if (take) {
                        o = po, d = pd, ud = pud, time = ptime;
                        q = pq; // bvh_begin left cur at the root and the stack empty
                    }
                    if (give) {
                        po = to, pd = td, pud = tud, ptime = tt;
                        pq.tbest = ttb, pq.ibest = tib;
                    }
                    p_state = give ? kParkDone : kParkIdle; // (the context a walker without one leaves behind holds no item)
                    w_has = take;
                    wsel = !wsel;
                }
                RAYZ_PROF2_T(1)
            }
        }
        if (__ballot(q.cur != kBvhDone) == 0ull) {
            if (!progressed) break; // nobody walks, nothing to swap, nothing to service: the wave is done
            continue;
        }

        // ---- rounds of (N) box steps, (L) leaf tests, (C) candidate roots on the walkers ----
        const unsigned long long m_ready = __ballot(p_state == kParkReady || (w_has && p_state == kParkDead)),
                                 m_service2 = __ballot(p_state == kParkIdle || p_state == kParkDone);
        const int n_service2 = __popcll(m_service2);
        for (;;) {
            if constexpr (sizeof(R) == 8) q.tb32 = round_up_f32(q.tbest);
            for (;;) { // phase N
                const bool can_step = q.cur < kBvhDone;
                const int n_can = __popcll(__ballot(can_step));
                if (n_can == 0) break;
                if (n_can < keep_stepping && __ballot((int32_t)q.cur < 0) != 0ull) break;
                RAYZ_PROF2_L(4, n_can)
#ifdef RAYZ_BVH_PROFILE
                px3[0] += __popcll(__ballot((int32_t)q.cur < 0));
                px3[1] += __popcll(__ballot(q.cur == kBvhDone));
                unsigned long long ft_ = 0;
                if (can_step) bvh_node_step<R, kBvh2Wg, QUANT>(A.sc, nodes_base, q, tmin32, stack, ft_);
#else
                if (can_step) bvh_node_step<R, kBvh2Wg, QUANT>(A.sc, nodes_base, q, tmin32, stack);
#endif
                node_tests += 2u * (uint32_t)n_can;
            }
            RAYZ_PROF2_T(2)
            const bool parked = (int32_t)q.cur < 0;
            if (__ballot(parked) != 0ull) {
                RAYZ_PROF2_L(6, __popcll(__ballot(parked)))
                uint32_t cand0 = 0, cand1 = 0;
                if (parked) { // phase L
                    const uint32_t leaf = q.cur & ~kBvhLeafFlag;
                    sphere_tests += leaf & 3u;
                    cand0 = bvh_leaf_entry<R>(A.sc, q, leaf, 0u, o, d, ud, time, A.tmin);
                    if ((leaf & 3u) > 1u) cand1 = bvh_leaf_entry<R>(A.sc, q, leaf, 1u, o, d, ud, time, A.tmin);
                    bvh_pop<R, kBvh2Wg>(q, stack);
                }
                RAYZ_PROF2_T(3)
                if (__ballot((cand0 | cand1) != 0u) != 0ull) { // phase C
                    RAYZ_PROF2_L(8, __popcll(__ballot((cand0 | cand1) != 0u)))
                    const uint32_t k0 = cand0 != 0u ? cand0 : cand1, k1 = cand0 != 0u ? cand1 : 0u;
                    if (k0 != 0u) bvh_candidate<R>(A.sc, q, k0 - 1u, o, d, time, A.tmin);
                    if (__ballot(k1 != 0u) != 0ull) {
                        if (k1 != 0u) bvh_candidate<R>(A.sc, q, k1 - 1u, o, d, time, A.tmin);
                    }
                }
                RAYZ_PROF2_T(4)
            }
            // leave the rounds when the walkers that ran out make a swap or a service pass due
            const unsigned long long m_idle = __ballot(q.cur == kBvhDone);
            if (~m_idle == 0ull) break;
            if ((int)__popcll(m_idle & m_ready) >= t_swap) break;
            if (n_service2 != 0 && (n_service2 >= t_service || (int)__popcll(m_idle & m_service2) >= t_blocked)) break;
        }
    }
#ifdef RAYZ_BVH_PROFILE
    if (lane == 0) {
        for (int k = 0; k < 5; ++k) atomicAdd(&A.counters[4 + k], pt[k]);
        for (int k = 0; k < 10; ++k) atomicAdd(&A.counters[9 + k], pl[k]);
        for (int k = 0; k < 2; ++k) atomicAdd(&A.counters[19 + k], px3[k]);
    }
#endif
    unsigned long long t0 = nseg, t1 = node_tests, t2 = sphere_tests;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        t0 += __shfl_xor(t0, off);
        t2 += __shfl_xor(t2, off);
    }
    if (lane == 0) {
        atomicAdd(&A.counters[1], t0);
        atomicAdd(&A.counters[2], t1);
        atomicAdd(&A.counters[3], t2);
    }
}
#endif // RAYZ_EXPERIMENTS

#ifdef RAYZ_EXPERIMENTS
// ---- persistent trace kernel, BVH traversal, WALKER and SHADER waves (round-4 experiment) --------------------------------
// trace_kernel_bvh runs its shading pass (shade -> retire / pop / start -> per-segment set-up: ~1,700 vector instructions) with
// ~48 of 64 lanes, and its box steps with ~42 while ~13 lanes wait, finished, for that pass (profiles/r03, r04).  Here the 16
// waves of the workgroup take fixed roles: kXWalkers waves only walk, kXShaders waves only run the pass, and rays travel
// between them through small per-walker-wave slot sets in LDS — a walker lane whose walk is complete hands its whole path
// (ray + context, 7 x 16 B) to a FREE slot of its wave and takes a READY one (11 x 16 B: + the set-up's constants); the shader
// wave bound to that walker collects FINISHED slots from its 2-3 walker waves until it has a full wave of them, runs the pass
// on 64 lanes and writes each result back IN PLACE (FINISHED -> READY).  Every slot state has one writer per transition and
// a wave's LDS operations complete in order, so the hand-over needs no atomics: data first, then the state word.  Births and
// deaths of paths happen only in the shader (a finished chunk is retired and the next item popped in place), so the number of
// paths bound to a walker wave is constant, 64 + x_slots / 2: some slot is always FREE or about to become READY (no deadlock).
// Same per-path arithmetic, queue and summation tree as the other kernels: the image is identical bit for bit.
#ifndef RAYZ_BVHX_SHADERS
#define RAYZ_BVHX_SHADERS 4
#endif
constexpr uint32_t kXShaders = RAYZ_BVHX_SHADERS, kXWalkers = 16 - kXShaders, kXWalkerLanes = kXWalkers * 64;
constexpr uint32_t kXFree = 0, kXFinished = 1, kXReady = 2; // slot states
constexpr uint32_t kXChunksF = 7, kXChunksR = 11;           // 16-byte pieces of a record: walker -> shader, shader -> walker
// exchange area (u32 words from A.x_words): [s] done flag of shader s | [16 + w] rays held in the lanes of walker w | [32] abort |
// [64 + 64 w + i] state of slot i of walker w | from kXScratchWords: two 64-byte rank -> slot tables per wave | from kXSlotWords:
// the slots, piece c of slot i of walker w = the f4 at (w * 11 + c) * x_slots + i (piece-major: the lanes of a wave touch
// consecutive 16-byte pieces — conflict-free ds_read_b128 / ds_write_b128)
constexpr uint32_t kXScratchWords = 64 + 16 * 64, kXSlotWords = kXScratchWords + 16 * 32;
__host__ __device__ constexpr size_t bvhx_exchange_bytes(uint32_t ns) { return (size_t)kXSlotWords * 4 + (size_t)kXWalkers * kXChunksR * ns * 16; }
constexpr uint32_t kXSpinLimit = 1u << 22; // idle polls (s_sleep) before a wave gives up and aborts the launch: bounded, never a hang

__device__ __forceinline__ void set_prio(uint32_t p) { // s_setprio takes an immediate; p is wave-uniform
    if (p == 3u) __builtin_amdgcn_s_setprio(3);
    else if (p == 2u) __builtin_amdgcn_s_setprio(2);
    else if (p == 1u) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}
template <bool QUANT> __global__ __launch_bounds__(1024, 1) void trace_kernel_bvhx(const TraceArgs<float> A) {
    typedef float R;
    typedef f4 r4;
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t n_nodes = A.sc.bvh_n_nodes, NS = A.x_slots;
    const float tmin32 = round_down_f32(A.tmin);
    extern __shared__ uint32_t lds_words[];
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)lds_words != 0u) { // see trace_kernel_bvh
        if (threadIdx.x == 0) A.counters[31] = 1ull;
        return;
    }
    f4* top = (f4*)lds_words;
    for (uint32_t k = threadIdx.x; k < A.sc.bvh_top / 16u; k += 1024u) top[k] = A.sc.bvh_nodes[k];
    const f4* nodes_base = scalar_base(A.sc.bvh_nodes);
    unsigned char* big_lds = (unsigned char*)(lds_words + A.bvh_big_words);
    if (threadIdx.x < 2u * A.sc.bvh_n_big_leaves) { // (as trace_kernel_bvh)
        const uint32_t desc = A.sc.bvh_big[threadIdx.x >> 1], j = threadIdx.x & 1u;
        if (j < (desc & 3u)) {
            const uint32_t slot = (desc >> 4) + j;
            d4* dst64 = (d4*)(big_lds + threadIdx.x * bvh_big_entry_bytes<R>());
            dst64[0] = A.sc.bvh_sph64[2 * slot];
            dst64[1] = A.sc.bvh_sph64[2 * slot + 1];
            r4* dst = (r4*)(dst64 + 2);
            const r4* rec = A.sc.bvh_leaf + (size_t)A.sc.bvh_leaf_stride * slot;
            dst[0] = rec[0];
            dst[1] = rec[1];
            dst[2] = A.sc.bvh_leaf_stride > 2u ? rec[2] : rec[1];
        }
    }
    // the exchange area: every walker lane starts with a finished "null" path (no item, not alive: the shader pops one into it),
    // and half of every walker's slots start FINISHED with such a path too
    volatile uint32_t* xw = lds_words + A.x_words;
    f4* slots = (f4*)(lds_words + A.x_words + kXSlotWords);
    for (uint32_t k = threadIdx.x; k < kXSlotWords; k += 1024u) {
        uint32_t v = 0u;
        if (k >= 16u && k < 16u + kXWalkers) v = 64u;
        if (k >= 64u && k < 64u + 64u * kXWalkers) v = ((k & 63u) < NS / 2u) ? kXFinished : kXFree;
        xw[k] = v;
    }
    for (uint32_t k = threadIdx.x; k < kXWalkers * kXChunksR * NS; k += 1024u) slots[k] = f4{0.0f, 0.0f, 0.0f, 0.0f};
    if (wv < kXWalkers) lds_words[A.bvh_top_words + kXWalkerLanes + threadIdx.x] = kBvhDone; // the sentinel under every walker lane's stack
    __syncthreads();

    const uint32_t x_min = A.x_cfg & 0xffu, x_batch = (A.x_cfg >> 8) & 0xffu, x_patience = (A.x_cfg >> 16) & 0xffu, x_prio = (A.x_cfg >> 24) & 3u,
                   w_prio_n = (A.x_cfg >> 26) & 3u, w_prio_lc = (A.x_cfg >> 28) & 3u, w_prio_x = (A.x_cfg >> 30) & 3u;
    volatile unsigned char* scr = (volatile unsigned char*)(xw + kXScratchWords) + 128u * wv;
    // a path, as it travels
    Pcg32 g{0, 1};
    V<R> o{0, 0, 0}, d{0, 0, 1}, ud{0, 0, 1}, thr{1, 1, 1}, acc{0, 0, 0};
    R time = 0;
    uint32_t item = 0, px = 0, py = 0, s_cur = 0, s_end = 0, segflags = 0; // segflags: segments so far | alive << 30 | has_item << 31
    BvhQuery<R> q;
    q.qa = {1.0f, 1.0f, 1.0f};
    q.qb = {0.0f, 0.0f, 0.0f};
    q.tb32 = 0.0f;
    q.inv_a2 = 1.0;
    q.tbest = R(0);
    q.ibest = -1;
    q.cur = kBvhDone;
    q.sp = 0;
    q.top = kBvhDone;
    q.lb.make(ud, o);
    uint32_t nseg = 0, sphere_tests = 0, node_tests = 0;
#ifdef RAYZ_BVH_PROFILE
    unsigned long long x_fetch = 0;
#define RAYZ_XPROF_FETCH , x_fetch
    unsigned long long xp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, xt0 = __builtin_amdgcn_s_memtime();
#define RAYZ_XPROF_T(k) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); xp[k] += now_ - xt0; xt0 = now_; }
#define RAYZ_XPROF_N(k, n) { xp[k] += (unsigned long long)(n); }
#else
#define RAYZ_XPROF_FETCH
#define RAYZ_XPROF_T(k)
#define RAYZ_XPROF_N(k, n)
#endif

    if (wv < kXWalkers) {
        // ================================================ WALKER ================================================
        uint32_t* stack = lds_words + A.bvh_top_words + kXWalkerLanes + threadIdx.x; // (one guard row under entry 0)
        volatile uint32_t* xs = xw + 64u + 64u * wv;
        f4* my_slots = slots + (size_t)wv * kXChunksR * NS;
        const uint32_t my_shader = wv % kXShaders;
        bool has_ray = true;
        for (uint32_t spins = 0;;) {
            // ---- exchange: finished paths out, ready paths in ----
            {
                const uint32_t st = lane < NS ? xs[lane] : 3u;
                const bool fin = has_ray && q.cur == kBvhDone;
                const unsigned long long m_fin = __ballot(fin), m_free = __ballot(st == kXFree), m_ready = __ballot(st == kXReady);
                const uint32_t n_fin = (uint32_t)__popcll(m_fin), n_free = (uint32_t)__popcll(m_free), n_dep = n_fin < n_free ? n_fin : n_free;
                if (n_dep != 0u) {
                    if (st == kXFree) scr[__builtin_amdgcn_mbcnt_hi((uint32_t)(m_free >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_free, 0u))] = (unsigned char)lane;
                    const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_fin >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_fin, 0u));
                    asm volatile("" ::: "memory");
                    if (fin && r < n_dep) {
                        const uint32_t slot = scr[r];
                        f4* rec = my_slots + slot;
                        rec[0 * NS] = f4{o.x, o.y, o.z, time};
                        rec[1 * NS] = f4{d.x, d.y, d.z, q.tbest};
                        rec[2 * NS] = f4{ud.x, ud.y, ud.z, Bits<float>::from((uint32_t)q.ibest)};
                        rec[3 * NS] = f4{thr.x, thr.y, thr.z, Bits<float>::from(segflags)};
                        rec[4 * NS] = f4{acc.x, acc.y, acc.z, Bits<float>::from(item)};
                        rec[5 * NS] = f4{Bits<float>::from((uint32_t)g.state), Bits<float>::from((uint32_t)(g.state >> 32)), Bits<float>::from((uint32_t)g.inc),
                                         Bits<float>::from((uint32_t)(g.inc >> 32))};
                        rec[6 * NS] = f4{Bits<float>::from(px), Bits<float>::from(py), Bits<float>::from(s_cur), Bits<float>::from(s_end)};
                        asm volatile("" ::: "memory");
                        xs[slot] = kXFinished; // after the record (LDS keeps a wave's order)
                        has_ray = false;
                    }
                }
                const unsigned long long m_empty = __ballot(!has_ray);
                const uint32_t n_empty = (uint32_t)__popcll(m_empty), n_ready = (uint32_t)__popcll(m_ready), n_take = n_empty < n_ready ? n_empty : n_ready;
                // the rays this wave's lanes hold, as its shader's end test sees them: never below the truth — written after the
                // deposits' state words and before the takes'
                asm volatile("" ::: "memory");
                if (lane == 0u) xw[16u + wv] = 64u - n_empty + n_take;
                asm volatile("" ::: "memory");
                if (n_take != 0u) {
                    if (st == kXReady) scr[64u + __builtin_amdgcn_mbcnt_hi((uint32_t)(m_ready >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_ready, 0u))] = (unsigned char)lane;
                    const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_empty >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_empty, 0u));
                    asm volatile("" ::: "memory");
                    if (!has_ray && r < n_take) {
                        const uint32_t slot = scr[64u + r];
                        const f4* rec = my_slots + slot;
                        const f4 c0 = rec[0 * NS], c1 = rec[1 * NS], c2 = rec[2 * NS], c3 = rec[3 * NS], c4 = rec[4 * NS], c5 = rec[5 * NS],
                                 c6 = rec[6 * NS], c7 = rec[7 * NS], c8 = rec[8 * NS], c9 = rec[9 * NS], c10 = rec[10 * NS];
                        asm volatile("" ::: "memory");
                        xs[slot] = kXFree; // after the reads
                        o = {c0.x, c0.y, c0.z}, time = c0.w;
                        d = {c1.x, c1.y, c1.z}, q.tbest = c1.w;
                        ud = {c2.x, c2.y, c2.z}, q.ibest = (int)bits(c2.w);
                        thr = {c3.x, c3.y, c3.z}, segflags = bits(c3.w);
                        acc = {c4.x, c4.y, c4.z}, item = bits(c4.w);
                        g.state = (unsigned long long)bits(c5.x) | ((unsigned long long)bits(c5.y) << 32);
                        g.inc = (unsigned long long)bits(c5.z) | ((unsigned long long)bits(c5.w) << 32);
                        px = bits(c6.x), py = bits(c6.y), s_cur = bits(c6.z), s_end = bits(c6.w);
                        q.qa = {c7.x, c7.y, c7.z};
                        q.qb = {c7.w, c8.x, c8.y};
                        q.inv_a2 = __builtin_bit_cast(double, (unsigned long long)bits(c8.z) | ((unsigned long long)bits(c8.w) << 32));
                        q.lb.b.e1x = c9.x, q.lb.b.e1z = c9.y, q.lb.b.e2x = c9.z, q.lb.b.e2y = c9.w;
                        q.lb.b.e2z = c10.x, q.lb.b.k1 = c10.y, q.lb.b.k2 = c10.z;
                        q.cur = n_nodes ? 0u : kBvhDone;
                        q.sp = 0;
                        q.top = kBvhDone;
                        has_ray = true;
                    }
                }
            }
            RAYZ_XPROF_T(0)
            RAYZ_XPROF_N(3, 1)
            if (__ballot(q.cur != kBvhDone) == 0ull) { // nobody to walk: wait for ready paths, or for free slots — or the end
                if (__ballot(has_ray) == 0ull && xw[my_shader] != 0u) break;
                if (xw[32] != 0u) break;
                if (++spins > kXSpinLimit) {
                    xw[32] = 1u;
                    if (lane == 0u) A.counters[31] = 2ull;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                RAYZ_XPROF_T(1)
                continue;
            }
            spins = 0;
            RAYZ_XPROF_N(4, __popcll(__ballot(q.cur != kBvhDone)))
            RAYZ_XPROF_N(5, __popcll(__ballot(has_ray)))
            RAYZ_XPROF_N(6, 1)
            // ---- rounds of (N) box steps, (L) leaf tests, (C) candidate roots, as in trace_kernel_bvh ----
            const int keep_stepping = (int)((A.bvh_keep >> 8) & 0xffu);
            for (;;) {
                bool can_step = q.cur < kBvhDone;
                int n_can = __popcll(__ballot(can_step));
                bool run = n_can != 0 && (n_can >= keep_stepping || __ballot((int32_t)q.cur < 0) == 0ull);
                set_prio(w_prio_n);
                while (run) {
                    if (can_step) bvh_node_step<R, kXWalkerLanes, QUANT>(A.sc, nodes_base, q, tmin32, stack RAYZ_XPROF_FETCH);
                    {
                        const bool again = q.cur < kBvhDone;
                        node_tests += 2u * (uint32_t)__popcll(__ballot(again));
                        if (again) bvh_node_step<R, kXWalkerLanes, QUANT>(A.sc, nodes_base, q, tmin32, stack RAYZ_XPROF_FETCH);
                    }
                    node_tests += 2u * (uint32_t)n_can;
                    can_step = q.cur < kBvhDone;
                    n_can = __popcll(__ballot(can_step));
                    run = n_can != 0 && (n_can >= keep_stepping || __ballot((int32_t)q.cur < 0) == 0ull);
                }
                set_prio(w_prio_lc);
                const bool parked = (int32_t)q.cur < 0;
                if (__ballot(parked) != 0ull) {
                    uint32_t cand0 = 0, cand1 = 0;
                    int pool0 = 0, pool1 = 0;
                    if (parked) { // phase L
                        const uint32_t leaf = q.cur & ~kBvhLeafFlag;
                        sphere_tests += leaf & 3u;
                        bvh_pop<R, kXWalkerLanes>(q, stack);
                        bvh_leaf_pair<R>(A.sc, q, leaf, o, d, ud, time, A.tmin, cand0, cand1, pool0, pool1);
                    }
                    if (__ballot((cand0 | cand1) != 0u) != 0ull) { // phase C
                        const uint32_t c0 = cand0 != 0u ? cand0 : cand1, c1 = cand0 != 0u ? cand1 : 0u;
                        const int p0 = cand0 != 0u ? pool0 : pool1;
                        if (c0 != 0u) bvh_candidate<R>(A.sc, q, c0 - 1u, p0, o, d, time, A.tmin);
                        if (__ballot(c1 != 0u) != 0ull) {
                            if (c1 != 0u) bvh_candidate<R>(A.sc, q, c1 - 1u, pool1, o, d, time, A.tmin);
                        }
                    }
                }
                if (__ballot(q.cur != kBvhDone) == 0ull) break;
                if ((uint32_t)__popcll(__ballot(has_ray && q.cur == kBvhDone)) >= x_min) break; // enough finished paths to hand over
            }
            set_prio(w_prio_x);
            RAYZ_XPROF_T(2)
        }
#ifdef RAYZ_BVH_PROFILE
        if (lane == 0) for (int k = 0; k < 7; ++k) atomicAdd(&A.counters[4 + k], xp[k]);
#endif
    } else {
        // ================================================ SHADER ================================================
        const uint32_t sh = wv - kXWalkers;
        WaveQueue wq;
        uint32_t k0 = 0, waited = 0, spins = 0;
        set_prio(x_prio);
        uint32_t nw = 0;
        for (uint32_t k = 0; k < 3u; ++k) nw += (sh + k * kXShaders < kXWalkers) ? 1u : 0u;
        for (;;) {
            // ---- collect FINISHED slots of my walkers (starting with a different walker each time) ----
            uint32_t st[3];
            unsigned long long m[3];
            uint32_t n = 0;
#pragma unroll
            for (uint32_t j = 0; j < 3u; ++j) {
                const uint32_t k = (k0 + j) % 3u, w = sh + k * kXShaders;
                st[j] = (k < nw && lane < NS) ? xw[64u + 64u * w + lane] : 3u;
                m[j] = __ballot(st[j] == kXFinished);
                n += (uint32_t)__popcll(m[j]);
            }
            if (n == 0u) {
                bool over = false;
                if (queue_empty<R>(wq, A)) { // nothing is born any more: over when no path is left — slots, lanes, slots again
                    bool busy = false;
#pragma unroll
                    for (uint32_t j = 0; j < 3u; ++j) busy = busy || (st[j] == kXFinished || st[j] == kXReady);
                    uint32_t rays = 0;
                    for (uint32_t k = 0; k < nw; ++k) rays += xw[16u + sh + k * kXShaders];
                    asm volatile("" ::: "memory");
                    for (uint32_t k = 0; k < nw; ++k) {
                        const uint32_t s2 = lane < NS ? xw[64u + 64u * (sh + k * kXShaders) + lane] : 0u;
                        busy = busy || s2 != kXFree;
                    }
                    over = __ballot(busy) == 0ull && rays == 0u;
                }
                if (over) {
                    if (lane == 0u) xw[sh] = 1u;
                    break;
                }
                if (xw[32] != 0u) break;
                if (++spins > kXSpinLimit) {
                    xw[32] = 1u;
                    if (lane == 0u) A.counters[31] = 2ull;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                RAYZ_XPROF_T(0)
                continue;
            }
            spins = 0;
            if (n < x_batch && waited < x_patience) { // let the batch fill
                ++waited;
                __builtin_amdgcn_s_sleep(1);
                RAYZ_XPROF_T(1)
                continue;
            }
            RAYZ_XPROF_T(1)
            RAYZ_XPROF_N(3, 1)
            RAYZ_XPROF_N(4, n < 64u ? n : 64u)
            RAYZ_XPROF_N(5, n)
            waited = 0;
            uint32_t off = 0;
#pragma unroll
            for (uint32_t j = 0; j < 3u; ++j) {
                const uint32_t k = (k0 + j) % 3u;
                const uint32_t r = off + __builtin_amdgcn_mbcnt_hi((uint32_t)(m[j] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[j], 0u));
                if (st[j] == kXFinished && r < 64u) scr[r] = (unsigned char)((k << 6) | lane); // (x_slots <= 64 and k <= 2: fits a byte... k << 6 | lane < 192)
                off += (uint32_t)__popcll(m[j]);
            }
            k0 = (k0 + 1u) % 3u;
            asm volatile("" ::: "memory");
            const bool have = lane < (n < 64u ? n : 64u);
            uint32_t code = 0;
            if (have) code = scr[lane];
            const uint32_t w_of = sh + (code >> 6) * kXShaders, slot = code & 63u;
            f4* rec = slots + (size_t)w_of * kXChunksR * NS + slot;
            bool alive = false, has_item = false, fresh = false;
            uint32_t seg = 0;
            if (have) {
                const f4 c0 = rec[0 * NS], c1 = rec[1 * NS], c2 = rec[2 * NS], c3 = rec[3 * NS], c4 = rec[4 * NS], c5 = rec[5 * NS], c6 = rec[6 * NS];
                o = {c0.x, c0.y, c0.z}, time = c0.w;
                d = {c1.x, c1.y, c1.z}, q.tbest = c1.w;
                ud = {c2.x, c2.y, c2.z}, q.ibest = (int)bits(c2.w);
                thr = {c3.x, c3.y, c3.z}, segflags = bits(c3.w);
                acc = {c4.x, c4.y, c4.z}, item = bits(c4.w);
                g.state = (unsigned long long)bits(c5.x) | ((unsigned long long)bits(c5.y) << 32);
                g.inc = (unsigned long long)bits(c5.z) | ((unsigned long long)bits(c5.w) << 32);
                px = bits(c6.x), py = bits(c6.y), s_cur = bits(c6.z), s_end = bits(c6.w);
                alive = (segflags >> 30) & 1u, has_item = (segflags >> 31) & 1u, seg = segflags & 0x3fffffffu;
            }
            // ---- the pass, as in trace_kernel_bvh: shade, retire / pop / start, per-segment set-up ----
            if (alive) {
                nseg++;
                seg++;
                bool cont = shade<R>(A.sc, g, o, d, ud, time, q.tbest, q.ibest, thr, acc);
                if (seg >= A.max_bounces) cont = false;
                alive = cont;
                fresh = cont;
            }
            if (have && !alive && has_item && s_cur == s_end) {
                A.partial[item] = r4{acc.x, acc.y, acc.z, R(0)};
                has_item = false;
            }
            {
                const bool popping = __ballot(have && !alive && !has_item && !queue_empty<R>(wq, A)) != 0ull;
                uint32_t got_item = 0;
                if (queue_pop<R>(A, wq, lane, have && !alive && !has_item && !queue_empty<R>(wq, A), got_item)) {
                    item = got_item;
                    has_item = true;
                    const uint32_t k = place_item<R, true>(A, item, px, py);
                    chunk_bounds<float>(A, k, s_cur, s_end);
                    acc = {R(0), R(0), R(0)};
                }
                if (popping && nseg > RAYZ_STAT_SPILL) {
                    atomicAdd(&A.counters[3], (unsigned long long)sphere_tests);
                    atomicAdd(&A.counters[1], (unsigned long long)nseg);
                    sphere_tests = nseg = 0;
                }
            }
            if (have && !alive && has_item) {
                const unsigned long long pixel_index = (unsigned long long)py * A.width + px;
                g.seed_path(A.seed, pixel_index * A.spp + s_cur);
                camera_ray<R>(A.cam, g, px, py, o, d, time);
                thr = {R(1), R(1), R(1)};
                seg = 0;
                s_cur++;
                alive = true;
                fresh = true;
            }
            if (fresh) {
                ud = unit(d);
                bvh_begin<R, QUANT>(q, A.sc, o, d, ud, n_nodes);
            }
            if (A.sc.bvh_n_big_leaves != 0u && __ballot(fresh) != 0ull) {
                if (fresh) {
                    for (uint32_t k = 0; k < A.sc.bvh_n_big_leaves; ++k) {
                        const uint32_t desc = A.sc.bvh_big[k];
                        sphere_tests += desc & 3u;
                        for (uint32_t j = 0; j < (desc & 3u); ++j) {
                            const d4* rec64 = (const d4*)(big_lds + (2u * k + j) * bvh_big_entry_bytes<R>());
                            const r4* brec = (const r4*)(rec64 + 2);
                            const r4 c = brec[0], v = brec[1], w3 = brec[2];
                            if (bvh_leaf_eval<R>(A.sc, q, desc, j, c, v, o, d, ud, time, A.tmin, &w3) != 0u)
                                bvh_candidate_eval<R>(q, rec64[0], rec64[1], (int)bits(v.w), o, d, time, A.tmin);
                        }
                    }
                }
            }
            // ---- back, in place: a living path READY for its walker; a path that ended with the queue dry frees its slot ----
            if (have) {
                volatile uint32_t* state = xw + 64u + 64u * w_of + slot;
                if (alive) {
                    segflags = seg | (1u << 30) | ((has_item ? 1u : 0u) << 31);
                    const unsigned long long ia = __builtin_bit_cast(unsigned long long, q.inv_a2);
                    rec[0 * NS] = f4{o.x, o.y, o.z, time};
                    rec[1 * NS] = f4{d.x, d.y, d.z, q.tbest};
                    rec[2 * NS] = f4{ud.x, ud.y, ud.z, Bits<float>::from((uint32_t)q.ibest)};
                    rec[3 * NS] = f4{thr.x, thr.y, thr.z, Bits<float>::from(segflags)};
                    rec[4 * NS] = f4{acc.x, acc.y, acc.z, Bits<float>::from(item)};
                    rec[5 * NS] = f4{Bits<float>::from((uint32_t)g.state), Bits<float>::from((uint32_t)(g.state >> 32)), Bits<float>::from((uint32_t)g.inc),
                                     Bits<float>::from((uint32_t)(g.inc >> 32))};
                    rec[6 * NS] = f4{Bits<float>::from(px), Bits<float>::from(py), Bits<float>::from(s_cur), Bits<float>::from(s_end)};
                    rec[7 * NS] = f4{q.qa.x, q.qa.y, q.qa.z, q.qb.x};
                    rec[8 * NS] = f4{q.qb.y, q.qb.z, Bits<float>::from((uint32_t)ia), Bits<float>::from((uint32_t)(ia >> 32))};
                    rec[9 * NS] = f4{q.lb.b.e1x, q.lb.b.e1z, q.lb.b.e2x, q.lb.b.e2y};
                    rec[10 * NS] = f4{q.lb.b.e2z, q.lb.b.k1, q.lb.b.k2, 0.0f};
                    asm volatile("" ::: "memory");
                    *state = kXReady;
                } else {
                    *state = kXFree;
                }
            }
            asm volatile("" ::: "memory");
            RAYZ_XPROF_T(2)
        }
#ifdef RAYZ_BVH_PROFILE
        if (lane == 0) for (int k = 0; k < 6; ++k) atomicAdd(&A.counters[12 + k], xp[k]);
#endif
    }
    unsigned long long t0 = nseg, t1 = node_tests, t2 = sphere_tests;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        t0 += __shfl_xor(t0, off);
        t2 += __shfl_xor(t2, off);
    }
    if (lane == 0) {
        atomicAdd(&A.counters[1], t0);
        atomicAdd(&A.counters[2], t1);
        atomicAdd(&A.counters[3], t2);
    }
}
#endif // RAYZ_EXPERIMENTS (exchange kernel)

// ---- pixel = (Σ_chunks partial) · (1/spp), chunk order: src/renderer.zig:94-95 ---------------------
template <class R>
__global__ __launch_bounds__(256) void resolve_kernel(const typename VecOf<R>::type* __restrict__ partial,
                                                      R* __restrict__ out, uint32_t shard_pixels,
                                                      uint32_t chunks_per_px, uint32_t spp) {
    typedef typename VecOf<R>::type r4;
    const uint32_t lp = blockIdx.x * blockDim.x + threadIdx.x;
    if (lp >= shard_pixels) return;
    R x = 0, y = 0, z = 0;
    for (uint32_t k = 0; k < chunks_per_px; ++k) {
        const r4 p = partial[(size_t)k * shard_pixels + lp];
        x = x + p.x;
        y = y + p.y;
        z = z + p.z;
    }
    const R inv = R(1) / (R)spp;
    out[3 * (size_t)lp + 0] = x * inv;
    out[3 * (size_t)lp + 1] = y * inv;
    out[3 * (size_t)lp + 2] = z * inv;
}

// ---- `writePPM`'s per-pixel transform, src/image.zig:35-38 + src/vec.zig:79-93 -------------------
__global__ __launch_bounds__(256) void tonemap_kernel(const float* __restrict__ rgb, uint8_t* __restrict__ out,
                                                      size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // in f64, as writePPM computes it on the widened pixel: bit-identical to the host writer
    const double v = (double)rgb[i];
    double s = v > 0.0 ? __builtin_sqrt(v) : 0.0;
    s = s > 0.0 ? s : 0.0; // utils.max(x, low)
    s = s < 1.0 ? s : 1.0; // utils.min(.., high)
    out[i] = (uint8_t)(s * 255.0);
}

// ---- known-answer entry (rayz_hip_kat): the kernel's own device functions on caller-supplied inputs -----------------
// One thread per record; records are RAYZ_KAT_IN_STRIDE doubles in, RAYZ_KAT_OUT_STRIDE doubles out (layouts in
// include/rayz_hip.h).  Inputs are narrowed to R exactly as the scene and camera are when they cross the ABI.
constexpr int kKatIn = 48, kKatOut = 12;
template <class R> __global__ __launch_bounds__(64) void kat_kernel(uint32_t op, const double* in, uint32_t n, double* out) {
    typedef typename VecOf<R>::type r4;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* a = in + (size_t)i * kKatIn;
    double* r = out + (size_t)i * kKatOut;
    for (int k = 0; k < kKatOut; ++k) r[k] = 0.0;
    auto v3 = [&](int k) { return V<R>{(R)a[k], (R)a[k + 1], (R)a[k + 2]}; };
    auto put3 = [&](int k, V<R> v) { r[k] = (double)v.x, r[k + 1] = (double)v.y, r[k + 2] = (double)v.z; };
    switch (op) {
    case 0: { // REFRACT: ud(3) n(3) eta -> dir(3)
        const V<R> ud = v3(0), nrm = v3(3);
        put3(0, refract<R>(ud, nrm, -dot3(ud, nrm), (R)a[6]));
        break;
    }
    case 1: // REFLECTANCE: cos, eta -> r
        r[0] = (double)reflectance<R>((R)a[0], (R)a[1]);
        break;
    case 2: { // GET_RAY: from du dv pxo defu defv (18) defocus px py n_u u[..] -> origin(3) dir(3) time draws
        DevCamera<R> cam;
        for (int k = 0; k < 3; ++k) {
            cam.from[k] = (R)a[k], cam.du[k] = (R)a[3 + k], cam.dv[k] = (R)a[6 + k], cam.pxo[k] = (R)a[9 + k];
            cam.defu[k] = (R)a[12 + k], cam.defv[k] = (R)a[15 + k];
        }
        cam.defocus = a[18] != 0.0 ? 1u : 0u;
        cam._pad = 0;
        ListRng g{a + 22, a[21] < 0.0 ? 0u : (uint32_t)a[21], 0u};
        V<R> o, d;
        R time;
        if (a[21] < 0.0) camera_ray_no_rng<R>(cam, (uint32_t)a[19], (uint32_t)a[20], o, d, time); // n_u = -1: getRay(px, py, null)
        else camera_ray<R>(cam, g, (uint32_t)a[19], (uint32_t)a[20], o, d, time);
        put3(0, o);
        put3(3, d);
        r[6] = (double)time;
        r[7] = (double)g.i;
        break;
    }
    case 3: { // BOX_HIT: lo(3) hi(3) o(3) d(3) tmin tmax ... format[26] -> hit, t_entry.  The box as the device holds it, in
              // either record format (DevScene::bvh_nodes), made by the host from the record as the scene upload makes it
              // (rayz_hip_kat): format 0: a[14..19] = 16-bit plane indices lo.xyz hi.xyz, a[20..25] = grid origin and cell size;
              // format 1: a[14..19] = the padded planes rounded outward to f32, the grid = origin 0, cell 1
        DevScene<R> g{};
        for (int k = 0; k < 3; ++k) g.bvh_glo[k] = (float)a[20 + k], g.bvh_cell[k] = (float)a[23 + k];
        BvhQuery<R> q;
        const V<R> d = v3(9);
        bvh_begin<R>(q, g, v3(6), d, unit(d), 1u);
        q.tbest = (R)a[13];
        q.tb32 = round_up_f32(q.tbest);
        float t0;
        if (a[26] != 0.0) {
            r[0] = bvh_box_hit_planes<R, false>(V<float>{(float)a[14], (float)a[15], (float)a[16]}, V<float>{(float)a[17], (float)a[18], (float)a[19]},
                                                q, round_down_f32((R)a[12]), t0) ? 1.0 : 0.0;
        } else {
            const uint32_t wx = (uint32_t)a[14] | ((uint32_t)a[17] << 16), wy = (uint32_t)a[15] | ((uint32_t)a[18] << 16),
                           wz = (uint32_t)a[16] | ((uint32_t)a[19] << 16);
            r[0] = bvh_box_hit<R>(wx, wy, wz, q, round_down_f32((R)a[12]), t0) ? 1.0 : 0.0;
        }
        r[1] = (double)t0;
        break;
    }
    case 4: { // SPHERE_HIT: c(3) v(3) radius o(3) d(3) time tmin tmax -> hit t point(3) normal(3) front filter
        const V<R> o = v3(7), d = v3(10);
        const R time = (R)a[13], tmin = (R)a[14];
        const r4 c = {(R)a[0], (R)a[1], (R)a[2], (R)a[16]}, v = {(R)a[3], (R)a[4], (R)a[5], R(0)}; // a[16]: padded r² (f32), from the host
        const V<R> ud = unit(d);
        // the reject test as the kernels run it: f32 for both precisions, the ray narrowed to f32
        const RayBasis<float> b = make_basis<float>(V<float>{(float)ud.x, (float)ud.y, (float)ud.z}, V<float>{(float)o.x, (float)o.y, (float)o.z});
        const bool cand = leaf_reject_test(b, (float)time, V<float>{(float)c.x, (float)c.y, (float)c.z},
                                           V<float>{(float)v.x, (float)v.y, (float)v.z}, (float)c.w);
        r[9] = cand ? 1.0 : 0.0;
        R tbest = (R)a[15];
        int ibest = -1;
        if (cand) {
            const double ddx = d.x, ddy = d.y, ddz = d.z;
            const double inv_a2 = 1.0 / fm(ddz, ddz, fm(ddy, ddy, ddx * ddx));
            const d4 c64 = {a[0], a[1], a[2], a[6] * a[6]}, v64 = {a[3], a[4], a[5], 0.0};
            // the reference accepts t <= tmax (src/geom.zig:56-58); the kernel's running best is exclusive with a tie
            // rule on the pool index: index 1 > the initial -1 makes a tie with tmax an accept here too
            narrow_eval<R>(c64, v64, 1, o, d, time, inv_a2, tmin, tbest, ibest);
        }
        if (ibest >= 0) {
            V<R> pt, nrm;
            sphere_hit_record<R>(c, v, o, d, time, tbest, pt, nrm);
            const bool front = face_forward<R>(d, nrm);
            r[0] = 1.0, r[1] = (double)tbest;
            put3(2, pt);
            put3(5, nrm);
            r[8] = front ? 1.0 : 0.0;
        }
        break;
    }
    case 5: { // SCATTER: kind method param o(3) d(3) point(3) normal(3) front n_u u[..] -> ok dir(3) draws
        const uint32_t kind = (uint32_t)a[0], method = (uint32_t)a[1];
        const R param = (R)a[2];
        const V<R> d = v3(6);
        ListRng g{a + 17, (uint32_t)a[16], 0u};
        V<R> nd{R(0), R(0), R(0)};
        const bool ok = scatter_dir<R>(kind, method, param, R(1) / param, g, d, unit(d), v3(9), v3(12), a[15] != 0.0, nd);
        r[0] = ok ? 1.0 : 0.0;
        put3(1, nd);
        r[4] = (double)g.i;
        break;
    }
    case 6: // CHECKER: p(3) scale -> parity
        r[0] = (double)checker_parity<R>(v3(0), (R)a[3]);
        break;
    case 7: { // BACKGROUND: d(3) -> colour(3)
        put3(0, background<R>(unit(v3(0))));
        break;
    }
    case 8: { // TRIANGLE_HIT (build-defined): v0(3) v1(3) v2(3) o(3) d(3) tmin tmax -> hit t filter>=0
        const V<R> v0 = v3(0), o = v3(9), d = v3(12);
        const V<R> e1{(R)(a[3] - a[0]), (R)(a[4] - a[1]), (R)(a[5] - a[2])}, e2{(R)(a[6] - a[0]), (R)(a[7] - a[1]), (R)(a[8] - a[2])};
        const R f = tri_filter<R>(v0, e1, e2, o, d);
        R tbest = (R)a[16];
        int ibest = -1;
        tri_accept<R>(f, v0, e1, e2, o, d, (R)a[15], 1, tbest, ibest);
        r[0] = ibest >= 0 ? 1.0 : 0.0;
        r[1] = ibest >= 0 ? (double)tbest : 0.0;
        r[2] = f >= R(0) ? 1.0 : 0.0;
        break;
    }
    case 9: { // SCAN_DISCS: one block of the flat list's scan, THROUGH the packed-FMA form the scan loop runs
              // (ScanGroup<float, cls>::discs: two spheres per v_pk_fma_f32), plus the general-velocity form of the BVH
              // leaves on the same spheres: cx[4] cy[4] cz[4] radius[4] vy[4] o(3) d(3) time cls (+ padded r²[4] at 28, from the
              // host) -> disc[4] leaf_disc[4]
        const V<R> o = v3(20), d = v3(23);
        const V<R> ud = unit(d);
        const RayBasis<float> b = make_basis<float>(V<float>{(float)ud.x, (float)ud.y, (float)ud.z}, V<float>{(float)o.x, (float)o.y, (float)o.z});
        const float ft = (float)(R)a[26];
        float out4[4];
        if (a[27] == 0.0) {
            ScanGroup<float, 0> g;
            for (int q = 0; q < 2; ++q) {
                g.cx[q] = f2{(float)a[2 * q], (float)a[2 * q + 1]}, g.cy[q] = f2{(float)a[4 + 2 * q], (float)a[5 + 2 * q]};
                g.cz[q] = f2{(float)a[8 + 2 * q], (float)a[9 + 2 * q]}, g.r2[q] = f2{(float)a[28 + 2 * q], (float)a[29 + 2 * q]};
            }
            g.discs(out4, b, ft);
        } else {
            ScanGroup<float, 1> g;
            for (int q = 0; q < 2; ++q) {
                g.cx[q] = f2{(float)a[2 * q], (float)a[2 * q + 1]}, g.cy[q] = f2{(float)a[4 + 2 * q], (float)a[5 + 2 * q]};
                g.cz[q] = f2{(float)a[8 + 2 * q], (float)a[9 + 2 * q]}, g.r2[q] = f2{(float)a[28 + 2 * q], (float)a[29 + 2 * q]};
                g.vy[q] = f2{(float)a[16 + 2 * q], (float)a[17 + 2 * q]};
            }
            g.discs(out4, b, ft);
        }
        for (int k = 0; k < 4; ++k) {
            r[k] = (double)out4[k];
            const float vy = a[27] == 0.0 ? 0.0f : (float)a[16 + k];
            const float p1 = fm(0.0f, ft * b.e1z, fm(0.0f, ft * b.e1x, basis_p1<float>(b, (float)a[k], (float)a[8 + k])));
            const float p2 = fm(0.0f, ft * b.e2z, fm(vy, ft * b.e2y, fm(0.0f, ft * b.e2x, basis_p2<float>(b, (float)a[k], (float)a[4 + k], (float)a[8 + k]))));
            r[4 + k] = (double)basis_disc<float>(p1, p2, (float)a[28 + k]);
        }
        break;
    }
    default: break;
    }
}

} // namespace rayz_dev
