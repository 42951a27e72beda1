// rayz_hip.hip — the C ABI of include/rayz_hip.h: scene upload, workspace, kernel launches.
//
// Replaces the body of `Tracer.render()` (src/renderer.zig:72-101 of jlucier/rayz).  A scene handle is bound to
// ONE device (its own context: stream, CU count); a process may drive several devices — one scene per device,
// rows dealt in interleaved tiles (params.shard_*) — either itself (rayz_hip_multi_*: one host thread, one stream
// per device, one RCCL gather of the row tiles to the first device) or as one process per GPU with the gather
// done by the caller (torch.distributed in bench.py).
#include "../../include/rayz_hip.h"
#include "rayz_device.hpp"
#include "bvh_build.hpp"

#include <rccl/rccl.h> // types and prototypes only: the library is opened with dlopen at the first multi-device call

#include <dlfcn.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <limits>
#include <mutex>
#include <algorithm>
#include <atomic>
#include <new>
#include <queue>
#include <type_traits>
#include <vector>

using namespace rayz_dev;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// No exception crosses the C ABI: every extern "C" body that can allocate runs inside guarded().
template <class F> int guarded(F&& f) noexcept {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return fail(RAYZ_ERR_OOM, "host allocation failed");
    } catch (const std::exception& e) {
        return fail(RAYZ_ERR_HIP, "unexpected exception: %s", e.what());
    } catch (...) {
        return fail(RAYZ_ERR_HIP, "unexpected exception");
    }
}

// One context per HIP device ordinal, created by rayz_hip_init(device) (or lazily by the *_on / multi entries).
struct DeviceCtx {
    bool ok = false;
    hipStream_t stream = nullptr;
    int num_cu = 0;
};
DeviceCtx g_ctx[RAYZ_MAX_DEVICES];

// Measurement knobs (rayz_hip_debug_set; they change scheduling or the walked tree, never an image).  The library reads
// no environment variable: a stray one cannot change a production render.  -1 = the built-in default.
struct Tuning {
    std::atomic<long long> v[RAYZ_DEBUG_KNOBS];
    Tuning() { for (auto& x : v) x.store(-1, std::memory_order_relaxed); }
};
Tuning g_tune; // written by rayz_hip_debug_set, read (once per knob) by the render / scene build that starts next
long long tuning(int knob, long long dflt) {
    const long long x = g_tune.v[knob].load(std::memory_order_relaxed);
    return x < 0 ? dflt : x;
}
int g_default = -1; // device of the last successful rayz_hip_init: what entry points without a device argument use
std::mutex g_mu;    // guards g_ctx / g_default

// HIP's current device is per host thread: every entry point that touches a device selects it and restores the
// caller's on return (the host may be torch, with its own idea of the current device).
struct DeviceScope {
    int prev = -1, dev;
    explicit DeviceScope(int d) : dev(d) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceScope() {
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};

#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return fail(e_ == hipErrorOutOfMemory ? RAYZ_ERR_OOM : RAYZ_ERR_HIP, "%s: %s (%s:%d)", #expr,     \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                         \
    } while (0)

// Device copy of the scene in one precision: scan streams + pool-indexed shading tables.
template <class R> struct SceneBuffers {
    typedef typename VecOf<R>::type r4;
    float* stat = nullptr;  // blocks of G = 4 spheres: cx[G] cy[G] cz[G] r²[G]   (scan streams: f32 for both precisions)
    float* movy = nullptr;  // blocks of G: cx[G] cy[G] cz[G] r²[G] vy[G]
    f4* movg = nullptr;
    r4* sph_pool = nullptr;
    r4* mat = nullptr;
    r4* tex = nullptr;
    r4* tri = nullptr;
    u4* bvh_nodes = nullptr; // BVH traversal only (f32 planes or 16-bit plane indices, for both precisions: the box test only culls)
    bool quantized = false;  // .. which: DevScene::bvh_nodes
    r4* bvh_leaf = nullptr;
    uint32_t nt_pad = 0, bvh_leaf_stride = 2, bvh_n_inner = 0;
    uint32_t n_big_leaves = 0, big_desc[4] = {0, 0, 0, 0};
    uint32_t bvh_top = 0; // inner-node records the BVH kernel copies to LDS
    rayz_bvh::PlaneGrid grid; // the grid the node records' 16-bit plane indices live on (f32 planes: origin 0, cell 1)
    double pad_S = 0; // the origin bound S the filter radii of these buffers were padded for
    bool ready = false, bvh_ready = false;
    void release() {
        (void)hipFree(tri);
        (void)hipFree(bvh_nodes);
        (void)hipFree(bvh_leaf);
        tri = bvh_leaf = nullptr;
        bvh_nodes = nullptr;
        bvh_ready = false;
        (void)hipFree(stat);
        (void)hipFree(movy);
        (void)hipFree(movg);
        (void)hipFree(sph_pool);
        (void)hipFree(mat);
        (void)hipFree(tex);
        sph_pool = mat = tex = nullptr;
        movg = nullptr;
        stat = movy = nullptr;
        ready = false;
    }
};

// The pool's own f64 records in slot order (narrow phase) + slot → pool index; shared by both precisions.
struct NarrowBuffers {
    d4* slot64 = nullptr;
    uint32_t* slot_pool = nullptr;
    d4* bvh_sph64 = nullptr; // leaf order (BVH traversal only)
    uint32_t ns_pad = 0, ny_pad = 0, ng_pad = 0;
    bool ready = false, bvh_ready = false;
    void release() {
        (void)hipFree(bvh_sph64);
        bvh_sph64 = nullptr;
        bvh_ready = false;
        (void)hipFree(slot64);
        (void)hipFree(slot_pool);
        slot64 = nullptr;
        slot_pool = nullptr;
        ready = false;
    }
};

uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

// ---- chunk schedule: which samples of a pixel are summed together ------------------------------
// params.chunk_spp != 0: uniform chunks of that many samples (the last one shorter).  0 = automatic: frames of fewer
// than 2^19 pixels, or fewer than 64 samples per pixel, use uniform chunks of 16; larger renders use chunks of C samples
// while at least 2 C remain, and split the rest by halving down to 16 (.. C, C/2, C/4, .., 16, 16): the work queue — which
// hands out chunk 0 of every pixel, then chunk 1, .. — ends in SHORT items, so no lane is left with a long item while the
// others have run dry.  C (auto_chunk) is sized for the frame being DEALT TO 8 GPUs (round 4): no work item larger than 1/8
// of what a lane of a 2^18-lane GPU gets of an 8-way deal, C = pixels · spp / 2^24 held to [64, 256] (a power of two, at most
// spp / 2) — 64 for 1920x1080x1024, 256 for 3840x2160x4096.  Measured (profiles/r04/multi/): with C = 256 one GPU's share of
// the 1080p frame runs at 80 - 87 % of the whole-frame rate (≈1 pixel per lane: three items of a quarter of a lane's work
// each, nothing left to balance with), with 64 at 94 %; the whole frame on ONE GPU changes by +1.0 % (flat list) / −1.2 % (BVH).
// The price is partial sums: 18 per pixel instead of 8 at 1024 spp.  Depends on the full frame's size, never on the shard or
// the GPU count: the image is the same for every deal.
uint32_t auto_chunk(uint64_t pixels, uint32_t spp) {
    auto pow2floor = [](uint64_t v) {
        uint64_t r = 1;
        while (r <= v / 2) r *= 2;
        return r;
    };
    const uint64_t share = pixels >= (1ull << 32) ? 256 : (pixels * spp) >> 24;
    long long cap = (long long)pow2floor(std::min<uint64_t>(256, std::max<uint64_t>(64, share)));
#ifdef RAYZ_EXPERIMENTS // tools/chunk_cap_sweep.py only: CHANGES the summation tree (the oracle does not follow it)
    cap = tuning(RAYZ_DEBUG_CHUNK_CAP, cap);
#endif
    return (uint32_t)std::min<uint64_t>((uint64_t)cap, pow2floor(spp / 2));
}
void chunk_schedule(const RayzRenderParams* p, std::vector<uint32_t>& starts) {
    const uint32_t spp = p->samples_per_px;
    starts.clear();
    starts.push_back(0);
    const bool uniform = p->chunk_spp != 0 || (uint64_t)p->width * p->height < (1ull << 19) || spp < 64;
    if (uniform) {
        const uint32_t c = p->chunk_spp ? p->chunk_spp : 16u;
        for (uint64_t s0 = c; s0 < spp; s0 += c) starts.push_back((uint32_t)s0);
        starts.push_back(spp);
        return;
    }
    auto pow2floor = [](uint32_t v) {
        uint32_t r = 1;
        while (r <= v / 2) r *= 2;
        return r;
    };
    const uint32_t C = auto_chunk((uint64_t)p->width * p->height, spp);
    uint32_t at = 0, rem = spp;
    while (rem >= 2 * C) at += C, rem -= C, starts.push_back(at);
    while (rem > 16) {
        const uint32_t c = std::max(16u, pow2floor(rem / 2));
        at += c, rem -= c, starts.push_back(at);
    }
    if (rem) starts.push_back(at + rem);
}

// Number of chunks of that schedule, without building it (a pixel may have at most kMaxChunksPerPx: the table is a host
// vector, a device array and the divisor of every work item's index).
constexpr uint64_t kMaxChunksPerPx = 1ull << 20;
uint64_t chunk_count(const RayzRenderParams* p) {
    const uint64_t spp = p->samples_per_px;
    const bool uniform = p->chunk_spp != 0 || (uint64_t)p->width * p->height < (1ull << 19) || spp < 64;
    if (uniform) {
        const uint64_t c = p->chunk_spp ? p->chunk_spp : 16u;
        return (spp + c - 1) / c;
    }
    // the automatic schedule, counted exactly by chunk_schedule's own rule (the full chunks in closed form, the halving
    // tail by its ≤ 10 steps): `spp / 256 + 8` undercounted tails of 256 .. 511 samples by one (spp = 497: 10 chunks)
    auto pow2floor = [](uint64_t v) {
        uint64_t r = 1;
        while (r <= v / 2) r *= 2;
        return r;
    };
    const uint64_t C = auto_chunk((uint64_t)p->width * p->height, (uint32_t)spp);
    uint64_t n = spp >= 2 * C ? (spp - 2 * C) / C + 1 : 0, rem = spp - n * C;
    while (rem > 16) rem -= std::max<uint64_t>(16, pow2floor(rem / 2)), ++n;
    return n + (rem ? 1 : 0);
}

double norm3(const double* v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

// ---- conservative reject filter ---------------------------------------------------------------
// The scan and the BVH leaves test r_pad² − p1² − p2² ≥ 0 in R with r_pad = r + E, E = 32·u·(|c| + |v| + r + S):
// u = unit roundoff of R, S = a bound on |o| of every ray the render can produce (camera lens, hit points on any
// hittable).  The padded square is rounded UP to R.  Candidates are decided by the f64 narrow phase, so the padding
// changes no image — it only guarantees that no sphere with an f64 discriminant ≥ 0 is filtered out.
template <class R> constexpr double unit_roundoff() { return sizeof(R) == 4 ? 5.9604644775390625e-08 : 1.1102230246251565e-16; }
// The scan streams' filter runs in f32 for both precisions.  For R = double the ray reaches it narrowed to f32 (origin,
// unit direction, time: ≤ u·S + u·(|c| + S) + u·|v| more on the line's distance to the centre), so its pad is 40u, not 32u.
template <class R> float pad_radius2_scan(const RayzSphere& q, double S) {
    const double E = (sizeof(R) == 4 ? 32.0 : 40.0) * unit_roundoff<float>() *
                     (norm3(q.center) + norm3(q.velocity) + std::fabs(q.radius) + S);
    const double rp = std::fabs(q.radius) + E;
    return rayz_bvh::roundUp<float>(rp * rp);
}
double scene_origin_bound(const RayzScene* s);
double camera_origin_bound(const RayzCameraDesc* c) { return norm3(c->look_from) + norm3(c->defocus_u) + norm3(c->defocus_v); }

} // namespace

struct RayzScene {
    int device = -1; // HIP ordinal this scene's buffers live on; bound at creation (_on) or at the first render
    double origin_bound = -1; // max |hit point| over the pool (lazily)
    std::vector<RayzSphere> spheres;
    std::vector<RayzMaterial> materials;
    std::vector<RayzTexture> textures;
    std::vector<RayzTriangle> triangles;
    SceneBuffers<float> f32;
    SceneBuffers<double> f64;
    NarrowBuffers narrow;
    std::vector<uint32_t> cls[3]; // pool indices by velocity class: static, mov-Y, mov-G (pool order inside)
    rayz_bvh::FlatBvh bvh;        // host build of the reference's BVH (lazily, first BVH render / export)
    bool bvh_built = false;
    rayz_bvh::FlatBvh bvh_dev;    // the tree the GPU walks: the same build with the oversized hittables kept out (bvh_build.hpp)
    bool bvh_dev_built = false;
    void* partial = nullptr; // chunk sums, grow-only
    size_t partial_bytes = 0;
    uint32_t* chunk_start = nullptr; // device copy of the chunk schedule of the last render
    size_t chunk_start_cap = 0;
    std::vector<uint32_t> chunk_start_host;
    unsigned long long* counters = nullptr; // [0] queue head, [1] segments
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t last_stream = nullptr;
    bool rendered = false, last_bvh = false, last_two_paths = false, last_exchange = false;
    RayzRenderStats last{};
};

namespace {

// Every scattered ray starts at a hit point: on a sphere (|p| ≤ |c| + |v| + r, t ∈ [0,1)) or on a triangle.
double scene_origin_bound(const RayzScene* s) {
    double S = 0;
    for (const RayzSphere& q : s->spheres) S = std::max(S, norm3(q.center) + norm3(q.velocity) + std::fabs(q.radius));
    for (const RayzTriangle& q : s->triangles) S = std::max({S, norm3(q.v0), norm3(q.v1), norm3(q.v2)});
    return S * (1.0 + 1e-3);
}

template <class T> hipError_t put(T** dst, const std::vector<T>& v) {
    const size_t bytes = v.size() * sizeof(T);
    hipError_t e = hipMalloc((void**)dst, bytes ? bytes : 16);
    if (e != hipSuccess) return e;
    return bytes ? hipMemcpy(*dst, v.data(), bytes, hipMemcpyHostToDevice) : hipSuccess;
}

// velocity class of a pool sphere: 0 static, 1 v = (0, vy, 0), 2 anything else
int velocity_class(const RayzSphere& q) {
    const bool x = q.velocity[0] != 0, y = q.velocity[1] != 0, z = q.velocity[2] != 0;
    if (!x && !y && !z) return 0;
    return (!x && !z) ? 1 : 2;
}

void classify(RayzScene* s) {
    for (auto& c : s->cls) c.clear();
    for (uint32_t i = 0; i < s->spheres.size(); ++i) s->cls[velocity_class(s->spheres[i])].push_back(i);
}

// scanned length of a stream (whole group pairs) and its allocated length (+ two spare groups for the prefetch)
uint32_t scan_len(size_t n, uint32_t group) { return round_up((uint32_t)n, 2 * group); }
uint32_t stream_len(size_t n, uint32_t group) { return scan_len(n, group) + 2 * group; }

int upload_narrow_body(RayzScene* s) {
    NarrowBuffers& nb = s->narrow;
    classify(s);
    // slot numbering is shared by both precisions: pad to the larger (f32) group size
    nb.ns_pad = scan_len(s->cls[0].size(), kStaticGroup);
    nb.ny_pad = scan_len(s->cls[1].size(), kMovYGroup);
    nb.ng_pad = scan_len(s->cls[2].size(), kMovGGroup);
    const size_t slots = (size_t)nb.ns_pad + nb.ny_pad + nb.ng_pad;
    std::vector<d4> slot64(2 * slots, d4{0, 0, 0, 0});
    std::vector<uint32_t> slot_pool(slots, 0u);
    auto place = [&](size_t slot, uint32_t pool) {
        const RayzSphere& q = s->spheres[pool];
        slot64[2 * slot] = d4{q.center[0], q.center[1], q.center[2], q.radius * q.radius}; // radius * radius in f64, src/geom.zig:45
        slot64[2 * slot + 1] = d4{q.velocity[0], q.velocity[1], q.velocity[2], 0.0};
        slot_pool[slot] = pool;
    };
    for (size_t k = 0; k < s->cls[0].size(); ++k) place(k, s->cls[0][k]);
    for (size_t k = 0; k < s->cls[1].size(); ++k) place(nb.ns_pad + k, s->cls[1][k]);
    for (size_t k = 0; k < s->cls[2].size(); ++k) place((size_t)nb.ns_pad + nb.ny_pad + k, s->cls[2][k]);
    HIP_TRY(put(&nb.slot64, slot64));
    HIP_TRY(put(&nb.slot_pool, slot_pool));
    nb.ready = true;
    return RAYZ_OK;
}

// A failed upload leaves nothing behind: the partly filled buffer set is released, so a retry starts clean.
int upload_narrow(RayzScene* s) {
    if (s->narrow.ready) return RAYZ_OK;
    const int rc = upload_narrow_body(s);
    if (rc != RAYZ_OK) s->narrow.release();
    return rc;
}

template <class R> int upload_body(RayzScene* s, SceneBuffers<R>& b) {
    typedef typename VecOf<R>::type r4;
    int rc = upload_narrow(s);
    if (rc != RAYZ_OK) return rc;
    auto rec = [&](uint32_t pool) { // w = the PADDED r² of the conservative filter
        const RayzSphere& q = s->spheres[pool];
        return r4{(R)q.center[0], (R)q.center[1], (R)q.center[2], (R)pad_radius2_scan<R>(q, b.pad_S)};
    };
    // static / mov-Y streams (f32 for both precisions): blocks of G spheres, SoA inside a block (field f of sphere k of
    // block g at g·F·G + f·G + k), pad spheres {0, 0, 0, r² = -inf, vy = 0}
    const float ninf32 = -std::numeric_limits<float>::infinity();
    auto rec32 = [&](uint32_t pool) { // w = the PADDED r² of the conservative filter
        const RayzSphere& q = s->spheres[pool];
        return f4{(float)q.center[0], (float)q.center[1], (float)q.center[2], pad_radius2_scan<R>(q, b.pad_S)};
    };
    auto blocks = [&](const std::vector<uint32_t>& cls, uint32_t G, uint32_t F, uint32_t scanned) {
        const uint32_t n = scanned + 2 * G; // the scanned slots + two spare groups
        std::vector<float> v((size_t)n * F, 0.0f);
        for (uint32_t k = 0; k < n; ++k) v[(size_t)(k / G) * F * G + 3 * G + k % G] = ninf32;
        for (size_t k = 0; k < cls.size(); ++k) {
            const RayzSphere& q = s->spheres[cls[k]];
            const f4 c = rec32(cls[k]);
            float* blk = v.data() + (k / G) * F * G + k % G;
            blk[0] = c.x, blk[G] = c.y, blk[2 * G] = c.z, blk[3 * G] = c.w;
            if (F == 5) blk[4 * G] = (float)q.velocity[1];
        }
        return v;
    };
    const std::vector<float> stat = blocks(s->cls[0], group_size<float>(), 4, s->narrow.ns_pad),
                             movy = blocks(s->cls[1], group_size<float>(), 5, s->narrow.ny_pad);
    std::vector<f4> movg(2 * (size_t)stream_len(s->cls[2].size(), kMovGGroup), f4{0.0f, 0.0f, 0.0f, 0.0f});
    for (size_t k = 0; k < movg.size(); k += 2) movg[k] = f4{0.0f, 0.0f, 0.0f, ninf32};
    for (size_t k = 0; k < s->cls[2].size(); ++k) {
        const RayzSphere& q = s->spheres[s->cls[2][k]];
        movg[2 * k] = rec32(s->cls[2][k]);
        movg[2 * k + 1] = f4{(float)q.velocity[0], (float)q.velocity[1], (float)q.velocity[2], 0.0f};
    }
    // triangles: {v0, bits(material)}, {e1, 0}, {e2, 0}; edges subtracted in f64, then narrowed
    b.nt_pad = scan_len(s->triangles.size(), kTriGroup);
    std::vector<r4> tri(3 * (size_t)(b.nt_pad + kTriGroup), r4{R(0), R(0), R(0), R(0)});
    for (size_t k = 0; k < s->triangles.size(); ++k) {
        const RayzTriangle& q = s->triangles[k];
        tri[3 * k] = r4{(R)q.v0[0], (R)q.v0[1], (R)q.v0[2], Bits<R>::from(q.material)};
        tri[3 * k + 1] = r4{(R)(q.v1[0] - q.v0[0]), (R)(q.v1[1] - q.v0[1]), (R)(q.v1[2] - q.v0[2]), R(0)};
        tri[3 * k + 2] = r4{(R)(q.v2[0] - q.v0[0]), (R)(q.v2[1] - q.v0[1]), (R)(q.v2[2] - q.v0[2]), R(0)};
    }
    HIP_TRY(put(&b.tri, tri));
    std::vector<r4> sph_pool, mat, tex;
    for (uint32_t i = 0; i < s->spheres.size(); ++i) {
        const RayzSphere& q = s->spheres[i];
        sph_pool.push_back(rec(i));
        sph_pool.push_back(r4{(R)q.velocity[0], (R)q.velocity[1], (R)q.velocity[2], Bits<R>::from(q.material)});
    }
    for (const RayzMaterial& m : s->materials) {
        const R p = (R)m.param;
        mat.push_back(r4{Bits<R>::from(m.kind | (m.method << 8)), Bits<R>::from(m.texture), p, R(1) / p});
    }
    for (const RayzTexture& t : s->textures) {
        tex.push_back(r4{Bits<R>::from(t.kind), Bits<R>::from(t.even), Bits<R>::from(t.odd), (R)t.scale});
        tex.push_back(r4{(R)t.color[0], (R)t.color[1], (R)t.color[2], R(0)});
    }
    HIP_TRY(put(&b.stat, stat));
    HIP_TRY(put(&b.movy, movy));
    HIP_TRY(put(&b.movg, movg));
    HIP_TRY(put(&b.sph_pool, sph_pool));
    HIP_TRY(put(&b.mat, mat));
    HIP_TRY(put(&b.tex, tex));
    b.ready = true;
    return RAYZ_OK;
}

// `S` = the origin bound this render needs.  Buffers padded for a smaller bound are rebuilt (for twice the bound, so
// that a moving camera does not rebuild every frame); the padding changes no image.
template <class R> int upload(RayzScene* s, SceneBuffers<R>& b, double S) {
    if (b.ready && b.pad_S >= S) return RAYZ_OK;
    const bool again = b.ready;
    if (again) HIP_TRY(hipDeviceSynchronize());
    b.release();
    b.pad_S = again ? 2.0 * S : S;
    const int rc = upload_body<R>(s, b);
    if (rc != RAYZ_OK) b.release();
    return rc;
}

void ensure_bvh(RayzScene* s) {
    if (!s->bvh_built) {
        s->bvh = rayz_bvh::build(s->spheres, s->triangles); // replaces initHittables + bvh.build, src/renderer.zig:76-78
        s->bvh_built = true;
    }
}

void ensure_bvh_dev(RayzScene* s) {
    if (!s->bvh_dev_built) {
        // RAYZ_DEBUG_BVH_PEEL = 0 (measurement only): walk the reference's full tree
        s->bvh_dev = rayz_bvh::build(s->spheres, s->triangles, tuning(RAYZ_DEBUG_BVH_PEEL, 1) != 0, tuning(RAYZ_DEBUG_BVH_SPLIT, 0) == 0);
        s->bvh_dev_built = true;
    }
}

template <class R> int upload_bvh_body(RayzScene* s, SceneBuffers<R>& b) {
    typedef typename VecOf<R>::type r4;
    ensure_bvh_dev(s);
    const rayz_bvh::FlatBvh& t = s->bvh_dev;
    const uint32_t ns = (uint32_t)s->spheres.size();
    // leaf-order slots: the tree's hittables, then the oversized ones kept out of it
    std::vector<uint32_t> slots(t.order);
    slots.insert(slots.end(), t.big.begin(), t.big.end());
    b.n_big_leaves = 0;
    for (size_t k = 0; k < t.big.size(); k += 2) {
        const uint32_t first = (uint32_t)(t.order.size() + k), count = (uint32_t)std::min<size_t>(2, t.big.size() - k);
        uint32_t desc = (first << 4) | count;
        for (uint32_t j = 0; j < count; ++j)
            if (t.big[k + j] >= ns) desc |= 1u << (2 + j);
        b.big_desc[b.n_big_leaves++] = desc;
    }
    if (!s->narrow.bvh_ready) {
        std::vector<d4> sph64;
        for (uint32_t prim : slots) {
            if (prim < ns) {
                const RayzSphere& q = s->spheres[prim];
                sph64.push_back(d4{q.center[0], q.center[1], q.center[2], q.radius * q.radius});
                sph64.push_back(d4{q.velocity[0], q.velocity[1], q.velocity[2], 0.0});
            } else { // triangle slot: unused by the narrow phase
                sph64.push_back(d4{0, 0, 0, 0});
                sph64.push_back(d4{0, 0, 0, 0});
            }
        }
        HIP_TRY(put(&s->narrow.bvh_sph64, sph64));
        s->narrow.bvh_ready = true;
    }
    if (b.bvh_ready) return RAYZ_OK;
    b.bvh_leaf_stride = s->triangles.empty() ? 2u : 3u;
    // one record per INNER node holding its two children's boxes (narrowed outward to f32: never smaller than the f64
    // box) + in lo.w where each child leads: an inner index, or kBvhLeafFlag | leaf descriptor
    // (first << 4 | type1 << 3 | type0 << 2 | count)
    std::vector<u4> nodes;
    std::vector<r4> leaf;
    // the inner nodes the kernel keeps in LDS (the tree's "top") are numbered first, the rest in pre-order
    std::vector<uint32_t> inner_index(t.nodes.size(), 0xffffffffu), inner_order;
    uint32_t n_inner = 0;
    {
        // top-of-tree records kept in LDS: as many as fit beside the stacks of the one-path kernel's workgroup (the two-path
        // kernel, with its smaller workgroups, keeps a prefix of them); RAYZ_DEBUG_BVH_TOP lowers the cap
        const size_t stacks = ((size_t)t.depth + 3) * kBvhWg * sizeof(uint32_t) + (t.big.empty() ? 0 : kBvhBigLdsBytes); // (+ the oversized hittables' records)
        const size_t lds_for_top = stacks < kBvhLdsBudget ? kBvhLdsBudget - stacks : 0;
        // WHICH record format (DevScene::bvh_nodes): 16-bit plane indices halve the bytes a step fetches and double the
        // records the LDS top holds, for 12 conversions per step — worth it only when most steps fetch from global memory,
        // i.e. for a tree much larger than the f32 top (measured: profiles/r03/lds_top).  RAYZ_DEBUG_BVH_NODES forces one.
        size_t inner_total = 0;
        for (const rayz_bvh::FlatNode& n : t.nodes) inner_total += n.count == 0;
        const long long format = tuning(RAYZ_DEBUG_BVH_NODES, 0);
        b.quantized = format == 2 || (format == 0 && inner_total > kQuantizeAboveTops * (lds_for_top / 64));
        const uint32_t fit = (uint32_t)(lds_for_top / (b.quantized ? 32 : 64));
        const uint32_t top_cap = (uint32_t)std::min<long long>(tuning(RAYZ_DEBUG_BVH_TOP, fit), fit);
        // WHICH records: grown from the root, always taking the candidate whose box has the largest surface area next — the
        // chance that a ray visits a node goes with its box's area, and never exceeds its parent's (RAYZ_DEBUG_BVH_TOP_ORDER
        // = 1: plain breadth-first, the order of rounds 2-3a)
        auto area = [&](size_t i) {
            const rayz_bvh::Box& x = t.nodes[i].box;
            const double dx = x.hi[0] - x.lo[0], dy = x.hi[1] - x.lo[1], dz = x.hi[2] - x.lo[2];
            return dx * dy + dy * dz + dz * dx;
        };
        const bool by_area = tuning(RAYZ_DEBUG_BVH_TOP_ORDER, 0) == 0;
        typedef std::pair<double, size_t> Cand; // (priority, node): largest first; breadth-first = decreasing sequence numbers
        std::priority_queue<Cand> frontier;
        double seq = 0;
        if (!t.nodes.empty() && t.nodes[0].count == 0) frontier.push({by_area ? area(0) : seq--, 0});
        while (!frontier.empty() && n_inner < top_cap) {
            const size_t i = frontier.top().second;
            frontier.pop();
            inner_index[i] = n_inner++;
            inner_order.push_back((uint32_t)i);
            for (size_t c : {i + 1, (size_t)t.nodes[i + 1].skip})
                if (t.nodes[c].count == 0) frontier.push({by_area ? area(c) : seq--, c});
        }
        b.bvh_top = n_inner;
        for (size_t i = 0; i < t.nodes.size(); ++i)
            if (t.nodes[i].count == 0 && inner_index[i] == 0xffffffffu) {
                inner_index[i] = n_inner++;
                inner_order.push_back((uint32_t)i);
            }
    }
    // the boxes go to the device PADDED: the slab test carries no slack of its own (rayz_device.hpp: bvh_box_hit;
    // E = 16u·max(S, B) for f32 planes rounded outward, 16u·(max(S, B) + X) for plane indices on a grid of extent X)
    double box_B = 0;
    rayz_bvh::Box all;
    for (const rayz_bvh::FlatNode& n : t.nodes) {
        all.enclose(n.box);
        for (int k = 0; k < 3; ++k) box_B = std::max({box_B, std::fabs(n.box.lo[k]), std::fabs(n.box.hi[k])});
    }
    if (t.nodes.empty())
        for (int k = 0; k < 3; ++k) all.lo[k] = all.hi[k] = 0;
    double box_pad = kBoxPadUlps * unit_roundoff<float>() * std::max(b.pad_S, box_B);
    b.grid = rayz_bvh::PlaneGrid{}; // origin 0, cell 1: a plane is its own index
    if (b.quantized) {
        box_pad = kBoxPadUlps * unit_roundoff<float>() * (std::max(b.pad_S, box_B) + 2.0 * box_B);
        b.grid = rayz_bvh::PlaneGrid::over(all.lo, all.hi, 2.0 * box_pad);
        box_pad = kBoxPadUlps * unit_roundoff<float>() * (std::max(b.pad_S, box_B) + b.grid.extent); // (extent <= 2 B + 4 pad: within the margin)
    }
    auto leaf_info = [&](const rayz_bvh::FlatNode& n) {
        uint32_t info = (n.first << 4) | n.count;
        for (uint32_t k = 0; k < n.count; ++k)
            if (t.order[n.first + k] >= ns) info |= 1u << (2 + k);
        return info;
    };
    auto fbits = [](float f) {
        uint32_t w;
        std::memcpy(&w, &f, 4);
        return w;
    };
    auto child = [&](size_t c) {
        const rayz_bvh::FlatNode& n = t.nodes[c];
        const bool is_leaf = n.count != 0;
        const uint32_t ref = is_leaf ? (kBvhLeafFlag | leaf_info(n)) : (inner_index[c] << (b.quantized ? 5 : 6)); // inner: byte offset
        if (b.quantized) {
            uint32_t w[3];
            b.grid.quantize(n.box, box_pad, w);
            nodes.push_back(u4{w[0], w[1], w[2], ref});
        } else {
            nodes.push_back(u4{fbits(rayz_bvh::roundDown<float>(n.box.lo[0] - box_pad)), fbits(rayz_bvh::roundDown<float>(n.box.lo[1] - box_pad)),
                               fbits(rayz_bvh::roundDown<float>(n.box.lo[2] - box_pad)), ref});
            nodes.push_back(u4{fbits(rayz_bvh::roundUp<float>(n.box.hi[0] + box_pad)), fbits(rayz_bvh::roundUp<float>(n.box.hi[1] + box_pad)),
                               fbits(rayz_bvh::roundUp<float>(n.box.hi[2] + box_pad)), 0u});
        }
    };
    if (!t.nodes.empty() && t.nodes[0].count != 0) { // the whole pool fits one leaf: a root record whose two child
        child(0);                                     // slots both name it (the repeat cannot change the result)
        child(0);
        n_inner = 1;
    } else {
        for (uint32_t i : inner_order) { // records in index order
            child((size_t)i + 1);             // left child follows its parent in pre-order
            child(t.nodes[i + 1].skip);       // right child = where the left subtree ends
        }
    }
    if (n_inner >= (1u << 25)) // the walk addresses a record by a 32-bit byte offset (index << 6)
        return fail(RAYZ_ERR_BAD_ARG, "BVH of %u inner nodes exceeds the device layout (2^25)", n_inner);
    b.bvh_n_inner = t.nodes.empty() ? 0u : n_inner;
    for (uint32_t prim : slots) {
        if (prim < ns) {
            const RayzSphere& q = s->spheres[prim];
            leaf.push_back(r4{(R)q.center[0], (R)q.center[1], (R)q.center[2], (R)pad_radius2_scan<R>(q, b.pad_S)});
            leaf.push_back(r4{(R)q.velocity[0], (R)q.velocity[1], (R)q.velocity[2], Bits<R>::from(prim)});
            if (b.bvh_leaf_stride == 3) leaf.push_back(r4{R(0), R(0), R(0), R(0)});
        } else {
            const RayzTriangle& q = s->triangles[prim - ns];
            leaf.push_back(r4{(R)q.v0[0], (R)q.v0[1], (R)q.v0[2], Bits<R>::from(prim)});
            leaf.push_back(r4{(R)(q.v1[0] - q.v0[0]), (R)(q.v1[1] - q.v0[1]), (R)(q.v1[2] - q.v0[2]), R(0)});
            leaf.push_back(r4{(R)(q.v2[0] - q.v0[0]), (R)(q.v2[1] - q.v0[1]), (R)(q.v2[2] - q.v0[2]), R(0)});
        }
    }
    HIP_TRY(put(&b.bvh_nodes, nodes));
    HIP_TRY(put(&b.bvh_leaf, leaf));
    b.bvh_ready = true;
    return RAYZ_OK;
}

template <class R> int upload_bvh(RayzScene* s, SceneBuffers<R>& b) {
    if (b.bvh_ready && s->narrow.bvh_ready) return RAYZ_OK;
    const int rc = upload_bvh_body<R>(s, b);
    if (rc != RAYZ_OK) { // drop the partly built BVH buffers (the scan streams stay valid)
        (void)hipFree(b.bvh_nodes);
        (void)hipFree(b.bvh_leaf);
        b.bvh_nodes = nullptr;
        b.bvh_leaf = nullptr;
        b.bvh_ready = false;
        if (!s->narrow.bvh_ready) {
            (void)hipFree(s->narrow.bvh_sph64);
            s->narrow.bvh_sph64 = nullptr;
        }
    }
    return rc;
}

// Nesting depth of every texture (solid = 1, checker = 1 + deeper child); 0 marks a cycle.  The device walks a
// checker chain with a bounded loop (kMaxTextureDepth lookups) where the reference recurses without a limit
// (src/material.zig:36-37): a pool the loop cannot resolve is refused here instead of rendering black.
int texture_depths(const RayzSceneDesc* d, std::vector<uint32_t>& depth) {
    const uint32_t n = d->n_textures;
    depth.assign(n, 0u);
    std::vector<uint8_t> state(n, 0); // 0 unvisited, 1 on the stack, 2 done
    std::vector<uint32_t> stack;
    for (uint32_t root = 0; root < n; ++root) {
        if (state[root]) continue;
        stack.push_back(root);
        while (!stack.empty()) {
            const uint32_t i = stack.back();
            const RayzTexture& t = d->textures[i];
            if (t.kind == RAYZ_TEX_SOLID) {
                depth[i] = 1, state[i] = 2;
                stack.pop_back();
                continue;
            }
            if (state[i] == 0) {
                state[i] = 1;
                bool pushed = false;
                for (uint32_t c : {t.even, t.odd}) {
                    if (state[c] == 1) return fail(RAYZ_ERR_BAD_ARG, "texture %u: checker chain contains a cycle (through %u)", i, c);
                    if (state[c] == 0) stack.push_back(c), pushed = true;
                }
                if (pushed) continue;
            }
            // both children done (or were done already)
            if (state[t.even] != 2 || state[t.odd] != 2) { // a child is still on the stack below us: a cycle
                return fail(RAYZ_ERR_BAD_ARG, "texture %u: checker chain contains a cycle", i);
            }
            depth[i] = 1 + (depth[t.even] > depth[t.odd] ? depth[t.even] : depth[t.odd]);
            state[i] = 2;
            stack.pop_back();
        }
    }
    return RAYZ_OK;
}

int validate_scene(const RayzSceneDesc* d) {
    if (!d) return fail(RAYZ_ERR_BAD_ARG, "scene is null");
    if ((d->n_spheres && !d->spheres) || (d->n_materials && !d->materials) || (d->n_textures && !d->textures) ||
        (d->n_triangles && !d->triangles))
        return fail(RAYZ_ERR_BAD_ARG, "scene list pointer is null");
    for (uint32_t i = 0; i < d->n_triangles; ++i)
        if (d->triangles[i].material >= d->n_materials)
            return fail(RAYZ_ERR_BAD_ARG, "triangle %u: material handle %u out of range", i, d->triangles[i].material);
    for (uint32_t i = 0; i < d->n_textures; ++i) {
        const RayzTexture& t = d->textures[i];
        if (t.kind > RAYZ_TEX_SOLID) return fail(RAYZ_ERR_BAD_ARG, "texture %u: bad kind %u", i, t.kind);
        if (t.kind == RAYZ_TEX_CHECKER && (t.even >= d->n_textures || t.odd >= d->n_textures))
            return fail(RAYZ_ERR_BAD_ARG, "texture %u: checker handle out of range", i);
    }
    {
        std::vector<uint32_t> depth;
        const int rc = texture_depths(d, depth);
        if (rc != RAYZ_OK) return rc;
        for (uint32_t i = 0; i < d->n_textures; ++i)
            if (depth[i] > (uint32_t)kMaxTextureDepth)
                return fail(RAYZ_ERR_BAD_ARG, "texture %u: checker nesting depth %u exceeds the device limit %d", i, depth[i],
                            kMaxTextureDepth);
    }
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const RayzMaterial& m = d->materials[i];
        if (m.kind > RAYZ_MAT_DIELECTRIC) return fail(RAYZ_ERR_BAD_ARG, "material %u: bad kind %u", i, m.kind);
        if (m.kind != RAYZ_MAT_DIELECTRIC && m.texture >= d->n_textures)
            return fail(RAYZ_ERR_BAD_ARG, "material %u: texture handle %u out of range", i, m.texture);
        if (m.kind == RAYZ_MAT_DIFFUSE && m.method > RAYZ_DIFFUSE_HEMISPHERE)
            return fail(RAYZ_ERR_BAD_ARG, "material %u: bad diffuse method %u", i, m.method);
    }
    for (uint32_t i = 0; i < d->n_spheres; ++i)
        if (d->spheres[i].material >= d->n_materials)
            return fail(RAYZ_ERR_BAD_ARG, "sphere %u: material handle %u out of range", i, d->spheres[i].material);
    return RAYZ_OK;
}

int validate_params(const RayzRenderParams* p) {
    if (!p) return fail(RAYZ_ERR_BAD_ARG, "params is null");
    if (!p->width || !p->height || !p->samples_per_px) return fail(RAYZ_ERR_BAD_ARG, "width, height and samples_per_px must be > 0");
    if (p->precision > RAYZ_PRECISION_F64) return fail(RAYZ_ERR_BAD_ARG, "bad precision %u", p->precision);
    if (p->traversal > RAYZ_TRAVERSAL_AUTO) return fail(RAYZ_ERR_BAD_ARG, "bad traversal %u", p->traversal);
    const uint32_t sc = p->shard_count ? p->shard_count : 1;
    if (p->shard_index >= sc) return fail(RAYZ_ERR_BAD_ARG, "shard_index %u >= shard_count %u", p->shard_index, sc);
    if (!(p->tmin == p->tmin)) return fail(RAYZ_ERR_BAD_ARG, "tmin is NaN");
    if (chunk_count(p) >= kMaxChunksPerPx) // whatever chunk_spp is, 0 (the automatic schedule) included
        return fail(RAYZ_ERR_BAD_ARG, "more than 2^20 chunks per pixel (%llu): raise chunk_spp", (unsigned long long)chunk_count(p));
    return RAYZ_OK;
}

template <class R> void fill_camera(const RayzCameraDesc* c, DevCamera<R>& o) {
    for (int k = 0; k < 3; ++k) {
        o.from[k] = (R)c->look_from[k];
        o.du[k] = (R)c->px_du[k];
        o.dv[k] = (R)c->px_dv[k];
        o.pxo[k] = (R)c->px_origin[k];
        o.defu[k] = (R)c->defocus_u[k];
        o.defv[k] = (R)c->defocus_v[k];
    }
    o.defocus = c->defocus ? 1u : 0u;
    o._pad = 0;
}

// ---- device contexts -----------------------------------------------------------------------------------
int ensure_ctx(int device) { // creates the context of `device` if needed; g_mu held by the caller
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(RAYZ_ERR_NO_DEVICE, "no HIP device: %s", hipGetErrorString(e));
    if (device < 0 || device >= n || device >= RAYZ_MAX_DEVICES)
        return fail(RAYZ_ERR_BAD_ARG, "device %d out of range [0,%d)", device, n < RAYZ_MAX_DEVICES ? n : RAYZ_MAX_DEVICES);
    DeviceCtx& c = g_ctx[device];
    if (c.ok) return RAYZ_OK;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RAYZ_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);
    DeviceScope scope(device);
    HIP_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    c.num_cu = prop.multiProcessorCount;
    c.ok = true;
    return RAYZ_OK;
}

// The context a scene renders on.  A scene created without a device is bound to the default device here.
int scene_ctx(RayzScene* s, DeviceCtx** out) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (s->device < 0) {
        if (g_default < 0) return fail(RAYZ_ERR_NO_DEVICE, "rayz_hip_init has not succeeded");
        s->device = g_default;
    }
    if (!g_ctx[s->device].ok) return fail(RAYZ_ERR_NO_DEVICE, "device %d is not initialised (rayz_hip_init / shutdown order)", s->device);
    *out = &g_ctx[s->device];
    return RAYZ_OK;
}

// Launches one render of `p`'s shard on the scene's device.  The caller has selected that device (DeviceScope).
// A scene supports ONE render in flight: a second call first waits for the previous one (its workspace and
// counters are reused).
template <class R>
int render_impl(RayzScene* s, const DeviceCtx& ctx, SceneBuffers<R>& b, const RayzCameraDesc* cam, const RayzRenderParams* p,
                R* d_out, hipStream_t stream) {
    typedef typename VecOf<R>::type r4;
    const bool use_bvh = p->traversal == RAYZ_TRAVERSAL_BVH ||
                         (p->traversal == RAYZ_TRAVERSAL_AUTO && s->spheres.size() + s->triangles.size() > RAYZ_AUTO_BVH_MIN);
    if (s->spheres.size() + s->triangles.size() >= (1u << 27))
        return fail(RAYZ_ERR_BAD_ARG, "too many hittables for the device layout");
    if (s->last_stream && s->last_stream != stream) HIP_TRY(hipStreamSynchronize(s->last_stream)); // previous render done
    if (s->origin_bound < 0) s->origin_bound = scene_origin_bound(s);
    int rc = upload<R>(s, b, std::max(s->origin_bound, camera_origin_bound(cam)));
    if (rc != RAYZ_OK) return rc;
    if (use_bvh) {
        rc = upload_bvh<R>(s, b);
        if (rc != RAYZ_OK) return rc;
        if (s->bvh_dev.depth > (uint32_t)kBvhStackDepth)
            return fail(RAYZ_ERR_BAD_ARG, "BVH depth %u exceeds the traversal stack (%d)", s->bvh_dev.depth, kBvhStackDepth);
    }

    const uint32_t rows = rayz_hip_shard_rows(p);
    const uint64_t shard_pixels64 = (uint64_t)rows * p->width;
    std::vector<uint32_t> starts;
    chunk_schedule(p, starts);
    const uint32_t chunks_per_px = (uint32_t)starts.size() - 1;
    const uint64_t items64 = shard_pixels64 * chunks_per_px;
    if (shard_pixels64 >= (1ull << 31) || items64 >= (1ull << 32) - (1ull << 26))
        return fail(RAYZ_ERR_BAD_ARG, "too many work items (%llu): raise chunk_spp", (unsigned long long)items64);
    s->last = RayzRenderStats{};
    s->last.primary_rays = shard_pixels64 * p->samples_per_px;
    s->last_bvh = use_bvh;
    s->last_two_paths = false;
    if (items64 == 0) {
        s->rendered = false;
        return RAYZ_OK;
    }
    if (!d_out) return fail(RAYZ_ERR_BAD_ARG, "output pointer is null");
    s->last_stream = stream;
    if (p->max_bounces == 0) { // bounceRay(ray, 0) is black, src/renderer.zig:104-105
        HIP_TRY(hipMemsetAsync(d_out, 0, shard_pixels64 * 3 * sizeof(R), stream));
        s->rendered = false;
        return RAYZ_OK;
    }
    const size_t need = (size_t)items64 * sizeof(r4);
    if (need > s->partial_bytes) {
        HIP_TRY(hipStreamSynchronize(stream));
        (void)hipFree(s->partial);
        s->partial = nullptr;
        s->partial_bytes = 0;
        HIP_TRY(hipMalloc(&s->partial, need));
        s->partial_bytes = need;
    }
    if (!s->counters) HIP_TRY(hipMalloc((void**)&s->counters, 32 * sizeof(unsigned long long)));
    if (starts != s->chunk_start_host) { // the schedule table, kept on the device until it changes
        HIP_TRY(hipStreamSynchronize(stream));
        if (starts.size() > s->chunk_start_cap) {
            (void)hipFree(s->chunk_start);
            s->chunk_start = nullptr;
            s->chunk_start_cap = 0;
            HIP_TRY(hipMalloc((void**)&s->chunk_start, starts.size() * sizeof(uint32_t)));
            s->chunk_start_cap = starts.size();
        }
        HIP_TRY(hipMemcpy(s->chunk_start, starts.data(), starts.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        s->chunk_start_host = starts;
    }
    if (!s->ev0) {
        HIP_TRY(hipEventCreate(&s->ev0));
        HIP_TRY(hipEventCreate(&s->ev1));
    }

    TraceArgs<R> A{};
    A.sc.stat = b.stat;
    A.sc.movy = b.movy;
    A.sc.movg = b.movg;
    A.sc.slot64 = s->narrow.slot64;
    A.sc.slot_pool = s->narrow.slot_pool;
    A.sc.sph_pool = b.sph_pool;
    A.sc.mat = b.mat;
    A.sc.tex = b.tex;
    A.sc.ns_pad = s->narrow.ns_pad;
    A.sc.ny_pad = s->narrow.ny_pad;
    A.sc.ng_pad = s->narrow.ng_pad;
    A.sc.n_spheres = (uint32_t)s->spheres.size();
    A.sc.tri = b.tri;
    A.sc.nt_pad = b.nt_pad;
    A.sc.n_triangles = (uint32_t)s->triangles.size();
    A.sc.bvh_nodes = (const f4*)b.bvh_nodes;
    for (int k = 0; k < 3; ++k) A.sc.bvh_glo[k] = b.grid.glo[k], A.sc.bvh_cell[k] = b.grid.cell[k];
    A.sc.bvh_leaf = b.bvh_leaf;
    A.sc.bvh_sph64 = s->narrow.bvh_sph64;
    A.sc.bvh_n_nodes = use_bvh ? b.bvh_n_inner : 0u;
    A.sc.bvh_leaf_stride = b.bvh_leaf_stride;
    A.sc.bvh_n_big_leaves = use_bvh ? b.n_big_leaves : 0u;
    for (int k = 0; k < 4; ++k) A.sc.bvh_big[k] = b.big_desc[k];
    A.sc.bvh_top = 0u; // (bytes: set below, once the launch knows how many records its workgroup keeps in LDS)
    fill_camera<R>(cam, A.cam);
    A.partial = (r4*)s->partial;
    A.counters = s->counters;
    A.seed = p->seed;
    A.tmin = (R)p->tmin;
    A.width = p->width;
    A.height = p->height;
    A.spp = p->samples_per_px;
    A.max_bounces = p->max_bounces;
    A.chunk_start = s->chunk_start;
    A.chunks_per_px = chunks_per_px;
    A.chunk_uniform = starts[1]; // the table's uniform prefix (chunk_bounds): chunks of starts[1] samples while the table keeps that stride
    A.chunk_n_uniform = 0;
    while (A.chunk_n_uniform < chunks_per_px && starts[A.chunk_n_uniform + 1] == (A.chunk_n_uniform + 1) * A.chunk_uniform) A.chunk_n_uniform++;
    A.tile_rows = p->tile_rows ? p->tile_rows : RAYZ_DEFAULT_TILE_ROWS;
    A.shard_index = p->shard_index;
    A.shard_count = p->shard_count ? p->shard_count : 1u;
    A.shard_pixels = (uint32_t)shard_pixels64;
    A.tiled_pixels = p->width % 8 == 0 ? (uint32_t)((uint64_t)(rows / 8 * 8) * p->width) : 0u; // whole 8x8 tiles of the local rows (place_item)
    A.total_items = (uint32_t)items64;
    A.queue_grab = (uint32_t)std::max(1ll, tuning(RAYZ_DEBUG_QUEUE_GRAB, kQueueGrab));
    // Which walk: one path per lane (trace_kernel_bvh).  The two-paths-per-lane form (trace_kernel_bvh2, f32 only: same image,
    // 19 % slower) is a retired experiment: only a -DRAYZ_EXPERIMENTS build contains it (RAYZ_DEBUG_BVH_KERNEL = 2).
#ifdef RAYZ_EXPERIMENTS
    const bool two_paths = use_bvh && sizeof(R) == 4 && tuning(RAYZ_DEBUG_BVH_KERNEL, 1) == 2;
    // .. and the walker / shader-wave form (trace_kernel_bvhx, f32 only): round 4's experiment, RAYZ_DEBUG_BVH_KERNEL = 3
    const bool exchange = use_bvh && sizeof(R) == 4 && tuning(RAYZ_DEBUG_BVH_KERNEL, 1) == 3;
#else
    const bool two_paths = false, exchange = false;
#endif
    // scheduling thresholds of the BVH kernel (no effect on results; rayz_hip_debug_set refuses values outside 1 .. 64 lanes)
    A.bvh_keep = (uint32_t)tuning(RAYZ_DEBUG_BVH_KEEP, kBvhKeepActive | (kBvhKeepStepping << 8));
#ifdef RAYZ_EXPERIMENTS
    if (two_paths)
        A.bvh_keep = (uint32_t)tuning(RAYZ_DEBUG_BVH2_KEEP, kBvh2Service | (kBvh2Blocked << 8) | (kBvh2Swap << 16) | (kBvhKeepStepping << 24));
#endif

    const int block = (use_bvh && !two_paths) ? (int)kBvhWg : 256;
    int blocks_per_cu = 0;
    // the BVH kernel's LDS stack holds one entry per tree level below the root (nearer child first: the stack never
    // holds more than one entry per level); sized from THIS tree, so a shallow tree does not cap the occupancy
    // (+ one guard row under entry 0: a lane that has popped its sentinel reads ahead at index −1)
    size_t bvh_stack_bytes = use_bvh ? ((size_t)s->bvh_dev.depth + 3) * block * sizeof(uint32_t) : 0;
    size_t x_bytes = 0; // the exchange kernel's slot area (after the oversized hittables' records); only its walker waves have stacks
#ifdef RAYZ_EXPERIMENTS
    if (exchange) {
        const long long xk = tuning(RAYZ_DEBUG_BVHX, -1);
        const uint32_t ns = xk < 0 ? 24u : (uint32_t)(xk & 0xff), xmin = xk < 0 ? 12u : (uint32_t)((xk >> 8) & 0xff),
                       xbatch = xk < 0 ? 48u : (uint32_t)((xk >> 16) & 0xff), xpat = xk < 0 ? 8u : (uint32_t)((xk >> 24) & 0xff),
                       xprio = xk < 0 ? 0x6eu : (uint32_t)((xk >> 32) & 0xff); // shader | walker box steps << 2 | leaf / root phases << 4 | exchange << 6
        A.x_slots = ns;
        A.x_cfg = xmin | (xbatch << 8) | (xpat << 16) | (xprio << 24);
        bvh_stack_bytes = ((size_t)s->bvh_dev.depth + 3) * kXWalkerLanes * sizeof(uint32_t);
        x_bytes = bvhx_exchange_bytes(ns);
        if (p->max_bounces >= (1u << 30)) return fail(RAYZ_ERR_BAD_ARG, "the exchange kernel packs flags into the segment count: max_bounces < 2^30");
    }
#endif
    // the tree's top: first in LDS.  The scene numbered b.bvh_top records breadth-first for the one-path kernel's workgroup;
    // a kernel whose workgroup has less LDS to spare (two paths per lane: three 256-thread workgroups per CU) keeps a prefix
    uint32_t top_records = use_bvh ? b.bvh_top : 0u;
    if (two_paths) top_records = std::min<uint32_t>(top_records, b.quantized ? 512u : 256u);
    if (exchange) { // what the walkers' stacks and the slots leave of the budget
        const size_t fixed = bvh_stack_bytes + (b.n_big_leaves ? kBvhBigLdsBytes : 0) + x_bytes;
        const size_t room = fixed < kBvhLdsBudget ? kBvhLdsBudget - fixed : 0;
        top_records = std::min<uint32_t>(top_records, (uint32_t)(room / (b.quantized ? 32 : 64)));
        const long long cap = tuning(RAYZ_DEBUG_BVH_TOP, -1);
        if (cap >= 0) top_records = std::min<uint32_t>(top_records, (uint32_t)cap);
    }
    typedef void (*Kernel)(const TraceArgs<R>);
    Kernel kernel = trace_kernel<R, 1>;
    if (use_bvh) kernel = b.quantized ? trace_kernel_bvh<R, true> : trace_kernel_bvh<R, false>;
#ifdef RAYZ_EXPERIMENTS
    if (two_paths) {
        if constexpr (sizeof(R) == 4) kernel = b.quantized ? trace_kernel_bvh2<float, true> : trace_kernel_bvh2<float, false>;
    }
    if (exchange) {
        if constexpr (sizeof(R) == 4) kernel = b.quantized ? trace_kernel_bvhx<true> : trace_kernel_bvhx<false>;
    }
#endif
    // The LDS request: top | stacks | oversized hittables' records (+ RAYZ_DEBUG_LDS_PAD unused bytes: an occupancy experiment —
    // fewer workgroups per CU, the same code).  The top was sized for kBvhLdsBudget, which this driver stack accepts; should a
    // stack refuse the request (hipFuncSetAttribute fails, or the occupancy query finds room for no workgroup), the launch keeps
    // a SHORTER PREFIX of the top instead of failing every BVH render — the walk works with any prefix (records beyond it are
    // read from global memory), only slower.
    const size_t rec_bytes = b.quantized ? 32 : 64;
    const size_t bvh_fixed = use_bvh ? bvh_stack_bytes + (b.n_big_leaves ? kBvhBigLdsBytes : 0) + x_bytes + (size_t)tuning(RAYZ_DEBUG_LDS_PAD, 0) : 0;
    size_t bvh_top_bytes = 0, bvh_lds = 0;
    for (;;) {
        bvh_top_bytes = (size_t)top_records * rec_bytes;
        bvh_lds = bvh_top_bytes + bvh_fixed;
        hipError_t e = hipSuccess;
        if (use_bvh && bvh_lds > 64 * 1024) // a workgroup that asks for more than 64 KB of LDS has to say so first
            e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bvh_lds);
        if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, kernel, block, use_bvh ? bvh_lds : 0);
        if (e == hipSuccess && (blocks_per_cu >= 1 || !use_bvh)) break;
        (void)hipGetLastError(); // (clear the sticky error of the refused request)
        if (!use_bvh || top_records == 0)
            return fail(RAYZ_ERR_HIP, "the trace kernel cannot be launched with %zu bytes of LDS: %s", bvh_lds,
                        e == hipSuccess ? "no workgroup fits a CU" : hipGetErrorString(e));
        const uint32_t step = (uint32_t)(8192 / rec_bytes); // give back 8 KB of the top per try
        top_records = top_records > step ? top_records - step : 0u;
    }
    A.bvh_top_words = (uint32_t)(bvh_top_bytes / sizeof(uint32_t));
    A.bvh_big_words = (uint32_t)((bvh_top_bytes + bvh_stack_bytes) / sizeof(uint32_t));
    A.sc.bvh_top = (uint32_t)bvh_top_bytes; // the walk compares byte offsets
#ifdef RAYZ_EXPERIMENTS
    A.x_words = (uint32_t)((bvh_top_bytes + bvh_stack_bytes + (b.n_big_leaves ? kBvhBigLdsBytes : 0)) / sizeof(uint32_t));
#endif
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    uint64_t grid = (uint64_t)ctx.num_cu * blocks_per_cu;
    // (a lane of the two-path kernel holds two items)
    const uint64_t want = (items64 + (two_paths ? 2 : 1) * block - 1) / ((two_paths ? 2 : 1) * block);
    if (grid > want) grid = want;

    s->last_two_paths = two_paths;
    s->last_exchange = exchange;
    HIP_TRY(hipMemsetAsync(s->counters, 0, 32 * sizeof(unsigned long long), stream));
    HIP_TRY(hipEventRecord(s->ev0, stream));
    hipLaunchKernelGGL(kernel, dim3((uint32_t)grid), dim3(block), use_bvh ? bvh_lds : 0, stream, A);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s->ev1, stream));
    hipLaunchKernelGGL(resolve_kernel<R>, dim3((A.shard_pixels + 255) / 256), dim3(256), 0, stream,
                       (const r4*)s->partial, d_out, A.shard_pixels, chunks_per_px, A.spp);
    HIP_TRY(hipGetLastError());
    s->rendered = true;
    return RAYZ_OK;
}

int check_render_args(RayzScene* s, const RayzCameraDesc* cam, const RayzRenderParams* p, uint32_t precision) {
    if (!s) return fail(RAYZ_ERR_STATE, "scene handle is null");
    if (!cam) return fail(RAYZ_ERR_BAD_ARG, "camera is null");
    int rc = validate_params(p);
    if (rc != RAYZ_OK) return rc;
    if (p->precision != precision)
        return fail(RAYZ_ERR_BAD_ARG, "params.precision %u does not match this entry point", p->precision);
    return RAYZ_OK;
}

template <class R>
int render_device(RayzScene* s, const RayzCameraDesc* cam, const RayzRenderParams* p, R* d_out, void* stream, uint32_t precision) {
    int rc = check_render_args(s, cam, p, precision);
    if (rc != RAYZ_OK) return rc;
    DeviceCtx* ctx = nullptr;
    rc = scene_ctx(s, &ctx);
    if (rc != RAYZ_OK) return rc;
    DeviceScope scope(s->device);
    SceneBuffers<R>* b;
    if constexpr (sizeof(R) == 4) b = &s->f32;
    else b = &s->f64;
    return render_impl<R>(s, *ctx, *b, cam, p, d_out, stream ? (hipStream_t)stream : ctx->stream);
}

int scene_new(const RayzSceneDesc* scene, int device, RayzScene** out) {
    if (!out) return fail(RAYZ_ERR_BAD_ARG, "out handle pointer is null");
    *out = nullptr;
    int rc = validate_scene(scene);
    if (rc != RAYZ_OK) return rc;
    RayzScene* s = new (std::nothrow) RayzScene();
    if (!s) return fail(RAYZ_ERR_OOM, "host allocation failed");
    try {
        s->spheres.assign(scene->spheres, scene->spheres + scene->n_spheres);
        s->materials.assign(scene->materials, scene->materials + scene->n_materials);
        s->textures.assign(scene->textures, scene->textures + scene->n_textures);
        s->triangles.assign(scene->triangles, scene->triangles + scene->n_triangles);
    } catch (...) {
        delete s;
        return fail(RAYZ_ERR_OOM, "host allocation failed");
    }
    s->device = device;
    *out = s;
    return RAYZ_OK;
}

int scene_sync(RayzScene* s, RayzRenderStats* stats) {
    if (!s) return fail(RAYZ_ERR_STATE, "scene handle is null");
    if (s->device < 0) { // never rendered: nothing to wait for
        if (stats) *stats = s->last;
        return RAYZ_OK;
    }
    DeviceScope scope(s->device);
    if (s->last_stream) HIP_TRY(hipStreamSynchronize(s->last_stream));
    if (s->rendered) {
        unsigned long long c[32] = {};
        HIP_TRY(hipMemcpy(c, s->counters, sizeof(c), hipMemcpyDeviceToHost));
#ifdef RAYZ_FLAT_PROFILE // measurement build only: wave time per phase of trace_kernel
        if (!s->last_bvh && c[9]) {
            const double tot = (double)(c[4] + c[5] + c[6] + c[7] + c[8]);
            std::fprintf(stderr, "flat phases (share of wave time; ticks per wave-iteration %.0f): refill %.1f%% | setup %.1f%% | scan %.1f%% | "
                                 "flush %.1f%% | shade %.1f%%\n", tot / (double)c[9], 100.0 * c[4] / tot, 100.0 * c[5] / tot,
                         100.0 * c[6] / tot, 100.0 * c[7] / tot, 100.0 * c[8] / tot);
        }
#endif
#ifdef RAYZ_BVH_PROFILE // measurement build only: per-phase wave time and lane occupancy of the BVH kernels
        if (s->last_bvh && s->last_exchange) {
            const double wt = (double)(c[4] + c[5] + c[6]), stt = (double)(c[12] + c[13] + c[14]);
            std::fprintf(stderr, "bvhx walkers (share of wave time): exchange %.1f%% | idle (nobody walks) %.1f%% | rounds %.1f%%; per exchange: %.0f ticks; "
                                 "at the start of a run of rounds: %.1f lanes walking, %.1f lanes hold a path; runs of rounds %.3g, exchanges %.3g\n",
                         100.0 * c[4] / wt, 100.0 * c[5] / wt, 100.0 * c[6] / wt, (double)c[4] / (double)(c[7] ? c[7] : 1), (double)c[8] / (double)(c[10] ? c[10] : 1),
                         (double)c[9] / (double)(c[10] ? c[10] : 1), (double)c[10], (double)c[7]);
            std::fprintf(stderr, "bvhx shaders (share of wave time): idle, nothing finished %.1f%% | waiting for a batch %.1f%% | pass %.1f%% (%.0f ticks per pass); "
                                 "%.1f paths per pass, %.1f finished slots seen; passes %.3g\n",
                         100.0 * c[12] / stt, 100.0 * c[13] / stt, 100.0 * c[14] / stt, (double)c[14] / (double)(c[15] ? c[15] : 1),
                         (double)c[16] / (double)(c[15] ? c[15] : 1), (double)c[17] / (double)(c[15] ? c[15] : 1), (double)c[15]);
        } else
        if (s->last_bvh && s->last_two_paths) {
            const double tot = (double)(c[4] + c[5] + c[6] + c[7] + c[8]);
            auto per = [&](int k) { return (double)c[9 + k] / (double)(c[10 + k] ? c[10 + k] : 1); };
            std::fprintf(stderr,
                         "bvh2 phases (share of wave time | mean lanes): service %.1f%% %.1f | swap %.1f%% %.1f | N %.1f%% %.1f | L %.1f%% %.1f | "
                         "C %.1f%% %.1f\n",
                         100.0 * c[4] / tot, per(0), 100.0 * c[5] / tot, per(2), 100.0 * c[6] / tot, per(4), 100.0 * c[7] / tot, per(6),
                         100.0 * c[8] / tot, per(8));
            const double it = (double)(c[14] ? c[14] : 1);
            std::fprintf(stderr, "  box steps: lanes per wave-step — stepping %.1f | parked at a leaf %.1f | walker idle %.1f; wave-steps per segment "
                                 "%.2f; service passes %.3g, swaps %.3g, leaf phases %.3g\n",
                         per(4), (double)c[19] / it, (double)c[20] / it, it / (double)(c[1] ? c[1] : 1), (double)c[10], (double)c[12], (double)c[16]);
        } else
        if (s->last_bvh) {
            const double tot = (double)(c[4] + c[5] + c[6] + c[7] + c[8]);
            std::fprintf(stderr,
                         "bvh phases (share of wave time | mean active lanes): refill %.1f%% | N %.1f%% %.1f | L %.1f%% %.1f | C %.1f%% "
                         "%.1f | shade %.1f%% %.1f\n",
                         100.0 * c[4] / tot, 100.0 * c[5] / tot, (double)c[9] / (double)(c[10] ? c[10] : 1), 100.0 * c[6] / tot,
                         (double)c[11] / (double)(c[12] ? c[12] : 1), 100.0 * c[7] / tot, (double)c[13] / (double)(c[14] ? c[14] : 1),
                         100.0 * c[8] / tot, (double)c[1] / (double)(c[15] ? c[15] : 1));
            const double it = (double)(c[10] ? c[10] : 1);
            std::fprintf(stderr, "  box steps: lanes per wave-step — stepping %.1f | parked at a leaf %.1f | walk finished, waiting for the shading pass %.1f | "
                                 "no path %.1f; wave-steps per segment %.2f; shading passes %.3g, rounds %.3g\n",
                         (double)c[9] / it, (double)c[16] / it, (double)c[17] / it, (double)c[18] / it, it / (double)(c[1] ? c[1] : 1) * 1.0,
                         (double)c[15], (double)c[12]);
            std::fprintf(stderr, "  node fetch (issue -> data): %.1f%% of the box-step phase, %.0f ticks per wave-step\n",
                         100.0 * (double)c[19] / (double)(c[5] ? c[5] : 1), (double)c[19] / it);
        }
#endif
        if (s->last_bvh && c[31] == 2)
            return fail(RAYZ_ERR_STATE, "the exchange kernel gave up waiting for a hand-over between its waves (bounded wait, no result)");
        if (s->last_bvh && c[31])
            return fail(RAYZ_ERR_STATE, "trace_kernel_bvh refused to run: its dynamic LDS segment does not start at LDS address 0");
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
        s->last.segments = c[1];
        s->last.sphere_tests = s->last_bvh ? c[3] : c[1] * (unsigned long long)(s->spheres.size() + s->triangles.size());
        s->last.node_tests = s->last_bvh ? c[2] : 0;
        s->last.kernel_ms = ms;
    }
    if (stats) *stats = s->last;
    return RAYZ_OK;
}

int scene_free(RayzScene* s) {
    if (!s) return RAYZ_OK;
    if (s->device >= 0) {
        DeviceScope scope(s->device);
        if (s->last_stream) (void)hipStreamSynchronize(s->last_stream);
        s->f32.release();
        s->f64.release();
        s->narrow.release();
        (void)hipFree(s->partial);
        (void)hipFree(s->counters);
        (void)hipFree(s->chunk_start);
        if (s->ev0) (void)hipEventDestroy(s->ev0);
        if (s->ev1) (void)hipEventDestroy(s->ev1);
    }
    delete s;
    return RAYZ_OK;
}

template <class R>
int render_oneshot(const RayzSceneDesc* scene, const RayzCameraDesc* cam, const RayzRenderParams* p, R* out,
                   RayzRenderStats* stats, uint32_t precision) {
    if (!out) return fail(RAYZ_ERR_BAD_ARG, "output pointer is null");
    int rc = validate_params(p);
    if (rc != RAYZ_OK) return rc;
    int device;
    {
        std::unique_lock<std::mutex> lock(g_mu);
        device = g_default;
    }
    if (device < 0) {
        rc = rayz_hip_init(0);
        if (rc != RAYZ_OK) return rc;
        device = 0;
    }
    RayzScene* s = nullptr;
    rc = scene_new(scene, device, &s);
    if (rc != RAYZ_OK) return rc;
    DeviceScope scope(device);
    const size_t n = (size_t)rayz_hip_shard_rows(p) * p->width * 3;
    R* d_out = nullptr;
    hipError_t e = hipMalloc((void**)&d_out, n ? n * sizeof(R) : 16);
    if (e != hipSuccess) {
        scene_free(s);
        return fail(RAYZ_ERR_OOM, "hipMalloc(output): %s", hipGetErrorString(e));
    }
    rc = render_device<R>(s, cam, p, d_out, nullptr, precision);
    if (rc == RAYZ_OK) rc = scene_sync(s, stats);
    if (rc == RAYZ_OK && n) {
        e = hipMemcpy(out, d_out, n * sizeof(R), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(RAYZ_ERR_HIP, "hipMemcpy(output): %s", hipGetErrorString(e));
    }
    (void)hipFree(d_out);
    scene_free(s);
    return rc;
}

// ---- RCCL, opened at run time ------------------------------------------------------------------------------
// The single-device entry points must not depend on RCCL being loadable, and a host that already carries an RCCL
// (torch ships one with the same soname) must not get a second copy: dlopen by soname reuses what is mapped.
struct Rccl {
    void* handle = nullptr;
    bool tried = false;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGather) Gather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl g_rccl;

int rccl_load() { // g_mu held
    Rccl& r = g_rccl;
    if (r.handle) return RAYZ_OK;
    if (r.tried) return fail(RAYZ_ERR_STATE, "RCCL is not available (librccl.so.1 could not be loaded)");
    r.tried = true;
    const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) return fail(RAYZ_ERR_STATE, "RCCL is not available: %s", dlerror());
    bool ok = true;
    auto sym = [&](const char* name) {
        void* p = dlsym(h, name);
        if (!p) ok = false;
        return p;
    };
    r.GetVersion = (decltype(r.GetVersion))sym("ncclGetVersion");
    r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.Gather = (decltype(r.Gather))sym("ncclGather");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!ok) {
        dlclose(h);
        return fail(RAYZ_ERR_STATE, "RCCL is not available: librccl lacks a required symbol");
    }
    r.handle = h;
    return RAYZ_OK;
}

#define NCCL_TRY(expr)                                                                                      \
    do {                                                                                                    \
        ncclResult_t r_ = (expr);                                                                           \
        if (r_ != ncclSuccess)                                                                              \
            return fail(RAYZ_ERR_HIP, "%s: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

// Interleaved row tiles back into the frame: gathered[rank][local row][w*3] -> frame[row][w*3] (on the root device).
template <class T>
__global__ __launch_bounds__(256) void unshard_kernel(const T* __restrict__ gathered, T* __restrict__ frame, uint32_t height,
                                                      uint32_t row_elems, uint32_t tile_rows, uint32_t n_ranks,
                                                      uint32_t max_rows) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)height * row_elems) return;
    const uint32_t y = (uint32_t)(i / row_elems), x = (uint32_t)(i - (size_t)y * row_elems);
    const uint32_t tile = y / tile_rows, rank = tile % n_ranks, local = (tile / n_ranks) * tile_rows + (y - tile * tile_rows);
    frame[i] = gathered[((size_t)rank * max_rows + local) * row_elems + x];
}

} // namespace

// One scene per device + the buffers of the gather; everything is driven by the calling host thread.
struct RayzMulti {
    std::vector<int> devices;
    std::vector<RayzScene*> scenes;
    std::vector<ncclComm_t> comms; // RAYZ_GATHER_RCCL
    std::vector<void*> tile;       // per device: this device's rows, grow-only
    std::vector<void*> tile8;      // per device: the same rows tone-mapped to u8 (render_u8 only)
    std::vector<size_t> tile_have, tile8_have;
    std::vector<hipEvent_t> done;  // per device: tile ready (peer-copy transport)
    void* gathered = nullptr; // root: [n][max_rows][row bytes]
    void* frame = nullptr;    // root: the assembled frame
    size_t gathered_bytes = 0, frame_bytes = 0;
    uint32_t transport = RAYZ_GATHER_RCCL;
    int rccl_version = 0;
    std::vector<RayzRenderStats> last_dev; // per device: counters of the last frame (rayz_hip_multi_device_stats)
    hipEvent_t g0 = nullptr, g1 = nullptr; // on the root's stream: its own tile done / frame assembled
    double last_gather_ms = 0, last_frame_ms = 0;
};

namespace {

int multi_free(RayzMulti* m) {
    if (!m) return RAYZ_OK;
    for (size_t i = 0; i < m->devices.size(); ++i) {
        if (i < m->scenes.size()) scene_free(m->scenes[i]); // waits for the device's last render
        DeviceScope scope(m->devices[i]);
        if (i < m->comms.size() && m->comms[i]) (void)g_rccl.CommDestroy(m->comms[i]);
        if (i < m->tile.size()) (void)hipFree(m->tile[i]);
        if (i < m->tile8.size()) (void)hipFree(m->tile8[i]);
        if (i < m->done.size() && m->done[i]) (void)hipEventDestroy(m->done[i]);
    }
    if (!m->devices.empty()) {
        DeviceScope scope(m->devices[0]);
        (void)hipFree(m->gathered);
        (void)hipFree(m->frame);
        if (m->g0) (void)hipEventDestroy(m->g0);
        if (m->g1) (void)hipEventDestroy(m->g1);
    }
    delete m;
    return RAYZ_OK;
}

// `dup_ok`: RAYZ_GATHER_ALLOW_DUPLICATE_DEVICES was passed with the peer-copy transport (tests on a one-GPU box: the N-way
// sharding, the gather into N slots and the un-interleave then run for real, every "device" being the same one)
int check_device_list(const int* devices, int n, bool dup_ok) {
    if (!devices) return fail(RAYZ_ERR_BAD_ARG, "device list is null");
    if (n < 1 || n > RAYZ_MAX_DEVICES) return fail(RAYZ_ERR_BAD_ARG, "n_devices %d out of range [1,%d]", n, RAYZ_MAX_DEVICES);
    for (int i = 0; i < n; ++i) {
        if (devices[i] < 0 || devices[i] >= RAYZ_MAX_DEVICES) return fail(RAYZ_ERR_BAD_ARG, "device %d out of range", devices[i]);
        for (int j = 0; j < i && !dup_ok; ++j)
            if (devices[j] == devices[i]) return fail(RAYZ_ERR_BAD_ARG, "device %d is listed twice", devices[i]);
    }
    return RAYZ_OK;
}

int grow(void** buf, size_t* have, size_t need) {
    if (need <= *have) return RAYZ_OK;
    (void)hipFree(*buf);
    *buf = nullptr;
    *have = 0;
    HIP_TRY(hipMalloc(buf, need));
    *have = need;
    return RAYZ_OK;
}

// T = element type of the frame that crosses the ABI (float, double; uint8_t for the tone-mapped form, rendered in f32).
template <class T>
int multi_render(RayzMulti* m, const RayzCameraDesc* cam, const RayzRenderParams* p, T* out, RayzRenderStats* stats) {
    typedef typename std::conditional<sizeof(T) == 8, double, float>::type R;
    constexpr bool to_u8 = sizeof(T) == 1;
    if (!m) return fail(RAYZ_ERR_STATE, "multi handle is null");
    if (!cam) return fail(RAYZ_ERR_BAD_ARG, "camera is null");
    if (!out) return fail(RAYZ_ERR_BAD_ARG, "output pointer is null");
    int rc = validate_params(p);
    if (rc != RAYZ_OK) return rc;
    if (p->precision != (sizeof(R) == 8 ? RAYZ_PRECISION_F64 : RAYZ_PRECISION_F32))
        return fail(RAYZ_ERR_BAD_ARG, "params.precision %u does not match this entry point", p->precision);
    if (p->shard_index != 0 || p->shard_count > 1)
        return fail(RAYZ_ERR_BAD_ARG, "the multi-device entry shards the frame itself: shard_index / shard_count must be 0");
    const uint32_t n = (uint32_t)m->devices.size();
    RayzRenderParams q = *p;
    q.tile_rows = p->tile_rows ? p->tile_rows : RAYZ_DEFAULT_TILE_ROWS; // ONE default for every entry point (include/rayz_hip.h)
    q.shard_count = n;
    uint32_t max_rows = 0;
    for (uint32_t i = 0; i < n; ++i) {
        q.shard_index = i;
        const uint32_t r = rayz_hip_shard_rows(&q);
        max_rows = r > max_rows ? r : max_rows;
    }
    const size_t row_elems = (size_t)p->width * 3;
    const size_t tile_bytes = (size_t)max_rows * row_elems * sizeof(R), tile8_bytes = (size_t)max_rows * row_elems;
    const size_t send_bytes = to_u8 ? tile8_bytes : tile_bytes;
    const size_t frame_bytes = (size_t)p->height * row_elems * sizeof(T);
    if ((size_t)p->height * row_elems >= (1ull << 32)) return fail(RAYZ_ERR_BAD_ARG, "frame too large");

    // 1. every device traces its rows (asynchronous: the launches of all devices overlap)
    std::vector<DeviceCtx*> ctx(n, nullptr);
    for (uint32_t i = 0; i < n; ++i) {
        rc = scene_ctx(m->scenes[i], &ctx[i]);
        if (rc != RAYZ_OK) return rc;
        DeviceScope scope(m->devices[i]);
        HIP_TRY(hipStreamSynchronize(ctx[i]->stream)); // the previous frame's gather has left the tiles
        rc = grow(&m->tile[i], &m->tile_have[i], tile_bytes ? tile_bytes : 16);
        if (rc == RAYZ_OK && to_u8) rc = grow(&m->tile8[i], &m->tile8_have[i], tile8_bytes ? tile8_bytes : 16);
        if (rc != RAYZ_OK) return rc;
    }
    {
        DeviceScope scope(m->devices[0]);
        rc = grow(&m->gathered, &m->gathered_bytes, (size_t)n * send_bytes ? (size_t)n * send_bytes : 16);
        if (rc == RAYZ_OK) rc = grow(&m->frame, &m->frame_bytes, frame_bytes);
        if (rc != RAYZ_OK) return rc;
    }
    for (uint32_t i = 0; i < n; ++i) {
        DeviceScope scope(m->devices[i]);
        q.shard_index = i;
        SceneBuffers<R>* b;
        if constexpr (sizeof(R) == 4) b = &m->scenes[i]->f32;
        else b = &m->scenes[i]->f64;
        rc = render_impl<R>(m->scenes[i], *ctx[i], *b, cam, &q, (R*)m->tile[i], ctx[i]->stream);
        if (rc != RAYZ_OK) return rc;
        if constexpr (to_u8) { // writePPM's transform before the gather: the tiles travel as u8, 4x smaller (src/image.zig:35-38)
            const size_t ne = (size_t)rayz_hip_shard_rows(&q) * row_elems;
            if (ne) {
                hipLaunchKernelGGL(tonemap_kernel, dim3((uint32_t)((ne + 255) / 256)), dim3(256), 0, ctx[i]->stream,
                                   (const float*)m->tile[i], (uint8_t*)m->tile8[i], ne);
                HIP_TRY(hipGetLastError());
            }
        }
    }
    // 2. one gather of the row tiles to the first device.  g0 .. g1 on the root's stream = from "the root's own rows are
    //    done" to "the frame is assembled": the transfer plus whatever the root waited for slower devices
    const auto wall0 = std::chrono::steady_clock::now();
    {
        DeviceScope scope(m->devices[0]);
        if (!m->g0) {
            HIP_TRY(hipEventCreate(&m->g0));
            HIP_TRY(hipEventCreate(&m->g1));
        }
        HIP_TRY(hipEventRecord(m->g0, ctx[0]->stream));
    }
    auto src = [&](uint32_t i) { return to_u8 ? m->tile8[i] : m->tile[i]; };
    if (m->transport == RAYZ_GATHER_RCCL) {
        NCCL_TRY(g_rccl.GroupStart());
        for (uint32_t i = 0; i < n; ++i) {
            ncclResult_t r = g_rccl.Gather(src(i), m->gathered, send_bytes, ncclUint8, 0, m->comms[i], ctx[i]->stream);
            if (r != ncclSuccess) {
                (void)g_rccl.GroupEnd();
                return fail(RAYZ_ERR_HIP, "ncclGather: %s", g_rccl.GetErrorString(r));
            }
        }
        NCCL_TRY(g_rccl.GroupEnd());
    } else { // peer copies, each on its source device's stream; the root's stream waits for all of them
        for (uint32_t i = 0; i < n; ++i) {
            DeviceScope scope(m->devices[i]);
            HIP_TRY(hipMemcpyPeerAsync((char*)m->gathered + (size_t)i * send_bytes, m->devices[0], src(i), m->devices[i], send_bytes,
                                       ctx[i]->stream));
            HIP_TRY(hipEventRecord(m->done[i], ctx[i]->stream));
        }
        DeviceScope scope(m->devices[0]);
        for (uint32_t i = 1; i < n; ++i) HIP_TRY(hipStreamWaitEvent(ctx[0]->stream, m->done[i], 0));
    }
    // 3. un-interleave on the root, copy out
    {
        DeviceScope scope(m->devices[0]);
        const size_t ne = (size_t)p->height * row_elems;
        hipLaunchKernelGGL(unshard_kernel<T>, dim3((uint32_t)((ne + 255) / 256)), dim3(256), 0, ctx[0]->stream,
                           (const T*)m->gathered, (T*)m->frame, p->height, (uint32_t)row_elems, q.tile_rows, n, max_rows);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(m->g1, ctx[0]->stream));
        HIP_TRY(hipMemcpyAsync(out, m->frame, frame_bytes, hipMemcpyDeviceToHost, ctx[0]->stream));
        HIP_TRY(hipStreamSynchronize(ctx[0]->stream));
        float gms = 0;
        HIP_TRY(hipEventElapsedTime(&gms, m->g0, m->g1));
        m->last_gather_ms = gms;
        m->last_frame_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    }
    // 4. counters: sums over the devices; kernel_ms is the slowest device's trace kernel
    RayzRenderStats tot{};
    m->last_dev.assign(n, RayzRenderStats{});
    for (uint32_t i = 0; i < n; ++i) {
        RayzRenderStats st{};
        rc = scene_sync(m->scenes[i], &st);
        if (rc != RAYZ_OK) return rc;
        m->last_dev[i] = st;
        tot.primary_rays += st.primary_rays;
        tot.segments += st.segments;
        tot.sphere_tests += st.sphere_tests;
        tot.node_tests += st.node_tests;
        tot.kernel_ms = st.kernel_ms > tot.kernel_ms ? st.kernel_ms : tot.kernel_ms;
    }
    if (stats) *stats = tot;
    return RAYZ_OK;
}

template <class T>
int render_multi_oneshot(const int* devices, int n, const RayzSceneDesc* scene, const RayzCameraDesc* cam, const RayzRenderParams* p,
                         T* out, RayzRenderStats* stats) {
    RayzMulti* m = nullptr;
    int rc = rayz_hip_multi_create(devices, n, scene, RAYZ_GATHER_RCCL, &m);
    if (rc != RAYZ_OK) return rc;
    rc = multi_render<T>(m, cam, p, out, stats);
    multi_free(m);
    return rc;
}

} // namespace

extern "C" {

uint32_t rayz_hip_abi_version(void) { return RAYZ_HIP_ABI_VERSION; }

int rayz_hip_debug_set(uint32_t knob, long long value) {
    if (knob >= RAYZ_DEBUG_KNOBS) return fail(RAYZ_ERR_BAD_ARG, "bad debug knob %u", knob);
    // A knob changes scheduling, never a result — and must never be able to hang the device: values a kernel's loop
    // control cannot take are refused here (negative = back to the built-in default, always accepted).
    if (value >= 0) {
        auto lane_count = [](long long b) { return b >= 1 && b <= 64; }; // a threshold counted in lanes of a wave
        switch (knob) {
        case RAYZ_DEBUG_QUEUE_GRAB:
            if (value < 1 || value > (1 << 20)) return fail(RAYZ_ERR_BAD_ARG, "QUEUE_GRAB %lld outside 1 .. 2^20", value);
            break;
        case RAYZ_DEBUG_BVH_KEEP: // keep_active | keep_stepping << 8
            if (value >> 16 || !lane_count(value & 0xff) || !lane_count((value >> 8) & 0xff))
                return fail(RAYZ_ERR_BAD_ARG, "BVH_KEEP 0x%llx: both thresholds must be 1 .. 64 lanes", (unsigned long long)value);
            break;
        case RAYZ_DEBUG_BVH_KERNEL:
#ifdef RAYZ_EXPERIMENTS
            if (value < 1 || value > 3) return fail(RAYZ_ERR_BAD_ARG, "BVH_KERNEL %lld: 1, 2 or 3", value);
#else
            if (value != 1) return fail(RAYZ_ERR_BAD_ARG, "BVH_KERNEL %lld: this build holds the one-path kernel only (the retired two-path "
                                                          "experiment needs -DRAYZ_EXPERIMENTS)", value);
#endif
            break;
        case RAYZ_DEBUG_BVH2_KEEP: // service | blocked << 8 | swap << 16 | keep_stepping << 24
#ifdef RAYZ_EXPERIMENTS
            if (value >> 32 || !lane_count(value & 0xff) || !lane_count((value >> 8) & 0xff) || !lane_count((value >> 16) & 0xff) ||
                !lane_count((value >> 24) & 0xff))
                return fail(RAYZ_ERR_BAD_ARG, "BVH2_KEEP 0x%llx: every threshold must be 1 .. 64 lanes", (unsigned long long)value);
#else
            return fail(RAYZ_ERR_BAD_ARG, "BVH2_KEEP: the two-path kernel is not in this build (-DRAYZ_EXPERIMENTS)");
#endif
            break;
        case RAYZ_DEBUG_BVHX: { // slots | exchange threshold << 8 | minimum batch << 16 | patience << 24 | priority << 32
#ifdef RAYZ_EXPERIMENTS
            const long long ns = value & 0xff;
            if (value >> 40 || ns < 4 || ns > 64 || (ns & 1) || !lane_count((value >> 8) & 0xff) || !lane_count((value >> 16) & 0xff))
                return fail(RAYZ_ERR_BAD_ARG, "BVHX 0x%llx: slots even 4 .. 64, thresholds 1 .. 64 lanes", (unsigned long long)value);
#else
            return fail(RAYZ_ERR_BAD_ARG, "BVHX: the exchange kernel is not in this build (-DRAYZ_EXPERIMENTS)");
#endif
            break;
        }
        case RAYZ_DEBUG_CHUNK_CAP:
#ifdef RAYZ_EXPERIMENTS
            if (value < 16 || value > 4096 || (value & (value - 1))) return fail(RAYZ_ERR_BAD_ARG, "CHUNK_CAP %lld: a power of two, 16 .. 4096", value);
#else
            return fail(RAYZ_ERR_BAD_ARG, "CHUNK_CAP changes the image's summation tree: -DRAYZ_EXPERIMENTS builds only");
#endif
            break;
        case RAYZ_DEBUG_LDS_PAD:
            if (value > 160 * 1024) return fail(RAYZ_ERR_BAD_ARG, "LDS_PAD %lld exceeds a CU's LDS", value);
            break;
        default: break;
        }
    }
    g_tune.v[knob].store(value, std::memory_order_relaxed);
    return RAYZ_OK;
}
const char* rayz_hip_last_error(void) { return g_err; }

int rayz_hip_init(int device) {
    return guarded([&] {
        std::lock_guard<std::mutex> lock(g_mu);
        const int rc = ensure_ctx(device);
        if (rc == RAYZ_OK) g_default = device;
        return rc;
    });
}

void rayz_hip_shutdown(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    for (int d = 0; d < RAYZ_MAX_DEVICES; ++d) {
        DeviceCtx& c = g_ctx[d];
        if (!c.ok) continue;
        DeviceScope scope(d);
        (void)hipStreamSynchronize(c.stream);
        (void)hipStreamDestroy(c.stream);
        c = DeviceCtx{};
    }
    g_default = -1;
}

uint32_t rayz_hip_shard_rows(const RayzRenderParams* p) {
    if (!p) return 0;
    const uint32_t tr = p->tile_rows ? p->tile_rows : RAYZ_DEFAULT_TILE_ROWS, sc = p->shard_count ? p->shard_count : 1u;
    if (p->shard_index >= sc) return 0;
    uint32_t n = 0;
    for (uint32_t t = p->shard_index; (uint64_t)t * tr < p->height; t += sc) {
        const uint32_t r0 = t * tr;
        n += (p->height - r0 < tr) ? p->height - r0 : tr;
    }
    return n;
}

uint32_t rayz_hip_chunk_schedule(const RayzRenderParams* p, uint32_t* starts, uint32_t capacity) {
    if (!p || !p->samples_per_px || !p->width || !p->height) return 0;
    if (chunk_count(p) >= kMaxChunksPerPx) return 0; // a schedule no render accepts (validate_params)
    std::vector<uint32_t> v;
    try {
        chunk_schedule(p, v);
    } catch (...) {
        return 0;
    }
    if (v.size() - 1 != chunk_count(p)) return 0; // chunk_count is exact by construction: a disagreement is a bug, not a schedule
    if (starts)
        for (size_t i = 0; i < v.size() && i < capacity; ++i) starts[i] = v[i];
    return (uint32_t)v.size() - 1;
}

int rayz_hip_scene_create(const RayzSceneDesc* scene, RayzScene** out) {
    return guarded([&] { return scene_new(scene, -1, out); });
}

int rayz_hip_scene_create_on(int device, const RayzSceneDesc* scene, RayzScene** out) {
    return guarded([&] {
        if (out) *out = nullptr;
        {
            std::lock_guard<std::mutex> lock(g_mu);
            const int rc = ensure_ctx(device);
            if (rc != RAYZ_OK) return rc;
        }
        return scene_new(scene, device, out);
    });
}

int rayz_hip_scene_destroy(RayzScene* s) {
    return guarded([&] { return scene_free(s); });
}

int rayz_hip_render_device(RayzScene* s, const RayzCameraDesc* cam, const RayzRenderParams* p, float* d_out,
                           void* stream) {
    return guarded([&] { return render_device<float>(s, cam, p, d_out, stream, RAYZ_PRECISION_F32); });
}

int rayz_hip_render_device_f64(RayzScene* s, const RayzCameraDesc* cam, const RayzRenderParams* p, double* d_out,
                               void* stream) {
    return guarded([&] { return render_device<double>(s, cam, p, d_out, stream, RAYZ_PRECISION_F64); });
}

int rayz_hip_scene_sync(RayzScene* s, RayzRenderStats* stats) {
    return guarded([&] { return scene_sync(s, stats); });
}

int rayz_hip_scene_bvh(RayzScene* s, uint32_t* n_nodes, uint32_t* depth, double* boxes, uint32_t* skip, uint32_t* first,
                       uint32_t* count, uint32_t* order) {
    return guarded([&] {
        if (!s || !n_nodes) return fail(RAYZ_ERR_BAD_ARG, "null argument");
        ensure_bvh(s);
        const rayz_bvh::FlatBvh& t = s->bvh;
        *n_nodes = (uint32_t)t.nodes.size();
        if (depth) *depth = t.depth;
        for (size_t i = 0; i < t.nodes.size(); ++i) {
            if (boxes)
                for (int k = 0; k < 3; ++k) boxes[6 * i + k] = t.nodes[i].box.lo[k], boxes[6 * i + 3 + k] = t.nodes[i].box.hi[k];
            if (skip) skip[i] = t.nodes[i].skip;
            if (first) first[i] = t.nodes[i].first;
            if (count) count[i] = t.nodes[i].count;
        }
        if (order) std::copy(t.order.begin(), t.order.end(), order);
        return (int)RAYZ_OK;
    });
}

int rayz_hip_render(const RayzSceneDesc* scene, const RayzCameraDesc* cam, const RayzRenderParams* p, float* out,
                    RayzRenderStats* stats) {
    return guarded([&] { return render_oneshot<float>(scene, cam, p, out, stats, RAYZ_PRECISION_F32); });
}

int rayz_hip_render_f64(const RayzSceneDesc* scene, const RayzCameraDesc* cam, const RayzRenderParams* p, double* out,
                        RayzRenderStats* stats) {
    return guarded([&] { return render_oneshot<double>(scene, cam, p, out, stats, RAYZ_PRECISION_F64); });
}

int rayz_hip_tonemap_u8(const float* d_rgb, uint8_t* d_rgb8, size_t n_pixels, void* stream) {
    return guarded([&] {
        int device;
        hipStream_t own;
        {
            std::lock_guard<std::mutex> lock(g_mu);
            device = g_default;
            if (device < 0) return fail(RAYZ_ERR_NO_DEVICE, "rayz_hip_init has not succeeded");
            own = g_ctx[device].stream;
        }
        if (!n_pixels) return (int)RAYZ_OK;
        if (!d_rgb || !d_rgb8) return fail(RAYZ_ERR_BAD_ARG, "null buffer");
        DeviceScope scope(device);
        const size_t n = n_pixels * 3;
        hipLaunchKernelGGL(tonemap_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream ? (hipStream_t)stream : own,
                           d_rgb, d_rgb8, n);
        HIP_TRY(hipGetLastError());
        return (int)RAYZ_OK;
    });
}

// ---- known answers: the kernel's device functions on caller inputs ---------------------------------------------
int rayz_hip_kat(uint32_t op, uint32_t precision, const double* in, uint32_t n, double* out) {
    return guarded([&] {
        if (op > RAYZ_KAT_SCAN_DISCS) return fail(RAYZ_ERR_BAD_ARG, "bad known-answer op %u", op);
        if (precision > RAYZ_PRECISION_F64) return fail(RAYZ_ERR_BAD_ARG, "bad precision %u", precision);
        if (!n) return (int)RAYZ_OK;
        if (!in || !out) return fail(RAYZ_ERR_BAD_ARG, "null buffer");
        int device;
        hipStream_t stream;
        {
            std::lock_guard<std::mutex> lock(g_mu);
            device = g_default;
            if (device < 0) return fail(RAYZ_ERR_NO_DEVICE, "rayz_hip_init has not succeeded");
            stream = g_ctx[device].stream;
        }
        std::vector<double> host(in, in + (size_t)n * RAYZ_KAT_IN_STRIDE);
        for (uint32_t i = 0; i < n; ++i) { // what the scene upload would have prepared for these hittables
            double* a = host.data() + (size_t)i * RAYZ_KAT_IN_STRIDE;
            // the list of uniforms must lie inside the record: the device reads u[0 .. n_u)
            if (op == RAYZ_KAT_GET_RAY || op == RAYZ_KAT_SCATTER) {
                const int at = op == RAYZ_KAT_GET_RAY ? 21 : 16;
                const double nu = a[at];
                const bool no_rng = op == RAYZ_KAT_GET_RAY && nu == -1.0; // getRay(px, py, null)
                if (!no_rng && !(nu >= 0 && nu <= RAYZ_KAT_IN_STRIDE - (at + 1) && nu == std::floor(nu)))
                    return fail(RAYZ_ERR_BAD_ARG, "record %u: n_u = %g is not an integer in [0, %d]%s", i, nu, RAYZ_KAT_IN_STRIDE - (at + 1),
                                op == RAYZ_KAT_GET_RAY ? " (or -1: no generator)" : "");
            }
            if (op == RAYZ_KAT_BOX_HIT) { // the box as a scene upload would hold it (S = this ray's origin, B = this box), in the format a[26] names
                rayz_bvh::Box bx;
                double B = 0;
                for (int k = 0; k < 3; ++k) bx.lo[k] = a[k], bx.hi[k] = a[3 + k], B = std::max({B, std::fabs(a[k]), std::fabs(a[3 + k])});
                if (a[26] != 0.0) { // f32 planes
                    const double pad = kBoxPadUlps * unit_roundoff<float>() * std::max(norm3(a + 6), B);
                    for (int k = 0; k < 3; ++k) {
                        a[14 + k] = (double)rayz_bvh::roundDown<float>(bx.lo[k] - pad), a[17 + k] = (double)rayz_bvh::roundUp<float>(bx.hi[k] + pad);
                        a[20 + k] = 0.0, a[23 + k] = 1.0;
                    }
                } else { // 16-bit plane indices on the grid over this box
                    double pad = kBoxPadUlps * unit_roundoff<float>() * (std::max(norm3(a + 6), B) + 2.0 * B);
                    const rayz_bvh::PlaneGrid g = rayz_bvh::PlaneGrid::over(bx.lo, bx.hi, 2.0 * pad);
                    pad = kBoxPadUlps * unit_roundoff<float>() * (std::max(norm3(a + 6), B) + g.extent);
                    uint32_t w[3];
                    g.quantize(bx, pad, w);
                    for (int k = 0; k < 3; ++k) a[14 + k] = w[k] & 0xffffu, a[17 + k] = w[k] >> 16, a[20 + k] = g.glo[k], a[23 + k] = g.cell[k];
                }
            }
            if (op == RAYZ_KAT_SCAN_DISCS) { // the padded squares the scan streams would hold for these four spheres
                double S = norm3(a + 20);
                RayzSphere q[4] = {};
                for (int k = 0; k < 4; ++k) {
                    q[k].center[0] = a[k], q[k].center[1] = a[4 + k], q[k].center[2] = a[8 + k];
                    q[k].radius = a[12 + k];
                    q[k].velocity[1] = a[27] != 0.0 ? a[16 + k] : 0.0;
                    S = std::max(S, norm3(q[k].center) + norm3(q[k].velocity) + std::fabs(q[k].radius));
                }
                for (int k = 0; k < 4; ++k)
                    a[28 + k] = precision == RAYZ_PRECISION_F32 ? (double)pad_radius2_scan<float>(q[k], S) : (double)pad_radius2_scan<double>(q[k], S);
            }
            if (op == RAYZ_KAT_SPHERE_HIT) {
                RayzSphere q{};
                for (int k = 0; k < 3; ++k) q.center[k] = a[k], q.velocity[k] = a[3 + k];
                q.radius = a[6];
                const double S = std::max(norm3(a + 7), norm3(q.center) + norm3(q.velocity) + std::fabs(q.radius));
                a[16] = precision == RAYZ_PRECISION_F32 ? (double)pad_radius2_scan<float>(q, S) : (double)pad_radius2_scan<double>(q, S);
            }
        }
        DeviceScope scope(device);
        double *d_in = nullptr, *d_out = nullptr;
        const size_t in_bytes = host.size() * sizeof(double), out_bytes = (size_t)n * RAYZ_KAT_OUT_STRIDE * sizeof(double);
        hipError_t e = hipMalloc((void**)&d_in, in_bytes);
        if (e == hipSuccess) e = hipMalloc((void**)&d_out, out_bytes);
        if (e == hipSuccess) e = hipMemcpyAsync(d_in, host.data(), in_bytes, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) {
            if (precision == RAYZ_PRECISION_F32) hipLaunchKernelGGL(kat_kernel<float>, dim3((n + 63) / 64), dim3(64), 0, stream, op, d_in, n, d_out);
            else hipLaunchKernelGGL(kat_kernel<double>, dim3((n + 63) / 64), dim3(64), 0, stream, op, d_in, n, d_out);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        (void)hipFree(d_in);
        (void)hipFree(d_out);
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? RAYZ_ERR_OOM : RAYZ_ERR_HIP, "rayz_hip_kat: %s", hipGetErrorString(e));
        return (int)RAYZ_OK;
    });
}

// ---- several devices behind one call ----------------------------------------------------------------------
int rayz_hip_multi_create(const int* devices, int n_devices, const RayzSceneDesc* scene, uint32_t transport, RayzMulti** out) {
    return guarded([&] {
        if (!out) return fail(RAYZ_ERR_BAD_ARG, "out handle pointer is null");
        *out = nullptr;
        const bool dup_ok = (transport & RAYZ_GATHER_ALLOW_DUPLICATE_DEVICES) != 0;
        transport &= ~(uint32_t)RAYZ_GATHER_ALLOW_DUPLICATE_DEVICES;
        if (transport > RAYZ_GATHER_PEER_COPY) return fail(RAYZ_ERR_BAD_ARG, "bad gather transport %u", transport);
        if (dup_ok && transport != RAYZ_GATHER_PEER_COPY)
            return fail(RAYZ_ERR_BAD_ARG, "RAYZ_GATHER_ALLOW_DUPLICATE_DEVICES needs the peer-copy transport (RCCL refuses a device twice)");
        int rc = check_device_list(devices, n_devices, dup_ok);
        if (rc != RAYZ_OK) return rc;
        rc = validate_scene(scene);
        if (rc != RAYZ_OK) return rc;
        {
            std::lock_guard<std::mutex> lock(g_mu);
            for (int i = 0; i < n_devices; ++i) {
                rc = ensure_ctx(devices[i]);
                if (rc != RAYZ_OK) return rc;
            }
            if (transport == RAYZ_GATHER_RCCL) {
                rc = rccl_load();
                if (rc != RAYZ_OK) return rc;
            }
        }
        RayzMulti* m = new RayzMulti();
        m->transport = transport;
        m->devices.assign(devices, devices + n_devices);
        m->tile.assign(n_devices, nullptr);
        m->tile8.assign(n_devices, nullptr);
        m->tile_have.assign(n_devices, 0);
        m->tile8_have.assign(n_devices, 0);
        m->done.assign(n_devices, nullptr);
        for (int i = 0; i < n_devices; ++i) {
            RayzScene* s = nullptr;
            rc = scene_new(scene, devices[i], &s);
            if (rc != RAYZ_OK) {
                multi_free(m);
                return rc;
            }
            m->scenes.push_back(s);
        }
        auto bail = [&](int code) {
            multi_free(m);
            return code;
        };
        if (transport == RAYZ_GATHER_RCCL) {
            m->comms.assign(n_devices, nullptr);
            ncclResult_t r = g_rccl.CommInitAll(m->comms.data(), n_devices, m->devices.data());
            if (r != ncclSuccess) {
                m->comms.clear();
                return bail(fail(RAYZ_ERR_HIP, "ncclCommInitAll(%d devices): %s", n_devices, g_rccl.GetErrorString(r)));
            }
            (void)g_rccl.GetVersion(&m->rccl_version);
        } else {
            for (int i = 0; i < n_devices; ++i) {
                DeviceScope scope(devices[i]);
                hipError_t e = hipEventCreateWithFlags(&m->done[i], hipEventDisableTiming);
                if (e != hipSuccess) return bail(fail(RAYZ_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(e)));
                if (i > 0) { // the root pulls nothing; sources push into the root's buffer
                    int can = 0;
                    (void)hipDeviceCanAccessPeer(&can, devices[i], devices[0]);
                    if (can) {
                        e = hipDeviceEnablePeerAccess(devices[0], 0);
                        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                            return bail(fail(RAYZ_ERR_HIP, "hipDeviceEnablePeerAccess(%d -> %d): %s", devices[i], devices[0],
                                             hipGetErrorString(e)));
                        (void)hipGetLastError();
                    }
                }
            }
        }
        *out = m;
        return (int)RAYZ_OK;
    });
}

int rayz_hip_multi_destroy(RayzMulti* m) {
    return guarded([&] { return multi_free(m); });
}

int rayz_hip_multi_info(const RayzMulti* m, int* n_devices, uint32_t* transport, int* rccl_version) {
    if (!m) return fail(RAYZ_ERR_STATE, "multi handle is null");
    if (n_devices) *n_devices = (int)m->devices.size();
    if (transport) *transport = m->transport;
    if (rccl_version) *rccl_version = m->rccl_version;
    return RAYZ_OK;
}

int rayz_hip_multi_device_stats(const RayzMulti* m, int index, RayzRenderStats* stats) {
    if (!m) return fail(RAYZ_ERR_STATE, "multi handle is null");
    if (!stats) return fail(RAYZ_ERR_BAD_ARG, "stats pointer is null");
    if (index < 0 || (size_t)index >= m->devices.size()) return fail(RAYZ_ERR_BAD_ARG, "device index %d out of range", index);
    if (m->last_dev.size() != m->devices.size()) return fail(RAYZ_ERR_STATE, "no frame has been rendered on this handle");
    *stats = m->last_dev[(size_t)index];
    return RAYZ_OK;
}

int rayz_hip_multi_timing(const RayzMulti* m, double* gather_ms, double* frame_ms) {
    if (!m) return fail(RAYZ_ERR_STATE, "multi handle is null");
    if (m->last_dev.size() != m->devices.size()) return fail(RAYZ_ERR_STATE, "no frame has been rendered on this handle");
    if (gather_ms) *gather_ms = m->last_gather_ms;
    if (frame_ms) *frame_ms = m->last_frame_ms;
    return RAYZ_OK;
}

int rayz_hip_multi_render(RayzMulti* m, const RayzCameraDesc* cam, const RayzRenderParams* p, float* rgb_out, RayzRenderStats* stats) {
    return guarded([&] { return multi_render<float>(m, cam, p, rgb_out, stats); });
}
int rayz_hip_multi_render_f64(RayzMulti* m, const RayzCameraDesc* cam, const RayzRenderParams* p, double* rgb_out,
                              RayzRenderStats* stats) {
    return guarded([&] { return multi_render<double>(m, cam, p, rgb_out, stats); });
}
int rayz_hip_multi_render_u8(RayzMulti* m, const RayzCameraDesc* cam, const RayzRenderParams* p, uint8_t* rgb8_out,
                             RayzRenderStats* stats) {
    return guarded([&] { return multi_render<uint8_t>(m, cam, p, rgb8_out, stats); });
}

int rayz_hip_render_multi(const int* devices, int n_devices, const RayzSceneDesc* scene, const RayzCameraDesc* cam,
                          const RayzRenderParams* p, float* rgb_out, RayzRenderStats* stats) {
    return guarded([&] { return render_multi_oneshot<float>(devices, n_devices, scene, cam, p, rgb_out, stats); });
}
int rayz_hip_render_multi_f64(const int* devices, int n_devices, const RayzSceneDesc* scene, const RayzCameraDesc* cam,
                              const RayzRenderParams* p, double* rgb_out, RayzRenderStats* stats) {
    return guarded([&] { return render_multi_oneshot<double>(devices, n_devices, scene, cam, p, rgb_out, stats); });
}

} // extern "C"
