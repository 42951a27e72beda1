// rayz_hip.hip — the C ABI of include/rayz_hip.h: scene upload, workspace, kernel launches.
//
// Replaces the body of `Tracer.render()` (src/renderer.zig:72-101 of jlucier/rayz).  One process
// drives one GPU; multi-GPU sharding is by interleaved row tiles (params.shard_*), the gather is the
// caller's (RCCL through torch.distributed in bench.py).
#include "../../include/rayz_hip.h"
#include "rayz_device.hpp"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <new>
#include <vector>

using namespace rayz_dev;

namespace {

thread_local char g_err[512] = "";
int g_device = -1;
hipStream_t g_stream = nullptr;
int g_num_cu = 0;

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return fail(e_ == hipErrorOutOfMemory ? RAYZ_ERR_OOM : RAYZ_ERR_HIP, "%s: %s (%s:%d)", #expr,     \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                         \
    } while (0)

template <class R> struct Bits;
template <> struct Bits<float> {
    static float from(uint32_t u) {
        float f;
        std::memcpy(&f, &u, 4);
        return f;
    }
};
template <> struct Bits<double> {
    static double from(uint32_t u) {
        uint64_t w = u;
        double d;
        std::memcpy(&d, &w, 8);
        return d;
    }
};

// Device copy of the scene in one precision.
template <class R> struct SceneBuffers {
    typedef typename VecOf<R>::type r4;
    r4* stat = nullptr;
    r4* mov = nullptr;
    uint32_t* sphere_mat = nullptr;
    r4* mat = nullptr;
    r4* tex = nullptr;
    uint32_t ns_pad = 0, nm_pad = 0;
    bool ready = false;
    void release() {
        (void)hipFree(stat);
        (void)hipFree(mov);
        (void)hipFree(sphere_mat);
        (void)hipFree(mat);
        (void)hipFree(tex);
        stat = mov = mat = tex = nullptr;
        sphere_mat = nullptr;
        ready = false;
    }
};

uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

} // namespace

struct RayzScene {
    std::vector<RayzSphere> spheres;
    std::vector<RayzMaterial> materials;
    std::vector<RayzTexture> textures;
    SceneBuffers<float> f32;
    SceneBuffers<double> f64;
    void* partial = nullptr; // chunk sums, grow-only
    size_t partial_bytes = 0;
    unsigned long long* counters = nullptr; // [0] queue head, [1] segments
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t last_stream = nullptr;
    bool rendered = false;
    RayzRenderStats last{};
};

namespace {

template <class R> int upload(RayzScene* s, SceneBuffers<R>& b) {
    typedef typename VecOf<R>::type r4;
    if (b.ready) return RAYZ_OK;
    const R ninf = -std::numeric_limits<R>::infinity();
    std::vector<r4> stat, mov;
    std::vector<uint32_t> smat, mmat;
    for (const RayzSphere& q : s->spheres) {
        const bool moving = q.velocity[0] != 0 || q.velocity[1] != 0 || q.velocity[2] != 0;
        const R r = (R)q.radius; // r² in R: for R = double this is the reference's radius * radius, src/geom.zig:45
        const r4 c = {(R)q.center[0], (R)q.center[1], (R)q.center[2], r * r};
        if (!moving) {
            stat.push_back(c);
            smat.push_back(q.material);
        } else {
            mov.push_back(c);
            mov.push_back(r4{(R)q.velocity[0], (R)q.velocity[1], (R)q.velocity[2], R(0)});
            mmat.push_back(q.material);
        }
    }
    b.ns_pad = round_up((uint32_t)stat.size(), kStaticUnroll);
    b.nm_pad = round_up((uint32_t)mmat.size(), kMovingUnroll);
    const r4 pad = {R(0), R(0), R(0), ninf}; // r² = -inf: discriminant is -inf (or NaN), never ≥ 0
    while (stat.size() < b.ns_pad) {
        stat.push_back(pad);
        smat.push_back(0);
    }
    while (mmat.size() < b.nm_pad) {
        mov.push_back(pad);
        mov.push_back(r4{R(0), R(0), R(0), R(0)});
        mmat.push_back(0);
    }
    std::vector<uint32_t> sphere_mat(smat);
    sphere_mat.insert(sphere_mat.end(), mmat.begin(), mmat.end());
    std::vector<r4> mat, tex;
    for (const RayzMaterial& m : s->materials) {
        const R p = (R)m.param;
        mat.push_back(r4{Bits<R>::from(m.kind | (m.method << 8)), Bits<R>::from(m.texture), p, R(1) / p});
    }
    for (const RayzTexture& t : s->textures) {
        tex.push_back(r4{Bits<R>::from(t.kind), Bits<R>::from(t.even), Bits<R>::from(t.odd), (R)t.scale});
        tex.push_back(r4{(R)t.color[0], (R)t.color[1], (R)t.color[2], R(0)});
    }
    auto put = [](auto** dst, const auto& v) -> hipError_t {
        const size_t bytes = v.size() * sizeof(v[0]);
        hipError_t e = hipMalloc((void**)dst, bytes ? bytes : 16);
        if (e != hipSuccess) return e;
        return bytes ? hipMemcpy(*dst, v.data(), bytes, hipMemcpyHostToDevice) : hipSuccess;
    };
    HIP_TRY(put(&b.stat, stat));
    HIP_TRY(put(&b.mov, mov));
    HIP_TRY(put(&b.sphere_mat, sphere_mat));
    HIP_TRY(put(&b.mat, mat));
    HIP_TRY(put(&b.tex, tex));
    b.ready = true;
    return RAYZ_OK;
}

int validate_scene(const RayzSceneDesc* d) {
    if (!d) return fail(RAYZ_ERR_BAD_ARG, "scene is null");
    if ((d->n_spheres && !d->spheres) || (d->n_materials && !d->materials) || (d->n_textures && !d->textures))
        return fail(RAYZ_ERR_BAD_ARG, "scene list pointer is null");
    for (uint32_t i = 0; i < d->n_textures; ++i) {
        const RayzTexture& t = d->textures[i];
        if (t.kind > RAYZ_TEX_SOLID) return fail(RAYZ_ERR_BAD_ARG, "texture %u: bad kind %u", i, t.kind);
        if (t.kind == RAYZ_TEX_CHECKER && (t.even >= d->n_textures || t.odd >= d->n_textures))
            return fail(RAYZ_ERR_BAD_ARG, "texture %u: checker handle out of range", i);
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
    if (p->traversal > RAYZ_TRAVERSAL_BVH) return fail(RAYZ_ERR_BAD_ARG, "bad traversal %u", p->traversal);
    const uint32_t sc = p->shard_count ? p->shard_count : 1;
    if (p->shard_index >= sc) return fail(RAYZ_ERR_BAD_ARG, "shard_index %u >= shard_count %u", p->shard_index, sc);
    if (!(p->tmin == p->tmin)) return fail(RAYZ_ERR_BAD_ARG, "tmin is NaN");
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

template <class R>
int render_impl(RayzScene* s, SceneBuffers<R>& b, const RayzCameraDesc* cam, const RayzRenderParams* p, R* d_out,
                hipStream_t stream) {
    typedef typename VecOf<R>::type r4;
    if (g_device < 0) return fail(RAYZ_ERR_NO_DEVICE, "rayz_hip_init has not succeeded");
    if (p->traversal != RAYZ_TRAVERSAL_LINEAR) return fail(RAYZ_ERR_BAD_ARG, "BVH traversal is not built yet");
    int rc = upload<R>(s, b);
    if (rc != RAYZ_OK) return rc;

    const uint32_t rows = rayz_hip_shard_rows(p);
    const uint64_t shard_pixels64 = (uint64_t)rows * p->width;
    const uint32_t chunk = p->chunk_spp ? p->chunk_spp : 16u;
    const uint32_t chunks_per_px = (p->samples_per_px + chunk - 1) / chunk;
    const uint64_t items64 = shard_pixels64 * chunks_per_px;
    if (shard_pixels64 >= (1ull << 31) || items64 >= (1ull << 32) - (1ull << 26))
        return fail(RAYZ_ERR_BAD_ARG, "too many work items (%llu): raise chunk_spp", (unsigned long long)items64);
    s->last = RayzRenderStats{};
    s->last.primary_rays = shard_pixels64 * p->samples_per_px;
    s->last_stream = stream;
    if (items64 == 0) {
        s->rendered = false;
        return RAYZ_OK;
    }
    if (!d_out) return fail(RAYZ_ERR_BAD_ARG, "output pointer is null");
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
    if (!s->counters) HIP_TRY(hipMalloc((void**)&s->counters, 4 * sizeof(unsigned long long)));
    if (!s->ev0) {
        HIP_TRY(hipEventCreate(&s->ev0));
        HIP_TRY(hipEventCreate(&s->ev1));
    }

    TraceArgs<R> A{};
    A.sc.stat = b.stat;
    A.sc.mov = b.mov;
    if (sizeof(R) == sizeof(double)) {
        A.sc.stat64 = (const d4*)b.stat;
        A.sc.mov64 = (const d4*)b.mov;
    } else {
        rc = upload<double>(s, s->f64); // narrow phase reads the pool's f64 records
        if (rc != RAYZ_OK) return rc;
        if (s->f64.ns_pad != b.ns_pad || s->f64.nm_pad != b.nm_pad) return fail(RAYZ_ERR_STATE, "f32/f64 layouts differ");
        A.sc.stat64 = s->f64.stat;
        A.sc.mov64 = s->f64.mov;
    }
    A.sc.sphere_mat = b.sphere_mat;
    A.sc.mat = b.mat;
    A.sc.tex = b.tex;
    A.sc.ns_pad = b.ns_pad;
    A.sc.nm_pad = b.nm_pad;
    A.sc.n_spheres = (uint32_t)s->spheres.size();
    fill_camera<R>(cam, A.cam);
    A.partial = (r4*)s->partial;
    A.counters = s->counters;
    A.seed = p->seed;
    A.tmin = (R)p->tmin;
    A.width = p->width;
    A.height = p->height;
    A.spp = p->samples_per_px;
    A.max_bounces = p->max_bounces;
    A.chunk_spp = chunk;
    A.chunks_per_px = chunks_per_px;
    A.tile_rows = p->tile_rows ? p->tile_rows : 8u;
    A.shard_index = p->shard_index;
    A.shard_count = p->shard_count ? p->shard_count : 1u;
    A.shard_pixels = (uint32_t)shard_pixels64;
    A.total_items = (uint32_t)items64;

    int blocks_per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, trace_kernel<R>, 256, 0));
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    uint64_t grid = (uint64_t)g_num_cu * blocks_per_cu;
    const uint64_t want = (items64 + 255) / 256;
    if (grid > want) grid = want;

    HIP_TRY(hipMemsetAsync(s->counters, 0, 4 * sizeof(unsigned long long), stream));
    HIP_TRY(hipEventRecord(s->ev0, stream));
    hipLaunchKernelGGL(trace_kernel<R>, dim3((uint32_t)grid), dim3(256), 0, stream, A);
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
int render_oneshot(const RayzSceneDesc* scene, const RayzCameraDesc* cam, const RayzRenderParams* p, R* out,
                   RayzRenderStats* stats, uint32_t precision) {
    if (!out) return fail(RAYZ_ERR_BAD_ARG, "output pointer is null");
    int rc = validate_params(p);
    if (rc != RAYZ_OK) return rc;
    if (g_device < 0) {
        rc = rayz_hip_init(0);
        if (rc != RAYZ_OK) return rc;
    }
    RayzScene* s = nullptr;
    rc = rayz_hip_scene_create(scene, &s);
    if (rc != RAYZ_OK) return rc;
    const size_t n = (size_t)rayz_hip_shard_rows(p) * p->width * 3;
    R* d_out = nullptr;
    hipError_t e = hipMalloc((void**)&d_out, n ? n * sizeof(R) : 16);
    if (e != hipSuccess) {
        rayz_hip_scene_destroy(s);
        return fail(RAYZ_ERR_OOM, "hipMalloc(output): %s", hipGetErrorString(e));
    }
    rc = check_render_args(s, cam, p, precision);
    if (rc == RAYZ_OK) {
        if (precision == RAYZ_PRECISION_F32) rc = render_impl<float>(s, s->f32, cam, p, (float*)d_out, g_stream);
        else rc = render_impl<double>(s, s->f64, cam, p, (double*)d_out, g_stream);
    }
    if (rc == RAYZ_OK) rc = rayz_hip_scene_sync(s, stats);
    if (rc == RAYZ_OK && n) {
        e = hipMemcpy(out, d_out, n * sizeof(R), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(RAYZ_ERR_HIP, "hipMemcpy(output): %s", hipGetErrorString(e));
    }
    (void)hipFree(d_out);
    rayz_hip_scene_destroy(s);
    return rc;
}

} // namespace

extern "C" {

uint32_t rayz_hip_abi_version(void) { return RAYZ_HIP_ABI_VERSION; }
const char* rayz_hip_last_error(void) { return g_err; }

int rayz_hip_init(int device) {
    if (g_device == device && g_stream) return RAYZ_OK;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(RAYZ_ERR_NO_DEVICE, "no HIP device: %s", hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(RAYZ_ERR_BAD_ARG, "device %d out of range [0,%d)", device, n);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RAYZ_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);
    if (g_stream) {
        (void)hipStreamDestroy(g_stream);
        g_stream = nullptr;
    }
    HIP_TRY(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
    g_num_cu = prop.multiProcessorCount;
    g_device = device;
    return RAYZ_OK;
}

void rayz_hip_shutdown(void) {
    if (g_stream) (void)hipStreamDestroy(g_stream);
    g_stream = nullptr;
    g_device = -1;
}

uint32_t rayz_hip_shard_rows(const RayzRenderParams* p) {
    if (!p) return 0;
    const uint32_t tr = p->tile_rows ? p->tile_rows : 8u, sc = p->shard_count ? p->shard_count : 1u;
    if (p->shard_index >= sc) return 0;
    uint32_t n = 0;
    for (uint32_t t = p->shard_index; (uint64_t)t * tr < p->height; t += sc) {
        const uint32_t r0 = t * tr;
        n += (p->height - r0 < tr) ? p->height - r0 : tr;
    }
    return n;
}

int rayz_hip_scene_create(const RayzSceneDesc* scene, RayzScene** out) {
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
    } catch (...) {
        delete s;
        return fail(RAYZ_ERR_OOM, "host allocation failed");
    }
    *out = s;
    return RAYZ_OK;
}

int rayz_hip_scene_destroy(RayzScene* s) {
    if (!s) return RAYZ_OK;
    if (s->last_stream || g_stream) (void)hipStreamSynchronize(s->last_stream ? s->last_stream : g_stream);
    s->f32.release();
    s->f64.release();
    (void)hipFree(s->partial);
    (void)hipFree(s->counters);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    delete s;
    return RAYZ_OK;
}

int rayz_hip_render_device(RayzScene* s, const RayzCameraDesc* cam, const RayzRenderParams* p, float* d_out,
                           void* stream) {
    int rc = check_render_args(s, cam, p, RAYZ_PRECISION_F32);
    if (rc != RAYZ_OK) return rc;
    return render_impl<float>(s, s->f32, cam, p, d_out, stream ? (hipStream_t)stream : g_stream);
}

int rayz_hip_render_device_f64(RayzScene* s, const RayzCameraDesc* cam, const RayzRenderParams* p, double* d_out,
                               void* stream) {
    int rc = check_render_args(s, cam, p, RAYZ_PRECISION_F64);
    if (rc != RAYZ_OK) return rc;
    return render_impl<double>(s, s->f64, cam, p, d_out, stream ? (hipStream_t)stream : g_stream);
}

int rayz_hip_scene_sync(RayzScene* s, RayzRenderStats* stats) {
    if (!s) return fail(RAYZ_ERR_STATE, "scene handle is null");
    if (g_device < 0) return fail(RAYZ_ERR_NO_DEVICE, "rayz_hip_init has not succeeded");
    HIP_TRY(hipStreamSynchronize(s->last_stream ? s->last_stream : g_stream));
    if (s->rendered) {
        unsigned long long c[4] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpy(c, s->counters, sizeof(c), hipMemcpyDeviceToHost));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
        s->last.segments = c[1];
        s->last.sphere_tests = c[1] * (unsigned long long)s->spheres.size();
        s->last.node_tests = 0;
        s->last.kernel_ms = ms;
    }
    if (stats) *stats = s->last;
    return RAYZ_OK;
}

int rayz_hip_render(const RayzSceneDesc* scene, const RayzCameraDesc* cam, const RayzRenderParams* p, float* out,
                    RayzRenderStats* stats) {
    return render_oneshot<float>(scene, cam, p, out, stats, RAYZ_PRECISION_F32);
}

int rayz_hip_render_f64(const RayzSceneDesc* scene, const RayzCameraDesc* cam, const RayzRenderParams* p, double* out,
                        RayzRenderStats* stats) {
    return render_oneshot<double>(scene, cam, p, out, stats, RAYZ_PRECISION_F64);
}

int rayz_hip_tonemap_u8(const float* d_rgb, uint8_t* d_rgb8, size_t n_pixels, void* stream) {
    if (g_device < 0) return fail(RAYZ_ERR_NO_DEVICE, "rayz_hip_init has not succeeded");
    if (!n_pixels) return RAYZ_OK;
    if (!d_rgb || !d_rgb8) return fail(RAYZ_ERR_BAD_ARG, "null buffer");
    const size_t n = n_pixels * 3;
    hipLaunchKernelGGL(tonemap_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0,
                       stream ? (hipStream_t)stream : g_stream, d_rgb, d_rgb8, n);
    HIP_TRY(hipGetLastError());
    return RAYZ_OK;
}

} // extern "C"
