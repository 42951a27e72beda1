// rayz.hpp — host-side mirror of rayz's Scene/Camera/Image/Tracer API, in C++ because the
// reference's own language (Zig) has no toolchain in this image.  Same names, argument meaning and
// error behaviour as the reference for the path `main → randomBouncing → Tracer.render → writePPM`;
// `render()` flattens the pool and the camera into the PODs of include/rayz_hip.h and calls the HIP
// library instead of running the loop nest.  Nothing here traces rays: there is no CPU fallback.
//
// file:line citations are into jlucier/rayz (src/...).
#pragma once

#include "../../include/rayz_hip.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

namespace rayz {

// ---- src/utils.zig:3-13 --------------------------------------------------------------------------
namespace utils {
template <class T> inline T min(T a, T b) { return a < b ? a : b; }
template <class T> inline T max(T a, T b) { return a > b ? a : b; }
template <class T> inline T clamp(T x, T low, T high) { return min<T>(max<T>(x, low), high); }
} // namespace utils

// ---- src/vec.zig:4-157 (the subset host code needs; device math lives in csrc/) -----------------
struct V3 {
    double x = 0, y = 0, z = 0;
    static V3 of(double v) { return {v, v, v}; }
    static V3 ones() { return of(1); }
    static V3 x_hat() { return {1, 0, 0}; }
    static V3 y_hat() { return {0, 1, 0}; }
    static V3 z_hat() { return {0, 0, 1}; }
    double at(unsigned axis) const { return axis == 0 ? x : (axis == 1 ? y : z); }
    V3 add(V3 o) const { return {x + o.x, y + o.y, z + o.z}; }
    V3 sub(V3 o) const { return {x - o.x, y - o.y, z - o.z}; }
    V3 mul(double v) const { return {x * v, y * v, z * v}; }
    V3 div(double v) const { return mul(1 / v); } // multiply by reciprocal, src/vec.zig:67-69
    double dot(V3 o) const { return x * o.x + y * o.y + z * o.z; }
    double mag() const { return std::sqrt(dot(*this)); }
    V3 unit() const { return div(mag()); }
    V3 cross(V3 o) const { return {y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x}; }
    V3 vmul(V3 o) const { return {x * o.x, y * o.y, z * o.z}; }
    V3 vmin(V3 o) const { return {std::fmin(x, o.x), std::fmin(y, o.y), std::fmin(z, o.z)}; }
    V3 vmax(V3 o) const { return {std::fmax(x, o.x), std::fmax(y, o.y), std::fmax(z, o.z)}; }
    bool nearZero() const {
        const double tol = 1e-8;
        return std::fabs(x) <= tol && std::fabs(y) <= tol && std::fabs(z) <= tol;
    }
    bool close(V3 o) const { return sub(o).nearZero(); }
    V3 sqrt() const { return {x > 0 ? std::sqrt(x) : 0, y > 0 ? std::sqrt(y) : 0, z > 0 ? std::sqrt(z) : 0}; }
    V3 clamp(double lo, double hi) const {
        return {utils::clamp<double>(x, lo, hi), utils::clamp<double>(y, lo, hi), utils::clamp<double>(z, lo, hi)};
    }
    unsigned amax() const { // src/vec.zig:150-156
        if (x > y) return x > z ? 0 : 2;
        return y > z ? 1 : 2;
    }
};

struct Ray { // src/vec.zig:159-167
    V3 origin, dir;
    double time = 0;
    V3 at(double t) const { return origin.add(dir.mul(t)); }
};

// ---- std.Random.DefaultPrng as the reference uses it (src/renderer.zig:22,55-59) ---------------
// xoshiro256++ seeded through SplitMix64; float() is Zig 0.13/0.14's Random.float(f64).
struct DefaultPrng {
    uint64_t s[4];
    static DefaultPrng init(uint64_t seed) {
        DefaultPrng g;
        uint64_t x = seed;
        for (int i = 0; i < 4; ++i) {
            x += 0x9e3779b97f4a7c15ull;
            uint64_t z = x;
            z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
            z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
            g.s[i] = z ^ (z >> 31);
        }
        return g;
    }
    static uint64_t rotl(uint64_t v, int k) { return (v << k) | (v >> (64 - k)); }
    uint64_t next() {
        const uint64_t r = rotl(s[0] + s[3], 23) + s[0];
        const uint64_t t = s[1] << 17;
        s[2] ^= s[0], s[3] ^= s[1], s[1] ^= s[2], s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return r;
    }
    double float64() {
        const uint64_t r = next();
        unsigned lz = r ? (unsigned)__builtin_clzll(r) : 64u;
        if (lz >= 12) {
            lz = 12;
            for (;;) {
                const uint64_t a = next();
                const unsigned alz = a ? (unsigned)__builtin_clzll(a) : 64u;
                lz += alz;
                if (alz != 64) break;
                if (lz >= 1022) {
                    lz = 1022;
                    break;
                }
            }
        }
        const uint64_t b = ((uint64_t)(1022 - lz) << 52) | (r & 0xFFFFFFFFFFFFFull);
        double d;
        std::memcpy(&d, &b, 8);
        return d;
    }
    V3 v3(double low, double high) { // V3.random, src/vec.zig:9-16
        const double scale = high - low;
        V3 v;
        v.x = float64() * scale + low;
        v.y = float64() * scale + low;
        v.z = float64() * scale + low;
        return v;
    }
};

// ---- src/ecs.zig:6-20 -----------------------------------------------------------------------------
struct TextureHandle { size_t idx = 0; };
struct MaterialHandle { size_t idx = 0; };

// ---- src/material.zig:19-51, 67-75, 104-106, 134-135, 162-165 (data only; scatter runs on the GPU) --
struct SolidTexture { V3 color; };
struct CheckerTexture { double scale = 1; TextureHandle even, odd; };
struct Texture {
    RayzTextureKind kind = RAYZ_TEX_SOLID;
    CheckerTexture checker;
    SolidTexture solid;
    static Texture Solid(V3 color) {
        Texture t;
        t.kind = RAYZ_TEX_SOLID;
        t.solid.color = color;
        return t;
    }
    static Texture Checker(double scale, TextureHandle even, TextureHandle odd) {
        Texture t;
        t.kind = RAYZ_TEX_CHECKER;
        t.checker = {scale, even, odd};
        return t;
    }
};
typedef RayzDiffuseMethod DiffuseScatterMethod;
struct DiffuseMaterial { DiffuseScatterMethod method = RAYZ_DIFFUSE_HEMISPHERE; TextureHandle texture; };
struct MetallicMaterial { double fuzz = 0; TextureHandle texture; };
struct DielectricMaterial { double refractive_index = 1.0; };
struct Material {
    RayzMaterialKind kind = RAYZ_MAT_DIFFUSE;
    DiffuseMaterial diffuse;
    MetallicMaterial metallic;
    DielectricMaterial dielectric;
    static Material Diffuse(TextureHandle t, DiffuseScatterMethod m = RAYZ_DIFFUSE_HEMISPHERE) {
        Material r;
        r.kind = RAYZ_MAT_DIFFUSE;
        r.diffuse = {m, t};
        return r;
    }
    static Material Metallic(TextureHandle t, double fuzz = 0) {
        Material r;
        r.kind = RAYZ_MAT_METALLIC;
        r.metallic = {fuzz, t};
        return r;
    }
    static Material Dielectric(double ri = 1.0) {
        Material r;
        r.kind = RAYZ_MAT_DIELECTRIC;
        r.dielectric = {ri};
        return r;
    }
};

// ---- src/geom.zig:11-22 ---------------------------------------------------------------------------
struct Sphere {
    Ray center;
    double radius = 0;
    MaterialHandle material;
    static Sphere stationary(V3 c, double radius, MaterialHandle m) {
        Sphere s;
        s.center.origin = c;
        s.center.dir = V3{};
        s.radius = radius;
        s.material = m;
        return s;
    }
};

// ---- build-defined: a triangle hittable (the reference's geom.zig has only Sphere) --------------------
struct Triangle {
    V3 v0, v1, v2;
    MaterialHandle material;
};

// ---- src/ecs.zig:22-69 (+ the triangle list) ----------------------------------------------------------
struct MemPool {
    std::vector<Sphere> spheres;
    std::vector<Material> materials;
    std::vector<Texture> textures;
    std::vector<Triangle> triangles;
    void add(const Sphere& s) { spheres.push_back(s); }
    void add(const Triangle& t) { triangles.push_back(t); }
    size_t addAndReturnHandle(const Triangle& t) {
        triangles.push_back(t);
        return triangles.size() - 1;
    }
    TextureHandle addAndReturnHandle(const Texture& t) {
        textures.push_back(t);
        return {textures.size() - 1};
    }
    MaterialHandle addAndReturnHandle(const Material& m) {
        materials.push_back(m);
        return {materials.size() - 1};
    }
    size_t addAndReturnHandle(const Sphere& s) {
        spheres.push_back(s);
        return spheres.size() - 1;
    }
};

// ---- src/camera.zig:8-77 --------------------------------------------------------------------------
struct Camera {
    V3 look_from, px_du, px_dv, px_origin, defocus_u, defocus_v;
    bool defocus = false;
    static Camera init(double vfov, double focus_dist, double defocus_angle, V3 look_from, V3 look_at, V3 vup,
                       size_t img_height, size_t img_width) {
        const double DEG_TO_RAD = M_PI / 180.0;
        const double fimg_h = (double)img_height, fimg_w = (double)img_width;
        const double vp_height = 2 * std::tan(vfov * DEG_TO_RAD / 2.0) * focus_dist;
        const double vp_width = vp_height * fimg_w / fimg_h;
        const V3 w = look_from.sub(look_at).unit();
        const V3 u = vup.cross(w).unit();
        const V3 v = w.cross(u);
        const V3 vp_u = u.mul(vp_width), vp_v = v.mul(-vp_height);
        Camera c;
        c.px_du = vp_u.div(fimg_w);
        c.px_dv = vp_v.div(fimg_h);
        const double defocus_radius = std::tan(defocus_angle * DEG_TO_RAD / 2) * focus_dist;
        c.px_origin = look_from.sub(w.mul(focus_dist)).sub(vp_u.div(2)).sub(vp_v.div(2)).add(
            c.px_du.add(c.px_dv).mul(0.5));
        c.look_from = look_from;
        c.defocus_u = u.mul(defocus_radius);
        c.defocus_v = v.mul(defocus_radius);
        c.defocus = defocus_angle > 0;
        return c;
    }
    // getRay(px, py, null): the rng-less form (src/camera.zig:59-77 with rng == null).  The jittered
    // form is device code (csrc/rayz_device.hpp camera_ray).
    Ray getRay(size_t px, size_t py) const {
        const double x = (double)px, y = (double)py;
        Ray r;
        r.origin = look_from;
        r.dir = px_du.mul(x).add(px_dv.mul(y)).add(px_origin).sub(look_from);
        r.time = 0;
        return r;
    }
    RayzCameraDesc desc() const {
        RayzCameraDesc d{};
        auto put = [](double* p, V3 v) { p[0] = v.x, p[1] = v.y, p[2] = v.z; };
        put(d.look_from, look_from);
        put(d.px_du, px_du);
        put(d.px_dv, px_dv);
        put(d.px_origin, px_origin);
        put(d.defocus_u, defocus_u);
        put(d.defocus_v, defocus_v);
        d.defocus = defocus ? 1u : 0u;
        return d;
    }
};

// ---- src/image.zig:4-41 ---------------------------------------------------------------------------
struct Image {
    size_t h = 0, w = 0;
    std::vector<V3> pixels;
    static Image initEmpty(size_t h, size_t w) {
        Image im;
        im.h = h, im.w = w;
        im.pixels.resize(h * w);
        return im;
    }
    static void toU8(V3 px, uint8_t out[3]) { // src/image.zig:35-38
        const V3 clm = px.sqrt().clamp(0, 1);
        out[0] = (uint8_t)(clm.x * 255), out[1] = (uint8_t)(clm.y * 255), out[2] = (uint8_t)(clm.z * 255);
    }
    void writePPM(FILE* f) const {
        std::fprintf(f, "P3\n%zu %zu\n%d\n", w, h, 255);
        std::string buf;
        buf.reserve(1 << 20);
        char line[48];
        for (const V3& px : pixels) {
            uint8_t c[3];
            toU8(px, c);
            const int n = std::snprintf(line, sizeof(line), "%u %u %u\n", c[0], c[1], c[2]);
            buf.append(line, (size_t)n);
            if (buf.size() > (1 << 20) - 64) {
                std::fwrite(buf.data(), 1, buf.size(), f);
                buf.clear();
            }
        }
        std::fwrite(buf.data(), 1, buf.size(), f);
    }
};

struct GpuRenderFailed : std::runtime_error {
    int status;
    GpuRenderFailed(int st, const std::string& m) : std::runtime_error(m), status(st) {}
};

// What the MI355X path adds to the reference's Tracer fields (none of these exist in src/renderer.zig).
struct GpuOptions {
    RayzPrecision precision = RAYZ_PRECISION_F32;
    RayzTraversal traversal = RAYZ_TRAVERSAL_AUTO; // the reference always walks its BVH; AUTO picks what is faster here
    double tmin = -1;         // < 0: 1e-3 for f32 (1e-10 is unusable in f32), the reference's 1e-10 for f64
    bool has_render_seed = false;
    uint64_t render_seed = 0; // else drawn from `rng` when render() starts
    uint32_t chunk_spp = 0;
    uint32_t tile_rows = 0, shard_index = 0, shard_count = 0;
    std::vector<int> devices; // empty: the default device; else render() drives all of these (one kept RayzMulti, rayz_hip_multi_render)
};

static const double ASPECT_RATIO = 16.0 / 9.0; // src/renderer.zig:16

// ---- src/renderer.zig:18-101 ----------------------------------------------------------------------
struct Tracer {
    Camera camera;
    Image img;
    DefaultPrng rng;
    size_t max_bounces = 50;
    size_t samples_per_px = 10;
    MemPool pool;
    GpuOptions gpu;
    RayzRenderStats stats{};
    // Several GPUs (gpu.devices): the per-device scenes and the RCCL communicators (`ncclCommInitAll`: tens of
    // milliseconds per device) are kept across render() calls and rebuilt only when the pool or the device list changes,
    // instead of paying rayz_hip_render_multi's create + destroy per frame.  Shared by copies of the Tracer.
    struct MultiCache {
        RayzMulti* handle = nullptr;
        std::vector<int> devices;
        std::vector<unsigned char> pool_bytes; // the flattened pool the handle was created from
        ~MultiCache() {
            if (handle) rayz_hip_multi_destroy(handle);
        }
    };
    std::shared_ptr<MultiCache> multi_cache;

    // `seed` == nullptr seeds from the OS as the reference does (std.posix.getrandom, :55-59)
    static Tracer init(size_t img_w, double vfov, double focus_dist, double defocus_angle, V3 look_from, V3 look_at,
                       V3 vup, const uint64_t* seed = nullptr) {
        const double fimg_w = (double)img_w;
        const size_t height = (size_t)(fimg_w / ASPECT_RATIO);
        Tracer t;
        t.camera = Camera::init(vfov, focus_dist, defocus_angle, look_from, look_at, vup, height, img_w);
        t.img = Image::initEmpty(height, img_w);
        uint64_t sd;
        if (seed) sd = *seed;
        else {
            std::random_device rd;
            sd = ((uint64_t)rd() << 32) ^ rd();
        }
        t.rng = DefaultPrng::init(sd);
        return t;
    }

    struct Flat {
        std::vector<RayzSphere> spheres;
        std::vector<RayzMaterial> materials;
        std::vector<RayzTexture> textures;
        std::vector<RayzTriangle> triangles;
        RayzSceneDesc desc() const {
            RayzSceneDesc d{};
            d.spheres = spheres.data(), d.materials = materials.data(), d.textures = textures.data();
            d.n_spheres = (uint32_t)spheres.size(), d.n_materials = (uint32_t)materials.size();
            d.n_textures = (uint32_t)textures.size();
            d.triangles = triangles.data(), d.n_triangles = (uint32_t)triangles.size();
            return d;
        }
    };
    // field-by-field copy of the pool into extern-compatible PODs (the Zig side must do the same)
    Flat flatten() const {
        Flat f;
        for (const Sphere& s : pool.spheres) {
            RayzSphere q{};
            q.center[0] = s.center.origin.x, q.center[1] = s.center.origin.y, q.center[2] = s.center.origin.z;
            q.velocity[0] = s.center.dir.x, q.velocity[1] = s.center.dir.y, q.velocity[2] = s.center.dir.z;
            q.radius = s.radius;
            q.material = (uint32_t)s.material.idx;
            f.spheres.push_back(q);
        }
        for (const Triangle& t : pool.triangles) {
            RayzTriangle q{};
            q.v0[0] = t.v0.x, q.v0[1] = t.v0.y, q.v0[2] = t.v0.z;
            q.v1[0] = t.v1.x, q.v1[1] = t.v1.y, q.v1[2] = t.v1.z;
            q.v2[0] = t.v2.x, q.v2[1] = t.v2.y, q.v2[2] = t.v2.z;
            q.material = (uint32_t)t.material.idx;
            f.triangles.push_back(q);
        }
        for (const Material& m : pool.materials) {
            RayzMaterial q{};
            q.kind = m.kind;
            q.method = RAYZ_DIFFUSE_HEMISPHERE;
            if (m.kind == RAYZ_MAT_DIFFUSE) q.texture = (uint32_t)m.diffuse.texture.idx, q.method = m.diffuse.method;
            else if (m.kind == RAYZ_MAT_METALLIC) q.texture = (uint32_t)m.metallic.texture.idx, q.param = m.metallic.fuzz;
            else q.param = m.dielectric.refractive_index;
            f.materials.push_back(q);
        }
        for (const Texture& t : pool.textures) {
            RayzTexture q{};
            q.kind = t.kind;
            if (t.kind == RAYZ_TEX_CHECKER) {
                q.scale = t.checker.scale;
                q.even = (uint32_t)t.checker.even.idx, q.odd = (uint32_t)t.checker.odd.idx;
            } else {
                q.color[0] = t.solid.color.x, q.color[1] = t.solid.color.y, q.color[2] = t.solid.color.z;
            }
            f.textures.push_back(q);
        }
        return f;
    }
    // the params render() passes; `seed` is filled by the caller
    RayzRenderParams params(uint64_t seed) const {
        RayzRenderParams p{};
        p.width = (uint32_t)img.w, p.height = (uint32_t)img.h;
        p.samples_per_px = (uint32_t)samples_per_px, p.max_bounces = (uint32_t)max_bounces;
        p.seed = seed;
        p.precision = gpu.precision, p.traversal = gpu.traversal;
        p.tmin = gpu.tmin >= 0 ? gpu.tmin : (gpu.precision == RAYZ_PRECISION_F32 ? 1e-3 : 1e-10);
        p.chunk_spp = gpu.chunk_spp;
        p.tile_rows = gpu.tile_rows, p.shard_index = gpu.shard_index, p.shard_count = gpu.shard_count;
        return p;
    }

    // src/renderer.zig:72-101.  Returns the primary-ray count; throws GpuRenderFailed where the Zig
    // drop-in returns error.GpuRenderFailed.
    size_t render() {
        const uint64_t seed = gpu.has_render_seed ? gpu.render_seed : rng.next();
        const Flat f = flatten();
        const RayzSceneDesc sd = f.desc();
        const RayzCameraDesc cd = camera.desc();
        RayzRenderParams p = params(seed);
        p.shard_index = 0, p.shard_count = 1; // a Tracer owns a whole image
        const size_t n = img.h * img.w;
        int rc = RAYZ_OK;
        const bool multi = !gpu.devices.empty();
        RayzMulti* mh = nullptr;
        if (multi) {
            p.shard_count = 0; // the library deals the rows to gpu.devices itself
            std::vector<unsigned char> bytes; // what the pool looks like across the ABI (the reference may edit it between renders)
            auto append = [&](const void* q, size_t nbytes) { bytes.insert(bytes.end(), (const unsigned char*)q, (const unsigned char*)q + nbytes); };
            append(f.spheres.data(), f.spheres.size() * sizeof(RayzSphere));
            append(f.materials.data(), f.materials.size() * sizeof(RayzMaterial));
            append(f.textures.data(), f.textures.size() * sizeof(RayzTexture));
            append(f.triangles.data(), f.triangles.size() * sizeof(RayzTriangle));
            if (!multi_cache) multi_cache = std::make_shared<MultiCache>();
            MultiCache& mc = *multi_cache;
            if (!mc.handle || mc.devices != gpu.devices || mc.pool_bytes != bytes) {
                if (mc.handle) rayz_hip_multi_destroy(mc.handle);
                mc.handle = nullptr;
                rc = rayz_hip_multi_create(gpu.devices.data(), (int)gpu.devices.size(), &sd, RAYZ_GATHER_RCCL, &mc.handle);
                if (rc != RAYZ_OK) throw GpuRenderFailed(rc, rayz_hip_last_error());
                mc.devices = gpu.devices;
                mc.pool_bytes.swap(bytes);
            }
            mh = mc.handle;
        }
        if (gpu.precision == RAYZ_PRECISION_F32) {
            std::vector<float> rgb(n * 3);
            rc = multi ? rayz_hip_multi_render(mh, &cd, &p, rgb.data(), &stats) : rayz_hip_render(&sd, &cd, &p, rgb.data(), &stats);
            if (rc == RAYZ_OK)
                for (size_t i = 0; i < n; ++i) img.pixels[i] = V3{rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]};
        } else {
            std::vector<double> rgb(n * 3);
            rc = multi ? rayz_hip_multi_render_f64(mh, &cd, &p, rgb.data(), &stats) : rayz_hip_render_f64(&sd, &cd, &p, rgb.data(), &stats);
            if (rc == RAYZ_OK)
                for (size_t i = 0; i < n; ++i) img.pixels[i] = V3{rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]};
        }
        if (rc != RAYZ_OK) throw GpuRenderFailed(rc, rayz_hip_last_error());
        return (size_t)stats.primary_rays;
    }
};

// ---- scenes ---------------------------------------------------------------------------------------
// `randomBouncing`, src/rayz.zig:45-168; the grid bounds (reference: a,b ∈ [-11,11)) are parameters so
// that the 10k-sphere benchmark scene (a,b ∈ [-50,50)) comes from the same generator.
inline Tracer randomBouncing(size_t img_w, int grid_lo = -11, int grid_hi = 11, const uint64_t* seed = nullptr) {
    Tracer tracer = Tracer::init(img_w, 20.0, 10.0, 0.6, V3{13, 2, 3}, V3{}, V3::y_hat(), seed);
    MemPool& pool = tracer.pool;
    { // ground: handles are created innermost-first (even, odd, checker, material), src/rayz.zig:58-74
        const TextureHandle even = pool.addAndReturnHandle(Texture::Solid(V3{0.2, 0.3, 0.1}));
        const TextureHandle odd = pool.addAndReturnHandle(Texture::Solid(V3::of(0.9)));
        const TextureHandle checker = pool.addAndReturnHandle(Texture::Checker(0.32, even, odd));
        pool.add(Sphere::stationary(V3{0, -1000, 0}, 1000, pool.addAndReturnHandle(Material::Diffuse(checker))));
    }
    pool.add(Sphere::stationary(V3{0, 1, 0}, 1.0, pool.addAndReturnHandle(Material::Dielectric(1.5))));
    pool.add(Sphere::stationary(
        V3{-4, 1, 0}, 1.0,
        pool.addAndReturnHandle(Material::Diffuse(pool.addAndReturnHandle(Texture::Solid(V3{0.4, 0.2, 0.1}))))));
    pool.add(Sphere::stationary(
        V3{4, 1, 0}, 1.0,
        pool.addAndReturnHandle(Material::Metallic(pool.addAndReturnHandle(Texture::Solid(V3{0.7, 0.6, 0.5}))))));
    DefaultPrng& rand = tracer.rng; // the renderer's own stream, src/rayz.zig:109
    for (int a = grid_lo; a < grid_hi; ++a) {
        for (int b = grid_lo; b < grid_hi; ++b) {
            const double rand_mat = rand.float64();
            const double fa = (double)a, fb = (double)b;
            V3 center;
            center.x = fa + 0.9 * rand.float64();
            center.y = 0.2;
            center.z = fb + 0.9 * rand.float64();
            if (center.sub(V3{4, 0.2, 0}).mag() <= 0.9) continue;
            Ray sphere_ray;
            sphere_ray.origin = center;
            MaterialHandle m;
            if (rand_mat < 0.8) {
                const V3 c1 = rand.v3(0, 1.0);
                const V3 c2 = rand.v3(0, 1.0);
                m = pool.addAndReturnHandle(Material::Diffuse(pool.addAndReturnHandle(Texture::Solid(c1.vmul(c2)))));
                sphere_ray.dir = V3::y_hat().mul(rand.float64() * 0.5);
            } else if (rand_mat < 0.95) {
                const double fuzz = rand.float64() * 0.5;
                const V3 col = rand.v3(0.5, 1.0);
                m = pool.addAndReturnHandle(Material::Metallic(pool.addAndReturnHandle(Texture::Solid(col)), fuzz));
            } else {
                m = pool.addAndReturnHandle(Material::Dielectric(1.5));
            }
            Sphere s;
            s.center = sphere_ray;
            s.radius = 0.2;
            s.material = m;
            pool.add(s);
        }
    }
    return tracer;
}

// BASELINE config 1: three stationary Lambertian spheres (positions from the dead `penultimateScene`,
// src/rayz.zig:182-203,225-237), vfov 20 from (-2,2,1) at (0,0,-1), focus 3.4, no defocus.
inline Tracer threeSpheres(size_t img_w, const uint64_t* seed = nullptr) {
    Tracer tracer = Tracer::init(img_w, 20.0, 3.4, 0.0, V3{-2, 2, 1}, V3{0, 0, -1}, V3::y_hat(), seed);
    MemPool& pool = tracer.pool;
    auto lambert = [&](V3 c) { return pool.addAndReturnHandle(Material::Diffuse(pool.addAndReturnHandle(Texture::Solid(c)))); };
    pool.add(Sphere::stationary(V3{0, -100.5, -1}, 100, lambert(V3{0.8, 0.8, 0.0})));
    pool.add(Sphere::stationary(V3{0, 0, -1.2}, 0.5, lambert(V3{0.1, 0.2, 0.5})));
    pool.add(Sphere::stationary(V3{1, 0, -1}, 0.5, lambert(V3{0.8, 0.6, 0.2})));
    return tracer;
}

// BASELINE config 5 (build-defined; no reference counterpart): an n x n-quad height field (2·n² triangles;
// n = 224 gives 100,352) with smooth bumps, Lambertian with a checker, three spheres resting above it.
// Heights come from an integer hash + smoothstep, so the mesh is identical on every host (no libm).
inline Tracer triangleMesh(size_t img_w, unsigned n = 224, const uint64_t* seed = nullptr) {
    Tracer tracer = Tracer::init(img_w, 35.0, 9.0, 0.0, V3{6.5, 4.0, 6.5}, V3{0, 0.3, 0}, V3::y_hat(), seed);
    MemPool& pool = tracer.pool;
    const TextureHandle a = pool.addAndReturnHandle(Texture::Solid(V3{0.75, 0.7, 0.6}));
    const TextureHandle b = pool.addAndReturnHandle(Texture::Solid(V3{0.35, 0.45, 0.3}));
    const MaterialHandle ground = pool.addAndReturnHandle(Material::Diffuse(pool.addAndReturnHandle(Texture::Checker(0.8, a, b))));
    auto hash = [](uint32_t x, uint32_t y) {
        uint32_t h = x * 0x9E3779B1u ^ (y + 0x7F4A7C15u) * 0x85EBCA6Bu;
        h ^= h >> 15, h *= 0x2C1B3C6Du, h ^= h >> 12, h *= 0x297A2D39u, h ^= h >> 15;
        return (double)(h >> 8) * (1.0 / 16777216.0);
    };
    const double extent = 5.0, cell = 2 * extent / n;
    const unsigned coarse = 14; // bump lattice
    auto height = [&](unsigned i, unsigned j) {
        const double fx = (double)i * coarse / n, fy = (double)j * coarse / n;
        const unsigned x0 = (unsigned)fx, y0 = (unsigned)fy;
        double tx = fx - x0, ty = fy - y0;
        tx = tx * tx * (3 - 2 * tx), ty = ty * ty * (3 - 2 * ty);
        const double h00 = hash(x0, y0), h10 = hash(x0 + 1, y0), h01 = hash(x0, y0 + 1), h11 = hash(x0 + 1, y0 + 1);
        return 0.45 * ((h00 * (1 - tx) + h10 * tx) * (1 - ty) + (h01 * (1 - tx) + h11 * tx) * ty);
    };
    auto vert = [&](unsigned i, unsigned j) { return V3{-extent + i * cell, height(i, j), -extent + j * cell}; };
    for (unsigned j = 0; j < n; ++j)
        for (unsigned i = 0; i < n; ++i) {
            const V3 p00 = vert(i, j), p10 = vert(i + 1, j), p01 = vert(i, j + 1), p11 = vert(i + 1, j + 1);
            pool.add(Triangle{p00, p01, p11, ground});
            pool.add(Triangle{p00, p11, p10, ground});
        }
    pool.add(Sphere::stationary(V3{0, 1.1, 0}, 0.7, pool.addAndReturnHandle(Material::Dielectric(1.5))));
    pool.add(Sphere::stationary(
        V3{-1.9, 0.95, 0.6}, 0.55,
        pool.addAndReturnHandle(Material::Metallic(pool.addAndReturnHandle(Texture::Solid(V3{0.8, 0.7, 0.5})), 0.05))));
    pool.add(Sphere::stationary(
        V3{1.7, 0.9, -0.8}, 0.5,
        pool.addAndReturnHandle(Material::Diffuse(pool.addAndReturnHandle(Texture::Solid(V3{0.7, 0.15, 0.1}))))));
    return tracer;
}

} // namespace rayz
