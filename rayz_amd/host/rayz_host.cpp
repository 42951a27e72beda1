// rayz_host.cpp — include/rayz_host.h on top of the C++ mirror in rayz.hpp.
#include "../../include/rayz_host.h"
#include "rayz.hpp"

#include <new>

using namespace rayz;

struct RayzTracer {
    Tracer t;
    Tracer::Flat flat;
    std::vector<double> pixels; // img.pixels as packed doubles (valid after render)
    bool flat_valid = false;
};

namespace {
V3 v3(const double* p) { return {p[0], p[1], p[2]}; }
int wrap(Tracer&& t, RayzTracer** out) {
    RayzTracer* h = new (std::nothrow) RayzTracer();
    if (!h) return RAYZ_ERR_OOM;
    h->t = std::move(t);
    *out = h;
    return RAYZ_OK;
}
} // namespace

extern "C" {

int rayz_tracer_create(uint32_t img_w, double vfov, double focus_dist, double defocus_angle, const double* look_from,
                       const double* look_at, const double* vup, int has_seed, uint64_t seed, RayzTracer** out) {
    if (!out || !look_from || !look_at || !vup || img_w == 0) return RAYZ_ERR_BAD_ARG;
    *out = nullptr;
    try {
        return wrap(Tracer::init(img_w, vfov, focus_dist, defocus_angle, v3(look_from), v3(look_at), v3(vup),
                                 has_seed ? &seed : nullptr),
                    out);
    } catch (...) {
        return RAYZ_ERR_OOM;
    }
}

void rayz_tracer_destroy(RayzTracer* t) { delete t; }

int64_t rayz_tracer_add_texture_solid(RayzTracer* t, const double* color) {
    if (!t || !color) return RAYZ_ERR_BAD_ARG;
    t->flat_valid = false;
    return (int64_t)t->t.pool.addAndReturnHandle(Texture::Solid(v3(color))).idx;
}
int64_t rayz_tracer_add_texture_checker(RayzTracer* t, double scale, uint32_t even, uint32_t odd) {
    if (!t) return RAYZ_ERR_BAD_ARG;
    t->flat_valid = false;
    return (int64_t)t->t.pool.addAndReturnHandle(Texture::Checker(scale, {even}, {odd})).idx;
}
int64_t rayz_tracer_add_material_diffuse(RayzTracer* t, uint32_t texture, uint32_t method) {
    if (!t || method > RAYZ_DIFFUSE_HEMISPHERE) return RAYZ_ERR_BAD_ARG;
    t->flat_valid = false;
    return (int64_t)t->t.pool.addAndReturnHandle(Material::Diffuse({texture}, (RayzDiffuseMethod)method)).idx;
}
int64_t rayz_tracer_add_material_metallic(RayzTracer* t, uint32_t texture, double fuzz) {
    if (!t) return RAYZ_ERR_BAD_ARG;
    t->flat_valid = false;
    return (int64_t)t->t.pool.addAndReturnHandle(Material::Metallic({texture}, fuzz)).idx;
}
int64_t rayz_tracer_add_material_dielectric(RayzTracer* t, double ri) {
    if (!t) return RAYZ_ERR_BAD_ARG;
    t->flat_valid = false;
    return (int64_t)t->t.pool.addAndReturnHandle(Material::Dielectric(ri)).idx;
}
int64_t rayz_tracer_add_sphere(RayzTracer* t, const double* center, const double* velocity, double radius,
                               uint32_t material) {
    if (!t || !center) return RAYZ_ERR_BAD_ARG;
    t->flat_valid = false;
    Sphere s = Sphere::stationary(v3(center), radius, {material});
    if (velocity) s.center.dir = v3(velocity);
    return (int64_t)t->t.pool.addAndReturnHandle(s);
}

int64_t rayz_tracer_add_triangle(RayzTracer* t, const double* v0, const double* v1, const double* v2, uint32_t material) {
    if (!t || !v0 || !v1 || !v2) return RAYZ_ERR_BAD_ARG;
    t->flat_valid = false;
    return (int64_t)t->t.pool.addAndReturnHandle(Triangle{v3(v0), v3(v1), v3(v2), {material}});
}

int rayz_tracer_set_u64(RayzTracer* t, int field, uint64_t v) {
    if (!t) return RAYZ_ERR_BAD_ARG;
    switch (field) {
    case RAYZ_FIELD_MAX_BOUNCES: t->t.max_bounces = (size_t)v; break;
    case RAYZ_FIELD_SAMPLES_PER_PX: t->t.samples_per_px = (size_t)v; break;
    case RAYZ_FIELD_PRECISION:
        if (v > RAYZ_PRECISION_F64) return RAYZ_ERR_BAD_ARG;
        t->t.gpu.precision = (RayzPrecision)v;
        break;
    case RAYZ_FIELD_TRAVERSAL:
        if (v > RAYZ_TRAVERSAL_AUTO) return RAYZ_ERR_BAD_ARG;
        t->t.gpu.traversal = (RayzTraversal)v;
        break;
    case RAYZ_FIELD_CHUNK_SPP: t->t.gpu.chunk_spp = (uint32_t)v; break;
    case RAYZ_FIELD_RENDER_SEED:
        t->t.gpu.has_render_seed = true;
        t->t.gpu.render_seed = v;
        break;
    default: return RAYZ_ERR_BAD_ARG;
    }
    return RAYZ_OK;
}
int rayz_tracer_set_f64(RayzTracer* t, int field, double v) {
    if (!t || field != RAYZ_FIELD_TMIN) return RAYZ_ERR_BAD_ARG;
    t->t.gpu.tmin = v;
    return RAYZ_OK;
}
int rayz_tracer_set_devices(RayzTracer* t, const int* devices, int n) {
    if (!t || n < 0 || (n && !devices)) return RAYZ_ERR_BAD_ARG;
    try {
        t->t.gpu.devices.assign(devices, devices + n);
    } catch (...) {
        return RAYZ_ERR_OOM;
    }
    return RAYZ_OK;
}
int rayz_tracer_info(const RayzTracer* t, RayzTracerInfo* o) {
    if (!t || !o) return RAYZ_ERR_BAD_ARG;
    o->width = (uint32_t)t->t.img.w, o->height = (uint32_t)t->t.img.h;
    o->samples_per_px = (uint32_t)t->t.samples_per_px, o->max_bounces = (uint32_t)t->t.max_bounces;
    o->n_spheres = (uint32_t)t->t.pool.spheres.size(), o->n_materials = (uint32_t)t->t.pool.materials.size();
    o->n_textures = (uint32_t)t->t.pool.textures.size(), o->n_triangles = (uint32_t)t->t.pool.triangles.size();
    return RAYZ_OK;
}
int rayz_tracer_camera(const RayzTracer* t, RayzCameraDesc* out) {
    if (!t || !out) return RAYZ_ERR_BAD_ARG;
    *out = t->t.camera.desc();
    return RAYZ_OK;
}
int rayz_tracer_get_ray(const RayzTracer* t, uint32_t px, uint32_t py, double* origin, double* dir) {
    if (!t || !origin || !dir) return RAYZ_ERR_BAD_ARG;
    const Ray r = t->t.camera.getRay(px, py);
    origin[0] = r.origin.x, origin[1] = r.origin.y, origin[2] = r.origin.z;
    dir[0] = r.dir.x, dir[1] = r.dir.y, dir[2] = r.dir.z;
    return RAYZ_OK;
}
int rayz_tracer_scene(RayzTracer* t, RayzSceneDesc* out) {
    if (!t || !out) return RAYZ_ERR_BAD_ARG;
    try {
        if (!t->flat_valid) {
            t->flat = t->t.flatten();
            t->flat_valid = true;
        }
    } catch (...) {
        return RAYZ_ERR_OOM;
    }
    *out = t->flat.desc();
    return RAYZ_OK;
}
int rayz_tracer_params(const RayzTracer* t, RayzRenderParams* out) {
    if (!t || !out) return RAYZ_ERR_BAD_ARG;
    *out = t->t.params(t->t.gpu.has_render_seed ? t->t.gpu.render_seed : 0);
    return RAYZ_OK;
}
int rayz_tracer_rng_state(const RayzTracer* t, uint64_t* s) {
    if (!t || !s) return RAYZ_ERR_BAD_ARG;
    std::memcpy(s, t->t.rng.s, 32);
    return RAYZ_OK;
}
uint64_t rayz_tracer_rng_next(RayzTracer* t) { return t ? t->t.rng.next() : 0; }
double rayz_tracer_rng_float(RayzTracer* t) { return t ? t->t.rng.float64() : 0; }

int64_t rayz_tracer_render(RayzTracer* t) {
    if (!t) return RAYZ_ERR_BAD_ARG;
    try {
        const size_t rays = t->t.render();
        const size_t n = t->t.img.pixels.size();
        t->pixels.resize(n * 3);
        for (size_t i = 0; i < n; ++i) {
            t->pixels[3 * i] = t->t.img.pixels[i].x;
            t->pixels[3 * i + 1] = t->t.img.pixels[i].y;
            t->pixels[3 * i + 2] = t->t.img.pixels[i].z;
        }
        return (int64_t)rays;
    } catch (const GpuRenderFailed& e) {
        return e.status;
    } catch (...) {
        return RAYZ_ERR_OOM;
    }
}
int rayz_tracer_stats(const RayzTracer* t, RayzRenderStats* out) {
    if (!t || !out) return RAYZ_ERR_BAD_ARG;
    *out = t->t.stats;
    return RAYZ_OK;
}
const double* rayz_tracer_pixels(const RayzTracer* t) { return (t && !t->pixels.empty()) ? t->pixels.data() : nullptr; }

int rayz_tracer_write_ppm(const RayzTracer* t, const char* path) {
    if (!t || !path) return RAYZ_ERR_BAD_ARG;
    FILE* f = std::fopen(path, "w");
    if (!f) return RAYZ_ERR_BAD_ARG;
    t->t.img.writePPM(f);
    std::fclose(f);
    return RAYZ_OK;
}
int rayz_image_write_ppm(const double* rgb, uint32_t w, uint32_t h, const char* path) {
    if (!rgb || !path) return RAYZ_ERR_BAD_ARG;
    try {
        Image im = Image::initEmpty(h, w);
        for (size_t i = 0; i < (size_t)w * h; ++i) im.pixels[i] = V3{rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]};
        FILE* f = std::fopen(path, "w");
        if (!f) return RAYZ_ERR_BAD_ARG;
        im.writePPM(f);
        std::fclose(f);
    } catch (...) {
        return RAYZ_ERR_OOM;
    }
    return RAYZ_OK;
}
void rayz_image_to_u8(const double* rgb, size_t n_pixels, uint8_t* out) {
    for (size_t i = 0; i < n_pixels; ++i) Image::toU8(V3{rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]}, out + 3 * i);
}

int rayz_scene_random_bouncing(uint32_t img_w, int lo, int hi, int has_seed, uint64_t seed, RayzTracer** out) {
    if (!out || img_w == 0 || hi < lo) return RAYZ_ERR_BAD_ARG;
    *out = nullptr;
    try {
        return wrap(randomBouncing(img_w, lo, hi, has_seed ? &seed : nullptr), out);
    } catch (...) {
        return RAYZ_ERR_OOM;
    }
}
int rayz_scene_triangle_mesh(uint32_t img_w, uint32_t n, int has_seed, uint64_t seed, RayzTracer** out) {
    if (!out || img_w == 0 || n == 0 || n > 4096) return RAYZ_ERR_BAD_ARG;
    *out = nullptr;
    try {
        return wrap(triangleMesh(img_w, n, has_seed ? &seed : nullptr), out);
    } catch (...) {
        return RAYZ_ERR_OOM;
    }
}
int rayz_scene_three_spheres(uint32_t img_w, int has_seed, uint64_t seed, RayzTracer** out) {
    if (!out || img_w == 0) return RAYZ_ERR_BAD_ARG;
    *out = nullptr;
    try {
        return wrap(threeSpheres(img_w, has_seed ? &seed : nullptr), out);
    } catch (...) {
        return RAYZ_ERR_OOM;
    }
}

} // extern "C"
