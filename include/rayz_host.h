/*
 * rayz_host.h — C view of the host-side mirror of rayz's Tracer / MemPool / Camera / Image API
 * (rayz_amd/host/rayz.hpp), for callers that are neither Zig nor C++ (the Python tests and bench.py).
 *
 * Each function names the reference construct it mirrors (file:line into jlucier/rayz).  The render
 * itself always goes through include/rayz_hip.h; nothing here traces rays on the CPU.
 */
#ifndef RAYZ_HOST_H
#define RAYZ_HOST_H

#include "rayz_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct RayzTracer RayzTracer;

/* fields for rayz_tracer_set_u64 / rayz_tracer_set_f64 */
typedef enum RayzTracerField {
    RAYZ_FIELD_MAX_BOUNCES = 0,    /* Tracer.max_bounces, src/renderer.zig:23 (default 50) */
    RAYZ_FIELD_SAMPLES_PER_PX = 1, /* Tracer.samples_per_px, src/renderer.zig:24 (default 10) */
    RAYZ_FIELD_PRECISION = 2,      /* RayzPrecision */
    RAYZ_FIELD_TRAVERSAL = 3,      /* RayzTraversal */
    RAYZ_FIELD_CHUNK_SPP = 4,
    RAYZ_FIELD_RENDER_SEED = 5,    /* fixes the kernel seed; otherwise render() draws it from Tracer.rng */
    RAYZ_FIELD_TMIN = 6            /* f64; < 0 restores the default (1e-3 for f32, 1e-10 for f64) */
} RayzTracerField;

typedef struct RayzTracerInfo {
    uint32_t width, height;
    uint32_t samples_per_px, max_bounces;
    uint32_t n_spheres, n_materials, n_textures, n_triangles;
} RayzTracerInfo;

/* Tracer.init, src/renderer.zig:29-64.  has_seed == 0 seeds the Tracer's DefaultPrng from the OS as the
 * reference does (:55-59). */
int rayz_tracer_create(uint32_t img_w, double vfov, double focus_dist, double defocus_angle, const double* look_from,
                       const double* look_at, const double* vup, int has_seed, uint64_t seed, RayzTracer** out);
void rayz_tracer_destroy(RayzTracer* t);

/* MemPool.addAndReturnHandle, src/ecs.zig:57-69: return the new handle's idx, or a negative RayzStatus. */
int64_t rayz_tracer_add_texture_solid(RayzTracer* t, const double* color);
int64_t rayz_tracer_add_texture_checker(RayzTracer* t, double scale, uint32_t even, uint32_t odd);
int64_t rayz_tracer_add_material_diffuse(RayzTracer* t, uint32_t texture, uint32_t method);
int64_t rayz_tracer_add_material_metallic(RayzTracer* t, uint32_t texture, double fuzz);
int64_t rayz_tracer_add_material_dielectric(RayzTracer* t, double refractive_index);
int64_t rayz_tracer_add_sphere(RayzTracer* t, const double* center, const double* velocity, double radius,
                               uint32_t material);

/* build-defined triangle hittable (see RayzTriangle) */
int64_t rayz_tracer_add_triangle(RayzTracer* t, const double* v0, const double* v1, const double* v2, uint32_t material);

int rayz_tracer_set_u64(RayzTracer* t, int field, uint64_t value);
int rayz_tracer_set_f64(RayzTracer* t, int field, double value);
/* devices render() drives (rayz_hip_render_multi: rows dealt to them, one RCCL gather); n = 0: the default device */
int rayz_tracer_set_devices(RayzTracer* t, const int* devices, int n);
int rayz_tracer_info(const RayzTracer* t, RayzTracerInfo* out);

/* Tracer.camera after Camera.init, src/camera.zig:18-57 */
int rayz_tracer_camera(const RayzTracer* t, RayzCameraDesc* out);
/* Camera.getRay(px, py, null), src/camera.zig:59-77 */
int rayz_tracer_get_ray(const RayzTracer* t, uint32_t px, uint32_t py, double* origin, double* dir);

/* The flattened pool exactly as render() hands it to rayz_hip_render.  Pointers stay valid until the
 * tracer is mutated or destroyed. */
int rayz_tracer_scene(RayzTracer* t, RayzSceneDesc* out);
/* The params render() would pass (seed = RENDER_SEED if set, else 0 as a placeholder). */
int rayz_tracer_params(const RayzTracer* t, RayzRenderParams* out);

/* Tracer.rng (std.Random.DefaultPrng), src/renderer.zig:22: state peek and draws, for tests. */
int rayz_tracer_rng_state(const RayzTracer* t, uint64_t* state4);
uint64_t rayz_tracer_rng_next(RayzTracer* t);
double rayz_tracer_rng_float(RayzTracer* t);

/* Tracer.render, src/renderer.zig:72-101: primary-ray count, or a negative RayzStatus. */
int64_t rayz_tracer_render(RayzTracer* t);
int rayz_tracer_stats(const RayzTracer* t, RayzRenderStats* out);
/* Tracer.img.pixels, src/image.zig:7: h*w*3 doubles, row-major */
const double* rayz_tracer_pixels(const RayzTracer* t);
/* Image.writePPM, src/image.zig:29-41 */
int rayz_tracer_write_ppm(const RayzTracer* t, const char* path);
int rayz_image_write_ppm(const double* rgb, uint32_t w, uint32_t h, const char* path);
void rayz_image_to_u8(const double* rgb, size_t n_pixels, uint8_t* out);

/* Scenes: randomBouncing (src/rayz.zig:45-168) with its grid bounds as parameters; BASELINE config 1. */
int rayz_scene_random_bouncing(uint32_t img_w, int grid_lo, int grid_hi, int has_seed, uint64_t seed,
                               RayzTracer** out);
int rayz_scene_three_spheres(uint32_t img_w, int has_seed, uint64_t seed, RayzTracer** out);
/* BASELINE config 5 (build-defined): n x n-quad height field = 2 n^2 triangles + three spheres. */
int rayz_scene_triangle_mesh(uint32_t img_w, uint32_t n, int has_seed, uint64_t seed, RayzTracer** out);

#ifdef __cplusplus
}
#endif
#endif /* RAYZ_HOST_H */
