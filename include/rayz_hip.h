/*
 * rayz_hip.h — C ABI of the MI355X (gfx950) render path that replaces the loop
 * nest of rayz's `Tracer.render()`.
 *
 * Every entry point cites the reference interface it stands in for
 * (file:line into jlucier/rayz @ 2025-07-25).  The reference has no FFI of its
 * own: its boundary is the single Zig method `Tracer.render`
 * (src/renderer.zig:72-101).  A Zig caller copies its `MemPool` lists and its
 * `Camera` field by field into the `extern struct`-compatible PODs below
 * (Zig `struct`/`union(enum)` have no defined layout, so `items.ptr` cannot be
 * passed directly), calls `rayz_hip_render`, and widens the returned f32 RGB
 * into `img.pixels`.  See INTEGRATION.md for the Zig stub.
 *
 * Plain pointers and sizes only; no C++ types, exceptions or aborts cross this
 * boundary.  All functions return RAYZ_OK (0) or a negative RayzStatus and
 * leave a message retrievable through rayz_hip_last_error().
 */
#ifndef RAYZ_HIP_H
#define RAYZ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAYZ_HIP_ABI_VERSION 5u /* 2: RayzTriangle, shard fields; 3: RAYZ_TRAVERSAL_AUTO; 4: per-scene devices,
                                   rayz_hip_multi_* (several GPUs behind one call), rayz_hip_kat, chunk_spp auto;
                                   5: rayz_hip_debug_set (the library reads no environment variable),
                                   rayz_hip_multi_device_stats / _timing, RAYZ_GATHER_ALLOW_DUPLICATE_DEVICES */
#define RAYZ_DEFAULT_TILE_ROWS 8u /* what RayzRenderParams.tile_rows = 0 means (see there) */
#define RAYZ_MAX_DEVICES 64

typedef enum RayzStatus {
    RAYZ_OK = 0,
    RAYZ_ERR_BAD_ARG = -1,   /* null pointer, zero size, index out of range, bad enum */
    RAYZ_ERR_HIP = -2,       /* a HIP runtime call failed; text in rayz_hip_last_error() */
    RAYZ_ERR_OOM = -3,       /* host or device allocation failed */
    RAYZ_ERR_NO_DEVICE = -4, /* no usable gfx950 device / library not initialised */
    RAYZ_ERR_STATE = -5      /* bad handle, call order */
} RayzStatus;

/* Tag order follows `Texture = union(enum){ checker, solid }`, src/material.zig:41-44. */
typedef enum RayzTextureKind { RAYZ_TEX_CHECKER = 0, RAYZ_TEX_SOLID = 1 } RayzTextureKind;
/* `Material = union(enum){ diffuse, metallic, dielectric }`, src/material.zig:162-165. */
typedef enum RayzMaterialKind {
    RAYZ_MAT_DIFFUSE = 0,
    RAYZ_MAT_METALLIC = 1,
    RAYZ_MAT_DIELECTRIC = 2
} RayzMaterialKind;
/* `DiffuseScatterMethod`, src/material.zig:67-71 (default HEMISPHERE, :74). */
typedef enum RayzDiffuseMethod {
    RAYZ_DIFFUSE_UNIT_SPHERE = 0,
    RAYZ_DIFFUSE_UNIT_SPHERE_SURFACE = 1,
    RAYZ_DIFFUSE_HEMISPHERE = 2
} RayzDiffuseMethod;

typedef enum RayzPrecision {
    RAYZ_PRECISION_F32 = 0, /* the kernel arithmetic of DESIGN.md §4 (tmin 1e-3 recommended) */
    RAYZ_PRECISION_F64 = 1  /* fidelity mode: the reference's own scalar type, src/vec.zig:4-8 */
} RayzPrecision;

typedef enum RayzTraversal {
    RAYZ_TRAVERSAL_LINEAR = 0, /* flat hit list: every sphere tested per segment (north star) */
    RAYZ_TRAVERSAL_BVH = 1,    /* the reference's accelerator, src/hit.zig:101-217 */
    RAYZ_TRAVERSAL_AUTO = 2    /* flat list up to RAYZ_AUTO_BVH_MIN hittables, BVH above (same image either way) */
} RayzTraversal;
#define RAYZ_AUTO_BVH_MIN 160u /* measured crossover on MI355X: tools/crossover.py, profiles/r02/crossover.log */

/* One entry of `MemPool.textures` (src/ecs.zig:26): SolidTexture src/material.zig:19-25 or
 * CheckerTexture src/material.zig:27-39.  `even`/`odd` are TextureHandle.idx values. */
typedef struct RayzTexture {
    uint32_t kind; /* RayzTextureKind */
    uint32_t even; /* checker only */
    uint32_t odd;  /* checker only */
    uint32_t _pad;
    double scale;    /* checker only */
    double color[3]; /* solid only */
} RayzTexture;

/* One entry of `MemPool.materials` (src/ecs.zig:25): DiffuseMaterial src/material.zig:73-75,
 * MetallicMaterial :104-106, DielectricMaterial :134-135. */
typedef struct RayzMaterial {
    uint32_t kind;    /* RayzMaterialKind */
    uint32_t texture; /* TextureHandle.idx (diffuse, metallic) */
    uint32_t method;  /* RayzDiffuseMethod (diffuse) */
    uint32_t _pad;
    double param;     /* metallic: fuzz; dielectric: refractive_index */
} RayzMaterial;

/* One entry of `MemPool.spheres` (src/ecs.zig:24): `Sphere{center: Ray, radius, material}`,
 * src/geom.zig:11-14.  `center` = center.origin, `velocity` = center.dir (center.time unused). */
typedef struct RayzSphere {
    double center[3];
    double velocity[3];
    double radius;
    uint32_t material; /* MaterialHandle.idx */
    uint32_t _pad;
} RayzSphere;

/* A triangle hittable.  BUILD-DEFINED: the reference's geom.zig holds only `Sphere` (src/geom.zig:11-67);
 * BASELINE.json's config 5 asks for a triangle path, so this primitive is fitted to the `Hittable` / `Hit`
 * contract (src/hit.zig:8-42): two-sided, nearest root in [tmin, tmax], geometric normal flipped to face the
 * ray by `Hit.init`, stationary.  Its results are parity-unpinned (no reference code or test exists). */
typedef struct RayzTriangle {
    double v0[3];
    double v1[3];
    double v2[3];
    uint32_t material; /* MaterialHandle.idx */
    uint32_t _pad;
} RayzTriangle;

/* The `MemPool` lists (src/ecs.zig:22-27) plus the build-defined triangle list, borrowed for the duration of
 * the call.  Hittables are numbered spheres first, then triangles (the order `initHittables` would append
 * them, src/ecs.zig:43-51). */
typedef struct RayzSceneDesc {
    const RayzSphere* spheres;
    const RayzMaterial* materials;
    const RayzTexture* textures;
    uint32_t n_spheres;
    uint32_t n_materials;
    uint32_t n_textures;
    uint32_t n_triangles;
    const RayzTriangle* triangles;
} RayzSceneDesc;

/* The fields of `Camera` AFTER `Camera.init` (src/camera.zig:9-16): results, not look-at params. */
typedef struct RayzCameraDesc {
    double look_from[3];
    double px_du[3];
    double px_dv[3];
    double px_origin[3];
    double defocus_u[3];
    double defocus_v[3];
    uint32_t defocus; /* bool */
    uint32_t _pad;
} RayzCameraDesc;

/* `Tracer` fields that steer render() (src/renderer.zig:18-28) plus what the reference lacks:
 * a settable seed (it seeds from getrandom, :55-59), an explicit tmin (hard-coded 1e-10 at :107),
 * the precision/traversal selectors and the row-tile shard for multi-GPU. */
typedef struct RayzRenderParams {
    uint32_t width;  /* img.w, src/image.zig:6 */
    uint32_t height; /* img.h, src/image.zig:5 */
    uint32_t samples_per_px; /* src/renderer.zig:24 */
    uint32_t max_bounces;    /* src/renderer.zig:23 */
    uint64_t seed;           /* key of the per-(pixel,sample) PCG32 streams */
    double tmin;             /* src/renderer.zig:107 passes 1e-10 */
    uint32_t precision;      /* RayzPrecision */
    uint32_t traversal;      /* RayzTraversal */
    uint32_t chunk_spp;      /* samples summed per work item; 0 = the automatic schedule (rayz_hip_chunk_schedule);
                                part of the image's definition (fixes the f32 summation tree) */
    uint32_t tile_rows;      /* rows per shard tile; 0 = RAYZ_DEFAULT_TILE_ROWS (8) in EVERY entry point, single- and multi-device
                                alike.  Why 8: the BVH kernel deals its work as 8x8 pixel tiles of a shard's LOCAL rows, which are 8
                                consecutive image rows only with 8-row shard tiles; measured on an 8-way deal (every shard timed on one
                                GPU, profiles/r04/multi/tile_rows_ab.log): +8.6 % at 1920x1080 (rows 128..136 per rank) and +1.7 % at
                                3840x2160 against 1-row interleave, whose perfect row balance (135 each) does not make up for tiles
                                that span 57 image rows.  Irrelevant when shard_count <= 1 */
    uint32_t shard_index;    /* this call renders rows with (row / tile_rows) % shard_count == shard_index.  (Dealing the tiles in alternating
                                direction per band of shard_count — so that no shard's rows lie systematically lower in the frame, where paths
                                are longer — was measured in round 4 and NOT adopted: 7.09x instead of 6.91x for the flat list on 8 shards, but
                                6.99x instead of 7.17x through the BVH and worse at 2 and 4 shards, where it pairs adjacent tiles:
                                profiles/r04/multi/predicted_scaling_*.log) */
    uint32_t shard_count;    /* 0 or 1 = whole image */
} RayzRenderParams;

/* What `render()` returns (primary rays, src/renderer.zig:90,100) plus the counts the roofline needs. */
typedef struct RayzRenderStats {
    uint64_t primary_rays; /* rows_in_shard * width * samples_per_px */
    uint64_t segments;     /* findHit calls: one per ray segment, src/renderer.zig:107 */
    uint64_t sphere_tests; /* primitive tests: Sphere.hitInner evaluations (src/geom.zig:38-66) + triangle tests.  BVH: counted by
                              the kernel (leaf entries examined).  Flat list: DERIVED, segments x hittables — every segment scans
                              the whole list, so the kernel counts segments only */
    uint64_t node_tests;   /* AABB.hit evaluations (BVH traversal only), src/hit.zig:70-98 */
    double kernel_ms;      /* HIP-event time of the trace kernel(s) of the last render on this scene */
} RayzRenderStats;

typedef struct RayzScene RayzScene; /* opaque: device-resident scene + workspace */

/* Library / device lifetime.  rayz_hip_init(device) creates the context of that HIP ordinal (its stream, CU count;
 * gfx950 only) and makes it the DEFAULT device: the one the entry points without a device argument use.  Idempotent;
 * calling it for a second ordinal adds a context and moves the default, it does not re-target existing scenes —
 * a scene stays on the device it was bound to.  Every entry point selects its device itself and restores the
 * calling thread's current HIP device on return.  rayz_hip_shutdown destroys all contexts (scenes must be
 * destroyed first; a scene destroyed later still frees its memory). */
int rayz_hip_init(int device);
void rayz_hip_shutdown(void);
const char* rayz_hip_last_error(void);
uint32_t rayz_hip_abi_version(void);

/* Measurement knobs.  They change how the work is SCHEDULED or which (equivalent) tree the GPU walks — never an image —
 * and exist for the sweep tools under tools/ and for the tests that hold the kernels against each other.  Process-wide;
 * value < 0 restores the built-in default; takes effect for renders (BVH_PEEL / BVH_TOP: scenes) started afterwards.
 * The library reads NO environment variable. */
typedef enum RayzDebugKnob {
    RAYZ_DEBUG_QUEUE_GRAB = 0, /* work items a wave reserves per atomic on the queue head (default 64) */
    RAYZ_DEBUG_BVH_KEEP = 1,   /* one-path BVH kernel: keep_active | keep_stepping << 8 */
    RAYZ_DEBUG_BVH_PEEL = 2,   /* 0: walk the reference's full tree (oversized hittables stay in it) */
    RAYZ_DEBUG_BVH_TOP = 3,    /* cap on the inner-node records of the tree's top kept in LDS (default: what fits beside the stacks) */
    RAYZ_DEBUG_BVH_KERNEL = 4, /* f32 BVH renders: 1 = one path per lane (trace_kernel_bvh, default; the only one in the product library);
                                  -DRAYZ_EXPERIMENTS builds: 2 = two paths per lane (trace_kernel_bvh2), 3 = walker / shader waves (trace_kernel_bvhx) */
    RAYZ_DEBUG_BVH2_KEEP = 5,  /* two-path BVH kernel: service | blocked << 8 | swap << 16 | keep_stepping << 24 */
    RAYZ_DEBUG_LDS_PAD = 6,    /* BVH kernels: unused bytes added to the workgroup's LDS request (occupancy experiments) */
    RAYZ_DEBUG_BVH_TOP_ORDER = 7, /* which inner nodes the LDS top holds: 0 = by box surface area from the root (default), 1 = breadth-first */
    RAYZ_DEBUG_BVH_NODES = 8,     /* node record format of trees built from now on: 0 = by tree size (default), 1 = f32 planes (64 B), 2 = 16-bit plane indices (32 B) */
    RAYZ_DEBUG_BVH_SPLIT = 9,     /* how trees built from now on split a node: 0 = surface-area heuristic (default), 1 = the reference's median split */
    RAYZ_DEBUG_BVHX = 10,         /* exchange kernel (RAYZ_DEBUG_BVH_KERNEL = 3, -DRAYZ_EXPERIMENTS builds only): slots per walker wave | exchange threshold << 8 |
                                     shader's minimum batch << 16 | its patience << 24 | its priority << 32 */
    RAYZ_DEBUG_CHUNK_CAP = 11,    /* -DRAYZ_EXPERIMENTS builds only, refused otherwise — it CHANGES the image's summation tree (the one knob that
                                     does; tools/chunk_cap_sweep.py): largest chunk of the automatic schedule */
    RAYZ_DEBUG_KNOBS = 12
} RayzDebugKnob;
int rayz_hip_debug_set(uint32_t knob, long long value);

/* Number of rows the shard described by `p` owns (= rows of the compact output). */
uint32_t rayz_hip_shard_rows(const RayzRenderParams* p);

/* The chunk schedule the render of `p` will use — which consecutive samples of a pixel are summed by one work item;
 * the chunk sums are then added in chunk order (it fixes the f32 summation tree, so it is part of the
 * image's definition; it depends on width, height, samples_per_px and chunk_spp only, never on the shard fields).
 * Returns the number of chunks n; if `starts` is not NULL, fills starts[0..min(n, capacity-1)] with the first sample
 * of each chunk and, last, samples_per_px.  chunk_spp = 0 selects the automatic schedule (uniform 16 for small
 * renders; 256, 256, .., 128, 64, 32, 16, 16 for large ones). */
uint32_t rayz_hip_chunk_schedule(const RayzRenderParams* p, uint32_t* starts, uint32_t capacity);

/* Replaces src/renderer.zig:76-78 (initHittables + BVH build) and what follows it: validates the
 * handles (RAYZ_ERR_BAD_ARG for an index out of range, a checker chain that contains a cycle or nests deeper than
 * 8 lookups — the reference recurses without a limit, src/material.zig:36-37, the device walks a bounded loop),
 * lays the pool out in HBM and (for BVH traversal) builds the reference's BVH on the host.
 * rayz_hip_scene_create binds the scene to the default device at its first render; _create_on binds it to
 * `device` now (creating that device's context if needed).  One render in flight per scene. */
int rayz_hip_scene_create(const RayzSceneDesc* scene, RayzScene** out);
int rayz_hip_scene_create_on(int device, const RayzSceneDesc* scene, RayzScene** out);
int rayz_hip_scene_destroy(RayzScene* scene);

/* The BVH `render()` would build (src/renderer.zig:76-78 -> src/hit.zig:130-161), flattened in depth-first
 * pre-order with skip links as the GPU traverses it.  Host only (no device needed).  Call once with every array
 * NULL to get *n_nodes, then with arrays of n_nodes (boxes: 6 doubles per node, lo then hi; skip/first/count:
 * one u32 per node; order: n_spheres + n_triangles hittable indices in leaf order).  count == 0 marks an inner node. */
int rayz_hip_scene_bvh(RayzScene* scene, uint32_t* n_nodes, uint32_t* depth, double* boxes, uint32_t* skip,
                       uint32_t* first, uint32_t* count, uint32_t* order);

/* Replaces the loop nest src/renderer.zig:80-97.  Asynchronous on `hip_stream` (a hipStream_t, or
 * NULL for the library's own stream); `d_rgb_out` is DEVICE memory, rows_in_shard*width*3 floats,
 * row-major packed RGB, linear radiance means as `img.pixels` holds them (src/renderer.zig:94-95;
 * no gamma or clamp, those live in writePPM, src/image.zig:29-41). */
int rayz_hip_render_device(RayzScene* scene, const RayzCameraDesc* camera, const RayzRenderParams* params,
                           float* d_rgb_out, void* hip_stream);

/* Same, f64 output (precision F64): rows_in_shard*width*3 doubles. */
int rayz_hip_render_device_f64(RayzScene* scene, const RayzCameraDesc* camera, const RayzRenderParams* params,
                               double* d_rgb_out, void* hip_stream);

/* Waits for the last render on `scene` and returns its counters. */
int rayz_hip_scene_sync(RayzScene* scene, RayzRenderStats* stats_or_null);

/* Blocking one-shot form of `tracer.render()` as main() calls it (src/rayz.zig:26): upload, render,
 * download into caller-owned host memory (`rgb_out`: rows_in_shard*width*3 floats). */
int rayz_hip_render(const RayzSceneDesc* scene, const RayzCameraDesc* camera, const RayzRenderParams* params,
                    float* rgb_out, RayzRenderStats* stats_or_null);
int rayz_hip_render_f64(const RayzSceneDesc* scene, const RayzCameraDesc* camera, const RayzRenderParams* params,
                        double* rgb_out, RayzRenderStats* stats_or_null);

/* ---- several GPUs behind ONE call -------------------------------------------------------------------------
 * The reference's caller makes one call, `tracer.render()` (src/rayz.zig:26, src/renderer.zig:72-101).  These
 * entry points give that one call every GPU of the node: the pool is replicated (one scene per device), image
 * rows are dealt to the devices in interleaved tiles of `params->tile_rows` rows (0 = RAYZ_DEFAULT_TILE_ROWS = 8, as everywhere),
 * each device traces its rows on its own stream, and ONE collective — an RCCL gather of the row tiles to
 * devices[0] over xGMI (ncclCommInitAll + ncclGather), or peer copies — reassembles the frame, which is copied
 * to the caller's HOST buffer (height*width*3, row-major RGB).  The image is bit-identical for any device count
 * (the per-(pixel,sample) streams are keyed by global pixel coordinates).  `params->shard_index/shard_count`
 * must be 0: the library shards.  Blocking; driven by the calling thread; one call at a time per handle.
 * A device may be listed once (RAYZ_ERR_BAD_ARG otherwise); OR-ing RAYZ_GATHER_ALLOW_DUPLICATE_DEVICES into a
 * PEER_COPY transport lifts that for TESTS on a one-GPU box (refused with RCCL, which cannot take a device twice).
 * STATUS: with n > 1 DISTINCT devices this path has not yet run on hardware (the development pool has one GPU per
 * box): n = 1 through RCCL and n = 2/3/8 on one device through peer copies are what the GPU suite exercises. */
typedef enum RayzGatherTransport {
    RAYZ_GATHER_RCCL = 0,      /* ncclGather to devices[0] (librccl.so.1 is opened at the first multi-device call) */
    RAYZ_GATHER_PEER_COPY = 1, /* hipMemcpyPeerAsync into devices[0] */
    RAYZ_GATHER_ALLOW_DUPLICATE_DEVICES = 0x100 /* flag bit, peer-copy only: the same ordinal may be listed repeatedly */
} RayzGatherTransport;

typedef struct RayzMulti RayzMulti; /* opaque: one scene per device + communicators + gather buffers */

int rayz_hip_multi_create(const int* devices, int n_devices, const RayzSceneDesc* scene, uint32_t transport,
                          RayzMulti** out);
int rayz_hip_multi_destroy(RayzMulti* multi);
/* n_devices, the transport in use and RCCL's version code (0 with peer copies); any pointer may be NULL */
int rayz_hip_multi_info(const RayzMulti* multi, int* n_devices, uint32_t* transport, int* rccl_version);
/* After a render on the handle: device `index`'s own counters (its rows, its segments, ITS trace-kernel time — load
 * imbalance between row shards shows here), and the frame's timing: gather_ms = on the root's stream, from "the
 * root's rows are traced" to "the frame is assembled" (the transfer + the wait for slower devices + the
 * un-interleave); frame_ms = host wall time of gather + copy-out.  Either pointer of _timing may be NULL. */
int rayz_hip_multi_device_stats(const RayzMulti* multi, int index, RayzRenderStats* stats);
int rayz_hip_multi_timing(const RayzMulti* multi, double* gather_ms, double* frame_ms);
/* `tracer.render()` on all devices of the handle.  stats: counts summed over the devices, kernel_ms = slowest. */
int rayz_hip_multi_render(RayzMulti* multi, const RayzCameraDesc* camera, const RayzRenderParams* params,
                          float* rgb_out, RayzRenderStats* stats_or_null);
int rayz_hip_multi_render_f64(RayzMulti* multi, const RayzCameraDesc* camera, const RayzRenderParams* params,
                              double* rgb_out, RayzRenderStats* stats_or_null);
/* Same frame, but each device applies `Image.writePPM`'s per-pixel transform (src/image.zig:35-38) to its rows
 * BEFORE the gather, so the tiles travel as u8 (4x smaller) and `rgb8_out` (height*width*3 bytes) is what
 * writePPM would print.  f32 precision only. */
int rayz_hip_multi_render_u8(RayzMulti* multi, const RayzCameraDesc* camera, const RayzRenderParams* params,
                             uint8_t* rgb8_out, RayzRenderStats* stats_or_null);
/* One-shot forms: create, render, destroy (RCCL transport).  They pay communicator creation (ncclCommInitAll, tens
 * of milliseconds per device) on EVERY call: a caller that renders more than one frame keeps a RayzMulti. */
int rayz_hip_render_multi(const int* devices, int n_devices, const RayzSceneDesc* scene,
                          const RayzCameraDesc* camera, const RayzRenderParams* params, float* rgb_out,
                          RayzRenderStats* stats_or_null);
int rayz_hip_render_multi_f64(const int* devices, int n_devices, const RayzSceneDesc* scene,
                              const RayzCameraDesc* camera, const RayzRenderParams* params, double* rgb_out,
                              RayzRenderStats* stats_or_null);

/* The step after the path, `Image.writePPM`'s per-pixel transform (src/image.zig:35-38,
 * src/vec.zig:79-93): sqrt-gamma, clamp to [0,1], truncate x*255 to u8.  Device to device,
 * n_pixels*3 floats in, n_pixels*3 bytes out. */
int rayz_hip_tonemap_u8(const float* d_rgb, uint8_t* d_rgb8, size_t n_pixels, void* hip_stream); /* default device */

/* ---- known answers ---------------------------------------------------------------------------------------------
 * Evaluates the trace kernels' OWN device functions (the same inlined code the kernels run) on caller inputs, one
 * GPU thread per record, so that the reference's test vectors and a CPU restatement can be held against the
 * device code directly.  Covered: everything a path is assembled from, the BVH kernels' leaf reject test
 * (leaf_reject_test, shared with the walk) and the flat list's packed-FMA form of it (ScanGroup::discs, two spheres
 * per v_pk_fma_f32: RAYZ_KAT_SCAN_DISCS; in the scan loop its sphere operands come from scalar registers, here from
 * vector ones — the same arithmetic) included.  Vectors: src/material.zig:213-223 (refract), src/renderer.zig:129-149 (get ray),
 * src/hit.zig:247-279 (bbox hit); tests/test_kat_gpu.py.  Host buffers: `in` = n records of RAYZ_KAT_IN_STRIDE
 * doubles, `out` = n records of RAYZ_KAT_OUT_STRIDE doubles (unused slots 0).  Values are narrowed to `precision`
 * as a scene is when it crosses the ABI.  Random draws, where an op makes any, come from the record's list u[]
 * (0.5 beyond its end) in the order the path would make them.  Default device. */
typedef enum RayzKatOp {
    RAYZ_KAT_REFRACT = 0,     /* in: unit_dir[0..2] normal[3..5] eta[6]            out: dir[0..2]       src/material.zig:189-194 */
    RAYZ_KAT_REFLECTANCE = 1, /* in: cos[0] ri[1]                                  out: r[0]            src/material.zig:179-183 */
    RAYZ_KAT_GET_RAY = 2,     /* in: look_from px_du px_dv px_origin defocus_u defocus_v [0..17] defocus[18] px[19] py[20]
                                     n_u[21] (an integer in [0, 26], or -1; RAYZ_ERR_BAD_ARG otherwise) u[22..];
                                     n_u = -1 is getRay(px, py, null) — no jitter, lens centre, time 0: the call of the reference's own
                                     "get ray" test, src/renderer.zig:129-149; n_u = 0 draws 0.5 everywhere (the kernels always draw)
                                 out: origin[0..2] dir[3..5] time[6] draws[7]                           src/camera.zig:59-90 */
    RAYZ_KAT_BOX_HIT = 3,     /* in: low[0..2] high[3..5] origin[6..8] dir[9..11] tmin[12] tmax[13] format[26] (the node record
                                     the box is tested in: 0 = 16-bit plane indices, non-zero = f32 planes)
                                 out: hit[0] t_entry[1]                                                 src/hit.zig:70-98 */
    RAYZ_KAT_SPHERE_HIT = 4,  /* in: center[0..2] velocity[3..5] radius[6] origin[7..9] dir[10..12] time[13] tmin[14] tmax[15]
                                 out: hit[0] t[1] point[2..4] normal[5..7] front_face[8] passed_filter[9]
                                                                                        src/geom.zig:38-66, src/hit.zig:25-41 */
    RAYZ_KAT_SCATTER = 5,     /* in: kind[0] method[1] param[2] ray origin[3..5] dir[6..8] hit point[9..11] normal[12..14]
                                     front_face[15] n_u[16] (an integer in [0, 31]) u[17..]
                                 out: scattered[0] dir[1..3] draws[4]                                   src/material.zig:73-160 */
    RAYZ_KAT_CHECKER = 6,     /* in: point[0..2] scale[3]                          out: parity[0]       src/material.zig:32-36 */
    RAYZ_KAT_BACKGROUND = 7,  /* in: dir[0..2]                                     out: colour[0..2]    src/renderer.zig:124-125 */
    RAYZ_KAT_TRIANGLE_HIT = 8,/* in: v0[0..2] v1[3..5] v2[6..8] origin[9..11] dir[12..14] tmin[15] tmax[16]
                                 out: hit[0] t[1] passed_filter[2]                 build-defined */
    RAYZ_KAT_SCAN_DISCS = 9   /* one 4-sphere block of the flat list's scan streams, as the SCAN LOOP evaluates it (packed FMAs,
                                 two spheres per instruction; the values are f32 for both precisions):
                                 in: cx[0..3] cy[4..7] cz[8..11] radius[12..15] (padded and squared by the library as the scene
                                     upload does) vy[16..19] origin[20..22] dir[23..25] time[26] class[27] (0 static, 1 y-moving)
                                 out: r2 - p1^2 - p2^2 per sphere [0..3] (>= 0: candidate), the same value from the
                                      general-velocity form the BVH leaves use [4..7]     src/geom.zig:40-50, DESIGN.md 4.3 */
} RayzKatOp;
#define RAYZ_KAT_IN_STRIDE 48
#define RAYZ_KAT_OUT_STRIDE 12
int rayz_hip_kat(uint32_t op, uint32_t precision, const double* in, uint32_t n_records, double* out);

#ifdef __cplusplus
}
#endif
#endif /* RAYZ_HIP_H */
