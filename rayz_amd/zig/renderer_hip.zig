//! renderer_hip.zig — Zig-side drop-in for the loop nest of `Tracer.render()` (src/renderer.zig:72-101).
//!
//! SOURCE ONLY: there is no Zig toolchain in the build image, so this file has never been compiled.
//! The C++ mirror (rayz_amd/host/rayz.hpp `Tracer::render`) performs exactly these steps and is what the
//! test-suite exercises.  Written against the Zig 0.13/0.14 std the reference uses.  What CAN be checked without a
//! compiler is checked: tests/test_zig_binding_text.py parses every `extern struct` and `extern "c" fn` below and holds
//! field names, order and widths, and every prototype's parameter list, to include/rayz_hip.h.
//!
//! Usage in the reference tree: copy next to src/renderer.zig, link librayz_hip.so
//! (`exe.addLibraryPath(...)`, `exe.linkSystemLibrary("rayz_hip")`, `exe.linkLibC()` in build.zig), and
//! replace the body of `Tracer.render` by `return renderHip(self, .{});`.

const std = @import("std");
const vec = @import("./vec.zig");
const mat = @import("./material.zig");
const renderer = @import("./renderer.zig");

// ---- include/rayz_hip.h, field for field (extern struct = C layout) ----
pub const RayzTexture = extern struct {
    kind: u32, // 0 checker, 1 solid: tag order of `Texture = union(enum)`, src/material.zig:41-44
    even: u32,
    odd: u32,
    _pad: u32 = 0,
    scale: f64,
    color: [3]f64,
};
pub const RayzMaterial = extern struct {
    kind: u32, // 0 diffuse, 1 metallic, 2 dielectric: src/material.zig:162-165
    texture: u32,
    method: u32, // DiffuseScatterMethod, src/material.zig:67-71
    _pad: u32 = 0,
    param: f64, // fuzz | refractive_index
};
pub const RayzSphere = extern struct {
    center: [3]f64,
    velocity: [3]f64,
    radius: f64,
    material: u32,
    _pad: u32 = 0,
};
pub const RayzTriangle = extern struct { // build-defined hittable; the reference has none (leave the list empty)
    v0: [3]f64,
    v1: [3]f64,
    v2: [3]f64,
    material: u32,
    _pad: u32 = 0,
};
pub const RayzSceneDesc = extern struct {
    spheres: ?[*]const RayzSphere,
    materials: ?[*]const RayzMaterial,
    textures: ?[*]const RayzTexture,
    n_spheres: u32,
    n_materials: u32,
    n_textures: u32,
    n_triangles: u32 = 0,
    triangles: ?[*]const RayzTriangle = null,
};
pub const RayzCameraDesc = extern struct {
    look_from: [3]f64,
    px_du: [3]f64,
    px_dv: [3]f64,
    px_origin: [3]f64,
    defocus_u: [3]f64,
    defocus_v: [3]f64,
    defocus: u32,
    _pad: u32 = 0,
};
pub const RayzRenderParams = extern struct {
    width: u32,
    height: u32,
    samples_per_px: u32,
    max_bounces: u32,
    seed: u64,
    tmin: f64,
    precision: u32 = 0, // 0 f32 (tmin 1e-3), 1 f64 (the reference's 1e-10)
    traversal: u32 = 2, // 0 flat list, 1 BVH, 2 auto (flat list up to 160 hittables, BVH above)
    chunk_spp: u32 = 0,
    tile_rows: u32 = 0,
    shard_index: u32 = 0,
    shard_count: u32 = 0,
};
pub const RayzRenderStats = extern struct {
    primary_rays: u64,
    segments: u64,
    sphere_tests: u64,
    node_tests: u64,
    kernel_ms: f64,
};

extern "c" fn rayz_hip_init(device: c_int) c_int;
extern "c" fn rayz_hip_last_error() [*:0]const u8;
extern "c" fn rayz_hip_render(
    scene: *const RayzSceneDesc,
    camera: *const RayzCameraDesc,
    params: *const RayzRenderParams,
    rgb_out: [*]f32,
    stats: ?*RayzRenderStats,
) c_int;
/// The reference's own arithmetic (f64 path state, roots, hit records and shading: src/vec.zig:4-8): params.precision = 1.
extern "c" fn rayz_hip_render_f64(
    scene: *const RayzSceneDesc,
    camera: *const RayzCameraDesc,
    params: *const RayzRenderParams,
    rgb_out: [*]f64,
    stats: ?*RayzRenderStats,
) c_int;

/// All GPUs of the node behind the same single call: rows are dealt to `devices` in interleaved tiles, each device
/// traces its rows, one RCCL gather over xGMI reassembles the frame into `rgb_out` (host, h*w*3).
extern "c" fn rayz_hip_render_multi(
    devices: [*]const c_int,
    n_devices: c_int,
    scene: *const RayzSceneDesc,
    camera: *const RayzCameraDesc,
    params: *const RayzRenderParams,
    rgb_out: [*]f32,
    stats: ?*RayzRenderStats,
) c_int;
extern "c" fn rayz_hip_render_multi_f64(
    devices: [*]const c_int,
    n_devices: c_int,
    scene: *const RayzSceneDesc,
    camera: *const RayzCameraDesc,
    params: *const RayzRenderParams,
    rgb_out: [*]f64,
    stats: ?*RayzRenderStats,
) c_int;

// A caller that renders more than one frame on several GPUs keeps ONE handle (per-device scenes + the RCCL
// communicators, `ncclCommInitAll`: tens of milliseconds per device) instead of paying the one-shot form's
// create + destroy per frame: rayz_hip_multi_create once, rayz_hip_multi_render per frame, rayz_hip_multi_destroy at
// the end (HipOptions.multi).  NOTE: with more than one DISTINCT device this path has not yet run on hardware.
pub const RayzMulti = opaque {};
extern "c" fn rayz_hip_multi_create(
    devices: [*]const c_int,
    n_devices: c_int,
    scene: *const RayzSceneDesc,
    transport: u32, // 0 = RCCL gather, 1 = peer copies
    out: *?*RayzMulti,
) c_int;
extern "c" fn rayz_hip_multi_render(
    multi: *RayzMulti,
    camera: *const RayzCameraDesc,
    params: *const RayzRenderParams,
    rgb_out: [*]f32,
    stats: ?*RayzRenderStats,
) c_int;
extern "c" fn rayz_hip_multi_render_f64(
    multi: *RayzMulti,
    camera: *const RayzCameraDesc,
    params: *const RayzRenderParams,
    rgb_out: [*]f64,
    stats: ?*RayzRenderStats,
) c_int;
extern "c" fn rayz_hip_multi_destroy(multi: ?*RayzMulti) c_int;

fn v3(v: vec.V3) [3]f64 {
    return .{ v.x, v.y, v.z };
}

pub const Precision = enum(u32) { f32 = 0, f64 = 1 }; // RayzPrecision
pub const Traversal = enum(u32) { flat_list = 0, bvh = 1, auto = 2 }; // RayzTraversal

pub const HipOptions = struct {
    seed: ?u64 = null, // null: next u64 of the Tracer's own DefaultPrng
    // .f32: f32 path state and reject tests, f64 candidate roots, tmin 1e-3 (the fast default).  .f64: the reference's own scalar
    // type for path state, roots, hit records and shading (src/vec.zig:4-8) with its tmin of 1e-10 (src/renderer.zig:107);
    // the frame is written straight into img.pixels (f64, src/image.zig:4-17) without narrowing.
    precision: Precision = .f32,
    traversal: Traversal = .auto, // the reference always walks its BVH (src/renderer.zig:76-78, src/hit.zig:181-216)
    tmin: ?f64 = null, // null: 1e-3 for .f32, 1e-10 for .f64
    devices: []const c_int = &.{}, // empty: device 0; e.g. &.{ 0, 1, 2, 3, 4, 5, 6, 7 } for a whole MI355X node
    // in/out: a handle kept across frames for `devices` (null: created by the first renderHip call that has devices;
    // the caller releases it with rayz_hip_multi_destroy).  The pool must not change while it is kept.
    multi: ?*?*RayzMulti = null,
};

/// The body of `Tracer.render()`: flatten → one extern call → the frame into img.pixels (f32 widened, or f64 as it is).
pub fn renderHip(self: *renderer.Tracer, opt: HipOptions) !usize {
    const a = self.allocator;
    // Zig structs / unions have no defined layout: copy field by field (SURVEY.md §8b "Layout caveat").
    const spheres = try a.alloc(RayzSphere, self.pool.spheres.items.len);
    defer a.free(spheres);
    for (self.pool.spheres.items, spheres) |s, *o| o.* = .{
        .center = v3(s.center.origin),
        .velocity = v3(s.center.dir),
        .radius = s.radius,
        .material = @intCast(s.material.idx),
    };
    const materials = try a.alloc(RayzMaterial, self.pool.materials.items.len);
    defer a.free(materials);
    for (self.pool.materials.items, materials) |m, *o| o.* = switch (m) {
        .diffuse => |d| .{ .kind = 0, .texture = @intCast(d.texture.idx), .method = @intFromEnum(d.method), .param = 0 },
        .metallic => |d| .{ .kind = 1, .texture = @intCast(d.texture.idx), .method = 2, .param = d.fuzz },
        .dielectric => |d| .{ .kind = 2, .texture = 0, .method = 2, .param = d.refractive_index },
    };
    const textures = try a.alloc(RayzTexture, self.pool.textures.items.len);
    defer a.free(textures);
    for (self.pool.textures.items, textures) |t, *o| o.* = switch (t) {
        .checker => |c| .{ .kind = 0, .even = @intCast(c.even.idx), .odd = @intCast(c.odd.idx), .scale = c.scale, .color = .{ 0, 0, 0 } },
        .solid => |s| .{ .kind = 1, .even = 0, .odd = 0, .scale = 0, .color = v3(s.color) },
    };
    const scene = RayzSceneDesc{
        .spheres = spheres.ptr,
        .materials = materials.ptr,
        .textures = textures.ptr,
        .n_spheres = @intCast(spheres.len),
        .n_materials = @intCast(materials.len),
        .n_textures = @intCast(textures.len),
    };
    const cam = RayzCameraDesc{
        .look_from = v3(self.camera.look_from),
        .px_du = v3(self.camera.px_du),
        .px_dv = v3(self.camera.px_dv),
        .px_origin = v3(self.camera.px_origin),
        .defocus_u = v3(self.camera.defocus_u),
        .defocus_v = v3(self.camera.defocus_v),
        .defocus = @intFromBool(self.camera.defocus),
    };
    const params = RayzRenderParams{
        .width = @intCast(self.img.w),
        .height = @intCast(self.img.h),
        .samples_per_px = @intCast(self.samples_per_px),
        .max_bounces = @intCast(self.max_bounces),
        .seed = opt.seed orelse self.rng.random().int(u64),
        .tmin = opt.tmin orelse (if (opt.precision == .f64) @as(f64, 1e-10) else @as(f64, 1e-3)),
        .precision = @intFromEnum(opt.precision),
        .traversal = @intFromEnum(opt.traversal),
    };
    var stats: RayzRenderStats = undefined;
    const n_px = self.img.w * self.img.h;
    if (opt.precision == .f64) {
        // the reference's arithmetic: an f64 frame, copied field by field into img.pixels (V3 is a plain Zig struct: no
        // defined layout, so the library cannot write into it directly)
        const rgb = try a.alloc(f64, n_px * 3);
        defer a.free(rgb);
        const rc = if (opt.devices.len == 0)
            (if (rayz_hip_init(0) != 0) @as(c_int, -4) else rayz_hip_render_f64(&scene, &cam, &params, rgb.ptr, &stats))
        else if (opt.multi) |slot| blk: {
            if (slot.* == null) {
                const crc = rayz_hip_multi_create(opt.devices.ptr, @intCast(opt.devices.len), &scene, 0, slot);
                if (crc != 0) break :blk crc;
            }
            break :blk rayz_hip_multi_render_f64(slot.*.?, &cam, &params, rgb.ptr, &stats);
        } else rayz_hip_render_multi_f64(opt.devices.ptr, @intCast(opt.devices.len), &scene, &cam, &params, rgb.ptr, &stats);
        if (rc != 0) {
            std.debug.print("rayz_hip: {s}\n", .{rayz_hip_last_error()});
            return error.GpuRenderFailed;
        }
        for (self.img.pixels, 0..) |*px, i| px.* = .{ .x = rgb[3 * i], .y = rgb[3 * i + 1], .z = rgb[3 * i + 2] };
        return @intCast(stats.primary_rays);
    }
    const rgb = try a.alloc(f32, n_px * 3);
    defer a.free(rgb);
    const rc = if (opt.devices.len == 0)
        (if (rayz_hip_init(0) != 0) @as(c_int, -4) else rayz_hip_render(&scene, &cam, &params, rgb.ptr, &stats))
    else if (opt.multi) |slot| blk: {
        if (slot.* == null) {
            const crc = rayz_hip_multi_create(opt.devices.ptr, @intCast(opt.devices.len), &scene, 0, slot);
            if (crc != 0) break :blk crc;
        }
        break :blk rayz_hip_multi_render(slot.*.?, &cam, &params, rgb.ptr, &stats);
    } else rayz_hip_render_multi(opt.devices.ptr, @intCast(opt.devices.len), &scene, &cam, &params, rgb.ptr, &stats);
    if (rc != 0) {
        std.debug.print("rayz_hip: {s}\n", .{rayz_hip_last_error()});
        return error.GpuRenderFailed;
    }
    for (self.img.pixels, 0..) |*px, i| px.* = .{ .x = rgb[3 * i], .y = rgb[3 * i + 1], .z = rgb[3 * i + 2] }; // widen f32 -> f64
    return @intCast(stats.primary_rays); // what render() returns, src/renderer.zig:90,100
}
