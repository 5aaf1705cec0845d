//! hip_backend.zig -- Zig host side of the MI355X trace path (drop into src/ of Copper280z/raytrace).
//!
//! Binds include/rt_hip.h (librt_hip.so) and offers replacements with the reference's own signatures:
//!   render_and_write_hip  for rt.render_and_write   (raytracer.zig:580-633; call sites main.zig:317, raytracer.zig:742)
//!   start_trace_hip       for main.zig's start_trace (main.zig:307-318), progressive: the window keeps drawing
//! Shipped as source: this image has no Zig toolchain, so the file is NOT compiled here; the same ABI is exercised
//! from Python (ctypes) and C++ by the test suite.  build.zig: exe.addLibraryPath(...);
//! exe.linkSystemLibrary("rt_hip"); exe.linkLibC();
//! It touches the reference only through declarations that ARE `pub` there -- rt.Hittable (raytracer.zig:712),
//! rt.cam (:710), img_io.ImgType / Image_s / write_ppm_image (image.zig:6,12,18) -- so the reference's files stay
//! unmodified: `Sphere` (:38) and `shape_s` (:26) are private to raytracer.zig, hence `objects: anytype` and the
//! local `Shape` below (an anonymous `.{ .nx = .., .ny = .. }` literal, as both reference call sites pass, coerces
//! to it).  tests/test_zig_shim.py checks every `rt.` / `img_io.` name used here against that list.
const std = @import("std");
const rt = @import("raytracer.zig");
const img_io = @import("image.zig");

// ---- C ABI (include/rt_hip.h) ------------------------------------------------------------------------
pub const rt_ctx = opaque {};
pub const RT_OK: c_int = 0;
pub const RT_PROGRESS_MORE: c_int = 1;
pub const RT_SPP_CHUNK_AUTO: u32 = 0xFFFFFFFF;

pub const rt_camera = extern struct {
    lower_left: [3]f32,
    horizontal: [3]f32,
    vertical: [3]f32,
    origin: [3]f32,
    subtract_origin: u32, // 1 = MovableCamera.get_ray (camera.zig:39), 0 = BasicCamera (camera.zig:16)
};
pub const rt_spheres = extern struct {
    n: u32,
    cx: [*]const f32,
    cy: [*]const f32,
    cz: [*]const f32,
    radius: [*]const f32,
    kind: [*]const u32, // 0 lambertian, 1 metal, 2 dielectric (raytracer.zig:402-406)
    r: [*]const f32,
    g: [*]const f32,
    b: [*]const f32,
    param: [*]const f32, // roughness | index
};
pub const rt_triangles = extern struct {
    n: u32,
    v0: [*]const f32, // [n][3]
    v1: [*]const f32,
    v2: [*]const f32,
    material: [*]const u32, // index into the material table (an rt_spheres whose geometry is ignored)
};
pub const rt_config = extern struct {
    nx: u32,
    ny: u32,
    spp_begin: u32,
    spp_end: u32,
    spp_total: u32,
    max_depth: u32, // reference: 16 (raytracer.zig:556)
    t_min: f32, // reference: 0.001 (raytracer.zig:555)
    seed: u64,
    tile_x0: u32 = 0,
    tile_y0: u32 = 0,
    tile_x1: u32 = 0,
    tile_y1: u32 = 0,
    row_offset_q3: u32 = 1, // j = ny - jr as raytracer.zig:523
    spp_chunk: u32 = RT_SPP_CHUNK_AUTO,
};
pub const rt_stats = extern struct { rays: u64, primary_rays: u64, kernel_ms: f64, rays_by_depth: [20]u64 };

extern fn rt_create(device: c_int, out: *?*rt_ctx) c_int;
extern fn rt_destroy(ctx: ?*rt_ctx) void;
extern fn rt_device_count() c_int;
extern fn rt_strerror(code: c_int) [*:0]const u8;
extern fn rt_upload_scene(ctx: ?*rt_ctx, spheres: ?*const rt_spheres, triangles: ?*const rt_triangles, materials: ?*const rt_spheres) c_int;
extern fn rt_upload_timing(ctx: ?*const rt_ctx, host_build_ms: ?*f64, device_copy_ms: ?*f64, device_build_ms: ?*f64) c_int;
extern fn rt_render(ctx: ?*rt_ctx, cam: *const rt_camera, cfg: *const rt_config, accum_mean: ?[*]f32, rgb8: ?[*]u8, raycount: ?[*]u32, stats: ?*rt_stats) c_int;
extern fn rt_render_multi(ctxs: [*]const ?*rt_ctx, n_ctx: u32, cam: *const rt_camera, cfg: *const rt_config, accum_mean: ?[*]f32, rgb8: ?[*]u8, raycount: ?[*]u32, stats: ?*rt_stats) c_int;
extern fn rt_render_reference_style(ctx: ?*rt_ctx, cam: *const rt_camera, cfg: *const rt_config, n_images: u32, rgb8: [*]u8, stats: ?*rt_stats) c_int;
extern fn rt_progressive_begin(ctx: ?*rt_ctx, cam: *const rt_camera, cfg: *const rt_config, n_passes: u32) c_int;
extern fn rt_progressive_step(ctx: ?*rt_ctx, rgb8: ?[*]u8, accum_mean: ?[*]f32, spp_done: ?*u32, stats: ?*rt_stats) c_int;
extern fn rt_tessellate_spheres(spheres: *const rt_spheres, slices: c_int, stacks: c_int, cap: u32, v0: ?[*]f32, v1: ?[*]f32, v2: ?[*]f32, material: ?[*]u32) c_int;

pub const HipError = error{HipBackend};

/// raytracer.zig:26-29 `shape_s` (private there): same fields, same order.
pub const Shape = struct { ny: usize, nx: usize };

fn check(code: c_int) HipError!void {
    if (code < 0) {
        std.log.err("rt_hip: {s}", .{rt_strerror(code)});
        return HipError.HipBackend;
    }
}

// ---- repacking the reference's types ---------------------------------------------------------------------
/// `[]const Sphere` is not extern and has no stable layout (raytracer.zig:38-41): repack into SoA.
const SphereSoA = struct {
    cx: []f32,
    cy: []f32,
    cz: []f32,
    rad: []f32,
    kind: []u32,
    r: []f32,
    g: []f32,
    b: []f32,
    par: []f32,

    fn init(a: std.mem.Allocator, objects: anytype) !SphereSoA { // []const Sphere; the element type is private to raytracer.zig
        const n = objects.len;
        var s = SphereSoA{
            .cx = try a.alloc(f32, n), .cy = try a.alloc(f32, n), .cz = try a.alloc(f32, n), .rad = try a.alloc(f32, n),
            .kind = try a.alloc(u32, n), .r = try a.alloc(f32, n), .g = try a.alloc(f32, n), .b = try a.alloc(f32, n),
            .par = try a.alloc(f32, n),
        };
        for (objects, 0..) |o, i| {
            s.cx[i] = o.center[0];
            s.cy[i] = o.center[1];
            s.cz[i] = o.center[2];
            s.rad[i] = o.radius; // may be negative: hollow glass (raytracer.zig:677)
            switch (o.material) {
                .lambertian => |m| { s.kind[i] = 0; s.r[i] = m.albedo[0]; s.g[i] = m.albedo[1]; s.b[i] = m.albedo[2]; s.par[i] = 0; },
                .metal => |m| { s.kind[i] = 1; s.r[i] = m.albedo[0]; s.g[i] = m.albedo[1]; s.b[i] = m.albedo[2]; s.par[i] = m.roughness; },
                .dielectric => |m| { s.kind[i] = 2; s.r[i] = m.color[0]; s.g[i] = m.color[1]; s.b[i] = m.color[2]; s.par[i] = m.index; },
            }
        }
        return s;
    }
    fn view(s: *const SphereSoA) rt_spheres {
        return .{ .n = @intCast(s.cx.len), .cx = s.cx.ptr, .cy = s.cy.ptr, .cz = s.cz.ptr, .radius = s.rad.ptr,
                  .kind = s.kind.ptr, .r = s.r.ptr, .g = s.g.ptr, .b = s.b.ptr, .param = s.par.ptr };
    }
};

/// The implicit global camera (`pub var cam`, raytracer.zig:710) crosses the ABI by value.
fn cameraOf(c: anytype) rt_camera {
    return .{ .lower_left = c.lower_left, .horizontal = c.horizontal, .vertical = c.vertical, .origin = c.origin, .subtract_origin = 1 };
}

fn configOf(shape: Shape, num_samples: u32) rt_config {
    return .{ .nx = @intCast(shape.nx), .ny = @intCast(shape.ny), .spp_begin = 0, .spp_end = num_samples,
              .spp_total = num_samples, .max_depth = 16, .t_min = 0.001, .seed = 1 };
}

fn writeImage(a: std.mem.Allocator, fname: []const u8, img: []const u8, shape: Shape, filetype: img_io.ImgType) !void {
    if (filetype != .ppm) return;
    const n = shape.nx * shape.ny;
    var list = try std.ArrayList(@Vector(3, u8)).initCapacity(a, n);
    for (0..n) |k| list.appendAssumeCapacity(.{ img[3 * k], img[3 * k + 1], img[3 * k + 2] });
    try img_io.write_ppm_image(fname, .{ .arr = list, .nx = shape.nx, .ny = shape.ny });
}

// ---- render_and_write (raytracer.zig:580-633) ---------------------------------------------------------------
/// Same signature as rt.render_and_write.  num_samples is the TOTAL per pixel (the reference renders num_samples
/// on each of getCpuCount() threads).  Uses every GPU of the node: the reference's `for (0..n_cpu) Thread.spawn`
/// (raytracer.zig:594-601) with GPUs for threads, joined by one RCCL reduce inside rt_render_multi.
pub fn render_and_write_hip(fname: []const u8, num_samples: u16, world: rt.Hittable, shape: Shape,
                            filetype: img_io.ImgType, allocator: std.mem.Allocator) !void {
    var arena = std.heap.ArenaAllocator.init(allocator);
    defer arena.deinit();
    const a = arena.allocator();
    const soa = try SphereSoA.init(a, world.objects);
    const spheres = soa.view();
    const cam = cameraOf(rt.cam);
    const cfg = configOf(shape, num_samples);

    const n_gpu: usize = @intCast(@max(rt_device_count(), 1));
    var ctxs = try a.alloc(?*rt_ctx, n_gpu);
    @memset(ctxs, null);
    defer for (ctxs) |c| rt_destroy(c);
    for (0..n_gpu) |g| {
        try check(rt_create(@intCast(g), &ctxs[g]));
        try check(rt_upload_scene(ctxs[g], &spheres, null, null));
    }
    const img = try a.alloc(u8, shape.nx * shape.ny * 3);
    var stats: rt_stats = undefined;
    try check(rt_render_multi(ctxs.ptr, @intCast(n_gpu), &cam, &cfg, null, img.ptr, null, &stats));
    std.debug.print("HIP: {d} rays in {d:.1} ms on {d} GPU(s)\n", .{ stats.rays, stats.kernel_ms, n_gpu });
    try writeImage(a, fname, img, shape, filetype);
}

/// The reference's own output rule (quirk Q4: the integer mean of n_images gamma'd u8 images keyed i*123,
/// raytracer.zig:600,610-617) for an A/B against a CPU render; n_images plays getCpuCount().
pub fn render_and_write_reference_style(fname: []const u8, num_samples: u16, n_images: u32, world: rt.Hittable,
                                        shape: Shape, filetype: img_io.ImgType, allocator: std.mem.Allocator) !void {
    var arena = std.heap.ArenaAllocator.init(allocator);
    defer arena.deinit();
    const a = arena.allocator();
    const soa = try SphereSoA.init(a, world.objects);
    const spheres = soa.view();
    const cam = cameraOf(rt.cam);
    const cfg = configOf(shape, num_samples);
    var ctx: ?*rt_ctx = null;
    try check(rt_create(0, &ctx));
    defer rt_destroy(ctx);
    try check(rt_upload_scene(ctx, &spheres, null, null));
    const img = try a.alloc(u8, shape.nx * shape.ny * 3);
    try check(rt_render_reference_style(ctx, &cam, &cfg, n_images, img.ptr, null));
    try writeImage(a, fname, img, shape, filetype);
}

/// The preview's triangle meshes (zmesh parametric spheres, main.zig:128-138) traced instead of rasterised.
pub fn render_mesh_and_write_hip(fname: []const u8, num_samples: u16, world: rt.Hittable, shape: Shape,
                                 filetype: img_io.ImgType, allocator: std.mem.Allocator) !void {
    var arena = std.heap.ArenaAllocator.init(allocator);
    defer arena.deinit();
    const a = arena.allocator();
    const soa = try SphereSoA.init(a, world.objects);
    const spheres = soa.view();
    const n_tri_c = rt_tessellate_spheres(&spheres, 0, 0, 0, null, null, null, null); // query the count
    try check(n_tri_c);
    const n_tri: usize = @intCast(n_tri_c);
    const v0 = try a.alloc(f32, 3 * n_tri);
    const v1 = try a.alloc(f32, 3 * n_tri);
    const v2 = try a.alloc(f32, 3 * n_tri);
    const mat = try a.alloc(u32, n_tri);
    try check(rt_tessellate_spheres(&spheres, 0, 0, @intCast(n_tri), v0.ptr, v1.ptr, v2.ptr, mat.ptr));
    const tris = rt_triangles{ .n = @intCast(n_tri), .v0 = v0.ptr, .v1 = v1.ptr, .v2 = v2.ptr, .material = mat.ptr };
    var ctx: ?*rt_ctx = null;
    try check(rt_create(0, &ctx));
    defer rt_destroy(ctx);
    try check(rt_upload_scene(ctx, null, &tris, &spheres));
    const cam = cameraOf(rt.cam);
    const cfg = configOf(shape, num_samples);
    const img = try a.alloc(u8, shape.nx * shape.ny * 3);
    try check(rt_render(ctx, &cam, &cfg, null, img.ptr, null, null));
    try writeImage(a, fname, img, shape, filetype);
}

// ---- the GUI's Trace button (main.zig:307-318, 342-345) -------------------------------------------------------
/// Progressive trace for the GUI.  `TraceJob.begin` replaces the body of start_trace: it sets the camera exactly
/// as main.zig:308-315 does (quirk Q9 kept: `lookat` receives the forward vector and v_up = pos + (0,1,0)) and
/// starts the render; the main loop then calls `poll` once per frame (where `compute.onCompute` was meant to go,
/// main.zig:620-626) and shows `job.rgb8` until poll returns false; `finish` writes Scene.ppm.
pub const TraceJob = struct {
    ctx: ?*rt_ctx = null,
    rgb8: []u8 = &.{},
    shape: Shape = undefined,
    spp_done: u32 = 0,
    active: bool = false,

    pub fn begin(job: *TraceJob, allocator: std.mem.Allocator, world: rt.Hittable, cam_pos: anytype, cam_forward: anytype,
                 num_samples: u16, n_passes: u32) !void {
        const v_up = cam_pos + @as(@TypeOf(cam_pos), .{ 0, 1, 0 }); // main.zig:310
        job.shape = .{ .nx = 1200, .ny = 675 }; // main.zig:312-313
        rt.cam.SetPosition(cam_pos, cam_forward, v_up, 45, 1200.0 / 675.0); // main.zig:315
        var arena = std.heap.ArenaAllocator.init(allocator);
        defer arena.deinit();
        const soa = try SphereSoA.init(arena.allocator(), world.objects);
        const spheres = soa.view();
        if (job.ctx == null) try check(rt_create(0, &job.ctx));
        try check(rt_upload_scene(job.ctx, &spheres, null, null)); // the library copies: the arena may go
        job.rgb8 = try allocator.alloc(u8, job.shape.nx * job.shape.ny * 3);
        const cam = cameraOf(rt.cam);
        const cfg = configOf(job.shape, num_samples);
        try check(rt_progressive_begin(job.ctx, &cam, &cfg, n_passes));
        job.active = true;
    }

    /// One pass; returns true while more passes remain.  job.rgb8 holds the picture so far.
    pub fn poll(job: *TraceJob) !bool {
        if (!job.active) return false;
        const rc = rt_progressive_step(job.ctx, job.rgb8.ptr, null, &job.spp_done, null);
        try check(rc);
        job.active = rc == RT_PROGRESS_MORE;
        return job.active;
    }

    pub fn finish(job: *TraceJob, allocator: std.mem.Allocator, fname: []const u8) !void {
        var arena = std.heap.ArenaAllocator.init(allocator);
        defer arena.deinit();
        try writeImage(arena.allocator(), fname, job.rgb8, job.shape, .ppm); // main.zig:317 "Scene.ppm"
        allocator.free(job.rgb8);
        job.rgb8 = &.{};
    }

    pub fn deinit(job: *TraceJob) void {
        rt_destroy(job.ctx);
        job.ctx = null;
    }
};
