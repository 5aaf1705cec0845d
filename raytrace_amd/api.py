"""Python host mirror of the reference's camera / scene / render API over the C ABI.

Names follow the reference: MovableCamera.SetPosition (camera.zig:25), Random_Scene
(raytracer.zig:644), render_and_write (raytracer.zig:580), write_ppm_image (image.zig:18).
All rendering goes through librt_hip.so; there is no CPU path in this package.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import RtError, rt_camera, rt_config, rt_spheres, rt_stats, rt_tuning


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, ty):
    return a.ctypes.data_as(ty)


class Spheres:
    """SoA view of `world.objects` (raytracer.zig:38-41 Sphere + :402-449 Material)."""

    FIELDS = ("cx", "cy", "cz", "radius", "kind", "r", "g", "b", "param")

    def __init__(self, cx, cy, cz, radius, kind, r, g, b, param):
        self.cx, self.cy, self.cz, self.radius = _f32(cx), _f32(cy), _f32(cz), _f32(radius)
        self.kind = np.ascontiguousarray(kind, dtype=np.uint32)
        self.r, self.g, self.b, self.param = _f32(r), _f32(g), _f32(b), _f32(param)
        n = len(self.cx)
        for f in self.FIELDS:
            if len(getattr(self, f)) != n:
                raise ValueError("Spheres: field %s has length %d, expected %d" % (f, len(getattr(self, f)), n))

    def __len__(self):
        return len(self.cx)

    @classmethod
    def empty(cls, n):
        z = np.zeros(n, dtype=np.float32)
        return cls(z.copy(), z.copy(), z.copy(), z.copy(), np.zeros(n, np.uint32), z.copy(), z.copy(), z.copy(), z.copy())

    def view(self):
        return rt_spheres(len(self), _ptr(self.cx, _lib.f32p), _ptr(self.cy, _lib.f32p), _ptr(self.cz, _lib.f32p),
                          _ptr(self.radius, _lib.f32p), _ptr(self.kind, _lib.u32p), _ptr(self.r, _lib.f32p),
                          _ptr(self.g, _lib.f32p), _ptr(self.b, _lib.f32p), _ptr(self.param, _lib.f32p))

    def _out_ptrs(self):
        return [getattr(self, f).ctypes.data for f in self.FIELDS]

    def take(self, n):
        return Spheres(*[getattr(self, f)[:n].copy() for f in self.FIELDS])


def Random_Scene(half_size=11, seed=0):
    """Random_Scene (raytracer.zig:644-689) with std.Random.Sfc64 seeded `seed` (reference: 0, :732)."""
    cap = 2 * (2 * half_size) ** 2 + 8
    s = Spheres.empty(cap)
    n = _lib.load().rt_random_scene(half_size, seed, cap, *s._out_ptrs())
    if n < 0:
        raise RtError(n, "rt_random_scene")
    return s.take(n)


def Two_Sphere_Scene():
    """book_segments.zig:146-149,162."""
    s = Spheres.empty(2)
    n = _lib.load().rt_two_sphere_scene(2, *s._out_ptrs())
    if n < 0:
        raise RtError(n, "rt_two_sphere_scene")
    return s.take(n)


class Triangles:
    """Triangle soup + the material table it indexes (build-defined; main.zig:115-154 meshes)."""

    def __init__(self, v0, v1, v2, material):
        self.v0, self.v1, self.v2 = _f32(v0).reshape(-1, 3), _f32(v1).reshape(-1, 3), _f32(v2).reshape(-1, 3)
        self.material = np.ascontiguousarray(material, dtype=np.uint32)

    def __len__(self):
        return len(self.material)

    def view(self):
        return _lib.rt_triangles(len(self), _ptr(self.v0, _lib.f32p), _ptr(self.v1, _lib.f32p),
                                 _ptr(self.v2, _lib.f32p), _ptr(self.material, _lib.u32p))


def Tessellate_Spheres(spheres, slices=0, stacks=0):
    """zmesh parametric spheres as placed at main.zig:128-143; (0,0) = the reference's per-radius choice."""
    L = _lib.load()
    sv = spheres.view()
    n = L.rt_tessellate_spheres(C.byref(sv), slices, stacks, 0, None, None, None, None)
    if n < 0:
        raise RtError(n, "rt_tessellate_spheres")
    v0 = np.zeros((n, 3), np.float32)
    v1 = np.zeros((n, 3), np.float32)
    v2 = np.zeros((n, 3), np.float32)
    mat = np.zeros(n, np.uint32)
    m = L.rt_tessellate_spheres(C.byref(sv), slices, stacks, n, v0.ctypes.data, v1.ctypes.data, v2.ctypes.data,
                                mat.ctypes.data)
    if m != n:
        raise RtError(m if m < 0 else _lib.RT_E_CAP, "rt_tessellate_spheres")
    return Triangles(v0, v1, v2, mat)


class _CameraBase:
    def __init__(self):
        self._c = rt_camera()

    @property
    def lower_left(self):
        return np.array(self._c.lower_left, dtype=np.float32)

    @property
    def horizontal(self):
        return np.array(self._c.horizontal, dtype=np.float32)

    @property
    def vertical(self):
        return np.array(self._c.vertical, dtype=np.float32)

    @property
    def origin(self):
        return np.array(self._c.origin, dtype=np.float32)

    def as_array(self):
        return np.stack([self.lower_left, self.horizontal, self.vertical, self.origin])

    def to_abi(self):
        return self._c


class BasicCamera(_CameraBase):
    """camera.zig:9-19."""

    def __init__(self):
        super().__init__()
        _lib.load().rt_camera_basic(C.byref(self._c))


class MovableCamera(_CameraBase):
    """camera.zig:20-42; defaults equal BasicCamera's vectors until SetPosition is called."""

    def __init__(self):
        super().__init__()
        _lib.load().rt_camera_basic(C.byref(self._c))
        self._c.subtract_origin = 1

    def SetPosition(self, lookfrom, lookat, v_up, vfov, aspect):
        a3 = C.c_float * 3
        _lib.load().rt_camera_set_position(C.byref(self._c), a3(*map(float, lookfrom)), a3(*map(float, lookat)),
                                           a3(*map(float, v_up)), C.c_float(vfov), C.c_float(aspect))
        return self


def book_camera(nx, ny):
    """raytracer.zig:731 / main.zig:569: (13,2,3.5) -> (0,0,0), vfov 20, aspect f32(nx)/f32(ny)."""
    return MovableCamera().SetPosition((13.0, 2.0, 3.5), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 20.0,
                                       float(np.float32(nx) / np.float32(ny)))


def make_config(nx, ny, spp, max_depth=16, seed=1, t_min=0.001, spp_begin=0, spp_end=None, spp_total=None,
                tile=None, row_offset_q3=1, spp_chunk=0):
    """rt_config with the reference's constants as defaults (raytracer.zig:555-556)."""
    if spp_end is None:
        spp_end = spp
    if spp_total is None:
        spp_total = spp
    x0, y0, x1, y1 = tile if tile is not None else (0, 0, 0, 0)
    return rt_config(nx, ny, spp_begin, spp_end, spp_total, max_depth, t_min, seed, x0, y0, x1, y1,
                     row_offset_q3, spp_chunk)


@dataclass
class RenderResult:
    accum: np.ndarray      # [ny*nx, 3] f32 linear mean radiance
    rgb8: np.ndarray       # [ny*nx, 3] u8
    raycount: np.ndarray   # [ny*nx] u32 (None unless requested)
    rays: int
    primary_rays: int
    kernel_ms: float
    rays_by_depth: list


def _stats_tuple(st):
    return int(st.rays), int(st.primary_rays), float(st.kernel_ms), [int(v) for v in st.rays_by_depth]


class Context:
    """One GPU's render context (rt_ctx).  Mirrors compute.zig's create -> upload -> dispatch -> readback."""

    def __init__(self, device=0):
        self._L = _lib.load()
        h = C.c_void_p()
        rc = self._L.rt_create(device, C.byref(h))
        if rc != 0:
            raise RtError(rc, "rt_create")
        self._h = h
        self.device = device
        self._keep = None

    def close(self):
        if getattr(self, "_h", None):
            self._L.rt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, where):
        if rc != 0:
            raise RtError(rc, where, self._L.rt_last_hip_error(self._h))

    def set_flags(self, flags):
        self._check(self._L.rt_set_flags(self._h, flags), "rt_set_flags")

    def set_tuning(self, queue_batch=0, flush_lanes=0, keep_traversing=0, force_global_scene=0, bvh_depth_cap=0,
                   build_on_device=0, node_format=0):
        """rt_set_tuning: kernel tuning knobs of this ctx, 0 = default (bvh_depth_cap and build_on_device apply to the
        next upload)."""
        t = rt_tuning(queue_batch, flush_lanes, keep_traversing, force_global_scene, bvh_depth_cap, build_on_device, node_format)
        self._check(self._L.rt_set_tuning(self._h, C.byref(t)), "rt_set_tuning")

    def set_cu_mask(self, n_cus=None):
        """rt_set_cu_mask: run this ctx's renders on the first n_cus compute units (mask bits 0..n_cus-1: on MI355X
        spread evenly over the 8 XCDs); None = the whole GPU."""
        if n_cus is None:
            self._check(self._L.rt_set_cu_mask(self._h, None, 0), "rt_set_cu_mask")
            return
        if n_cus < 8:
            # fewer than 8 leading bits leave an XCD without a CU; the library refuses such a mask too (RT_E_ARG)
            raise ValueError("set_cu_mask: at least 8 CUs (one per XCD), got %d" % n_cus)
        words = (n_cus + 31) // 32
        mask = (C.c_uint32 * words)(*[0xFFFFFFFF if (w + 1) * 32 <= n_cus else (1 << (n_cus - w * 32)) - 1
                                      for w in range(words)])
        self._check(self._L.rt_set_cu_mask(self._h, mask, words), "rt_set_cu_mask")

    def set_cu_mask_words(self, words):
        """rt_set_cu_mask with an explicit mask (list of 32-bit words, bit i = CU i)."""
        mask = (C.c_uint32 * len(words))(*words)
        self._check(self._L.rt_set_cu_mask(self._h, mask, len(words)), "rt_set_cu_mask")

    def upload_scene(self, spheres=None, triangles=None, materials=None):
        sv = spheres.view() if spheres is not None else None
        tv = triangles.view() if triangles is not None else None
        mv = materials.view() if materials is not None else None
        rc = self._L.rt_upload_scene(self._h, C.byref(sv) if sv is not None else None,
                                     C.byref(tv) if tv is not None else None,
                                     C.byref(mv) if mv is not None else None)
        self._check(rc, "rt_upload_scene")

    def scene_info(self):
        vals = [C.c_uint32() for _ in range(4)]
        self._check(self._L.rt_scene_info(self._h, *[C.byref(x) for x in vals]), "rt_scene_info")
        return dict(zip(("n_spheres", "n_triangles", "n_bvh_nodes", "bvh_depth"), [int(x.value) for x in vals]))

    def upload_ms(self):
        """rt_upload_timing of the last upload: {host_build, device_copy, device_build} in ms."""
        a, b, c = C.c_double(0), C.c_double(0), C.c_double(0)
        self._check(self._L.rt_upload_timing(self._h, C.byref(a), C.byref(b), C.byref(c)), "rt_upload_timing")
        return {"host_build": a.value, "device_copy": b.value, "device_build": c.value}

    def wave_fill(self):
        """Fill of the BVH kernel's phases in the last RT_FLAG_DEPTH_HISTOGRAM render (profiling aid)."""
        out = (C.c_uint64 * 6)()
        self._check(self._L.rt_debug_wave_fill(self._h, out), "rt_debug_wave_fill")
        ws, ls, wf, lf, wsh, lsh = [int(v) for v in out]
        return {"wave_steps": ws, "lane_steps": ls, "traverse_fill": ls / max(1, 64 * ws),
                "wave_exact_iters": wf, "lane_exact_iters": lf, "exact_fill": lf / max(1, 64 * wf),
                "wave_shade_passes": wsh, "lanes_shaded": lsh, "shade_fill": lsh / max(1, 64 * wsh)}

    BLOCK_NAMES = ("surface fetch", "normalize(direction)", "sky", "work-item dequeue", "philox", "camera ray",
                   "random_in_unit_sphere", "cbrt fallback", "lambertian", "metal", "dielectric", "normal IEEE fallback")

    def block_counts(self):
        """Blocks of the shading pass in the last RT_FLAG_DEPTH_HISTOGRAM render: name -> (wave executions, lanes)."""
        out = (C.c_uint64 * 64)()
        n = self._L.rt_debug_block_counts(self._h, 64, out)
        if n < 0:
            raise RtError(n, "rt_debug_block_counts", 0)
        return {self.BLOCK_NAMES[k]: (int(out[2 * k]), int(out[2 * k + 1])) for k in range(n // 2)}

    def last_kernel_name(self):
        return self._L.rt_last_kernel_name(self._h).decode()

    def render(self, cam, cfg, want_raycount=False):
        """rt_render: host images out.  Blocking."""
        n = cfg.nx * cfg.ny
        accum = np.zeros((n, 3), np.float32)
        rgb8 = np.zeros((n, 3), np.uint8)
        rc_arr = np.zeros(n, np.uint32) if want_raycount else None
        st = rt_stats()
        rc = self._L.rt_render(self._h, C.byref(cam.to_abi()), C.byref(cfg), accum.ctypes.data, rgb8.ctypes.data,
                               rc_arr.ctypes.data if want_raycount else None, C.byref(st))
        self._check(rc, "rt_render")
        rays, prim, ms, hist = _stats_tuple(st)
        return RenderResult(accum, rgb8, rc_arr, rays, prim, ms, hist)

    def render_reference_style(self, cam, cfg, n_images):
        """rt_render_reference_style: the reference's N-images u8 integer mean (raytracer.zig:594-617)."""
        rgb8 = np.zeros((cfg.nx * cfg.ny, 3), np.uint8)
        st = rt_stats()
        rc = self._L.rt_render_reference_style(self._h, C.byref(cam.to_abi()), C.byref(cfg), n_images,
                                               rgb8.ctypes.data, C.byref(st))
        self._check(rc, "rt_render_reference_style")
        return rgb8, _stats_tuple(st)

    def render_progressive(self, cam, cfg, n_passes):
        """rt_progressive_begin / rt_progressive_step as a generator: yields (spp_done, rgb8, accum_mean, stats tuple)
        after every pass; the arrays are reused between passes."""
        n = cfg.nx * cfg.ny
        rgb8 = np.zeros((n, 3), np.uint8)
        accum = np.zeros((n, 3), np.float32)
        self._check(self._L.rt_progressive_begin(self._h, C.byref(cam.to_abi()), C.byref(cfg), n_passes),
                    "rt_progressive_begin")
        while True:
            done, st = C.c_uint32(0), rt_stats()
            rc = self._L.rt_progressive_step(self._h, rgb8.ctypes.data, accum.ctypes.data, C.byref(done), C.byref(st))
            if rc < 0:
                raise RtError(rc, "rt_progressive_step", self._L.rt_last_hip_error(self._h))
            yield int(done.value), rgb8, accum, _stats_tuple(st)
            if rc == 0:
                return

    def render_device(self, cam, cfg, d_accum_sum, d_raycount=0, stream=0, want_stats=True):
        """rt_render_device: adds into a device accumulation buffer (raw device pointers as ints)."""
        st = rt_stats()
        rc = self._L.rt_render_device(self._h, C.byref(cam.to_abi()), C.byref(cfg), C.c_void_p(d_accum_sum),
                                      C.c_void_p(d_raycount) if d_raycount else None,
                                      C.c_void_p(stream) if stream else None,
                                      C.byref(st) if want_stats else None)
        self._check(rc, "rt_render_device")
        return _stats_tuple(st) if want_stats else None

    def render_device_noalloc(self, cam, cfg, d_accum_sum, d_raycount=0, stream=0):
        """rt_render_device_noalloc: rt_render_device's work with the library's fixed scratch array instead of
        ctx-owned allocations; enqueue only (no stats)."""
        rc = self._L.rt_render_device_noalloc(self._h, C.byref(cam.to_abi()), C.byref(cfg), C.c_void_p(d_accum_sum),
                                              C.c_void_p(d_raycount) if d_raycount else None,
                                              C.c_void_p(stream) if stream else None)
        self._check(rc, "rt_render_device_noalloc")

    def tonemap_device(self, d_accum_sum, n_pixels, spp_total, d_mean=0, d_rgb8=0, stream=0):
        rc = self._L.rt_tonemap_device(self._h, C.c_void_p(d_accum_sum), n_pixels, spp_total,
                                       C.c_void_p(d_mean) if d_mean else None,
                                       C.c_void_p(d_rgb8) if d_rgb8 else None,
                                       C.c_void_p(stream) if stream else None)
        self._check(rc, "rt_tonemap_device")

    def render_and_write(self, fname, cam, cfg):
        st = rt_stats()
        rc = self._L.rt_render_and_write(self._h, fname.encode(), C.byref(cam.to_abi()), C.byref(cfg), C.byref(st))
        self._check(rc, "rt_render_and_write")
        return _stats_tuple(st)

    def selftest_rccl(self, n=4096):
        """rt_selftest_rccl: one-rank RCCL communicator + ncclReduce on this ctx's GPU."""
        self._check(self._L.rt_selftest_rccl(self._h, n), "rt_selftest_rccl")

    def selftest_cbrt_grid(self):
        """Production cbrt on all 2^24 inputs k * 2^-24; returns (values, number of inputs that took the fallback)."""
        out = np.zeros(1 << 24, np.float32)
        n = C.c_uint64(0)
        self._check(self._L.rt_selftest_cbrt_grid(self._h, out.ctypes.data, C.byref(n)), "rt_selftest_cbrt_grid")
        return out, int(n.value)

    def selftest_math(self, a, b, ctr0):
        a, b = _f32(a), _f32(b)
        ctr0 = np.ascontiguousarray(ctr0, dtype=np.uint32)
        n = len(a)
        od, os_, oc = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
        ph = np.zeros((n, 4), np.uint32)
        rc = self._L.rt_selftest_math(self._h, n, a.ctypes.data, b.ctypes.data, od.ctypes.data, os_.ctypes.data,
                                      oc.ctypes.data, ctr0.ctypes.data, ph.ctypes.data)
        self._check(rc, "rt_selftest_math")
        return od, os_, oc, ph


def device_count():
    """rt_device_count: GPUs visible to the library."""
    n = _lib.load().rt_device_count()
    if n < 0:
        raise RtError(n, "rt_device_count")
    return n


def render_multi(contexts, cam, cfg, want_raycount=False):
    """rt_render_multi: one process, several GPUs (one Context each, same scene uploaded); the sample range of
    cfg is split across them and summed on contexts[0]'s GPU (RCCL reduce; ordered add when contexts share a GPU).
    Blocking."""
    if not contexts:
        raise ValueError("render_multi needs at least one Context")
    L = contexts[0]._L
    n = cfg.nx * cfg.ny
    accum = np.zeros((n, 3), np.float32)
    rgb8 = np.zeros((n, 3), np.uint8)
    st = rt_stats()
    rc_arr = np.zeros(n, np.uint32) if want_raycount else None
    handles = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    rc = L.rt_render_multi(handles, len(contexts), C.byref(cam.to_abi()), C.byref(cfg), accum.ctypes.data,
                           rgb8.ctypes.data, rc_arr.ctypes.data if want_raycount else None, C.byref(st))
    if rc != 0:
        raise RtError(rc, "rt_render_multi", max(L.rt_last_hip_error(c._h) for c in contexts))
    rays, prim, ms, hist = _stats_tuple(st)
    return RenderResult(accum, rgb8, rc_arr, rays, prim, ms, hist)


def write_ppm_image(fname, rgb8, nx, ny):
    """write_ppm_image, image.zig:18-37."""
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    rc = _lib.load().rt_write_ppm(fname.encode(), rgb8.ctypes.data, nx, ny)
    if rc != 0:
        raise RtError(rc, "rt_write_ppm")


def render_and_write(fname, num_samples, world, shape, cam, device=0, seed=1, max_depth=16):
    """render_and_write (raytracer.zig:580-633): `world` is a Spheres list, `shape` = (ny, nx) as shape_s,
    `cam` replaces the reference's global `cam`.  num_samples is the total per pixel."""
    ny, nx = shape
    with Context(device) as ctx:
        ctx.upload_scene(world)
        return ctx.render_and_write(fname, cam, make_config(nx, ny, num_samples, max_depth=max_depth, seed=seed,
                                                            spp_chunk=_lib.RT_SPP_CHUNK_AUTO))
