"""ctypes binding of oracle/librt_oracle.so -- the CPU checker (test infrastructure only).

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product
package raytrace_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.environ.get("RT_ORACLE_LIB") or os.path.join(ORACLE_DIR, "librt_oracle.so")   # RT_ORACLE_LIB: bench.py's natively built copy

LAMBERTIAN, METAL, DIELECTRIC = 0, 1, 2
TRAV_LIST, TRAV_BVH_REF, TRAV_BRUTE = 0, 1, 2


class V3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def tup(self):
        return (self.x, self.y, self.z)


def v3(t):
    return V3(float(t[0]), float(t[1]), float(t[2]))


SPHERE_DTYPE = np.dtype(
    [("center", np.float32, 3), ("radius", np.float32), ("kind", np.uint32),
     ("albedo", np.float32, 3), ("param", np.float32)], align=False)
TRIANGLE_DTYPE = np.dtype(
    [("v0", np.float32, 3), ("v1", np.float32, 3), ("v2", np.float32, 3), ("material", np.uint32)],
    align=False)
assert SPHERE_DTYPE.itemsize == 36 and TRIANGLE_DTYPE.itemsize == 40


class Camera(C.Structure):
    _fields_ = [("lower_left", V3), ("horizontal", V3), ("vertical", V3), ("origin", V3),
                ("subtract_origin", C.c_uint32)]

    def as_array(self):
        return np.array([self.lower_left.tup(), self.horizontal.tup(), self.vertical.tup(),
                         self.origin.tup()], dtype=np.float32)


class Config(C.Structure):
    _fields_ = [("nx", C.c_uint32), ("ny", C.c_uint32),
                ("spp_begin", C.c_uint32), ("spp_end", C.c_uint32), ("spp_total", C.c_uint32),
                ("max_depth", C.c_uint32), ("t_min", C.c_float), ("seed", C.c_uint64),
                ("row_offset_q3", C.c_uint32), ("traversal", C.c_uint32),
                ("spp_chunk", C.c_uint32), ("threads", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("primary_rays", C.c_uint64), ("sphere_tests", C.c_uint64),
                ("aabb_tests", C.c_uint64), ("tri_tests", C.c_uint64), ("root_evals", C.c_uint64),
                ("hits", C.c_uint64), ("misses", C.c_uint64),
                ("scatter_lambertian", C.c_uint64), ("scatter_metal", C.c_uint64),
                ("scatter_dielectric", C.c_uint64), ("absorbed", C.c_uint64),
                ("depth_exceeded", C.c_uint64), ("rays_by_depth", C.c_uint64 * 40)]

    def as_dict(self):
        d = {k: int(getattr(self, k)) for k, _ in self._fields_ if k != "rays_by_depth"}
        d["rays_by_depth"] = [int(v) for v in self.rays_by_depth]
        return d


class Sfc64(C.Structure):
    _fields_ = [("a", C.c_uint64), ("b", C.c_uint64), ("c", C.c_uint64), ("counter", C.c_uint64)]


def build(force=False):
    """Compile the oracle (gcc only; seconds)."""
    if force or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(LIB_PATH)
            for f in ("rt_oracle.c", "rt_oracle.h", "Makefile")):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(LIB_PATH)
    f32p = C.POINTER(C.c_float)
    u32p = C.POINTER(C.c_uint32)
    L.orc_sfc64_init.argtypes = [C.POINTER(Sfc64), C.c_uint64]
    L.orc_sfc64_next.argtypes = [C.POINTER(Sfc64)]
    L.orc_sfc64_next.restype = C.c_uint64
    L.orc_sfc64_float.argtypes = [C.POINTER(Sfc64)]
    L.orc_sfc64_float.restype = C.c_float
    L.orc_philox4x32_10.argtypes = [u32p, u32p, u32p]
    L.orc_u32_to_unit_float.argtypes = [C.c_uint32]
    L.orc_u32_to_unit_float.restype = C.c_float
    L.orc_dot.argtypes = [V3, V3]
    L.orc_dot.restype = C.c_float
    L.orc_length.argtypes = [V3]
    L.orc_length.restype = C.c_float
    L.orc_normalize.argtypes = [V3]
    L.orc_normalize.restype = V3
    L.orc_cross.argtypes = [V3, V3]
    L.orc_cross.restype = V3
    L.orc_argmax.argtypes = [V3]
    L.orc_argmax.restype = C.c_uint32
    L.orc_cbrtf.argtypes = [C.c_float]
    L.orc_cbrtf.restype = C.c_float
    L.orc_cbrtf_array.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.orc_cbrtf_array.restype = None
    L.orc_pow5.argtypes = [C.c_float]
    L.orc_pow5.restype = C.c_float
    L.orc_schlick.argtypes = [C.c_float, C.c_float]
    L.orc_schlick.restype = C.c_float
    L.orc_reflect.argtypes = [V3, V3]
    L.orc_reflect.restype = V3
    L.orc_refract.argtypes = [V3, V3, C.c_float, C.POINTER(V3)]
    L.orc_refract.restype = C.c_int
    L.orc_sphere_hit.argtypes = [C.c_void_p, V3, V3, C.c_float, C.c_float, f32p, C.POINTER(V3), C.POINTER(V3)]
    L.orc_sphere_hit.restype = C.c_int
    L.orc_triangle_hit.argtypes = [C.c_void_p, V3, V3, C.c_float, C.c_float, f32p, C.POINTER(V3), C.POINTER(V3)]
    L.orc_triangle_hit.restype = C.c_int
    L.orc_canonical_triangle.argtypes = [C.c_void_p]
    L.orc_canonical_triangle.restype = None
    L.orc_sky.argtypes = [V3]
    L.orc_sky.restype = V3
    L.orc_camera_basic.argtypes = [C.POINTER(Camera)]
    L.orc_camera_set_position.argtypes = [C.POINTER(Camera), V3, V3, V3, C.c_float, C.c_float]
    L.orc_camera_get_ray.argtypes = [C.POINTER(Camera), C.c_float, C.c_float, C.POINTER(V3), C.POINTER(V3)]
    L.orc_random_scene.argtypes = [C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32]
    L.orc_random_scene.restype = C.c_int
    L.orc_two_sphere_scene.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_two_sphere_scene.restype = C.c_int
    L.orc_tessellate_spheres.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_uint32]
    L.orc_tessellate_spheres.restype = C.c_int
    L.orc_world_create.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    L.orc_world_create.restype = C.c_void_p
    L.orc_world_destroy.argtypes = [C.c_void_p]
    L.orc_world_bvh_nodes.argtypes = [C.c_void_p]
    L.orc_world_bvh_nodes.restype = C.c_uint32
    L.orc_render_counter.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Config), C.c_void_p, C.c_uint32,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Counters)]
    L.orc_render_counter.restype = C.c_int
    L.orc_render_reference_style.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Config), C.c_uint32,
                                             C.c_uint32, C.c_void_p, C.POINTER(Counters), C.POINTER(C.c_double)]
    L.orc_render_reference_style.restype = C.c_int
    L.orc_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
    L.orc_write_ppm.restype = C.c_int
    _lib = L
    return L


# ---- convenience wrappers -----------------------------------------------------------------

def random_scene(half_size=11, seed=0):
    cap = 2 * (2 * half_size) ** 2 + 8
    buf = np.zeros(cap, dtype=SPHERE_DTYPE)
    n = lib().orc_random_scene(half_size, seed, buf.ctypes.data, cap)
    assert n > 0
    return buf[:n].copy()


def two_sphere_scene():
    buf = np.zeros(2, dtype=SPHERE_DTYPE)
    assert lib().orc_two_sphere_scene(buf.ctypes.data, 2) == 2
    return buf


def tessellate(spheres, slices=0, stacks=0):
    spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
    if slices > 0:
        cap = len(spheres) * 2 * slices * stacks
    else:
        cap = sum(2 * 150 * 150 if r > 50 else (2 * 20 * 20 if r < 1.0 else 2 * 30 * 30)
                  for r in spheres["radius"])
    buf = np.zeros(cap, dtype=TRIANGLE_DTYPE)
    n = lib().orc_tessellate_spheres(spheres.ctypes.data, len(spheres), slices, stacks, buf.ctypes.data, cap)
    assert n >= 0
    return buf[:n].copy()


def camera_basic():
    cam = Camera()
    lib().orc_camera_basic(C.byref(cam))
    return cam


def camera_set_position(lookfrom, lookat, vup, vfov, aspect):
    cam = Camera()
    lib().orc_camera_set_position(C.byref(cam), v3(lookfrom), v3(lookat), v3(vup),
                                  C.c_float(vfov), C.c_float(aspect))
    return cam


def book_camera(nx, ny):
    """raytracer.zig:731 / main.zig:569: aspect is f32(nx)/f32(ny)."""
    return camera_set_position((13.0, 2.0, 3.5), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 20.0,
                               float(np.float32(nx) / np.float32(ny)))


class World:
    def __init__(self, spheres=None, triangles=None, materials=None):
        self.spheres = None if spheres is None else np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
        self.triangles = None if triangles is None else np.ascontiguousarray(triangles, dtype=TRIANGLE_DTYPE)
        self.materials = None if materials is None else np.ascontiguousarray(materials, dtype=SPHERE_DTYPE)
        ns = 0 if self.spheres is None else len(self.spheres)
        nt = 0 if self.triangles is None else len(self.triangles)
        nm = 0 if self.materials is None else len(self.materials)
        self.h = lib().orc_world_create(
            self.spheres.ctypes.data if ns else None, ns,
            self.triangles.ctypes.data if nt else None, nt,
            self.materials.ctypes.data if nm else None, nm)
        if not self.h:
            raise RuntimeError("orc_world_create failed")

    def close(self):
        if self.h:
            lib().orc_world_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_config(nx, ny, spp, max_depth=16, seed=1, t_min=0.001, spp_begin=0, spp_end=None, spp_total=None,
                row_offset_q3=1, traversal=TRAV_LIST, spp_chunk=0, threads=None):
    if spp_end is None:
        spp_end = spp
    if spp_total is None:
        spp_total = spp
    if threads is None:
        threads = min(8, os.cpu_count() or 1)
    return Config(nx, ny, spp_begin, spp_end, spp_total, max_depth, t_min, seed, row_offset_q3, traversal,
                  spp_chunk, threads)


def render_counter(world, cam, cfg, pixels=None):
    """Returns (accum_mean [n,3] f32, rgb8 [n,3] u8, raycount [n] u32, counters dict)."""
    if pixels is not None:
        pixels = np.ascontiguousarray(pixels, dtype=np.uint32)
        n = len(pixels)
    else:
        n = cfg.nx * cfg.ny
    accum = np.zeros((n, 3), dtype=np.float32)
    rgb8 = np.zeros((n, 3), dtype=np.uint8)
    rc = np.zeros(n, dtype=np.uint32)
    cnt = Counters()
    rcode = lib().orc_render_counter(world.h, C.byref(cam), C.byref(cfg),
                                     pixels.ctypes.data if pixels is not None else None, n,
                                     accum.ctypes.data, rgb8.ctypes.data, rc.ctypes.data, C.byref(cnt))
    if rcode != 0:
        raise RuntimeError("orc_render_counter -> %d" % rcode)
    return accum, rgb8, rc, cnt.as_dict()


def render_reference_style(world, cam, cfg, n_threads, spp_thread):
    """Returns (rgb8 [ny*nx,3], counters dict, seconds)."""
    rgb8 = np.zeros((cfg.nx * cfg.ny, 3), dtype=np.uint8)
    cnt = Counters()
    sec = C.c_double(0)
    rcode = lib().orc_render_reference_style(world.h, C.byref(cam), C.byref(cfg), n_threads, spp_thread,
                                             rgb8.ctypes.data, C.byref(cnt), C.byref(sec))
    if rcode != 0:
        raise RuntimeError("orc_render_reference_style -> %d" % rcode)
    return rgb8, cnt.as_dict(), sec.value
