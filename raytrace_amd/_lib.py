"""ctypes loader for the C-ABI library (include/rt_hip.h).

The library is built in-tree (raytrace_amd/librt_hip.so) by `make -C raytrace_amd/csrc` /
__graft_entry__.build().  There is no fallback: if it is missing, importing the product path fails.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_HIP_LIB") or os.path.join(_HERE, "librt_hip.so")   # RT_HIP_LIB: another build of the same library (dev aid)

RT_OK, RT_E_ARG, RT_E_HIP, RT_E_NOMEM, RT_E_STATE, RT_E_IO, RT_E_CAP, RT_E_NCCL = 0, -1, -2, -3, -4, -5, -6, -7
RT_LAMBERTIAN, RT_METAL, RT_DIELECTRIC = 0, 1, 2
RT_SPP_CHUNK_AUTO = 0xFFFFFFFF
RT_NOALLOC_SCRATCH_BYTES = 64 << 20
RT_FLAG_DEPTH_HISTOGRAM, RT_FLAG_FORCE_BRUTE_FORCE, RT_FLAG_FORCE_BVH = 1, 2, 4

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)


class rt_camera(C.Structure):
    _fields_ = [("lower_left", C.c_float * 3), ("horizontal", C.c_float * 3), ("vertical", C.c_float * 3),
                ("origin", C.c_float * 3), ("subtract_origin", C.c_uint32)]


class rt_spheres(C.Structure):
    _fields_ = [("n", C.c_uint32), ("cx", f32p), ("cy", f32p), ("cz", f32p), ("radius", f32p),
                ("kind", u32p), ("r", f32p), ("g", f32p), ("b", f32p), ("param", f32p)]


class rt_triangles(C.Structure):
    _fields_ = [("n", C.c_uint32), ("v0", f32p), ("v1", f32p), ("v2", f32p), ("material", u32p)]


class rt_config(C.Structure):
    _fields_ = [("nx", C.c_uint32), ("ny", C.c_uint32), ("spp_begin", C.c_uint32), ("spp_end", C.c_uint32),
                ("spp_total", C.c_uint32), ("max_depth", C.c_uint32), ("t_min", C.c_float), ("seed", C.c_uint64),
                ("tile_x0", C.c_uint32), ("tile_y0", C.c_uint32), ("tile_x1", C.c_uint32), ("tile_y1", C.c_uint32),
                ("row_offset_q3", C.c_uint32), ("spp_chunk", C.c_uint32)]


class rt_tuning(C.Structure):
    _fields_ = [("queue_batch", C.c_uint32), ("flush_lanes", C.c_uint32), ("keep_traversing", C.c_uint32),
                ("force_global_scene", C.c_uint32), ("bvh_depth_cap", C.c_uint32), ("build_on_device", C.c_uint32), ("node_format", C.c_uint32)]
RT_NODES_BINARY80, RT_NODES_BINARY72, RT_NODES_WIDE = 1, 2, 3


class rt_stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("primary_rays", C.c_uint64), ("kernel_ms", C.c_double),
                ("rays_by_depth", C.c_uint64 * 20)]


# every symbol include/rt_hip.h declares
SYMBOLS = [
    "rt_create", "rt_destroy", "rt_strerror", "rt_last_hip_error", "rt_last_rccl_error", "rt_selftest_rccl", "rt_set_flags", "rt_set_tuning", "rt_set_cu_mask", "rt_check_cu_mask", "rt_upload_scene",
    "rt_scene_info", "rt_upload_timing", "rt_last_kernel_name", "rt_debug_wave_fill", "rt_debug_block_counts", "rt_debug_round_plane",
    "rt_device_count", "rt_render", "rt_render_multi", "rt_render_reference_style", "rt_progressive_begin", "rt_progressive_step", "rt_render_device", "rt_render_device_noalloc", "rt_tonemap_device", "rt_camera_basic", "rt_camera_set_position",
    "rt_random_scene", "rt_two_sphere_scene", "rt_tessellate_spheres", "rt_write_ppm", "rt_render_and_write",
    "rt_selftest_math", "rt_selftest_cbrt_grid",
]

_lib = None


def load():
    """Load librt_hip.so; raises if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "raytrace_amd: %s is missing -- build it with `make -C raytrace_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64, and a second copy loaded
    # before it leaves torch without a device ("No HIP GPUs are available").  When torch is installed,
    # load it first so librt_hip.so binds to the runtime that is already resident (same SONAME).
    if os.environ.get("RT_HIP_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.rt_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.rt_create.restype = C.c_int
    L.rt_destroy.argtypes = [vp]
    L.rt_destroy.restype = None
    L.rt_strerror.argtypes = [C.c_int]
    L.rt_strerror.restype = C.c_char_p
    L.rt_last_hip_error.argtypes = [vp]
    L.rt_last_hip_error.restype = C.c_int
    L.rt_set_flags.argtypes = [vp, C.c_uint32]
    L.rt_set_flags.restype = C.c_int
    L.rt_set_tuning.argtypes = [vp, C.POINTER(rt_tuning)]
    L.rt_set_tuning.restype = C.c_int
    L.rt_set_cu_mask.argtypes = [vp, u32p, C.c_uint32]
    L.rt_set_cu_mask.restype = C.c_int
    L.rt_check_cu_mask.argtypes = [u32p, C.c_uint32, C.c_uint32]
    L.rt_check_cu_mask.restype = C.c_int
    L.rt_upload_scene.argtypes = [vp, C.POINTER(rt_spheres), C.POINTER(rt_triangles), C.POINTER(rt_spheres)]
    L.rt_upload_scene.restype = C.c_int
    L.rt_scene_info.argtypes = [vp, u32p, u32p, u32p, u32p]
    L.rt_scene_info.restype = C.c_int
    L.rt_upload_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.rt_upload_timing.restype = C.c_int
    L.rt_debug_wave_fill.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.rt_debug_wave_fill.restype = C.c_int
    L.rt_last_kernel_name.argtypes = [vp]
    L.rt_last_kernel_name.restype = C.c_char_p
    L.rt_render.argtypes = [vp, C.POINTER(rt_camera), C.POINTER(rt_config), vp, vp, vp, C.POINTER(rt_stats)]
    L.rt_render.restype = C.c_int
    L.rt_device_count.argtypes = []
    L.rt_device_count.restype = C.c_int
    L.rt_selftest_cbrt_grid.argtypes = [vp, vp, C.POINTER(C.c_uint64)]
    L.rt_selftest_cbrt_grid.restype = C.c_int
    L.rt_debug_block_counts.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint64)]
    L.rt_debug_block_counts.restype = C.c_int
    L.rt_debug_round_plane.argtypes = [C.c_float, C.c_uint32, C.c_int]
    L.rt_debug_round_plane.restype = C.c_float
    L.rt_render_multi.argtypes = [C.POINTER(vp), C.c_uint32, C.POINTER(rt_camera), C.POINTER(rt_config), vp, vp, vp,
                                  C.POINTER(rt_stats)]
    L.rt_last_rccl_error.argtypes = [vp]
    L.rt_last_rccl_error.restype = C.c_int
    L.rt_selftest_rccl.argtypes = [vp, C.c_uint32]
    L.rt_selftest_rccl.restype = C.c_int
    L.rt_render_multi.restype = C.c_int
    L.rt_render_reference_style.argtypes = [vp, C.POINTER(rt_camera), C.POINTER(rt_config), C.c_uint32, vp,
                                            C.POINTER(rt_stats)]
    L.rt_render_reference_style.restype = C.c_int
    L.rt_progressive_begin.argtypes = [vp, C.POINTER(rt_camera), C.POINTER(rt_config), C.c_uint32]
    L.rt_progressive_begin.restype = C.c_int
    L.rt_progressive_step.argtypes = [vp, vp, vp, u32p, C.POINTER(rt_stats)]
    L.rt_progressive_step.restype = C.c_int
    L.rt_render_device.argtypes = [vp, C.POINTER(rt_camera), C.POINTER(rt_config), vp, vp, vp, C.POINTER(rt_stats)]
    L.rt_render_device.restype = C.c_int
    L.rt_render_device_noalloc.argtypes = [vp, C.POINTER(rt_camera), C.POINTER(rt_config), vp, vp, vp]
    L.rt_render_device_noalloc.restype = C.c_int
    L.rt_tonemap_device.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, vp, vp]
    L.rt_tonemap_device.restype = C.c_int
    L.rt_camera_basic.argtypes = [C.POINTER(rt_camera)]
    L.rt_camera_basic.restype = None
    L.rt_camera_set_position.argtypes = [C.POINTER(rt_camera), C.c_float * 3, C.c_float * 3, C.c_float * 3,
                                         C.c_float, C.c_float]
    L.rt_camera_set_position.restype = None
    L.rt_random_scene.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32] + [vp] * 9
    L.rt_random_scene.restype = C.c_int
    L.rt_two_sphere_scene.argtypes = [C.c_uint32] + [vp] * 9
    L.rt_two_sphere_scene.restype = C.c_int
    L.rt_tessellate_spheres.argtypes = [C.POINTER(rt_spheres), C.c_int, C.c_int, C.c_uint32, vp, vp, vp, vp]
    L.rt_tessellate_spheres.restype = C.c_int
    L.rt_write_ppm.argtypes = [C.c_char_p, vp, C.c_uint32, C.c_uint32]
    L.rt_write_ppm.restype = C.c_int
    L.rt_render_and_write.argtypes = [vp, C.c_char_p, C.POINTER(rt_camera), C.POINTER(rt_config), C.POINTER(rt_stats)]
    L.rt_render_and_write.restype = C.c_int
    L.rt_selftest_math.argtypes = [vp, C.c_uint32] + [vp] * 7
    L.rt_selftest_math.restype = C.c_int
    _lib = L
    return L


class RtError(RuntimeError):
    def __init__(self, code, where, hip=0):
        self.code = code
        self.hip = hip
        msg = load().rt_strerror(code).decode()
        super().__init__("%s: %s (code %d, hip %d)" % (where, msg, code, hip))
