"""raytrace_amd -- MI355X (gfx950) per-pixel trace path behind the API of Copper280z/raytrace.

The product path is the HIP library raytrace_amd/librt_hip.so (C ABI: include/rt_hip.h); this
package is the thin Python host mirror used by tests and bench.py.  Nothing here imports oracle/.
"""
from .api import (BasicCamera, Context, MovableCamera, Spheres, Random_Scene, Two_Sphere_Scene,  # noqa: F401
                  make_config, render_and_write, render_multi, device_count, write_ppm_image, book_camera, RenderResult,
                  Triangles, Tessellate_Spheres)
from ._lib import (RT_LAMBERTIAN, RT_METAL, RT_DIELECTRIC, RT_SPP_CHUNK_AUTO, RT_FLAG_DEPTH_HISTOGRAM, RT_FLAG_FORCE_BRUTE_FORCE, RT_FLAG_FORCE_BVH, RtError,  # noqa: F401
                   rt_camera, rt_config, rt_stats)
