"""Dev aid: Random_Scene at several sizes, 1080p x 100 spp -- where the LDS-staged variant ends and what the
global-memory variant costs."""
import os, sys
sys.path.insert(0, ".")
import raytrace_amd as R
ctx = R.Context(0)
nx, ny, spp = 1920, 1080, 100
cam = R.book_camera(nx, ny)
for half in (8, 11, 12, 13, 14, 20):
    s = R.Random_Scene(half, 0)
    ctx.upload_scene(s)
    for g in (0, 1):
        ctx.set_tuning(force_global_scene=g)
        for _ in range(2):
            res = ctx.render(cam, R.make_config(nx, ny, spp, spp_chunk=R.RT_SPP_CHUNK_AUTO))
        print("half %2d: %4d spheres %s forced_global=%d kernel %7.2f ms %7.0f Mrays/s" % (
            half, len(s), ctx.scene_info(), g, res.kernel_ms, res.rays / res.kernel_ms / 1e3), flush=True)
