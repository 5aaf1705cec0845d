"""C4 (285 520 triangles): how much of the kernel's time is ray incoherence?  Same scene, 1080p x 64 spp, with the
depth cap at 0 (primary rays + one bounce ray each, the primaries coherent within 8x8-pixel waves) and at 16."""
import sys
sys.path.insert(0, ".")
import raytrace_amd as R
ctx = R.Context(0)
book = R.Random_Scene(11, 0)
tris = R.Tessellate_Spheres(book, 0, 0)
ctx.upload_scene(triangles=tris, materials=book)
print(ctx.scene_info(), ctx.upload_ms())
cam = R.book_camera(1920, 1080)
for depth in (0, 1, 16):
    for _ in range(2):
        res = ctx.render(cam, R.make_config(1920, 1080, 64, max_depth=depth, spp_chunk=R.RT_SPP_CHUNK_AUTO))
    ctx.set_flags(R.RT_FLAG_DEPTH_HISTOGRAM)
    h = ctx.render(cam, R.make_config(1920, 1080, 64, max_depth=depth, spp_chunk=R.RT_SPP_CHUNK_AUTO))
    ctx.set_flags(0)
    f = ctx.wave_fill()
    print("depth cap %2d: %6.2f ms %7.0f Mrays/s  rays/primary %.2f | lane steps/ray %.1f exact/ray %.2f | fill trav %.2f exact %.2f shade %.2f"
          % (depth, res.kernel_ms, res.rays / res.kernel_ms / 1e3, res.rays / res.primary_rays, f["lane_steps"] / h.rays,
             f["lane_exact_iters"] / h.rays, f["traverse_fill"], f["exact_fill"], f["shade_fill"]), flush=True)
