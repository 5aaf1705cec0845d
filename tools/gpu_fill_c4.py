import os, sys
sys.path.insert(0, ".")
import raytrace_amd as R
ctx = R.Context(0)
book = R.Random_Scene(11, 0)
tris = R.Tessellate_Spheres(book, 0, 0)
ctx.upload_scene(triangles=tris, materials=book)
ctx.set_flags(R.RT_FLAG_DEPTH_HISTOGRAM)
cam = R.book_camera(1920, 1080)
res = ctx.render(cam, R.make_config(1920, 1080, 32, spp_chunk=R.RT_SPP_CHUNK_AUTO))
f = ctx.wave_fill(); rays = res.rays
print("C4 rays %d kernel %.2f ms %s" % (rays, res.kernel_ms, ctx.scene_info()))
print("  per ray: lane steps %.2f, lane exact %.2f; per 64 rays: wave steps %.1f, wave exact iters %.1f, wave shade passes %.2f" % (
    f["lane_steps"] / rays, f["lane_exact_iters"] / rays, f["wave_steps"] * 64 / rays, f["wave_exact_iters"] * 64 / rays, f["wave_shade_passes"] * 64 / rays))
print("  fill: traverse %.3f exact %.3f shade %.3f" % (f["traverse_fill"], f["exact_fill"], f["shade_fill"]))
