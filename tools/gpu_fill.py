"""Wave-fill counters of the BVH kernel on C3 (dev aid)."""
import os, sys
sys.path.insert(0, ".")
import raytrace_amd as R
ctx = R.Context(0)
book = R.Random_Scene(11, 0)
ctx.upload_scene(book)
cam = R.book_camera(1920, 1080)
cfg = R.make_config(1920, 1080, 100, spp_chunk=R.RT_SPP_CHUNK_AUTO)
for _ in range(2):
    fast = ctx.render(cam, cfg)
ctx.set_flags(R.RT_FLAG_DEPTH_HISTOGRAM)
res = ctx.render(cam, cfg)
f = ctx.wave_fill()
rays = res.rays
print("%s: kernel %.2f ms (hist build %.2f)" % ({k: v for k, v in os.environ.items() if k.startswith("RT_HIP")}, fast.kernel_ms, res.kernel_ms))
print("  per ray: lane steps %.2f, lane exact %.2f; per 64 rays: wave steps %.1f, wave exact iters %.1f, wave shade passes %.2f; fill: traverse %.3f exact %.3f shade %.3f" % (
    f["lane_steps"] / rays, f["lane_exact_iters"] / rays, f["wave_steps"] * 64 / rays, f["wave_exact_iters"] * 64 / rays, f["wave_shade_passes"] * 64 / rays,
    f["traverse_fill"], f["exact_fill"], f["shade_fill"]))
