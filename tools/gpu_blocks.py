"""Dev aid: how often each block of the shading pass runs (wave level) and with how many lanes, C3 and C2."""
import sys
sys.path.insert(0, ".")
import raytrace_amd as R
ctx = R.Context(0)
for name, scene, cam in (("C3", R.Random_Scene(11, 0), R.book_camera(1920, 1080)), ("C2", R.Two_Sphere_Scene(), R.BasicCamera())):
    ctx.upload_scene(scene)
    ctx.set_flags(R.RT_FLAG_DEPTH_HISTOGRAM)
    res = ctx.render(cam, R.make_config(1920, 1080, 100, spp_chunk=R.RT_SPP_CHUNK_AUTO))
    ctx.set_flags(0)
    f = ctx.wave_fill()
    counts = ctx.block_counts()
    passes = f["wave_shade_passes"] or counts["philox"][0] or 1      # the brute-force kernel shades in every iteration
    print("%s %s: %d rays, %.2f ms (profiling build); wave shade passes %d" % (name, ctx.last_kernel_name(), res.rays, res.kernel_ms, f["wave_shade_passes"]))
    for k, (w, l) in counts.items():
        print("  %-24s runs in %5.1f %% of passes, %5.1f lanes per run, %6.2f runs per 64 rays" % (
            k, 100.0 * w / passes, l / w if w else 0.0, w * 64.0 / res.rays))
