import sys
sys.path.insert(0, ".")
import raytrace_amd as R
ctx = R.Context(0)
ctx.upload_scene(R.Random_Scene(11, 0))
res = ctx.render(R.book_camera(1920, 1080), R.make_config(1920, 1080, 100, spp_chunk=R.RT_SPP_CHUNK_AUTO))
print("C3 %.2f ms %d rays" % (res.kernel_ms, res.rays))
