import sys
sys.path.insert(0, ".")
import raytrace_amd as R
ctx = R.Context(0)
book = R.Random_Scene(11, 0)
tris = R.Tessellate_Spheres(book, 0, 0)
ctx.upload_scene(triangles=tris, materials=book)
res = ctx.render(R.book_camera(1920, 1080), R.make_config(1920, 1080, 64, spp_chunk=R.RT_SPP_CHUNK_AUTO))
print("C4/64spp %.2f ms %d rays" % (res.kernel_ms, res.rays))
