"""Dev aid: host-side cost of scene upload (tessellation, BVH build, copies) for C3 and C4."""
import sys, time
sys.path.insert(0, ".")
import raytrace_amd as R
ctx = R.Context(0)
book = R.Random_Scene(11, 0)
t = time.perf_counter(); ctx.upload_scene(book); t1 = time.perf_counter() - t
t = time.perf_counter(); tris = R.Tessellate_Spheres(book, 0, 0); t2 = time.perf_counter() - t
t = time.perf_counter(); ctx.upload_scene(triangles=tris, materials=book); t3 = time.perf_counter() - t
print("upload 314 spheres %.2f ms; tessellate %d triangles %.1f ms; upload+BVH triangles %.1f ms %s" % (
    t1 * 1e3, len(tris), t2 * 1e3, t3 * 1e3, ctx.scene_info()))
