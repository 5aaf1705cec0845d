"""Timing of the three BASELINE workloads C2/C3/C4 with the production settings (dev aid)."""
import os, sys
sys.path.insert(0, ".")
import raytrace_amd as R
ctx = R.Context(0)
book = R.Random_Scene(11, 0)
chunk = int(os.environ.get("TUNE_CHUNK", str(R.RT_SPP_CHUNK_AUTO)))
def run(name, cam, spp, reps=3):
    for _ in range(reps):
        res = ctx.render(cam, R.make_config(1920, 1080, spp, spp_chunk=chunk))
    print("%-4s %-32s %8.2f ms %8.0f Mrays/s (%d rays)" % (name, ctx.last_kernel_name(), res.kernel_ms, res.rays / res.kernel_ms / 1e3, res.rays), flush=True)
ctx.upload_scene(R.Two_Sphere_Scene()); run("C2", R.BasicCamera(), 100)
ctx.upload_scene(book); run("C3", R.book_camera(1920, 1080), 100)
tris = R.Tessellate_Spheres(book, 0, 0)
ctx.upload_scene(triangles=tris, materials=book); run("C4", R.book_camera(1920, 1080), 256, reps=2)
