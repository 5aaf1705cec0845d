"""Sweep one rt_tuning knob on C3 (and C4 with --c4) (dev aid).
Usage: gpu_tune.py [--c4] KNOB v1 v2 ...   KNOB in queue_batch flush_lanes keep_traversing force_global_scene bvh_depth_cap"""
import sys
sys.path.insert(0, ".")
import raytrace_amd as R
args = sys.argv[1:]
c4 = "--c4" in args
args = [a for a in args if a != "--c4"]
knob, vals = args[0], [int(v) for v in args[1:]]
ctx = R.Context(0)
book = R.Random_Scene(11, 0)
cam = R.book_camera(1920, 1080)
tris = R.Tessellate_Spheres(book, 0, 0) if c4 else None
for v in vals:
    ctx.set_tuning(**{knob: v})
    ctx.upload_scene(book)
    for _ in range(3):
        res = ctx.render(cam, R.make_config(1920, 1080, 100, spp_chunk=R.RT_SPP_CHUNK_AUTO))
    line = "%s=%d: C3 %8.2f ms %8.0f Mrays/s" % (knob, v, res.kernel_ms, res.rays / res.kernel_ms / 1e3)
    if c4:
        ctx.upload_scene(triangles=tris, materials=book)
        for _ in range(2):
            res = ctx.render(cam, R.make_config(1920, 1080, 64, spp_chunk=R.RT_SPP_CHUNK_AUTO))
        line += "   C4/64spp %8.2f ms %8.0f Mrays/s" % (res.kernel_ms, res.rays / res.kernel_ms / 1e3)
    print(line, flush=True)
