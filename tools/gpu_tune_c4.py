"""Sweep a tuning env var on C4 (dev aid).  Usage: gpu_tune_c4.py VAR v1 v2 ..."""
import os, sys, subprocess
var, vals = sys.argv[1], sys.argv[2:]
code = r'''
import os, sys
sys.path.insert(0, ".")
import raytrace_amd as R
ctx = R.Context(0)
book = R.Random_Scene(11, 0)
tris = R.Tessellate_Spheres(book, 0, 0)
ctx.upload_scene(triangles=tris, materials=book)
cam = R.book_camera(1920, 1080)
for _ in range(2):
    res = ctx.render(cam, R.make_config(1920, 1080, 64, spp_chunk=R.RT_SPP_CHUNK_AUTO))
print("C4/64spp %8.2f ms %8.0f Mrays/s" % (res.kernel_ms, res.rays / res.kernel_ms / 1e3))
'''
for v in vals:
    env = dict(os.environ); env[var] = v
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    print("%s=%s: %s" % (var, v, out.stdout.strip() or out.stderr[-300:]), flush=True)
