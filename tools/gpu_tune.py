"""Sweep a tuning env var on C3/C4 (dev aid).  Usage: gpu_tune.py VAR v1 v2 ..."""
import os, sys, subprocess, json
var, vals = sys.argv[1], sys.argv[2:]
code = r'''
import os, sys
sys.path.insert(0, ".")
import raytrace_amd as R
ctx = R.Context(0)
book = R.Random_Scene(11, 0)
ctx.upload_scene(book)
cam = R.book_camera(1920, 1080)
chunk = int(os.environ.get("TUNE_CHUNK", "0"))
for _ in range(3):
    res = ctx.render(cam, R.make_config(1920, 1080, 100, spp_chunk=chunk))
print("C3 %8.2f ms %8.0f Mrays/s" % (res.kernel_ms, res.rays / res.kernel_ms / 1e3), end="  ")
if os.environ.get("TUNE_C4"):
    tris = R.Tessellate_Spheres(book, 0, 0)
    ctx.upload_scene(triangles=tris, materials=book)
    for _ in range(2):
        res = ctx.render(cam, R.make_config(1920, 1080, 32))
    print("C4 %8.2f ms %8.0f Mrays/s" % (res.kernel_ms, res.rays / res.kernel_ms / 1e3), end="")
print()
'''
for v in vals:
    env = dict(os.environ); env[var] = v
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    print("%s=%s: %s" % (var, v, out.stdout.strip() or out.stderr[-300:]), flush=True)
