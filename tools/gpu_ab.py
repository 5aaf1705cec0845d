"""A/B two builds of the library on C3 and C4 (dev aid): python tools/gpu_ab.py [lib.so ...]; '-' = the in-tree build.
Each build runs in its own process (RT_HIP_LIB)."""
import os, subprocess, sys
code = r'''
import sys
sys.path.insert(0, ".")
import raytrace_amd as R
ctx = R.Context(0)
book = R.Random_Scene(11, 0)
cam = R.book_camera(1920, 1080)
ctx.upload_scene(book)
for _ in range(3):
    res = ctx.render(cam, R.make_config(1920, 1080, 100, spp_chunk=R.RT_SPP_CHUNK_AUTO))
print("C3 %8.2f ms %8.0f Mrays/s" % (res.kernel_ms, res.rays / res.kernel_ms / 1e3), end="   ")
ctx.upload_scene(R.Two_Sphere_Scene())
for _ in range(3):
    res = ctx.render(R.BasicCamera(), R.make_config(1920, 1080, 100, spp_chunk=R.RT_SPP_CHUNK_AUTO))
print("C2 %8.2f ms %8.0f Mrays/s" % (res.kernel_ms, res.rays / res.kernel_ms / 1e3), end="   ")
tris = R.Tessellate_Spheres(book, 0, 0)
ctx.upload_scene(triangles=tris, materials=book)
for _ in range(2):
    res = ctx.render(cam, R.make_config(1920, 1080, 64, spp_chunk=R.RT_SPP_CHUNK_AUTO))
print("C4/64spp %8.2f ms %8.0f Mrays/s %s" % (res.kernel_ms, res.rays / res.kernel_ms / 1e3, ctx.scene_info()))
'''
for lib in (sys.argv[1:] or ["-"]):
    env = dict(os.environ)
    if lib != "-":
        env["RT_HIP_LIB"] = os.path.abspath(lib)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=400)
    print("%-40s %s" % (lib, out.stdout.strip() or out.stderr[-400:]), flush=True)
