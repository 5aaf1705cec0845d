"""A/B two builds of the library on C3, C2 and C4 (dev aid): python tools/gpu_ab.py [lib.so ...]; '-' = the in-tree build.
Each build runs in its own process (RT_HIP_LIB).  Beside the times: the ray total and CRC32s of a 4-spp C4 render's ray
counts and accumulation bytes -- builds that differ only in scheduling or node layout must print the same ones."""
import os, subprocess, sys
code = r'''
import sys, zlib
sys.path.insert(0, ".")
import numpy as np
import raytrace_amd as R
ctx = R.Context(0)
book = R.Random_Scene(11, 0)
cam = R.book_camera(1920, 1080)
only = sys.argv[1] if len(sys.argv) > 1 else ""
if "3" in only or not only:
    ctx.upload_scene(book)
    for _ in range(3):
        res = ctx.render(cam, R.make_config(1920, 1080, 100, spp_chunk=R.RT_SPP_CHUNK_AUTO))
    r3 = ctx.render(cam, R.make_config(1920, 1080, 4), want_raycount=True)
    print("C3 %8.2f ms %8.0f Mrays/s [crc rc %08x acc %08x]" % (res.kernel_ms, res.rays / res.kernel_ms / 1e3, zlib.crc32(r3.raycount.tobytes()), zlib.crc32(np.ascontiguousarray(r3.accum).tobytes())), end="   ")
if "2" in only or not only:
    ctx.upload_scene(R.Two_Sphere_Scene())
    for _ in range(3):
        res = ctx.render(R.BasicCamera(), R.make_config(1920, 1080, 100, spp_chunk=R.RT_SPP_CHUNK_AUTO))
    print("C2 %8.2f ms %8.0f Mrays/s" % (res.kernel_ms, res.rays / res.kernel_ms / 1e3), end="   ")
if "4" in only or not only:
    tris = R.Tessellate_Spheres(book, 0, 0)
    ctx.upload_scene(triangles=tris, materials=book)
    for _ in range(3):
        res = ctx.render(cam, R.make_config(1920, 1080, 64, spp_chunk=R.RT_SPP_CHUNK_AUTO))
    print("C4/64spp %8.2f ms %8.0f Mrays/s %s" % (res.kernel_ms, res.rays / res.kernel_ms / 1e3, ctx.scene_info()), end="   ")
    r4 = ctx.render(cam, R.make_config(1920, 1080, 4), want_raycount=True)
    print("rays %d crc rc %08x acc %08x" % (r4.rays, zlib.crc32(r4.raycount.tobytes()), zlib.crc32(np.ascontiguousarray(r4.accum).tobytes())))
'''
only = os.environ.get("RT_AB_ONLY", "")
for lib in (sys.argv[1:] or ["-"]):
    env = dict(os.environ)
    if lib != "-":
        env["RT_HIP_LIB"] = os.path.abspath(lib)
    out = subprocess.run([sys.executable, "-c", code, only], env=env, capture_output=True, text=True, timeout=400)
    print("%-40s %s" % (lib, out.stdout.strip() or out.stderr[-400:]), flush=True)
