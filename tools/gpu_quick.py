"""Quick GPU check (dev aid): small renders against the oracle for every kernel flavour + C3 timing."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import oracle_lib as O
import raytrace_amd as R
from helpers import spheres_to_oracle, triangles_to_oracle

ctx = R.Context(0)

def check(name, res, ref):
    oa, o8, orc, cnt = ref
    print("%-22s %-34s kernel %8.3f ms rays %d (oracle %d) raycount_eq=%s max|d|=%.3g rgb8diff=%d" % (
        name, ctx.last_kernel_name(), res.kernel_ms, res.rays, cnt["rays"], np.array_equal(res.raycount, orc),
        np.abs(res.accum - oa).max(), int((res.rgb8 != o8).sum())), flush=True)

nx, ny, spp = 96, 54, 8
book = R.Random_Scene(11, 0)
w = O.World(spheres=spheres_to_oracle(book))
ref = O.render_counter(w, O.book_camera(nx, ny), O.make_config(nx, ny, spp))
for flag, nm in ((R.RT_FLAG_FORCE_BRUTE_FORCE, "book brute"), (R.RT_FLAG_FORCE_BVH, "book bvh")):
    ctx.set_flags(flag)
    ctx.upload_scene(book)
    check(nm, ctx.render(R.book_camera(nx, ny), R.make_config(nx, ny, spp), want_raycount=True), ref)
print(ctx.scene_info())
two = R.Two_Sphere_Scene()
w2 = O.World(spheres=spheres_to_oracle(two))
ref2 = O.render_counter(w2, O.camera_basic(), O.make_config(nx, ny, spp))
for flag, nm in ((R.RT_FLAG_FORCE_BRUTE_FORCE, "two brute"), (R.RT_FLAG_FORCE_BVH, "two bvh")):
    ctx.set_flags(flag)
    ctx.upload_scene(two)
    check(nm, ctx.render(R.BasicCamera(), R.make_config(nx, ny, spp), want_raycount=True), ref2)
ctx.set_flags(0)
# triangles: small mixed-material mesh scene
src = R.Spheres(cx=[0, 0, -1, 1], cy=[0, -100.5, 0, 0], cz=[-1, -1, -1, -1], radius=[0.5, 100, 0.5, 0.5],
                kind=[0, 0, 2, 1], r=[.5, .5, .95, .8], g=[.5, .5, .95, .6], b=[.5, .5, .95, .2], param=[0, 0, 1.5, .15])
tris = R.Tessellate_Spheres(src, 12, 8)
ctx.upload_scene(triangles=tris, materials=src)
print(ctx.scene_info())
wt = O.World(triangles=triangles_to_oracle(tris), materials=spheres_to_oracle(src))
check("mesh 12x8", ctx.render(R.BasicCamera(), R.make_config(nx, ny, spp), want_raycount=True),
      O.render_counter(wt, O.camera_basic(), O.make_config(nx, ny, spp)))

# timings at full size
for nm, scene, cam, flags in (("C3 bvh", book, R.book_camera(1920, 1080), R.RT_FLAG_FORCE_BVH),
                              ("C3 brute", book, R.book_camera(1920, 1080), R.RT_FLAG_FORCE_BRUTE_FORCE),
                              ("C2 brute", two, R.BasicCamera(), R.RT_FLAG_FORCE_BRUTE_FORCE),
                              ("C2 bvh", two, R.BasicCamera(), R.RT_FLAG_FORCE_BVH)):
    ctx.set_flags(flags)
    ctx.upload_scene(scene)
    for _ in range(2):
        res = ctx.render(cam, R.make_config(1920, 1080, 100))
    print("%-10s %-34s %9.2f ms  %9.0f Mrays/s" % (nm, ctx.last_kernel_name(), res.kernel_ms, res.rays / res.kernel_ms / 1e3), flush=True)
ctx.set_flags(0)
