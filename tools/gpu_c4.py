"""C4 timing + sampled parity (dev aid)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import oracle_lib as O
import raytrace_amd as R
from helpers import spheres_to_oracle, triangles_to_oracle
ctx = R.Context(0)
book = R.Random_Scene(11, 0)
t = time.time(); tris = R.Tessellate_Spheres(book, 0, 0); print("tessellate", len(tris), time.time() - t)
t = time.time(); ctx.upload_scene(triangles=tris, materials=book); print("upload+bvh", time.time() - t, ctx.scene_info(), flush=True)
nx, ny = 1920, 1080
cam = R.book_camera(nx, ny)
for spp in (4, 32):
    res = ctx.render(cam, R.make_config(nx, ny, spp), want_raycount=True)
    print("C4 spp %d: %.1f ms, %.0f Mrays/s, rays %d" % (spp, res.kernel_ms, res.rays / res.kernel_ms / 1e3, res.rays), flush=True)
t = time.time(); w = O.World(triangles=triangles_to_oracle(tris), materials=spheres_to_oracle(book)); print("oracle world", time.time() - t)
rng = np.random.default_rng(1)
pix = np.unique(rng.integers(0, nx * ny, 600)).astype(np.uint32)
t = time.time(); oa, o8, orc, cnt = O.render_counter(w, O.book_camera(nx, ny), O.make_config(nx, ny, 32), pixels=pix); print("oracle", time.time() - t)
print("raycount eq", np.array_equal(res.raycount[pix], orc), "max err", np.abs(res.accum[pix] - oa).max(), "tri tests/ray", cnt["tri_tests"] / cnt["rays"])
R.write_ppm_image("gpurun_out/c4.ppm", res.rgb8, nx, ny)
