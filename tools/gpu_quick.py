"""Quick GPU check: device math self-test + small renders against the oracle (dev aid)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import oracle_lib as O
import raytrace_amd as R

def to_oracle(s):
    a = np.zeros(len(s), dtype=O.SPHERE_DTYPE)
    a["center"][:, 0], a["center"][:, 1], a["center"][:, 2] = s.cx, s.cy, s.cz
    a["radius"], a["kind"], a["param"] = s.radius, s.kind, s.param
    a["albedo"][:, 0], a["albedo"][:, 1], a["albedo"][:, 2] = s.r, s.g, s.b
    return a

ctx = R.Context(0)
rng = np.random.default_rng(0)
n = 1 << 16
a = np.abs(rng.standard_normal(n).astype(np.float32)) * np.float32(10) ** rng.integers(-20, 20, n).astype(np.float32)
b = rng.standard_normal(n).astype(np.float32) * np.float32(10) ** rng.integers(-10, 10, n).astype(np.float32)
d, s, c, ph = ctx.selftest_math(a, b, (np.arange(n, dtype=np.uint64) * 2654435761 % (1 << 32)).astype(np.uint32))
print("div exact", np.array_equal(d.view(np.uint32), (a / b).view(np.uint32)))
print("sqrt exact", np.array_equal(s.view(np.uint32), np.sqrt(a).view(np.uint32)))
oc = np.array([O.lib().orc_cbrtf(float(x)) for x in a[:4096]], dtype=np.float32)
print("cbrt exact", np.array_equal(c[:4096].view(np.uint32), oc.view(np.uint32)))

for name, scene, cam, ocam, nx, ny, spp in [
        ("two", R.Two_Sphere_Scene(), R.BasicCamera(), O.camera_basic(), 64, 36, 8),
        ("book", R.Random_Scene(11, 0), R.book_camera(64, 36), O.book_camera(64, 36), 64, 36, 8)]:
    ctx.upload_scene(scene)
    cfg = R.make_config(nx, ny, spp)
    t = time.time(); res = ctx.render(cam, cfg, want_raycount=True); dt = time.time() - t
    w = O.World(spheres=to_oracle(scene))
    oa, o8, orc, cnt = O.render_counter(w, ocam, O.make_config(nx, ny, spp))
    print(name, "gpu %.3fs kernel %.3f ms rays %d (oracle %d)" % (dt, res.kernel_ms, res.rays, cnt["rays"]))
    print("  raycount equal:", np.array_equal(res.raycount, orc), " max|accum diff| = %.3g" % np.abs(res.accum - oa).max(),
          " rgb8 differing:", int((res.rgb8 != o8).sum()))
