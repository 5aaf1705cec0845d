"""Randomised parity sweep (GPU box, dev/evidence aid): random sphere scenes and tessellated meshes, cameras,
image sizes, sample ranges, chunk sizes, depth caps and seeds; every case compares per-pixel ray counts (exact)
and the linear mean (1e-3) with the oracle.  Prints one line per case and a summary; stops at the first failure.

    python tools/gpu_fuzz.py [n_cases] [seed] [first_case]      (cases before first_case are generated but not run)
"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import oracle_lib as O
import raytrace_amd as R
from helpers import spheres_to_oracle, triangles_to_oracle

def telephoto(frm, at, half_extent, nx, ny):
    """Far camera in BasicCamera.get_ray form (camera.zig:16): unit directions, frustum `half_extent` wide at `at`
    (MovableCamera's `- origin`, camera.zig:39, cancels every direction bit at these distances)."""
    w = at - frm; dist = np.linalg.norm(w); w = w / dist
    u = np.cross(w, [0.0, 1.0, 0.0]); u /= np.linalg.norm(u)
    v = np.cross(u, w)
    hw = half_extent / dist; hh = hw * ny / nx
    ll, hz, vt = w - hw * u - hh * v, 2 * hw * u, 2 * hh * v
    cam = R.BasicCamera()
    for k in range(3):
        cam._c.lower_left[k], cam._c.horizontal[k], cam._c.vertical[k], cam._c.origin[k] = ll[k], hz[k], vt[k], frm[k]
    cam._c.subtract_origin = 0
    return cam, O.Camera(O.v3(ll), O.v3(hz), O.v3(vt), O.v3(frm), 0)


n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
first_case = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx = R.Context(0)
t0 = time.time()
worst = 0.0
for case in range(n_cases):
    n = int(rng.choice([1, 2, 3, 5, 16, 17, 40, 150, 420, 700]))
    spread = float(rng.choice([1.5, 4.0, 12.0]))
    cx = rng.uniform(-spread, spread, n); cz = rng.uniform(-spread, spread, n) - 2.0
    rad = rng.choice([0.05, 0.2, 0.5, 1.0], n) * rng.uniform(0.5, 1.5, n)
    cy = rad + rng.uniform(-0.3, 0.6, n)
    kind = rng.integers(0, 3, n)
    rad = np.where((kind == 2) & (rng.random(n) < 0.3), -rad, rad)            # hollow glass (negative radius, Q2)
    param = np.where(kind == 2, rng.uniform(1.1, 2.5, n), rng.uniform(0.0, 0.5, n))
    if n > 2 and rng.random() < 0.5:                                            # a ground sphere
        cx[0], cy[0], cz[0], rad[0], kind[0] = 0.0, -1000.0, 0.0, 1000.0, 0
    if rng.random() < 0.15:                                                     # coincident centres / touching spheres
        cx[-1], cy[-1], cz[-1] = cx[0], cy[0], cz[0]
    s = R.Spheres(cx=cx, cy=cy, cz=cz, radius=rad, kind=kind, r=rng.random(n), g=rng.random(n), b=rng.random(n), param=param)
    nx, ny = int(rng.integers(1, 97)), int(rng.integers(1, 65))
    spp = int(rng.integers(1, 9))
    kw = dict(max_depth=int(rng.choice([0, 1, 3, 10, 16, 40])), seed=int(rng.integers(0, 2**63)),
              t_min=float(rng.choice([0.001, 0.0, 0.01])), row_offset_q3=int(rng.integers(0, 2)))
    b = int(rng.integers(0, spp)); e = int(rng.integers(b, spp + 1))
    kw.update(spp_begin=b, spp_end=e, spp_total=spp)
    chunk = int(rng.choice([0, 0, 1, 2, 3, R.RT_SPP_CHUNK_AUTO]))
    far = rng.random() < 0.15                # ray origins 10^3 .. 10^6 scene sizes away: the cone cull (RT_CONE_*)
    if far:
        dist = float(10.0 ** rng.uniform(3.0, 6.0)) * spread
        u = rng.standard_normal(3); u /= np.linalg.norm(u); u[1] = abs(u[1]) + 0.05
        cam, ocam = telephoto(u * dist + np.array([0.0, 0.0, -2.0]), np.array([0.0, 0.0, -2.0]), 1.5 * spread, nx, ny)
    elif rng.random() < 0.5:
        cam, ocam = R.BasicCamera(), O.camera_basic()
    else:
        frm = (float(rng.uniform(-6, 6)), float(rng.uniform(0.2, 4)), float(rng.uniform(2, 8)))
        at = (float(rng.uniform(-1, 1)), float(rng.uniform(0, 1)), -2.0)
        vfov, asp = float(rng.uniform(15, 80)), float(np.float32(nx) / np.float32(ny))
        cam = R.MovableCamera().SetPosition(frm, at, (0.0, 1.0, 0.0), vfov, asp)
        ocam = O.camera_set_position(frm, at, (0.0, 1.0, 0.0), vfov, asp)
    mesh = n <= 40 and rng.random() < 0.35
    flags = 0
    tuning = {}
    if mesh:
        slices, stacks = int(rng.integers(3, 24)), int(rng.integers(2, 16))
        tuning = dict(bvh_depth_cap=int(rng.choice([8, 12, 16, 24, 32])), node_format=int(rng.integers(0, 4)),
                      build_on_device=int(rng.random() < 0.25))
    else:
        flags = int(rng.choice([0, 0, R.RT_FLAG_FORCE_BRUTE_FORCE if n <= 150 else 0, R.RT_FLAG_FORCE_BVH]))
        tuning = dict(force_global_scene=int(rng.random() < 0.2))
    if case < first_case:
        continue
    if mesh:
        tris = R.Tessellate_Spheres(s, slices, stacks)
        ctx.set_tuning(**tuning)
        ctx.upload_scene(triangles=tris, materials=s)
        w = O.World(triangles=triangles_to_oracle(tris), materials=spheres_to_oracle(s))
        what = "mesh %6d tris" % len(tris)
    else:
        ctx.set_tuning(**tuning)
        ctx.set_flags(flags)
        ctx.upload_scene(s)
        w = O.World(spheres=spheres_to_oracle(s))
        what = "%4d spheres flags %d" % (n, flags)
    res = ctx.render(cam, R.make_config(nx, ny, spp, spp_chunk=chunk, **kw), want_raycount=True)
    ctx.set_flags(0)
    ochunk = 0 if chunk == R.RT_SPP_CHUNK_AUTO else chunk
    oa, o8, orc, cnt = O.render_counter(w, ocam, O.make_config(nx, ny, spp, spp_chunk=ochunk, **kw))
    # far cameras: the reference's arithmetic itself yields NaN radiance in some pixels (a hit normal that has lost
    # every bit); they must be the same pixels on both sides
    nan = np.isnan(oa)
    same_nan = np.array_equal(np.isnan(res.accum), nan) and (far or not nan.any())
    err = float(np.abs(res.accum - oa)[~nan].max()) if (~nan).any() else 0.0
    ok = res.rays == cnt["rays"] and np.array_equal(res.raycount, orc) and err <= 1e-3 and same_nan and not np.isinf(res.accum).any()
    worst = max(worst, err)
    print("case %3d %-24s %-30s %3dx%-3d spp [%d,%d)/%d chunk %s depth %2d tmin %.3g: rays %8d err %.2g %s" % (
        case, what, ctx.last_kernel_name(), nx, ny, b, e, spp, "auto" if chunk == R.RT_SPP_CHUNK_AUTO else chunk, kw["max_depth"], kw["t_min"],
        res.rays, err, "ok" if ok else "FAIL"), flush=True)
    if not ok:
        bad = np.flatnonzero(res.raycount != orc)
        print("  mismatching pixels:", bad[:10], res.raycount[bad[:10]], orc[bad[:10]])
        d = np.abs(np.nan_to_num(res.accum) - np.nan_to_num(oa)).max(axis=1)
        wp = np.argsort(d)[::-1][:5]
        print("  tuning", tuning, "flags", flags, "scene_info", ctx.scene_info(), "far", far)
        print("  worst pixels", wp, "gpu", res.accum[wp].tolist(), "oracle", oa[wp].tolist(), "raycounts", res.raycount[wp], orc[wp])
        if mesh:                                   # the checker's plain loop over all triangles; which tree variant deviates?
            ba, b8, brc, bcnt = O.render_counter(w, ocam, O.make_config(nx, ny, spp, spp_chunk=ochunk, traversal=O.TRAV_BRUTE, **kw))
            print("   oracle plain loop: vs oracle BVH err", float(np.abs(np.nan_to_num(ba) - np.nan_to_num(oa)).max()),
                  "; vs GPU err", float(np.abs(np.nan_to_num(ba) - np.nan_to_num(res.accum)).max()),
                  "raycounts equal GPU", np.array_equal(brc, res.raycount), "pixel", ba[wp[0]].tolist())
            for alt in (dict(node_format=1), dict(node_format=2), dict(), dict(build_on_device=1), dict(node_format=1, bvh_depth_cap=32)):
                ctx.set_tuning(**alt); ctx.upload_scene(triangles=tris, materials=s)
                r2 = ctx.render(cam, R.make_config(nx, ny, spp, spp_chunk=chunk, **kw), want_raycount=True)
                print("   variant", alt, "err", float(np.abs(np.nan_to_num(r2.accum) - np.nan_to_num(oa)).max()),
                      "raycounts equal", np.array_equal(r2.raycount, orc), "pixel", r2.accum[wp[0]].tolist())
        np.savez(os.path.join(ROOT, "gpurun_out", "fuzz_fail.npz"), cx=cx, cy=cy, cz=cz, rad=rad, kind=kind, param=param)
        sys.exit(1)
print("fuzz: %d cases ok, worst |err| %.3g, %.0f s" % (n_cases, worst, time.time() - t0))
