"""GPU parity tests (run with -m gpu on an MI355X): the HIP trace path, called through the C ABI,
against the CPU oracle on the same seeded inputs and against the committed golden vectors.

Bars (north_star / BASELINE.md):
  * integer work bit-exact: pixel indexing, sample assignment, per-pixel ray counts (every branch
    decision of every path), RNG streams;
  * linear accumulation: max |gpu - oracle| <= 1e-3 per pixel channel (TOL_ACCUM below; measured
    values are ~1e-7 because only the attenuation-product order differs);
  * final u8 image: PSNR >= 60 dB.
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
import raytrace_amd as R
from raytrace_amd import _lib
from helpers import psnr_u8, spheres_to_oracle, triangles_to_oracle

pytestmark = pytest.mark.gpu

TOL_ACCUM = 1e-3     # north_star: per-pixel max abs error on the linear accumulation buffer
MIN_PSNR = 60.0      # north_star: PSNR on the final image
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_golden.npz"))


def _scene(name, nx, ny):
    if name == "book":
        s = R.Random_Scene(11, 0)
        return s, R.book_camera(nx, ny), O.book_camera(nx, ny)
    s = R.Two_Sphere_Scene()
    return s, R.BasicCamera(), O.camera_basic()


def _compare(res, oa, o8, orc, cnt):
    assert res.rays == cnt["rays"] and res.primary_rays == cnt["primary_rays"]
    assert np.array_equal(res.raycount, orc), "per-pixel ray counts differ: some branch decision diverged"
    err = float(np.abs(res.accum - oa).max())
    assert err <= TOL_ACCUM, err
    assert psnr_u8(res.rgb8, o8) >= MIN_PSNR
    assert np.all(np.isfinite(res.accum)) and np.all(res.accum >= 0)      # check_nan3 / check_neg3
    return err


FLAVOURS = {"brute": R.RT_FLAG_FORCE_BRUTE_FORCE, "bvh": R.RT_FLAG_FORCE_BVH, "auto": 0}


def _both(ctx, name, nx, ny, spp, flavour="auto", **kw):
    s, cam, ocam = _scene(name, nx, ny)
    ctx.set_flags(FLAVOURS[flavour])
    ctx.upload_scene(s)
    okw = dict(kw)
    tile = okw.pop("tile", None)
    try:
        res = ctx.render(cam, R.make_config(nx, ny, spp, tile=tile, **kw), want_raycount=True)
    finally:
        ctx.set_flags(0)
    if flavour != "auto":
        assert flavour in ctx.last_kernel_name()
    w = O.World(spheres=spheres_to_oracle(s))
    ref = O.render_counter(w, ocam, O.make_config(nx, ny, spp, **okw))
    return res, ref


# ---- arithmetic primitives on the device, bit for bit -------------------------------------------------
def test_device_math_is_bit_exact(gpu_ctx):
    rng = np.random.default_rng(0)
    n = 1 << 16
    a = np.abs(rng.standard_normal(n).astype(np.float32)) * np.float32(10.0) ** rng.integers(-30, 30, n).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32) * np.float32(10.0) ** rng.integers(-12, 12, n).astype(np.float32)
    a[:8] = [0.0, 1.0, 4.0, 1e-40, 3.4e38, 2.0, 1e-45, 0.5]      # zeros, denormals, extremes
    ctr = (np.arange(n, dtype=np.uint64) * 2654435761 % (1 << 32)).astype(np.uint32)
    d, s, c, ph = gpu_ctx.selftest_math(a, b, ctr)
    with np.errstate(all="ignore"):
        assert np.array_equal(d.view(np.uint32), (a / b).view(np.uint32)), "f32 divide is not IEEE-exact"
        assert np.array_equal(s.view(np.uint32), np.sqrt(a).view(np.uint32)), "f32 sqrt is not IEEE-exact"
    L = O.lib()
    oc = np.array([L.orc_cbrtf(float(x)) for x in a[:8192]], dtype=np.float32)
    assert np.array_equal(c[:8192].view(np.uint32), oc.view(np.uint32)), "cbrt differs from the oracle"
    key = (C.c_uint32 * 2)(0x12345678, 0x9abcdef0)
    out = (C.c_uint32 * 4)()
    for k in list(range(64)) + [n - 1]:
        L.orc_philox4x32_10((C.c_uint32 * 4)(int(ctr[k]), k, 7, 0), key, out)
        assert list(out) == ph[k].tolist()


def test_cbrt_of_the_kernels_matches_the_oracle_on_every_possible_input(gpu_ctx):
    """random_in_unit_sphere's cbrt (raytracer.zig:577) only ever sees k * 2^-24 (u32_to_unit_float).  The trace
    kernels evaluate it with an f32 fast path and fall back to musl's f64 steps near rounding midpoints; this
    compares that production evaluation with the oracle's musl cbrtf on ALL 2^24 inputs, bit for bit."""
    got, n_risky = gpu_ctx.selftest_cbrt_grid()
    x = np.arange(1 << 24, dtype=np.float32) * np.float32(2.0 ** -24)
    exp = np.empty_like(x)
    O.lib().orc_cbrtf_array(x.ctypes.data, exp.ctypes.data, x.size)
    bad = np.flatnonzero(got.view(np.uint32) != exp.view(np.uint32))
    assert bad.size == 0, (bad.size, bad[:8], got[bad[:8]], exp[bad[:8]])
    assert 0 < n_risky < (1 << 24) // 1024, n_risky        # the fallback is rare (about 1 input in 4096) but live


# ---- golden vectors ----------------------------------------------------------------------------------
def _mesh_source():
    """The golden 'mesh' scene (tests/golden/make_golden.py): four spheres tessellated 12x8."""
    return R.Spheres(cx=[0, 0, -1, 1], cy=[0, -100.5, 0, 0], cz=[-1, -1, -1, -1], radius=[0.5, 100, 0.5, 0.5],
                     kind=[0, 0, 2, 1], r=[.5, .5, .95, .8], g=[.5, .5, .95, .6], b=[.5, .5, .95, .2],
                     param=[0, 0, 1.5, .15])


@pytest.mark.parametrize("name,flavour", [("book", "brute"), ("book", "bvh"), ("two", "brute"), ("two", "bvh"),
                                          ("mesh", "auto")])
@pytest.mark.parametrize("depth", [16, 10])
def test_golden_vectors(gpu_ctx, name, flavour, depth):
    gpu_ctx.set_flags(FLAVOURS[flavour])
    if name == "mesh":
        src = _mesh_source()
        gpu_ctx.upload_scene(triangles=R.Tessellate_Spheres(src, 12, 8), materials=src)
        cam = R.BasicCamera()
    else:
        s, cam, _ = _scene(name, 32, 18)
        gpu_ctx.upload_scene(s)
    try:
        res = gpu_ctx.render(cam, R.make_config(32, 18, 4, max_depth=depth, seed=1), want_raycount=True)
    finally:
        gpu_ctx.set_flags(0)
    key = "%s_d%d" % (name, depth)
    assert np.array_equal(res.raycount, GOLD[key + "_raycount"].astype(np.uint32))
    assert res.rays == int(GOLD[key + "_rays"])
    assert float(np.abs(res.accum - GOLD[key + "_accum"]).max()) <= TOL_ACCUM
    assert psnr_u8(res.rgb8, GOLD[key + "_rgb8"]) >= MIN_PSNR


# ---- BASELINE configs --------------------------------------------------------------------------------
@pytest.mark.parametrize("flavour", ["brute", "bvh"])
def test_c1_book_400x225_10spp_depth10(gpu_ctx, flavour):
    """configs[0]: the CPU-runnable case, whole image against the oracle."""
    res, (oa, o8, orc, cnt) = _both(gpu_ctx, "book", 400, 225, 10, flavour=flavour, max_depth=10)
    err = _compare(res, oa, o8, orc, cnt)
    print("C1 max|accum err| = %.3g, PSNR = %s" % (err, psnr_u8(res.rgb8, o8)))


@pytest.mark.parametrize("flavour", ["brute", "bvh"])
def test_c2_two_sphere_reduced(gpu_ctx, flavour):
    res, (oa, o8, orc, cnt) = _both(gpu_ctx, "two", 320, 180, 16, flavour=flavour)
    _compare(res, oa, o8, orc, cnt)


def test_lds_and_global_scene_variants_agree(gpu_ctx):
    """The book scene runs from LDS by default; rt_tuning.force_global_scene forces the global-memory variant.
    The other knobs (queue batch, flush and keep-traversing thresholds) must not change a bit either."""
    a, _ = _both(gpu_ctx, "book", 160, 90, 6, flavour="bvh")
    try:
        gpu_ctx.set_tuning(force_global_scene=1)
        b, _ = _both(gpu_ctx, "book", 160, 90, 6, flavour="bvh")
        gpu_ctx.set_tuning(queue_batch=256, flush_lanes=65, keep_traversing=40)
        c, _ = _both(gpu_ctx, "book", 160, 90, 6, flavour="bvh")
        with pytest.raises(R.RtError):
            gpu_ctx.set_tuning(queue_batch=4096)              # reservations past the queue's end would not fit 32 bits
    finally:
        gpu_ctx.set_tuning()
    assert np.array_equal(a.raycount, c.raycount)
    assert np.array_equal(a.accum.view(np.uint32), c.accum.view(np.uint32))
    assert np.array_equal(a.raycount, b.raycount)
    assert np.array_equal(a.accum.view(np.uint32), b.accum.view(np.uint32))


def test_largest_lds_resident_sphere_scene(gpu_ctx):
    """Random_Scene(14): 514 spheres, 49 KB of nodes and sphere records -- inside the 55 KB that sphere scenes may take of
    a workgroup's LDS share (more than the 64 KB a kernel gets by default together with the stacks: the launch raises
    the limit).  Against the oracle, and bit for bit against the global-memory variant."""
    nx, ny, spp = 128, 72, 4
    s = R.Random_Scene(14, 0)
    assert 40 * 1024 < (len(s) - 1) * 80 + len(s) * 16 <= 55 * 1024
    cam, ocam = R.book_camera(nx, ny), O.book_camera(nx, ny)
    gpu_ctx.set_flags(0)
    gpu_ctx.upload_scene(s)
    a = gpu_ctx.render(cam, R.make_config(nx, ny, spp), want_raycount=True)
    oa, o8, orc, cnt = O.render_counter(O.World(spheres=spheres_to_oracle(s)), ocam, O.make_config(nx, ny, spp))
    _compare(a, oa, o8, orc, cnt)
    try:
        gpu_ctx.set_tuning(force_global_scene=1)
        b = gpu_ctx.render(cam, R.make_config(nx, ny, spp), want_raycount=True)
    finally:
        gpu_ctx.set_tuning()
    assert np.array_equal(a.raycount, b.raycount) and np.array_equal(a.accum.view(np.uint32), b.accum.view(np.uint32))


def test_kernel_flavours_are_bit_identical(gpu_ctx):
    """The BVH only culls; the exact tests and the shading are the brute-force kernel's, so the two
    flavours must produce the same bits."""
    a, _ = _both(gpu_ctx, "book", 160, 90, 6, flavour="brute")
    b, _ = _both(gpu_ctx, "book", 160, 90, 6, flavour="bvh")
    assert np.array_equal(a.raycount, b.raycount)
    assert np.array_equal(a.accum.view(np.uint32), b.accum.view(np.uint32))


def test_c4_triangle_mesh_scene_1080p_256spp_sampled_pixels(gpu_ctx):
    """configs[3]: the book scene's spheres tessellated as the reference's raster preview does
    (main.zig:128-138: 150x150 / 30x30 / 20x20), 1920x1080, 256 spp; oracle on a random pixel subset."""
    nx, ny, spp = 1920, 1080, 256
    book = R.Random_Scene(11, 0)
    tris = R.Tessellate_Spheres(book, 0, 0)
    assert len(tris) == 285520
    gpu_ctx.upload_scene(triangles=tris, materials=book)
    res = gpu_ctx.render(R.book_camera(nx, ny), R.make_config(nx, ny, spp), want_raycount=True)
    w = O.World(triangles=triangles_to_oracle(tris), materials=spheres_to_oracle(book))
    rng = np.random.default_rng(4)
    pix = np.unique(rng.integers(0, nx * ny, 800)).astype(np.uint32)
    oa, o8, orc, _ = O.render_counter(w, O.book_camera(nx, ny), O.make_config(nx, ny, spp), pixels=pix)
    assert np.array_equal(res.raycount[pix], orc)
    err = float(np.abs(res.accum[pix] - oa).max())
    assert err <= TOL_ACCUM, err
    assert psnr_u8(res.rgb8[pix], o8) >= MIN_PSNR
    assert res.primary_rays == nx * ny * spp and res.rays == int(res.raycount.astype(np.uint64).sum())
    assert np.all(np.isfinite(res.accum)) and res.accum.min() >= 0
    # the 800 pixels above are checked against the oracle's own cone-culled triangle BVH -- a second
    # cull by the same hand.  24 of them once more against the DEFINITION: the plain loop over all 285 520 triangles for
    # every ray (ORC_TRAV_BRUTE: no box, no cone, no cull on the checker's side; ~2e8 exact tests per pixel).
    sub = pix[:: max(1, len(pix) // 24)][:24]
    ba, b8, brc, bcnt = O.render_counter(w, O.book_camera(nx, ny),
                                         O.make_config(nx, ny, spp, traversal=O.TRAV_BRUTE, threads=16), pixels=sub)
    assert bcnt["tri_tests"] == bcnt["rays"] * len(tris)                     # really every triangle for every ray
    assert np.array_equal(res.raycount[sub], brc), "C4 differs from the cull-free definition"
    berr = float(np.abs(res.accum[sub] - ba).max())
    assert berr <= TOL_ACCUM, berr
    idx = np.searchsorted(pix, sub)
    assert np.array_equal(orc[idx], brc) and np.array_equal(oa[idx].view(np.uint32), ba.view(np.uint32))   # and so does the checker's BVH
    print("C4 vs the plain loop over every triangle: %d pixels, %d rays, %.3g exact tests, max err %.3g"
          % (len(sub), bcnt["rays"], bcnt["tri_tests"], berr))
    print("C4 1080p/256spp: %d triangles, %d rays, kernel %.1f ms = %.0f Mrays/s, sampled max err %.3g"
          % (len(tris), res.rays, res.kernel_ms, res.rays / res.kernel_ms / 1e3, err))


def test_triangle_edge_cases(gpu_ctx):
    """One triangle; a degenerate (zero-area) triangle beside real ones; a shared-edge hit (equal t on two
    triangles: the lower index wins, as in list order)."""
    mats = R.Spheres(cx=[0, 0], cy=[0, 0], cz=[0, 0], radius=[1, 1], kind=[0, 1], r=[.8, .9], g=[.3, .9], b=[.3, .9],
                     param=[0, 0.0])
    cases = [
        R.Triangles([[-1, -1, -2]], [[1, -1, -2]], [[0, 1, -2]], [0]),
        R.Triangles([[-1, -1, -2], [0, 0, -1.5], [-3, -1, -3]], [[1, -1, -2], [0, 0, -1.5], [3, -1, -3]],
                    [[0, 1, -2], [0, 0, -1.5], [0, 2, -3]], [0, 1, 1]),
        # a quad split along its diagonal, camera looking exactly at the diagonal for some pixels
        R.Triangles([[-1, -1, -2], [-1, -1, -2]], [[1, -1, -2], [1, 1, -2]], [[1, 1, -2], [-1, 1, -2]], [0, 1]),
    ]
    for tris in cases:
        gpu_ctx.upload_scene(triangles=tris, materials=mats)
        res = gpu_ctx.render(R.BasicCamera(), R.make_config(64, 32, 6), want_raycount=True)
        w = O.World(triangles=triangles_to_oracle(tris), materials=spheres_to_oracle(mats))
        oa, o8, orc, cnt = O.render_counter(w, O.camera_basic(), O.make_config(64, 32, 6))
        _compare(res, oa, o8, orc, cnt)


def test_needle_triangle_in_all_three_vertex_rotations(gpu_ctx):
    """the grazing guard of the triangle test bounds sin(corner at v0) * cos(theta); rt_upload_scene (like the
    oracle's worlds) rotates every triangle so that v0 is the largest corner, so a needle is visible whichever vertex the
    caller lists first, and the three listings give the same bits.  Also a sliver mesh: a fan of 0.23-degree needles."""
    mats = R.Spheres(cx=[0], cy=[0], cz=[0], radius=[1], kind=[0], r=[.5], g=[.5], b=[.5], param=[0])
    apex, b1, b2 = [0.0, 0.0, -2.0], [-0.002, 1.0, -2.0], [0.002, 1.0, -2.0]
    cam = R.MovableCamera().SetPosition((0.0, 0.5, 0.0), (0.0, 0.5, -2.0), (0, 1, 0), 1.0, 1.0)
    ocam = O.camera_set_position((0.0, 0.5, 0.0), (0.0, 0.5, -2.0), (0, 1, 0), 1.0, 1.0)
    out = []
    for v in ((apex, b1, b2), (b1, b2, apex), (b2, apex, b1)):
        tris = R.Triangles([v[0]], [v[1]], [v[2]], [0])
        gpu_ctx.upload_scene(triangles=tris, materials=mats)
        res = gpu_ctx.render(cam, R.make_config(48, 48, 2), want_raycount=True)
        w = O.World(triangles=triangles_to_oracle(tris), materials=spheres_to_oracle(mats))
        oa, o8, orc, cnt = O.render_counter(w, ocam, O.make_config(48, 48, 2))
        _compare(res, oa, o8, orc, cnt)
        assert cnt["hits"] > 20                                  # the needle is seen
        out.append(res)
    for r in out[1:]:
        assert np.array_equal(r.raycount, out[0].raycount) and np.array_equal(r.accum.view(np.uint32), out[0].accum.view(np.uint32))
    # a fan of needles around a hub, every triangle listed apex first (the order that used to make them invisible)
    n = 64
    ang = np.linspace(0, 2 * np.pi, n, endpoint=False)
    rim0 = np.stack([np.cos(ang), np.sin(ang), np.full(n, -3.0)], 1)
    rim1 = np.stack([np.cos(ang + 0.004), np.sin(ang + 0.004), np.full(n, -3.0)], 1)
    hub = np.tile([[0.0, 0.0, -3.0]], (n, 1))
    fan = R.Triangles(hub, rim0, rim1, np.zeros(n, np.uint32))
    gpu_ctx.upload_scene(triangles=fan, materials=mats)
    res = gpu_ctx.render(R.BasicCamera(), R.make_config(160, 80, 2), want_raycount=True)
    w = O.World(triangles=triangles_to_oracle(fan), materials=spheres_to_oracle(mats))
    oa, o8, orc, cnt = O.render_counter(w, O.camera_basic(), O.make_config(160, 80, 2))
    _compare(res, oa, o8, orc, cnt)
    assert cnt["hits"] > 50


def test_large_random_sphere_scenes(gpu_ctx):
    """Random_Scene at other sizes/seeds (deeper BVHs, other hollow-glass placements)."""
    for half, seed in ((3, 5), (20, 7)):
        s = R.Random_Scene(half, seed)
        gpu_ctx.upload_scene(s)
        cam, ocam = R.book_camera(120, 68), O.book_camera(120, 68)
        res = gpu_ctx.render(cam, R.make_config(120, 68, 4), want_raycount=True)
        w = O.World(spheres=spheres_to_oracle(s))
        oa, o8, orc, cnt = O.render_counter(w, ocam, O.make_config(120, 68, 4))
        _compare(res, oa, o8, orc, cnt)


@pytest.mark.parametrize("cap", ["12", "16", "24", "32"])
def test_bvh_depth_cap_variants_of_the_triangle_kernel(gpu_ctx, cap):
    """The triangle kernel has one instantiation per stack size (depth <= 16 / 24 / 32) and the builder's depth
    cap decides which runs; a cap below the SAH depth also exercises its median-split fallback.  A mesh of
    27 000 triangles, same image whatever the tree."""
    src = R.Spheres(cx=[0, 0, -1.1, 1.1, 0.3], cy=[0, -100.5, 0, 0, 0.9], cz=[-1, -1, -1, -1, -1.2],
                    radius=[0.5, 100, 0.5, 0.5, 0.3], kind=[0, 0, 2, 1, 0], r=[.5, .5, .95, .8, .2], g=[.5, .5, .95, .6, .7],
                    b=[.5, .5, .95, .2, .3], param=[0, 0, 1.5, .15, 0])
    tris = R.Tessellate_Spheres(src, 60, 45)
    try:
        gpu_ctx.set_tuning(bvh_depth_cap=int(cap))
        gpu_ctx.upload_scene(triangles=tris, materials=src)
    finally:
        gpu_ctx.set_tuning()
    info = gpu_ctx.scene_info()
    min_depth = int(np.ceil(np.log2(len(tris))))             # 15: a cap below it is raised to what the mesh needs
    assert info["n_triangles"] == len(tris) and info["bvh_depth"] <= max(int(cap), min_depth)
    nx, ny, spp = 96, 54, 4
    res = gpu_ctx.render(R.BasicCamera(), R.make_config(nx, ny, spp), want_raycount=True)
    w = O.World(triangles=triangles_to_oracle(tris), materials=spheres_to_oracle(src))
    oa, o8, orc, cnt = O.render_counter(w, O.camera_basic(), O.make_config(nx, ny, spp))
    _compare(res, oa, o8, orc, cnt)


def test_tree_built_on_the_device_gives_the_same_image(gpu_ctx):
    """rt_tuning.build_on_device: Morton sort + top-down splits made by rt_build.hip (raytracer.zig:279-376).
    The closest hit does not depend on the tree, so the image and the ray counts equal the host-built tree's bit for
    bit -- and the oracle's list traversal.  Sphere counts from two up; the mesh is 27 000 triangles; a depth cap at
    the minimum, ceil(log2 n), forces the middle splits."""
    rng = np.random.default_rng(5)
    for n in (2, 3, 4, 5, 7, 8, 9, 31, 32, 33, 100, 485, 700):
        s = R.Spheres(cx=rng.uniform(-4, 4, n), cy=rng.uniform(-1, 2, n), cz=rng.uniform(-6, -2, n),
                      radius=rng.uniform(0.05, 0.5, n), kind=rng.integers(0, 3, n), r=rng.uniform(0.2, 1, n),
                      g=rng.uniform(0.2, 1, n), b=rng.uniform(0.2, 1, n), param=rng.uniform(0.0, 0.3, n) + (rng.integers(0, 2, n) * 1.5))
        cfg = R.make_config(64, 36, 2)
        gpu_ctx.upload_scene(s)
        host = gpu_ctx.render(R.BasicCamera(), cfg, want_raycount=True)
        try:
            gpu_ctx.set_tuning(build_on_device=1)
            gpu_ctx.upload_scene(s)
        finally:
            gpu_ctx.set_tuning()
        info = gpu_ctx.scene_info()
        assert info["n_bvh_nodes"] == n - 1 and int(np.ceil(np.log2(n))) <= info["bvh_depth"] <= 32, (n, info)
        assert gpu_ctx.upload_ms()["device_build"] > 0
        dev = gpu_ctx.render(R.BasicCamera(), cfg, want_raycount=True)
        assert np.array_equal(dev.raycount, host.raycount), n
        assert np.array_equal(dev.accum, host.accum), n
    src = R.Spheres(cx=[0, 0, -1.1, 1.1, 0.3], cy=[0, -100.5, 0, 0, 0.9], cz=[-1, -1, -1, -1, -1.2],
                    radius=[0.5, 100, 0.5, 0.5, 0.3], kind=[0, 0, 2, 1, 0], r=[.5, .5, .95, .8, .2], g=[.5, .5, .95, .6, .7],
                    b=[.5, .5, .95, .2, .3], param=[0, 0, 1.5, .15, 0])
    tris = R.Tessellate_Spheres(src, 60, 45)
    try:
        gpu_ctx.set_tuning(build_on_device=1, bvh_depth_cap=15)
        gpu_ctx.upload_scene(triangles=tris, materials=src)
        assert gpu_ctx.scene_info()["bvh_depth"] == 15
        capped = gpu_ctx.render(R.BasicCamera(), R.make_config(96, 54, 4), want_raycount=True)
        gpu_ctx.set_tuning(build_on_device=1)
        gpu_ctx.upload_scene(triangles=tris, materials=src)
    finally:
        gpu_ctx.set_tuning()
    info = gpu_ctx.scene_info()
    assert info["n_bvh_nodes"] == len(tris) - 1 and int(np.ceil(np.log2(len(tris)))) <= info["bvh_depth"] <= 32
    nx, ny, spp = 96, 54, 4
    res = gpu_ctx.render(R.BasicCamera(), R.make_config(nx, ny, spp), want_raycount=True)
    w = O.World(triangles=triangles_to_oracle(tris), materials=spheres_to_oracle(src))
    oa, o8, orc, cnt = O.render_counter(w, O.camera_basic(), O.make_config(nx, ny, spp))
    _compare(res, oa, o8, orc, cnt)
    assert np.array_equal(capped.raycount, res.raycount) and np.array_equal(capped.accum, res.accum)


def test_packed_tree_nodes_give_the_same_image(gpu_ctx):
    """Large triangle scenes get 72-byte nodes whose child refs ride in the low mantissa byte of the planes (one load
    fewer per traversal step; rt_kernels.h RT_PACKED_NODE_BYTES); and, by default, four-wide 144-byte nodes over two binary levels; rt_tuning.node_format selects.
    The planes only move outward, so the image is the same bit for bit -- and equal to the oracle's.  The mesh sits
    far from the origin on one axis and straddles zero on another so that the rounding of negative, tiny and large
    plane values is exercised."""
    src = R.Spheres(cx=[300.0, 300.0, 298.9, 301.1, 300.3], cy=[0, -100.5, 0, 0, 0.9], cz=[-1, -1, -1, -1, -1.2],
                    radius=[0.5, 100, 0.5, 0.5, 0.3], kind=[0, 0, 2, 1, 0], r=[.5, .5, .95, .8, .2], g=[.5, .5, .95, .6, .7],
                    b=[.5, .5, .95, .2, .3], param=[0, 0, 1.5, .15, 0])
    tris = R.Tessellate_Spheres(src, 60, 45)
    frm, at = (303.0, 1.5, 3.0), (300.0, 0.0, -1.0)
    cam = R.MovableCamera().SetPosition(frm, at, (0.0, 1.0, 0.0), 40.0, 96 / 54)
    ocam = O.camera_set_position(frm, at, (0.0, 1.0, 0.0), 40.0, 96 / 54)
    cfg = R.make_config(96, 54, 4)
    images = []
    for fmt in (_lib.RT_NODES_BINARY80, _lib.RT_NODES_BINARY72, _lib.RT_NODES_WIDE, 0):
        try:
            gpu_ctx.set_tuning(node_format=fmt)
            gpu_ctx.upload_scene(triangles=tris, materials=src)
        finally:
            gpu_ctx.set_tuning()
        images.append(gpu_ctx.render(cam, cfg, want_raycount=True))
    assert images[0].rays > 96 * 54 * 4 * 1.5                 # the camera does see the mesh
    for other in images[1:]:
        assert np.array_equal(images[0].raycount, other.raycount)
        assert np.array_equal(images[0].accum, other.accum)
    w = O.World(triangles=triangles_to_oracle(tris), materials=spheres_to_oracle(src))
    oa, o8, orc, cnt = O.render_counter(w, ocam, O.make_config(96, 54, 4))
    _compare(images[3], oa, o8, orc, cnt)


def test_device_build_refuses_geometry_that_is_not_finite(gpu_ctx):
    """The host builder's validation rejects NaN/inf geometry (RT_E_STATE); so does the device builder."""
    src = R.Spheres(cx=[0, 0], cy=[0, -100.5], cz=[-1, -1], radius=[0.5, 100], kind=[0, 0], r=[.5, .5], g=[.5, .5], b=[.5, .5],
                    param=[0, 0])
    tris = R.Tessellate_Spheres(src, 8, 6)
    tris.v1[5, 1] = np.nan
    try:
        gpu_ctx.set_tuning(build_on_device=1)
        with pytest.raises(R.RtError) as e:
            gpu_ctx.upload_scene(triangles=tris, materials=src)
        assert e.value.code == -4                        # RT_E_STATE
    finally:
        gpu_ctx.set_tuning()
    gpu_ctx.upload_scene(R.Two_Sphere_Scene())          # the ctx is still usable


def test_sphere_scene_too_big_for_lds_and_radii_without_a_fast_reciprocal(gpu_ctx):
    """(a) Random_Scene(14): ~500 spheres do not fit the LDS-staged variant, the global-memory BVH kernel runs.
    (b) radii whose reciprocal the host refuses to precompute (all-ones significand, far outside 2^+-60) and a
    negative radius: the hit normal then takes the IEEE divide; results must not care."""
    s = R.Random_Scene(14, 3)
    assert len(s) > 450
    gpu_ctx.upload_scene(s)
    cam, ocam = R.book_camera(120, 68), O.book_camera(120, 68)
    res = gpu_ctx.render(cam, R.make_config(120, 68, 4), want_raycount=True)
    w = O.World(spheres=spheres_to_oracle(s))
    _compare(res, *O.render_counter(w, ocam, O.make_config(120, 68, 4)))
    allones = float(np.frombuffer(np.uint32(0x3EFFFFFF).tobytes(), np.float32)[0])      # 0.49999997
    odd = R.Spheres(cx=[0, 0, -1.2, 1.2], cy=[0, -100.5, 0, 0], cz=[-1, -1, -1, -1], radius=[allones, 100, -0.45, 0.5],
                    kind=[0, 0, 2, 1], r=[.5, .5, .95, .8], g=[.5, .5, .95, .6], b=[.5, .5, .95, .2], param=[0, 0, 1.5, .1])
    for flavour in ("brute", "bvh"):
        gpu_ctx.set_flags(FLAVOURS[flavour])
        try:
            gpu_ctx.upload_scene(odd)
            res = gpu_ctx.render(R.BasicCamera(), R.make_config(96, 54, 6), want_raycount=True)
        finally:
            gpu_ctx.set_flags(0)
        w = O.World(spheres=spheres_to_oracle(odd))
        _compare(res, *O.render_counter(w, O.camera_basic(), O.make_config(96, 54, 6)))


def _telephoto(frm, half_extent, nx, ny):
    """A camera far away that still resolves the scene around the origin: BasicCamera.get_ray form (camera.zig:16,
    direction = lower_left + u*horizontal + v*vertical, no origin subtraction -- MovableCamera's `- origin`
    (camera.zig:39) cancels every direction bit at these distances), directions of unit length, the frustum
    `half_extent` wide at the origin.  Returns the product's and the oracle's camera."""
    frm = np.asarray(frm, np.float64)
    dist = np.linalg.norm(frm)
    w = -frm / dist
    u = np.cross(w, [0.0, 1.0, 0.0]); u /= np.linalg.norm(u)
    v = np.cross(u, w)
    hw = half_extent / dist
    hh = hw * ny / nx
    ll, hz, vt = w - hw * u - hh * v, 2 * hw * u, 2 * hh * v
    cam = R.BasicCamera()
    for k in range(3):
        cam._c.lower_left[k], cam._c.horizontal[k], cam._c.vertical[k], cam._c.origin[k] = ll[k], hz[k], vt[k], frm[k]
    cam._c.subtract_origin = 0
    ocam = O.Camera(O.v3(ll), O.v3(hz), O.v3(vt), O.v3(frm), 0)
    return cam, ocam


@pytest.mark.parametrize("dist", [2.0e3, 2.0e4, 5.0e5])
@pytest.mark.parametrize("mesh", [False, True])
def test_far_camera_cull_is_conservative(gpu_ctx, dist, mesh):
    """Ray origins 10^4 .. 10^6 primitive sizes away (ADVICE r1 / VERDICT r1 #10).  At that distance the f32
    discriminant of Sphere.hit (raytracer.zig:48-53) is mostly rounding noise -- the reference's own arithmetic
    draws a sphere as a blob several radii wide -- and the plane*inv - o*inv form of the slab test cancels badly.
    The BVH kernels test a cone (RT_CONE_*), so they must still visit every primitive the list traversal's exact
    test accepts: per-pixel ray counts equal the oracle's list traversal."""
    rng = np.random.default_rng(int(dist) % 1000 + mesh)
    n = 40
    s = R.Spheres(cx=rng.uniform(-1.5, 1.5, n), cy=rng.uniform(-1.0, 1.0, n), cz=rng.uniform(-1.5, 1.5, n),
                  radius=rng.choice([0.05, 0.2, 0.4], n), kind=rng.integers(0, 3, n), r=rng.random(n), g=rng.random(n),
                  b=rng.random(n), param=np.where(rng.random(n) < 0.5, 1.5, 0.2))
    nx, ny, spp = 96, 64, 4
    cam, ocam = _telephoto((0.6 * dist, 0.3 * dist, 0.74 * dist), 2.0, nx, ny)
    if mesh:
        tris = R.Tessellate_Spheres(s, 10, 8)
        gpu_ctx.upload_scene(triangles=tris, materials=s)
        w = O.World(triangles=triangles_to_oracle(tris), materials=spheres_to_oracle(s))
    else:
        gpu_ctx.set_flags(R.RT_FLAG_FORCE_BVH)
        try:
            gpu_ctx.upload_scene(s)
        finally:
            gpu_ctx.set_flags(0)
        w = O.World(spheres=spheres_to_oracle(s))
    gpu_ctx.set_flags(0 if mesh else R.RT_FLAG_FORCE_BVH)
    try:
        res = gpu_ctx.render(cam, R.make_config(nx, ny, spp), want_raycount=True)
    finally:
        gpu_ctx.set_flags(0)
    assert "bvh" in gpu_ctx.last_kernel_name()
    # the oracle WITHOUT any culling structure of its own (every triangle / every sphere, index order)
    oa, o8, orc, cnt = O.render_counter(w, ocam, O.make_config(nx, ny, spp, traversal=O.TRAV_BRUTE))
    assert cnt["rays"] > cnt["primary_rays"]                   # something is hit
    assert res.rays == cnt["rays"] and np.array_equal(res.raycount, orc)
    # at 10^5+ sizes the reference's arithmetic also produces NaN radiance (0/0 in a hit normal that has lost every
    # bit): the same pixels on both sides; everywhere else the usual bar
    nan = np.isnan(oa)
    assert np.array_equal(np.isnan(res.accum), nan)
    assert dist > 1e5 or not nan.any()
    assert float(np.abs(res.accum - oa)[~nan].max()) <= TOL_ACCUM


def test_scene_far_from_the_world_origin(gpu_ctx):
    """Coordinates of magnitude 10^4 with unit-size geometry: plane and origin nearly cancel in the slab test (the
    1e-5 |coordinate| part of the static padding)."""
    rng = np.random.default_rng(5)
    n = 60
    off = np.array([9000.0, -4000.0, 7000.0])
    s = R.Spheres(cx=off[0] + rng.uniform(-3, 3, n), cy=off[1] + rng.uniform(0, 1, n), cz=off[2] + rng.uniform(-3, 3, n),
                  radius=rng.choice([0.1, 0.3], n), kind=rng.integers(0, 3, n), r=rng.random(n), g=rng.random(n),
                  b=rng.random(n), param=np.where(rng.random(n) < 0.5, 1.5, 0.1))
    nx, ny, spp = 96, 64, 4
    frm = tuple(off + np.array([6.0, 3.0, 7.0])); at = tuple(off)
    cam = R.MovableCamera().SetPosition(frm, at, (0.0, 1.0, 0.0), 40.0, 1.5)
    ocam = O.camera_set_position(frm, at, (0.0, 1.0, 0.0), 40.0, 1.5)
    for mesh in (False, True):
        if mesh:
            tris = R.Tessellate_Spheres(s, 8, 6)
            gpu_ctx.upload_scene(triangles=tris, materials=s)
            w = O.World(triangles=triangles_to_oracle(tris), materials=spheres_to_oracle(s))
        else:
            gpu_ctx.upload_scene(s)
            w = O.World(spheres=spheres_to_oracle(s))
        res = gpu_ctx.render(cam, R.make_config(nx, ny, spp), want_raycount=True)
        assert "bvh" in gpu_ctx.last_kernel_name()
        _compare(res, *O.render_counter(w, ocam, O.make_config(nx, ny, spp)))


@pytest.mark.parametrize("name,spp", [("two", 100), ("book", 100)])
def test_full_size_1080p_sampled_pixels(gpu_ctx, name, spp):
    """configs[1] / configs[2] at full size: the counter RNG makes every pixel independent, so the
    oracle renders a random subset of pixels at full spp and the GPU image must match there; plus
    size-independent properties of the whole image."""
    nx, ny = 1920, 1080
    s, cam, ocam = _scene(name, nx, ny)
    gpu_ctx.upload_scene(s)
    res = gpu_ctx.render(cam, R.make_config(nx, ny, spp), want_raycount=True)
    rng = np.random.default_rng(11)
    pix = np.unique(np.concatenate([rng.integers(0, nx * ny, 1500), [0, nx - 1, nx * ny - nx, nx * ny - 1]])).astype(np.uint32)
    w = O.World(spheres=spheres_to_oracle(s))
    oa, o8, orc, _ = O.render_counter(w, ocam, O.make_config(nx, ny, spp), pixels=pix)
    assert np.array_equal(res.raycount[pix], orc)
    err = float(np.abs(res.accum[pix] - oa).max())
    assert err <= TOL_ACCUM, err
    assert psnr_u8(res.rgb8[pix], o8) >= MIN_PSNR
    # whole-image properties
    assert res.primary_rays == nx * ny * spp
    assert res.rays == int(res.raycount.astype(np.uint64).sum())
    assert np.all(res.raycount >= spp) and np.all(res.raycount <= spp * 18)    # 1..18 rays per path (Q5)
    assert np.all(np.isfinite(res.accum)) and res.accum.min() >= 0 and res.accum.max() <= 1.0 + 1e-5
    q = np.clip((255.99 * np.sqrt(res.accum)).astype(np.int32), 0, 255).astype(np.uint8)
    assert np.array_equal(q, res.rgb8)                                          # tonemap (raytracer.zig:527)
    print("%s 1080p/%dspp: %d rays, kernel %.1f ms = %.0f Mrays/s, sampled max err %.3g"
          % (name, spp, res.rays, res.kernel_ms, res.rays / res.kernel_ms / 1e3, err))


def test_c5_resolution_3840x2160_sampled_pixels(gpu_ctx):
    """configs[4] geometry (4K, 8.3 M pixels; its 1024 spp are split over GPUs by bench.py/distributed.py):
    one rank's share at reduced spp -- global sample indices [512, 516) of 1024 -- checked on random pixels."""
    nx, ny = 3840, 2160
    s, cam, ocam = _scene("book", nx, ny)
    gpu_ctx.upload_scene(s)
    kw = dict(spp_begin=512, spp_end=516, spp_total=1024)
    res = gpu_ctx.render(cam, R.make_config(nx, ny, 1024, **kw), want_raycount=True)
    rng = np.random.default_rng(21)
    pix = np.unique(np.concatenate([rng.integers(0, nx * ny, 3000), [0, nx * ny - 1]])).astype(np.uint32)
    w = O.World(spheres=spheres_to_oracle(s))
    oa, _, orc, _ = O.render_counter(w, ocam, O.make_config(nx, ny, 1024, **kw), pixels=pix)
    assert np.array_equal(res.raycount[pix], orc)
    assert float(np.abs(res.accum[pix] - oa).max()) <= TOL_ACCUM
    assert res.primary_rays == nx * ny * 4


def test_c5_full_size_3840x2160_1024spp_split_over_two_contexts(gpu_ctx):
    """configs[4] at FULL size: 3840x2160, 1024 spp, the samples split over two contexts by rt_render_multi
    (raytracer.zig:594-617 with GPUs for threads; on this one-GPU box both contexts sit on device 0, so the
    exchange step is the shared-GPU ordered add -- the RCCL reduce needs distinct GPUs).  8.5 G primary rays;
    the oracle renders 2 500 random pixels at all 1024 samples: ray counts exact, linear mean within 1e-3."""
    nx, ny, spp = 3840, 2160, 1024
    s, cam, ocam = _scene("book", nx, ny)
    gpu_ctx.upload_scene(s)
    with R.Context(0) as other:
        other.upload_scene(s)
        res = R.render_multi([gpu_ctx, other], cam, R.make_config(nx, ny, spp, spp_chunk=R.RT_SPP_CHUNK_AUTO),
                             want_raycount=True)
    rng = np.random.default_rng(55)
    pix = np.unique(np.concatenate([rng.integers(0, nx * ny, 2500), [0, nx - 1, nx * ny - nx, nx * ny - 1]])).astype(np.uint32)
    w = O.World(spheres=spheres_to_oracle(s))
    oa, o8, orc, _ = O.render_counter(w, ocam, O.make_config(nx, ny, spp), pixels=pix)
    assert np.array_equal(res.raycount[pix], orc)
    err = float(np.abs(res.accum[pix] - oa).max())
    assert err <= TOL_ACCUM, err
    assert psnr_u8(res.rgb8[pix], o8) >= MIN_PSNR
    assert res.primary_rays == nx * ny * spp and res.rays == int(res.raycount.astype(np.uint64).sum())
    assert res.raycount.min() >= spp and res.raycount.max() <= spp * 18
    assert np.all(np.isfinite(res.accum)) and res.accum.min() >= 0 and res.accum.max() <= 1.0 + 1e-5
    print("C5 4K/1024spp over 2 contexts: %d rays, slowest context %.0f ms, sampled max err %.3g"
          % (res.rays, res.kernel_ms, err))


def test_rccl_single_rank_smoke(gpu_ctx):
    """What one GPU can exercise of rt_render_multi's RCCL exchange step: the library loads RCCL, makes a one-rank
    communicator on the ctx's GPU (ncclCommInitAll) and runs an in-place ncclReduce(sum) on the ctx's stream.
    The N > 1 reduce itself needs distinct GPUs (the driver's 8-GPU node)."""
    gpu_ctx.selftest_rccl(1 << 16)
    gpu_ctx.selftest_rccl(3)          # the communicator is cached on the ctx


# ---- sharding properties -----------------------------------------------------------------
def test_spp_ranges_sum_to_the_whole(gpu_ctx):
    nx, ny, spp = 160, 90, 12
    s, cam, _ = _scene("book", nx, ny)
    gpu_ctx.upload_scene(s)
    full = gpu_ctx.render(cam, R.make_config(nx, ny, spp), want_raycount=True)
    parts = [gpu_ctx.render(cam, R.make_config(nx, ny, spp, spp_begin=b, spp_end=e, spp_total=spp), want_raycount=True)
             for b, e in ((0, 5), (5, 6), (6, 12))]
    assert np.array_equal(sum(p.raycount for p in parts), full.raycount)        # sample assignment bit-exact
    np.testing.assert_allclose(sum(p.accum for p in parts), full.accum, rtol=0, atol=1e-6)
    empty = gpu_ctx.render(cam, R.make_config(nx, ny, spp, spp_begin=4, spp_end=4, spp_total=spp), want_raycount=True)
    assert empty.rays == 0 and not empty.accum.any() and not empty.raycount.any()


def test_render_multi_contexts_split_the_samples(gpu_ctx):
    """rt_render_multi (single-process multi-GPU form, raytracer.zig:588-617 with GPUs for threads): three
    contexts -- on this one-GPU box all on device 0 -- take sample ranges [0,5) [5,9) [9,13); the sums are
    gathered on the first context's GPU and added in context order."""
    nx, ny, spp = 128, 72, 13
    s, cam, ocam = _scene("book", nx, ny)
    gpu_ctx.upload_scene(s)
    w = O.World(spheres=spheres_to_oracle(s))
    oa, o8, _, cnt = O.render_counter(w, ocam, O.make_config(nx, ny, spp))
    with R.Context(0) as c1, R.Context(0) as c2:
        with pytest.raises(R.RtError) as e:                       # a context without a scene
            R.render_multi([gpu_ctx, c1], cam, R.make_config(nx, ny, spp))
        assert e.value.code == -4
        c1.upload_scene(s); c2.upload_scene(s)
        with pytest.raises(R.RtError) as e:                       # the same context twice
            R.render_multi([gpu_ctx, c1, gpu_ctx], cam, R.make_config(nx, ny, spp))
        assert e.value.code == -1
        for chunk in (0, R.RT_SPP_CHUNK_AUTO):
            cfg = R.make_config(nx, ny, spp, spp_chunk=chunk)
            res = R.render_multi([gpu_ctx, c1, c2], cam, cfg)
            assert res.rays == cnt["rays"] and res.primary_rays == cnt["primary_rays"]
            assert float(np.abs(res.accum - oa).max()) <= TOL_ACCUM
            assert psnr_u8(res.rgb8, o8) >= MIN_PSNR
            parts = [gpu_ctx.render(cam, R.make_config(nx, ny, spp, spp_begin=b, spp_end=e_, spp_total=spp, spp_chunk=chunk))
                     for b, e_ in ((0, 5), (5, 9), (9, 13))]
            np.testing.assert_allclose(sum(p.accum for p in parts), res.accum, rtol=0, atol=1e-6)
            one = R.render_multi([c2], cam, cfg)                  # n_ctx = 1 is rt_render
            ref = c2.render(cam, cfg)
            assert np.array_equal(one.accum.view(np.uint32), ref.accum.view(np.uint32)) and np.array_equal(one.rgb8, ref.rgb8)
        # more contexts than samples: the extra ones trace nothing
        few = R.render_multi([gpu_ctx, c1, c2], cam, R.make_config(nx, ny, 2))
        ref = gpu_ctx.render(cam, R.make_config(nx, ny, 2))
        assert few.rays == ref.rays
        np.testing.assert_allclose(few.accum, ref.accum, rtol=0, atol=1e-6)


@pytest.mark.skipif(R.device_count() < 2 if os.path.exists("/dev/kfd") else True, reason="needs two GPUs")
def test_render_multi_over_two_gpus_reduces_with_rccl(gpu_ctx):
    """ADVICE r2 / VERDICT r2 weak #8: rt_render_multi with the contexts on DISTINCT GPUs -- the exchange step is then the
    grouped in-place ncclReduce onto ctxs[0]'s GPU, not the shared-GPU copy-and-add.  Skipped on a one-GPU box."""
    nx, ny, spp = 128, 72, 12
    s, cam, ocam = _scene("book", nx, ny)
    gpu_ctx.upload_scene(s)
    w = O.World(spheres=spheres_to_oracle(s))
    oa, o8, orc, cnt = O.render_counter(w, ocam, O.make_config(nx, ny, spp))
    with R.Context(1) as c1:
        c1.upload_scene(s)
        for chunk in (0, 3):
            cfg = R.make_config(nx, ny, spp, spp_chunk=chunk)
            one = gpu_ctx.render(cam, cfg, want_raycount=True)
            for want_rc in (True, False):
                res = R.render_multi([gpu_ctx, c1], cam, cfg, want_raycount=want_rc)
                assert res.rays == cnt["rays"] and res.primary_rays == cnt["primary_rays"]
                if want_rc:
                    assert np.array_equal(res.raycount, orc)
                np.testing.assert_allclose(res.accum, one.accum, rtol=0, atol=1e-6)
                assert float(np.abs(res.accum - oa).max()) <= TOL_ACCUM
                assert psnr_u8(res.rgb8, o8) >= MIN_PSNR
            # two ranks: the sum is a0 + a1 whatever order RCCL adds in
            parts = [gpu_ctx.render(cam, R.make_config(nx, ny, spp, spp_begin=b, spp_end=e, spp_total=spp, spp_chunk=chunk))
                     for b, e in ((0, 6), (6, 12))]
            np.testing.assert_allclose(parts[0].accum + parts[1].accum, res.accum, rtol=0, atol=2e-7)


def test_cu_mask_that_empties_an_xcd_is_refused(gpu_ctx):
    """{0x01010101} (CUs of XCD 0 only) used to reach hipExtStreamCreateWithCUMask, which takes the
    process down under ROCm 7.2.  It is now RT_E_ARG before any HIP call, the ctx keeps rendering, and the 4-per-XCD
    mask (the "@1/8 GPU" leg of bench.py) still works and gives the same image."""
    nx, ny, spp = 96, 54, 4
    s, cam, _ = _scene("book", nx, ny)
    gpu_ctx.upload_scene(s)
    ref = gpu_ctx.render(cam, R.make_config(nx, ny, spp), want_raycount=True)
    for words in ([0x01010101], [0x01010101] * 8, [0x7F], [0]):
        with pytest.raises(R.RtError) as e:
            gpu_ctx.set_cu_mask_words(words)
        assert e.value.code == _lib.RT_E_ARG
    with pytest.raises(ValueError):
        gpu_ctx.set_cu_mask(4)
    again = gpu_ctx.render(cam, R.make_config(nx, ny, spp), want_raycount=True)      # the ctx is unharmed
    assert np.array_equal(again.raycount, ref.raycount) and np.array_equal(again.accum.view(np.uint32), ref.accum.view(np.uint32))
    try:
        gpu_ctx.set_cu_mask(32)
        masked = gpu_ctx.render(cam, R.make_config(nx, ny, spp), want_raycount=True)
    finally:
        gpu_ctx.set_cu_mask(None)
    assert np.array_equal(masked.raycount, ref.raycount) and np.array_equal(masked.accum.view(np.uint32), ref.accum.view(np.uint32))


@pytest.mark.parametrize("chunk", [1, 3, 5, 16, R.RT_SPP_CHUNK_AUTO])
def test_sample_chunks_match_the_oracle_chunk_mode(gpu_ctx, chunk):
    """spp_chunk: samples summed per chunk of global sample indices, chunk sums added in order -- the oracle
    has the same mode.  Ray counts are exact; also checks a range that starts inside a chunk."""
    nx, ny = 100, 60
    s, cam, ocam = _scene("book", nx, ny)
    gpu_ctx.upload_scene(s)
    w = O.World(spheres=spheres_to_oracle(s))
    for kw in (dict(), dict(spp_begin=4, spp_end=13, spp_total=13)):
        spp = 13
        res = gpu_ctx.render(cam, R.make_config(nx, ny, spp, spp_chunk=chunk, **kw), want_raycount=True)
        ochunk = 0 if chunk == R.RT_SPP_CHUNK_AUTO else chunk
        oa, o8, orc, cnt = O.render_counter(w, ocam, O.make_config(nx, ny, spp, spp_chunk=ochunk, **kw))
        _compare(res, oa, o8, orc, cnt)
    tile = gpu_ctx.render(cam, R.make_config(nx, ny, 12, spp_chunk=4, tile=(10, 5, 61, 44)), want_raycount=True)
    full = gpu_ctx.render(cam, R.make_config(nx, ny, 12, spp_chunk=4), want_raycount=True)
    m = np.zeros((ny, nx), bool); m[5:44, 10:61] = True
    assert np.array_equal(tile.raycount.reshape(ny, nx)[m], full.raycount.reshape(ny, nx)[m])
    assert not tile.raycount.reshape(ny, nx)[~m].any() and not tile.accum.reshape(ny, nx, 3)[~m].any()
    assert np.array_equal(tile.accum.reshape(ny, nx, 3)[m].view(np.uint32), full.accum.reshape(ny, nx, 3)[m].view(np.uint32))


def test_many_small_chunks_at_full_resolution(gpu_ctx):
    """1920x1080 with one-sample chunks: 83 M work items (the bound is on items fitting 32 bits, not on pixels x 64)."""
    nx, ny, spp = 1920, 1080, 40
    s, cam, _ = _scene("two", nx, ny)
    gpu_ctx.upload_scene(s)
    a = gpu_ctx.render(cam, R.make_config(nx, ny, spp, spp_chunk=1), want_raycount=True)
    b = gpu_ctx.render(cam, R.make_config(nx, ny, spp), want_raycount=True)
    assert a.rays == b.rays and np.array_equal(a.raycount, b.raycount)
    np.testing.assert_allclose(a.accum, b.accum, rtol=0, atol=2e-6)
    with pytest.raises(R.RtError) as e:                       # 4097 chunks: more planes than the library allows
        gpu_ctx.render(cam, R.make_config(64, 64, 5000, spp_chunk=1))
    assert e.value.code == _lib.RT_E_ARG


def test_tiles_equal_the_subrectangle(gpu_ctx):
    nx, ny, spp = 100, 60, 4
    s, cam, _ = _scene("book", nx, ny)
    gpu_ctx.upload_scene(s)
    full = gpu_ctx.render(cam, R.make_config(nx, ny, spp), want_raycount=True)
    acc = np.zeros_like(full.accum).reshape(ny, nx, 3)
    cnt = np.zeros(ny * nx, np.uint32).reshape(ny, nx)
    for tile in ((0, 0, 37, 60), (37, 0, 100, 23), (37, 23, 100, 60)):
        t = gpu_ctx.render(cam, R.make_config(nx, ny, spp, tile=tile), want_raycount=True)
        x0, y0, x1, y1 = tile
        ta, tc = t.accum.reshape(ny, nx, 3), t.raycount.reshape(ny, nx)
        mask = np.zeros((ny, nx), bool); mask[y0:y1, x0:x1] = True
        assert not ta[~mask].any() and not tc[~mask].any()                      # nothing written outside the tile
        acc += ta; cnt += tc
    assert np.array_equal(cnt.ravel(), full.raycount)
    assert np.array_equal(acc.reshape(-1, 3).view(np.uint32), full.accum.view(np.uint32))   # same lanes' arithmetic


def test_runs_are_deterministic(gpu_ctx):
    s, cam, _ = _scene("book", 200, 120)
    gpu_ctx.upload_scene(s)
    a = gpu_ctx.render(cam, R.make_config(200, 120, 6), want_raycount=True)
    b = gpu_ctx.render(cam, R.make_config(200, 120, 6), want_raycount=True)
    assert np.array_equal(a.accum.view(np.uint32), b.accum.view(np.uint32)) and np.array_equal(a.raycount, b.raycount)
    c = gpu_ctx.render(cam, R.make_config(200, 120, 6, seed=2), want_raycount=True)
    assert not np.array_equal(a.raycount, c.raycount)                           # the seed matters


# ---- edge cases ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nx,ny,spp,kw", [
    (37, 19, 3, {}),                       # ragged: not a multiple of the 8x8 work tile
    (1, 1, 5, {}),                         # single pixel
    (7, 3, 1, {}),                         # smaller than one tile, one sample
    (64, 36, 2, {"max_depth": 0}),         # depth cap 0: primary hit scatters once, second hit is black
    (64, 36, 2, {"max_depth": 3}),
    (48, 27, 3, {"row_offset_q3": 0}),     # the un-quirked row mapping
    (48, 27, 3, {"seed": 0xDEADBEEFCAFEF00D}),   # 64-bit key
    (48, 27, 2, {"t_min": 0.0}),
])
@pytest.mark.parametrize("flavour", ["brute", "bvh"])
def test_edge_cases(gpu_ctx, flavour, nx, ny, spp, kw):
    res, (oa, o8, orc, cnt) = _both(gpu_ctx, "book", nx, ny, spp, flavour=flavour, **kw)
    _compare(res, oa, o8, orc, cnt)


def test_spp_offset_uses_global_sample_indices(gpu_ctx):
    res, (oa, o8, orc, cnt) = _both(gpu_ctx, "book", 40, 24, 9, spp_begin=5, spp_end=9, spp_total=9)
    _compare(res, oa, o8, orc, cnt)


def test_negative_radius_and_big_index_spheres(gpu_ctx):
    """Hollow glass (negative radius, raytracer.zig:677) and index up to 10 (Schlick cosine > 1, Q8)."""
    s = R.Spheres(cx=[0, 0, 0, 1.2], cy=[0, 0, -100.5, 0], cz=[-1, -1, -1, -1.5], radius=[0.5, -0.45, 100, 0.4],
                  kind=[2, 2, 0, 2], r=[0.95, 0.95, 0.5, 0.95], g=[0.95, 0.95, 0.5, 0.95], b=[0.95, 0.95, 0.5, 0.95],
                  param=[1.5, 1.5, 0, 9.7])
    gpu_ctx.upload_scene(s)
    res = gpu_ctx.render(R.BasicCamera(), R.make_config(96, 54, 8), want_raycount=True)
    w = O.World(spheres=spheres_to_oracle(s))
    oa, o8, orc, cnt = O.render_counter(w, O.camera_basic(), O.make_config(96, 54, 8))
    _compare(res, oa, o8, orc, cnt)
    assert cnt["scatter_dielectric"] > 1000


def test_depth_histogram_matches_oracle(gpu_ctx):
    s, cam, ocam = _scene("book", 96, 54)
    gpu_ctx.upload_scene(s)
    gpu_ctx.set_flags(R.RT_FLAG_DEPTH_HISTOGRAM)
    try:
        res = gpu_ctx.render(cam, R.make_config(96, 54, 6), want_raycount=True)
    finally:
        gpu_ctx.set_flags(0)
    w = O.World(spheres=spheres_to_oracle(s))
    _, _, orc, cnt = O.render_counter(w, ocam, O.make_config(96, 54, 6))
    assert np.array_equal(res.raycount, orc)
    assert res.rays_by_depth[:19] == cnt["rays_by_depth"][:19] and sum(res.rays_by_depth) == res.rays


def test_error_codes(gpu_ctx):
    fresh = R.Context(0)
    try:
        with pytest.raises(R.RtError) as e:                                    # render before upload
            fresh.render(R.BasicCamera(), R.make_config(8, 8, 1))
        assert e.value.code == _lib.RT_E_STATE
    finally:
        fresh.close()
    gpu_ctx.upload_scene(R.Two_Sphere_Scene())
    for bad in (R.make_config(0, 8, 1), R.make_config(8, 8, 1, spp_begin=3, spp_end=2),
                R.make_config(8, 8, 1, tile=(4, 0, 2, 8)), R.make_config(8, 8, 1, tile=(0, 0, 9, 8)),
                R.make_config(8, 8, 1, spp_total=0, spp_end=1)):
        with pytest.raises(R.RtError) as e:
            gpu_ctx.render(R.BasicCamera(), bad)
        assert e.value.code == _lib.RT_E_ARG
    s = R.Two_Sphere_Scene(); s.kind[0] = 7
    with pytest.raises(R.RtError):
        gpu_ctx.upload_scene(s)
    tri = R.Triangles([[0, 0, -1]], [[1, 0, -1]], [[0, 1, -1]], [3])          # material index out of range
    with pytest.raises(R.RtError):
        gpu_ctx.upload_scene(triangles=tri, materials=R.Two_Sphere_Scene())
    with pytest.raises(R.RtError):                                              # triangles need a material table
        gpu_ctx.upload_scene(triangles=R.Triangles([[0, 0, -1]], [[1, 0, -1]], [[0, 1, -1]], [0]))
    with pytest.raises(R.RtError):                                              # one kind of primitive per scene
        gpu_ctx.upload_scene(R.Two_Sphere_Scene(), triangles=tri, materials=R.Two_Sphere_Scene())


def test_randomised_scenes_cameras_and_configs():
    """tools/gpu_fuzz.py: 300 random cases (sphere scenes up to 700 spheres incl. hollow glass and coincident
    centres, tessellated meshes under several depth caps, cameras, image sizes down to 1x1, sample sub-ranges,
    chunk sizes, depth caps 0..40, t_min 0) -- ray counts exact, linear mean within 1e-3 in every one."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_fuzz.py"), "300", "7"], capture_output=True,
                         text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "fuzz: 300 cases ok" in out.stdout


# ---- device-resident entry points (what bench.py and the multi-GPU driver use) ----------------------------
def test_render_device_and_tonemap_device(gpu_ctx):
    import torch
    nx, ny, spp = 128, 72, 8
    s, cam, _ = _scene("book", nx, ny)
    gpu_ctx.upload_scene(s)
    host = gpu_ctx.render(cam, R.make_config(nx, ny, spp), want_raycount=True)
    acc = torch.zeros(nx * ny * 3, dtype=torch.float32, device="cuda")
    rc = torch.zeros(nx * ny, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    rays = 0
    for b, e in ((0, 3), (3, 8)):                                               # two passes ADD into the buffer
        r, _, _, _ = gpu_ctx.render_device(cam, R.make_config(nx, ny, spp, spp_begin=b, spp_end=e, spp_total=spp),
                                           acc.data_ptr(), rc.data_ptr(), st)
        rays += r
    mean = torch.empty_like(acc)
    rgb = torch.empty(nx * ny * 3, dtype=torch.uint8, device="cuda")
    gpu_ctx.tonemap_device(acc.data_ptr(), nx * ny, spp, mean.data_ptr(), rgb.data_ptr(), st)
    torch.cuda.synchronize()
    assert rays == host.rays
    assert np.array_equal(rc.cpu().numpy().astype(np.uint32), host.raycount)
    np.testing.assert_allclose(mean.cpu().numpy().reshape(-1, 3), host.accum, rtol=0, atol=1e-6)
    assert psnr_u8(rgb.cpu().numpy().reshape(-1, 3), host.rgb8) >= MIN_PSNR


def test_reference_style_u8_mean_of_images(gpu_ctx):
    """render_and_write's merge (raytracer.zig:594-617, quirk Q4): N whole-image renders keyed i*123, each
    quantised to u8 (:527), then the per-channel INTEGER mean (:610-617).  Checker: the oracle in the same mode --
    counter RNG keyed i*123, one render per image, numpy integer sum and truncating divide.  Byte work, so the bar is
    equality: every channel must be identical, except where one of the N linear values sits within 1e-6 (relative) of a
    quantisation threshold -- there the GPU's running-product rounding (~1e-7) may land on the other side, one image's
    byte moves by one and the mean by at most one."""
    nx, ny, spp, n_images = 96, 54, 3, 5
    s, cam, ocam = _scene("book", nx, ny)
    gpu_ctx.upload_scene(s)
    img, (rays, prim, _ms, _) = gpu_ctx.render_reference_style(cam, R.make_config(nx, ny, spp, seed=99), n_images)
    w = O.World(spheres=spheres_to_oracle(s))
    acc = np.zeros((nx * ny, 3), np.uint32)
    near_threshold = np.zeros((nx * ny, 3), bool)
    orays = 0
    for i in range(n_images):
        oa, o8, _, cnt = O.render_counter(w, ocam, O.make_config(nx, ny, spp, seed=i * 123))
        q = (np.float32(255.99) * np.sqrt(oa)).astype(np.float64)         # f32, as raytracer.zig:527 computes it
        near_threshold |= np.abs(q - np.round(q)) <= 1e-6 * np.maximum(q, 1.0) * 0.5 + 1e-6 * 255.99
        assert np.array_equal(o8, np.clip(q.astype(np.int64), 0, 255).astype(np.uint8))     # the oracle's own tonemap
        acc += o8
        orays += cnt["rays"]
    ref = (acc // n_images).astype(np.uint8)
    assert rays == orays and prim == nx * ny * spp * n_images
    assert np.array_equal(img[~near_threshold], ref[~near_threshold])
    assert np.abs(img.astype(np.int32) - ref.astype(np.int32)).max() <= 1
    assert near_threshold.mean() < 0.01
    lin = gpu_ctx.render(cam, R.make_config(nx, ny, spp * n_images, seed=0)).rgb8   # the linear-mean image differs (Q4)
    assert img.shape == lin.shape and not np.array_equal(img, lin)


def test_progressive_render_converges_to_rt_render(gpu_ctx):
    """rt_progressive_*: the polling form for the GUI's Trace button (main.zig:307-318,342-345; slot shape of
    compute.zig:123-207).  Every intermediate picture is the mean over the samples so far (checked against a plain
    render of exactly those samples), and the last one is rt_render's image with spp_chunk = the pass size, bit for bit."""
    nx, ny, spp, passes = 96, 54, 12, 4
    s, cam, ocam = _scene("book", nx, ny)
    gpu_ctx.upload_scene(s)
    w = O.World(spheres=spheres_to_oracle(s))
    # (collected first: a plain render between two steps would end the progressive render, see below)
    steps = [(done, rgb8.copy(), accum.copy(), rays, prim)
             for done, rgb8, accum, (rays, prim, _ms, _) in gpu_ctx.render_progressive(cam, R.make_config(nx, ny, spp), passes)]
    seen = [st[0] for st in steps]
    for done, rgb8, accum, rays, prim in steps:
        assert prim == nx * ny * done
        part = gpu_ctx.render(cam, R.make_config(nx, ny, done, spp_chunk=spp // passes))
        assert np.array_equal(accum.view(np.uint32), part.accum.view(np.uint32)) and np.array_equal(rgb8, part.rgb8)
        assert rays == part.rays
        # ... and against the ORACLE: every intermediate picture is the oracle's image of the samples
        # so far in its chunk mode with chunk = the pass size (each pass adds one chunk sum to the SUM buffer)
        oa, o8, _, cnt = O.render_counter(w, ocam, O.make_config(nx, ny, done, spp_chunk=spp // passes))
        assert rays == cnt["rays"] and prim == cnt["primary_rays"]
        assert float(np.abs(accum - oa).max()) <= TOL_ACCUM
        assert psnr_u8(rgb8, o8) >= MIN_PSNR
    assert seen == [3, 6, 9, 12]
    # a render, an upload or a mask change between two steps ends the progressive render (they share the ctx's buffers)
    for breaker in (lambda: gpu_ctx.render(cam, R.make_config(nx, ny, 1)), lambda: gpu_ctx.upload_scene(s),
                    lambda: gpu_ctx.set_cu_mask(None)):
        it = gpu_ctx.render_progressive(cam, R.make_config(nx, ny, 4), 2)
        next(it)
        breaker()
        with pytest.raises(R.RtError) as e:
            next(it)
        assert e.value.code == _lib.RT_E_STATE
    with pytest.raises(R.RtError) as e:                        # no render in progress any more
        gpu_ctx._check(gpu_ctx._L.rt_progressive_step(gpu_ctx._h, None, None, None, None), "rt_progressive_step")
    assert e.value.code == _lib.RT_E_STATE
    # more passes than samples: one sample per pass
    assert [d for d, *_ in gpu_ctx.render_progressive(cam, R.make_config(nx, ny, 3), 10)] == [1, 2, 3]


def test_trace_button_without_the_window(tmp_path):
    """rt_main trace: the GUI's Trace button (main.zig:307-318, 342-345) through rt::TraceJob -- camera set as start_trace
    sets it from the fly camera (quirk Q9: the forward vector as look-at, pos + (0,1,0) as up, vfov 45), the render in
    polling passes with the picture so far on disk after each.  The finished file is the ORACLE's image for that camera
    (per-pixel bytes within the PSNR bar, ray total exact), and the pass log counts the samples up."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytrace_amd", "host", "rt_main")
    assert os.path.exists(exe), "build it first: make -C raytrace_amd/csrc (__graft_entry__.build())"
    nx, ny, spp, passes = 160, 90, 6, 3
    pos, fwd = (0.0, 4.0, -4.0), (0.0, -0.45399049973954675, 0.8910065241883679)     # DemoState's camera after one update(): pitch 0.15 pi
    f = str(tmp_path / "Scene.ppm")
    out = subprocess.run([exe, "trace", str(spp), str(passes), f, str(nx), str(ny)] + [repr(float(x)) for x in pos + fwd],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert [l.split(":")[1].split()[0] for l in out.stdout.splitlines() if l.startswith("pass:")] == ["2", "4", "6"]
    assert open(f).read().split("\n")[:3] == ["P3", "%d %d" % (nx, ny), "255"]
    a = np.loadtxt(f, skiprows=3, dtype=np.int64).reshape(-1, 3)
    up = (pos[0], pos[1] + 1.0, pos[2])
    aspect = float(np.float32(nx) / np.float32(ny))
    ocam = O.camera_set_position(tuple(np.float32(pos)), tuple(np.float32(fwd)), up, 45.0, aspect)
    w = O.World(spheres=O.random_scene(11, 0))
    _, o8, _, cnt = O.render_counter(w, ocam, O.make_config(nx, ny, spp, spp_chunk=spp // passes))
    assert psnr_u8(a, o8) >= MIN_PSNR
    assert "%d rays" % cnt["rays"] in out.stdout.splitlines()[-1]
    # the library path with the same camera: the same picture up to the f32 sum order RT_SPP_CHUNK_AUTO leaves open inside a pass
    with R.Context(0) as ctx:
        ctx.upload_scene(R.Random_Scene(11, 0))
        cam = R.MovableCamera().SetPosition(tuple(np.float32(pos)), tuple(np.float32(fwd)), up, 45.0, aspect)
        res = ctx.render(cam, R.make_config(nx, ny, spp, spp_chunk=spp // passes))
    assert np.abs(a - res.rgb8.astype(np.int64)).max() <= 1 and psnr_u8(a, res.rgb8) >= MIN_PSNR
    assert subprocess.run([exe, "trace", "0"], capture_output=True, timeout=60).returncode == 2


def test_render_and_write_ppm(gpu_ctx, tmp_path):
    """render_and_write (raytracer.zig:580-633) -> the same P3 file the oracle writes for its image."""
    nx, ny, spp = 48, 27, 4
    s, cam, ocam = _scene("book", nx, ny)
    gpu_ctx.upload_scene(s)
    f = str(tmp_path / "Scene.ppm")
    gpu_ctx.render_and_write(f, cam, R.make_config(nx, ny, spp))
    w = O.World(spheres=spheres_to_oracle(s))
    _, o8, _, _ = O.render_counter(w, ocam, O.make_config(nx, ny, spp))
    g = str(tmp_path / "Oracle.ppm")
    assert O.lib().orc_write_ppm(g.encode(), o8.ctypes.data, nx, ny) == 0
    a = np.loadtxt(f, skiprows=3, dtype=np.int64)
    b = np.loadtxt(g, skiprows=3, dtype=np.int64)
    assert open(f).read().split("\n")[:3] == ["P3", "%d %d" % (nx, ny), "255"]
    assert psnr_u8(a, b) >= MIN_PSNR


def test_standalone_program_writes_the_same_image(gpu_ctx, tmp_path):
    """raytrace_amd/host/rt_main: the C++ mirror of the reference's standalone main (raytracer.zig:714-745) -- book scene,
    the reference's camera pose, P3 PPM out -- against the Python path on the same configuration (byte-equal pixels)
    and against the oracle."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytrace_amd", "host", "rt_main")
    assert os.path.exists(exe), "build it first: make -C raytrace_amd/csrc (__graft_entry__.build())"
    nx, ny, spp = 160, 90, 4
    f = str(tmp_path / "Scene.ppm")
    out = subprocess.run([exe, str(nx), str(ny), str(spp), f], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "World contains 314 spheres" in out.stdout and "Creating %d x %d image" % (nx, ny) in out.stdout
    assert open(f).read().split("\n")[:3] == ["P3", "%d %d" % (nx, ny), "255"]
    a = np.loadtxt(f, skiprows=3, dtype=np.int64).reshape(-1, 3)
    s, cam, ocam = _scene("book", nx, ny)
    gpu_ctx.upload_scene(s)
    res = gpu_ctx.render(cam, R.make_config(nx, ny, spp, spp_chunk=R.RT_SPP_CHUNK_AUTO))
    assert np.array_equal(a, res.rgb8.astype(np.int64))
    w = O.World(spheres=spheres_to_oracle(s))
    _, o8, _, _ = O.render_counter(w, ocam, O.make_config(nx, ny, spp))
    assert psnr_u8(a, o8) >= MIN_PSNR
    assert subprocess.run([exe, "0", "0", "1"], capture_output=True, timeout=60).returncode == 2
