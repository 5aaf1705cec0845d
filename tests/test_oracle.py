"""CPU tests of the oracle (oracle/rt_oracle.c): the reference's own known-answer tests
(vector.zig:60-102), hand-derived KATs for the rest of the path, the RNG pins, and the
committed golden vectors.  No GPU."""
import ctypes as C
import math
import os

import numpy as np
import pytest

import oracle_lib as O

L = O.lib()
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_golden.npz"))


# ---- the reference's four vector KATs (vector.zig:60-102), tolerance 1e-6 as there ---------------
def test_vec_length():       # vector.zig:60-67
    rt2_2 = math.sqrt(2.0) / 2.0
    assert abs(L.orc_length(O.v3((rt2_2, rt2_2, 0))) - 1.0) < 1e-6


def test_vec_normalize():    # vector.zig:68-80
    rt2_2 = math.sqrt(2.0) / 2.0
    n = L.orc_normalize(O.v3((10.0, 0, 10.0))).tup()
    for got, exp in zip(n, (rt2_2, 0, rt2_2)):
        assert abs(got - exp) < 1e-6


def test_vec_dot():          # vector.zig:81-88
    assert abs(L.orc_dot(O.v3((1, 2, 3)), O.v3((6, 5, 4))) - 28.0) < 1e-6


def test_vec_cross():        # vector.zig:90-102
    c = L.orc_cross(O.v3((1, 2, 3)), O.v3((6, 5, 4))).tup()
    for got, exp in zip(c, (-7.0, 14.0, -7.0)):
        assert abs(got - exp) < 1e-6


def test_argmax_ties_go_to_later_axis():   # vector.zig:40-56
    assert L.orc_argmax(O.v3((3, 1, 2))) == 0
    assert L.orc_argmax(O.v3((1, 3, 2))) == 1
    assert L.orc_argmax(O.v3((1, 2, 3))) == 2
    assert L.orc_argmax(O.v3((2, 2, 1))) == 1   # x > y false -> y > z
    assert L.orc_argmax(O.v3((2, 1, 2))) == 2   # x > z false
    assert L.orc_argmax(O.v3((1, 1, 1))) == 2


# ---- RNG pins --------------------------------------------------------------------------------------
def test_sfc64_matches_numpy_core():
    """sfc64 core pinned against numpy's SFC64 with Zig's seeding (a=b=c=seed, counter=1, 12 rounds)."""
    for seed in (0, 123, 246, 2 ** 63 + 5):
        bg = np.random.SFC64(0)
        st = bg.state
        st["state"]["state"] = np.array([seed, seed, seed, 1], dtype=np.uint64)
        st["has_uint32"] = 0
        bg.state = st
        ref = bg.random_raw(12 + 8)[12:]
        s = O.Sfc64()
        L.orc_sfc64_init(C.byref(s), seed)
        got = [L.orc_sfc64_next(C.byref(s)) for _ in range(8)]
        assert [int(x) for x in ref] == got


def test_sfc64_seed0_known_values():
    """first outputs for seed 0 and the float(f32) values under the restated bit recipe."""
    s = O.Sfc64()
    L.orc_sfc64_init(C.byref(s), 0)
    assert [L.orc_sfc64_next(C.byref(s)) for _ in range(4)] == [
        0x3acfa029e3cc6041, 0xf5b6515bf2ee419c, 0x1259635894a29b61, 0x0b6ae75395f8ebd6]
    L.orc_sfc64_init(C.byref(s), 0)
    got = [L.orc_sfc64_float(C.byref(s)) for _ in range(4)]
    np.testing.assert_allclose(got, [0.19958593, 0.93068862, 0.07939792, 0.06077179], rtol=0, atol=5e-9)


def test_sfc64_float_bit_recipe():
    """float(f32) = ((126 - clz(x)) << 23) | (x & 0x7FFFFF): recompute from the raw stream."""
    s1, s2 = O.Sfc64(), O.Sfc64()
    L.orc_sfc64_init(C.byref(s1), 99)
    L.orc_sfc64_init(C.byref(s2), 99)
    for _ in range(2000):
        x = L.orc_sfc64_next(C.byref(s1))
        lz = 64 - x.bit_length()
        assert lz < 41          # the refill branch has probability 2^-41
        bits = ((126 - lz) << 23) | (x & 0x7FFFFF)
        exp = np.array([bits], dtype=np.uint32).view(np.float32)[0]
        got = L.orc_sfc64_float(C.byref(s2))
        assert got == exp and 0.0 <= got < 1.0


def _philox(ctr, key):
    c, k, o = (C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), (C.c_uint32 * 4)()
    L.orc_philox4x32_10(c, k, o)
    return list(o)


def test_philox_random123_known_answers():
    """Random123 kat_vectors for philox4x32_10."""
    assert _philox([0] * 4, [0] * 2) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == [
        0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_unit_float():
    assert L.orc_u32_to_unit_float(0) == 0.0
    assert L.orc_u32_to_unit_float(0xFFFFFFFF) == float(np.float32(1.0) - np.float32(2.0 ** -24))
    assert L.orc_u32_to_unit_float(0x80000000) == 0.5


# ---- math primitives --------------------------------------------------------------------------------
def test_cbrt_against_float64():
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.random(2000).astype(np.float32), np.float32([0.0, 1.0, 8.0, 1e-30, 1e-42, 27.0, -8.0])])
    for x in xs:
        got = L.orc_cbrtf(float(x))
        exp = np.float32(np.cbrt(np.float64(x)))
        assert abs(np.float64(got) - np.float64(exp)) <= np.spacing(np.float32(abs(exp))), (x, got, exp)
    assert L.orc_cbrtf(8.0) == 2.0 and L.orc_cbrtf(27.0) == 3.0 and L.orc_cbrtf(-8.0) == -2.0


def test_pow5_order():
    for x in np.float32([0.3, -0.7, 1.5, -8.9, 1e-3]):
        x2 = np.float32(x * x)
        x4 = np.float32(x2 * x2)
        assert L.orc_pow5(float(x)) == float(np.float32(x * x4))


def test_reflect_refract_schlick():
    # reflect (raytracer.zig:420-422)
    r = L.orc_reflect(O.v3((1, -1, 0)), O.v3((0, 1, 0))).tup()
    assert r == (1.0, 1.0, 0.0)
    # refract straight down through index 1.5: direction unchanged (raytracer.zig:453-463)
    out = O.V3()
    assert L.orc_refract(O.v3((0, -1, 0)), O.v3((0, 1, 0)), C.c_float(1 / 1.5), C.byref(out)) == 1
    assert out.tup() == (0.0, -1.0, 0.0)
    # total internal reflection: grazing, ni/nt = 1.5 -> null
    assert L.orc_refract(O.v3((1, -0.1, 0)), O.v3((0, 1, 0)), C.c_float(1.5), C.byref(out)) == 0
    # schlick (raytracer.zig:464-468) at cos in {0, .5, 1, 1.7}, idx 1.5: r0 = 0.04
    for cos, exp in ((0.0, 1.0), (0.5, 0.04 + 0.96 * 0.5 ** 5), (1.0, 0.04), (1.7, 0.04 + 0.96 * (-0.7) ** 5)):
        assert abs(L.orc_schlick(cos, 1.5) - exp) < 1e-6


def _sphere(center, radius, kind=0, albedo=(0.5, 0.5, 0.5), param=0.0):
    a = np.zeros(1, dtype=O.SPHERE_DTYPE)
    a[0] = (center, radius, kind, albedo, param)
    return a


def _hit(sph, ro, rd, tmin=0.001, tmax=3.4028234663852886e38):
    t, p, n = C.c_float(), O.V3(), O.V3()
    h = L.orc_sphere_hit(sph.ctypes.data, O.v3(ro), O.v3(rd), C.c_float(tmin), C.c_float(tmax),
                         C.byref(t), C.byref(p), C.byref(n))
    return h, t.value, p.tup(), n.tup()


def test_sphere_hit_cases():
    """Sphere.hit, raytracer.zig:42-78."""
    s = _sphere((0, 0, -3), 1.0)
    h, t, p, n = _hit(s, (0, 0, 0), (0, 0, -1))          # front root
    assert h == 1 and t == 2.0 and p == (0.0, 0.0, -2.0) and n == (0.0, 0.0, 1.0)
    h, t, p, n = _hit(s, (0, 0, -3), (0, 0, -1))         # from inside: back root
    assert h == 1 and t == 1.0 and n == (0.0, 0.0, -1.0)
    h, t, _, _ = _hit(s, (0, 0, 0), (0, 0, -2))          # unnormalised direction: t scales
    assert h == 1 and t == 1.0
    neg = _sphere((0, 0, -3), -1.0)                       # negative radius flips the normal (:60)
    h, t, p, n = _hit(neg, (0, 0, 0), (0, 0, -1))
    assert h == 1 and t == 2.0 and n == (0.0, 0.0, -1.0)
    h, *_ = _hit(s, (0, 0, -2), (0, 0, 1))               # on the surface leaving: t1=-2... both <= t_min
    assert h == 0
    h, t, *_ = _hit(s, (0, 0, -2.0005), (0, 0, -1), tmin=0.001)   # near root 0.0005-ish < t_min -> far root
    assert h == 1 and abs(t - 1.9995) < 1e-5
    h, *_ = _hit(s, (1, 0, 0), (0, 0, -1))               # tangent: disc == 0 -> miss (test is > 0)
    assert h == 0
    h, *_ = _hit(s, (0, 0, 0), (0, 0, -1), tmax=1.5)     # beyond t_max
    assert h == 0
    h, *_ = _hit(s, (0, 0, 0), (0, 0, 1))                # behind
    assert h == 0


def test_triangle_hit_cases():
    tri = np.zeros(1, dtype=O.TRIANGLE_DTYPE)
    tri[0] = ((0, 0, -2), (1, 0, -2), (0, 1, -2), 0)
    t, p, n = C.c_float(), O.V3(), O.V3()

    def hit(ro, rd, tmin=0.001, tmax=1e30):
        return L.orc_triangle_hit(tri.ctypes.data, O.v3(ro), O.v3(rd), C.c_float(tmin), C.c_float(tmax),
                                  C.byref(t), C.byref(p), C.byref(n))
    assert hit((0.25, 0.25, 0), (0, 0, -1)) == 1 and t.value == 2.0 and n.tup() == (0.0, 0.0, 1.0)
    assert hit((0.25, 0.25, -4), (0, 0, 1)) == 1 and t.value == 2.0 and n.tup() == (0.0, 0.0, 1.0)   # two-sided, normal fixed
    assert hit((0.75, 0.75, 0), (0, 0, -1)) == 0          # u+v > 1
    assert hit((-0.1, 0.25, 0), (0, 0, -1)) == 0
    assert hit((0.25, 0.25, 0), (1, 0, 0)) == 0           # parallel
    assert hit((0.25, 0.25, 0), (0, 0, -1), tmax=1.5) == 0
    # grazing guard (part of the test's definition): sin(corner angle) * cos(angle to the normal) >= 2^-7.  The
    # corner at v0 is a right angle, so a ray that meets the plane at slope s = |d_z| / |d| hits iff s >= 2^-7.
    for slope, expect in ((0.02, 1), (0.0085, 1), (0.0075, 0), (0.001, 0)):
        dz = -slope / np.sqrt(1.0 - slope * slope)            # direction (1, 0, dz): slope against the plane z = -2
        ro = (0.25 - 1.0, 0.25, -2.0 - dz)                    # reaches (0.25, 0.25, -2) at t = 1
        assert hit(ro, (1.0, 0.0, dz)) == expect, slope
        if expect:
            assert abs(t.value - 1.0) < 1e-3


def test_triangle_vertex_order_is_canonical():
    """the grazing guard bounds sin(corner angle at v0) * cos(theta), so as written it depended on which
    vertex the caller listed first -- a needle whose sharp apex came first was invisible from every direction.  Worlds
    (and rt_upload_scene) rotate every triangle so that v0 faces the longest edge (the largest corner): all three
    rotations of a triangle become the same triple, winding kept, and the needle is visible."""
    apex, b1, b2 = (0.0, 0.0, -2.0), (-0.002, 1.0, -2.0), (0.002, 1.0, -2.0)     # apex angle 0.23 degrees: sin = 0.004 < 2^-7
    rots = [(apex, b1, b2), (b1, b2, apex), (b2, apex, b1)]
    canon = []
    for r in rots:
        tri = np.zeros(1, dtype=O.TRIANGLE_DTYPE)
        tri[0] = (r[0], r[1], r[2], 0)
        before = np.cross(np.array(r[1]) - r[0], np.array(r[2]) - r[0])
        L.orc_canonical_triangle(tri.ctypes.data)
        after = np.cross(tri["v1"][0] - tri["v0"][0], tri["v2"][0] - tri["v0"][0])
        assert np.dot(before, after) > 0                                          # same winding
        canon.append(tri.tobytes())
        # v0 now faces the longest edge
        v0, v1, v2 = tri["v0"][0], tri["v1"][0], tri["v2"][0]
        opp = np.sum((v2 - v1) ** 2)
        assert opp >= np.sum((v1 - v0) ** 2) and opp >= np.sum((v0 - v2) ** 2)
    assert canon[0] == canon[1] == canon[2]
    # a right-angled triangle listed with the right angle first and an equilateral one (three-way tie, broken by the
    # lexicographically smallest vertex) stay as given
    for verts in (((0, 0, 0), (1, 0, 0), (0, 1, 0)), ((0, 0, 0), (2, 0, 0), (1, 1.7320508, 0))):
        tri = np.zeros(1, dtype=O.TRIANGLE_DTYPE)
        tri[0] = (*verts, 0)
        keep = tri.tobytes()
        L.orc_canonical_triangle(tri.ctypes.data)
        assert tri.tobytes() == keep        # (0,0,0) is the lexicographically smallest vertex of both
    # as given with the apex first the test rejects a head-on ray; the world (which canonicalises) sees the needle
    t, p, n = C.c_float(), O.V3(), O.V3()
    tri = np.zeros(1, dtype=O.TRIANGLE_DTYPE)
    tri[0] = (apex, b1, b2, 0)
    args = (O.v3((0.0, 0.5, 0.0)), O.v3((0.0, 0.0, -1.0)), C.c_float(0.001), C.c_float(1e30), C.byref(t), C.byref(p), C.byref(n))
    assert L.orc_triangle_hit(tri.ctypes.data, *args) == 0
    L.orc_canonical_triangle(tri.ctypes.data)
    assert L.orc_triangle_hit(tri.ctypes.data, *args) == 1 and abs(t.value - 2.0) < 1e-6
    mats = np.zeros(1, dtype=O.SPHERE_DTYPE)
    mats[0] = ((0, 0, 0), 1.0, O.LAMBERTIAN, (0.5, 0.5, 0.5), 0.0)
    images = []
    for r in rots:
        tri = np.zeros(1, dtype=O.TRIANGLE_DTYPE)
        tri[0] = (r[0], r[1], r[2], 0)
        cam = O.camera_set_position((0.0, 0.5, 0.0), (0.0, 0.5, -2.0), (0, 1, 0), 1.0, 1.0)   # a 1-degree view of the needle
        acc, _, rc, cnt = O.render_counter(O.World(triangles=tri, materials=mats), cam, O.make_config(48, 48, 2))
        assert cnt["hits"] > 20                                                    # visible
        images.append((acc.tobytes(), rc.tobytes()))
    assert images[0] == images[1] == images[2]


def test_sky():
    """raytracer.zig:569-571."""
    np.testing.assert_allclose(L.orc_sky(O.v3((0, 1, 0))).tup(), (0.5, 0.7, 1.0), atol=1e-7)
    np.testing.assert_allclose(L.orc_sky(O.v3((0, -1, 0))).tup(), (1.0, 1.0, 1.0), atol=1e-7)
    np.testing.assert_allclose(L.orc_sky(O.v3((1, 0, 0))).tup(), (0.75, 0.85, 1.0), atol=1e-7)


def test_cameras():
    """camera.zig: BasicCamera defaults; MovableCamera at the reference pose (raytracer.zig:731)."""
    b = O.camera_basic()
    assert b.as_array().tolist() == [[-2, -1, -1], [4, 0, 0], [0, 2, 0], [0, 0, 0]] and b.subtract_origin == 0
    ro, rd = O.V3(), O.V3()
    L.orc_camera_get_ray(C.byref(b), C.c_float(0.5), C.c_float(0.5), C.byref(ro), C.byref(rd))
    assert rd.tup() == (0.0, 0.0, -1.0)
    m = O.book_camera(1200, 675)
    assert m.subtract_origin == 1
    a = m.as_array().astype(np.float64)
    f, at = np.array([13.0, 2.0, 3.5]), np.zeros(3)
    w = (f - at) / np.linalg.norm(f - at)
    u = np.cross([0, 1, 0], w); u /= np.linalg.norm(u)
    v = np.cross(w, u)
    hh = math.tan(math.radians(20) / 2); hw = (1200 / 675) * hh
    np.testing.assert_allclose(a[3], f)
    np.testing.assert_allclose(a[1], 2 * hw * u, atol=2e-6)
    np.testing.assert_allclose(a[2], 2 * hh * v, atol=2e-6)
    np.testing.assert_allclose(a[0], f - hw * u - hh * v - w, atol=2e-6)
    # centre ray points at the look-at point
    L.orc_camera_get_ray(C.byref(m), C.c_float(0.5), C.c_float(0.5), C.byref(ro), C.byref(rd))
    d = np.array(rd.tup()); d /= np.linalg.norm(d)
    np.testing.assert_allclose(d, -w, atol=2e-6)


# ---- scene generator -----------------------------------------------------------------------------------
def test_random_scene_structure():
    """Random_Scene (raytracer.zig:644-689): counts, fixed tail, hollow pairs, value ranges."""
    s = O.random_scene(11, 0)
    assert len(s) == int(GOLD["book_n"]) == 314
    tail = s[-4:]
    assert tail["center"].tolist() == [[0, -1000, 0], [0, 1, 0], [4, 1, 0], [-4, 1, 0]]
    assert tail["radius"].tolist() == [1000, 1, 1, 1]
    assert tail["kind"].tolist() == [O.LAMBERTIAN, O.DIELECTRIC, O.METAL, O.LAMBERTIAN]
    small = s[:-4]
    assert np.all(small["center"][:, 1] == np.float32(0.2))
    neg = np.where(small["radius"] < 0)[0]
    assert len(neg) == 27
    for k in neg:   # inner sphere first, outer right after, same centre and index (:677-679)
        assert small["radius"][k] == np.float32(-0.15) and small["radius"][k + 1] == np.float32(0.2)
        assert np.array_equal(small["center"][k], small["center"][k + 1]) and small["param"][k] == small["param"][k + 1]
        assert small["kind"][k] == O.DIELECTRIC and 1.5 <= small["param"][k] < 5.0
    metal = small[small["kind"] == O.METAL]
    assert np.all((metal["param"] >= 0) & (metal["param"] < 0.3))
    assert np.all((small["albedo"] >= 0) & (small["albedo"] <= 1.0))
    d = small["center"] - np.float32([4, 0.2, 0])
    assert np.all(np.sqrt((d.astype(np.float64) ** 2).sum(1)) > 0.9)
    # grid cell of every small sphere: centre in [a, a+0.9) x [b, b+0.9)
    fx = small["center"][:, 0] - np.floor(small["center"][:, 0])
    assert np.all(fx < 0.9001)


def test_random_scene_matches_golden():
    s = O.random_scene(11, 0)
    assert np.array_equal(s["center"][:5], GOLD["book_first5_center"])
    assert np.array_equal(s["albedo"][:5], GOLD["book_first5_albedo"])
    assert np.array_equal(s["kind"].astype(np.uint8), GOLD["book_kind"])
    assert np.array_equal(s["radius"], GOLD["book_radius"])
    assert np.array_equal(s["param"], GOLD["book_param"])


def test_reference_bvh_shape():
    """build_BVH depth 5 (raytracer.zig:701): first split is {ground} vs the rest, <= 2^6-1 nodes."""
    w = O.World(spheres=O.random_scene(11, 0))
    assert L.orc_world_bvh_nodes(w.h) == 33


def test_tessellation_counts_and_winding():
    two = O.two_sphere_scene()
    for sl, st in ((12, 8), (20, 20)):
        tri = O.tessellate(two[:1], sl, st)
        assert len(tri) == 2 * sl * (st - 1)            # pole-fan degenerates removed
        c = two["center"][0].astype(np.float64)
        e1 = tri["v1"].astype(np.float64) - tri["v0"]
        e2 = tri["v2"].astype(np.float64) - tri["v0"]
        nrm = np.cross(e1, e2)
        mid = (tri["v0"].astype(np.float64) + tri["v1"] + tri["v2"]) / 3 - c
        assert np.all((nrm * mid).sum(1) > 0)           # outward winding
        r = np.linalg.norm(tri["v0"].astype(np.float64) - c, axis=1)
        np.testing.assert_allclose(r, 0.5, atol=1e-6)
    neg = two[:1].copy(); neg["radius"] = -0.5
    tri = O.tessellate(neg, 12, 8)
    nrm = np.cross(tri["v1"].astype(np.float64) - tri["v0"], tri["v2"].astype(np.float64) - tri["v0"])
    mid = (tri["v0"].astype(np.float64) + tri["v1"] + tri["v2"]) / 3 - neg["center"][0]
    assert np.all((nrm * mid).sum(1) < 0)               # negative radius: normals point inward, like (p-c)/R
    # reference tessellation sizes (main.zig:128-134)
    assert len(O.tessellate(two, 0, 0)) == 2 * 20 * 19 + 2 * 150 * 149


# ---- renders: goldens, indexing, invariants ------------------------------------------------------------
def _golden_scene(name):
    if name == "book":
        return O.World(spheres=O.random_scene(11, 0)), O.book_camera(32, 18)
    if name == "two":
        return O.World(spheres=O.two_sphere_scene()), O.camera_basic()
    two = O.two_sphere_scene()
    extra = np.zeros(2, dtype=O.SPHERE_DTYPE)
    extra[0] = ((-1.0, 0.0, -1.0), 0.5, O.DIELECTRIC, (0.95, 0.95, 0.95), 1.5)
    extra[1] = ((1.0, 0.0, -1.0), 0.5, O.METAL, (0.8, 0.6, 0.2), 0.15)
    src = np.concatenate([two, extra])
    return O.World(triangles=O.tessellate(src, 12, 8), materials=src), O.camera_basic()


@pytest.mark.parametrize("name", ["book", "two", "mesh"])
@pytest.mark.parametrize("depth", [16, 10])
def test_render_matches_golden(name, depth):
    w, cam = _golden_scene(name)
    acc, rgb, rc, cnt = O.render_counter(w, cam, O.make_config(32, 18, 4, max_depth=depth, seed=1))
    key = "%s_d%d" % (name, depth)
    assert np.array_equal(acc.view(np.uint32), GOLD[key + "_accum"].view(np.uint32))
    assert np.array_equal(rgb, GOLD[key + "_rgb8"])
    assert np.array_equal(rc, GOLD[key + "_raycount"].astype(np.uint32))
    assert cnt["rays"] == int(GOLD[key + "_rays"])


def test_render_invariants_no_nan_no_negative():
    """check_nan3 / check_neg3 (raytracer.zig:502-516) as real assertions."""
    w, cam = _golden_scene("book")
    acc, rgb, rc, cnt = O.render_counter(w, cam, O.make_config(32, 18, 8))
    assert np.all(np.isfinite(acc)) and np.all(acc >= 0) and np.all(acc <= 1.0 + 1e-6)
    assert cnt["primary_rays"] == 32 * 18 * 8 and cnt["rays"] == int(rc.sum())
    assert cnt["hits"] + cnt["misses"] == cnt["rays"]
    assert max(k for k, v in enumerate(cnt["rays_by_depth"]) if v) <= 17   # rays exist at depth 0..17 (Q5)


def test_threads_and_pixel_lists_do_not_change_results():
    w, cam = _golden_scene("book")
    a1, _, r1, _ = O.render_counter(w, cam, O.make_config(32, 18, 4, threads=1))
    a8, _, r8, _ = O.render_counter(w, cam, O.make_config(32, 18, 4, threads=8))
    assert np.array_equal(a1.view(np.uint32), a8.view(np.uint32)) and np.array_equal(r1, r8)
    pix = np.array([0, 5, 31, 32, 300, 575], dtype=np.uint32)
    ap, _, rp, _ = O.render_counter(w, cam, O.make_config(32, 18, 4), pixels=pix)
    assert np.array_equal(ap.view(np.uint32), a1[pix].view(np.uint32)) and np.array_equal(rp, r1[pix])


def test_row_indexing_q3():
    """Render: j = ny - jr (raytracer.zig:523).  With q3 off the image is the q3 image shifted one row."""
    w, cam = _golden_scene("two")
    nx, ny = 16, 12
    a_q3, *_ = O.render_counter(w, cam, O.make_config(nx, ny, 2, row_offset_q3=1))
    a_fx, *_ = O.render_counter(w, cam, O.make_config(nx, ny, 2, row_offset_q3=0))
    # same j => same camera rays, but the RNG is keyed by the OUTPUT pixel index, so only compare statistics:
    # row jr of the fixed image looks through j = ny-1-jr, which the q3 image shows at row jr+1
    assert abs(a_q3[nx:].mean() - a_fx[:-nx].mean()) < 0.05


def test_spp_ranges_partition_the_samples():
    """Sample-index sharding: sums over disjoint ranges add up to the whole."""
    w, cam = _golden_scene("book")
    full, _, rc_full, _ = O.render_counter(w, cam, O.make_config(32, 18, 8))
    lo, _, rc_lo, _ = O.render_counter(w, cam, O.make_config(32, 18, 8, spp_begin=0, spp_end=3, spp_total=8))
    hi, _, rc_hi, _ = O.render_counter(w, cam, O.make_config(32, 18, 8, spp_begin=3, spp_end=8, spp_total=8))
    assert np.array_equal(rc_lo + rc_hi, rc_full)
    np.testing.assert_allclose(lo + hi, full, rtol=0, atol=1e-6)
    ch, _, rc_ch, _ = O.render_counter(w, cam, O.make_config(32, 18, 8, spp_chunk=3))
    assert np.array_equal(rc_ch, rc_full)
    np.testing.assert_allclose(ch, full, rtol=0, atol=1e-6)


def test_list_vs_reference_bvh_quirk_q2():
    """The reference BVH (2-axis slabs, negative-radius boxes) is not a perfect accelerator for the list:
    quantify the pixels that differ; everything else must be bit-identical."""
    w = O.World(spheres=O.random_scene(11, 0))
    cam = O.book_camera(96, 54)
    a_l, _, r_l, c_l = O.render_counter(w, cam, O.make_config(96, 54, 4, traversal=O.TRAV_LIST))
    a_b, _, r_b, c_b = O.render_counter(w, cam, O.make_config(96, 54, 4, traversal=O.TRAV_BVH_REF))
    differ = np.any(a_l != a_b, axis=1)
    assert differ.mean() < 0.01
    assert c_b["sphere_tests"] < c_l["sphere_tests"] and c_b["aabb_tests"] > 0 and c_l["aabb_tests"] == 0


def test_triangle_bvh_equals_brute_force():
    """The oracle's triangle BVH is only a speed-up: compare with a brute-force closest hit per ray."""
    rng = np.random.default_rng(5)
    two = O.two_sphere_scene()
    tri = O.tessellate(two, 10, 6)
    w = O.World(triangles=tri, materials=two)
    t, p, n = C.c_float(), O.V3(), O.V3()
    # brute force through orc_triangle_hit vs a 1-sample render's ray counts is indirect; instead check a
    # render with shuffled triangle order gives identical accumulation for lambertian-only scenes.
    perm = rng.permutation(len(tri))
    w2 = O.World(triangles=tri[perm], materials=two)
    a1, _, r1, _ = O.render_counter(w, O.camera_basic(), O.make_config(24, 14, 2))
    a2, _, r2, _ = O.render_counter(w2, O.camera_basic(), O.make_config(24, 14, 2))
    assert np.array_equal(r1, r2)
    assert np.array_equal(a1.view(np.uint32), a2.view(np.uint32))


def test_reference_style_render_runs_and_averages():
    """render_and_write restatement (raytracer.zig:580-618): N threads, seeds i*123, u8 integer mean."""
    w, cam = _golden_scene("two")
    cfg = O.make_config(24, 14, 1, traversal=O.TRAV_BVH_REF)
    img1, c1, _ = O.render_reference_style(w, cam, cfg, 1, 2)
    img3, c3, sec = O.render_reference_style(w, cam, cfg, 3, 2)
    assert img1.shape == (24 * 14, 3) and sec > 0
    assert c3["primary_rays"] == 3 * 24 * 14 * 2 and c1["primary_rays"] == 24 * 14 * 2
    # thread 0 has seed 0 in both runs; the 3-thread mean differs but stays close
    assert abs(img1.astype(float).mean() - img3.astype(float).mean()) < 12


def test_ppm_writer(tmp_path):
    """image.zig:28-34: 'P3\\n{nx} {ny}\\n255\\n' then 'r g b\\n' per pixel."""
    img = np.arange(2 * 3 * 3, dtype=np.uint8).reshape(6, 3)
    f = str(tmp_path / "t.ppm")
    assert L.orc_write_ppm(f.encode(), img.ctypes.data, 3, 2) == 0
    lines = open(f).read().split("\n")
    assert lines[:3] == ["P3", "3 2", "255"] and lines[3] == "0 1 2" and lines[8] == "15 16 17" and lines[9] == ""


def _telephoto_oracle(frm, half_extent, nx, ny):
    """Far camera in BasicCamera.get_ray form (camera.zig:16): unit directions, `half_extent` wide at the origin."""
    frm = np.asarray(frm, np.float64)
    dist = np.linalg.norm(frm)
    w = -frm / dist
    u = np.cross(w, [0.0, 1.0, 0.0]); u /= np.linalg.norm(u)
    v = np.cross(u, w)
    hw = half_extent / dist
    hh = hw * ny / nx
    return O.Camera(O.v3(w - hw * u - hh * v), O.v3(2 * hw * u), O.v3(2 * hh * v), O.v3(frm), 0)


@pytest.mark.parametrize("dist", [0.0, 3.0e3, 3.0e5])
def test_triangle_bvh_of_the_checker_equals_the_plain_loop(dist):
    """The checker's own triangle BVH is only a speed-up: it must return what the loop over every triangle returns,
    also with the ray origins 10^4 .. 10^6 triangle sizes away, where Moller-Trumbore's u, v carry visible rounding
    noise and a plain slab test would cull triangles the exact test accepts (tri_box_hit tests a cone)."""
    rng = np.random.default_rng(3)
    n = 12
    sph = np.zeros(n, dtype=O.SPHERE_DTYPE)
    sph["center"] = rng.uniform(-1.2, 1.2, (n, 3))
    sph["radius"] = rng.choice([0.1, 0.3, 0.5], n)
    sph["kind"] = rng.integers(0, 3, n)
    sph["albedo"] = rng.random((n, 3))
    sph["param"] = np.where(sph["kind"] == 2, 1.5, 0.2)
    tris = O.tessellate(sph, 9, 7)
    w = O.World(triangles=tris, materials=sph)
    nx, ny, spp = 48, 32, 3
    cam = O.camera_basic() if dist == 0.0 else _telephoto_oracle((0.5 * dist, 0.4 * dist, 0.77 * dist), 2.0, nx, ny)
    a = O.render_counter(w, cam, O.make_config(nx, ny, spp, traversal=O.TRAV_LIST))
    b = O.render_counter(w, cam, O.make_config(nx, ny, spp, traversal=O.TRAV_BRUTE))
    assert b[3]["rays"] > b[3]["primary_rays"]
    assert np.array_equal(a[2], b[2]) and a[3]["rays"] == b[3]["rays"]
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
    assert b[3]["tri_tests"] > a[3]["tri_tests"]
    if dist < 1e4:
        assert b[3]["tri_tests"] > 3 * a[3]["tri_tests"]              # and it is a speed-up (at 10^5 the cone spans the scene)
