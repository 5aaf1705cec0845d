"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol that
include/rt_hip.h declares, and the host mirrors of the reference API (camera, scene generator,
tessellator, PPM writer) agree bit-for-bit with the oracle.  No compute call is made without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as O
import raytrace_amd as R
from raytrace_amd import _lib
from helpers import spheres_to_oracle, triangles_to_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "rt_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s


def test_struct_layouts_match_header():
    assert C.sizeof(_lib.rt_camera) == 52
    assert C.sizeof(_lib.rt_config) == 64 and _lib.rt_config.seed.offset == 32
    assert C.sizeof(_lib.rt_stats) == 8 + 8 + 8 + 160
    assert C.sizeof(_lib.rt_spheres) == 8 + 9 * 8


def test_strerror():
    L = _lib.load()
    assert L.rt_strerror(0) == b"ok"
    assert b"argument" in L.rt_strerror(_lib.RT_E_ARG)
    assert b"HIP" in L.rt_strerror(_lib.RT_E_HIP)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful on a box without a GPU")
def test_no_gpu_fails_loudly():
    """There is no CPU fallback: creating a context without a device is an error, not a silent detour."""
    with pytest.raises(R.RtError) as e:
        R.Context(0)
    assert e.value.code == _lib.RT_E_HIP


def test_packed_node_plane_rounding_only_moves_outward():
    """The packed tree nodes (rt_kernels.h RT_PACKED_NODE_BYTES) carry a byte of a child reference in the low mantissa
    byte of every plane; the host rounds a lower plane down and an upper plane up to the nearest float with that byte.
    Checked here on the host: the byte is there, the plane never moves inward, and it moves by less than 512 ulps --
    across signs, zero, denormals and large magnitudes."""
    L = _lib.load()
    rng = np.random.default_rng(3)
    vals = np.concatenate([
        (rng.standard_normal(4000) * 10.0 ** rng.integers(-6, 7, 4000)).astype(np.float32),
        np.array([0.0, -0.0, 1e-45, -1e-45, 3e-39, -3e-39, 1.17549435e-38, -1.17549435e-38, 1.0, -1.0, 255.0, -1000.0, 1e30, -1e30],
                 dtype=np.float32)])
    for v in vals:
        for payload in (0, 1, 127, 128, 255, int(rng.integers(0, 256))):
            for down in (1, 0):
                r = np.float32(L.rt_debug_round_plane(float(v), payload, down))
                assert int(r.view(np.uint32)) & 0xFF == payload, (v, payload, down, r)
                assert (r <= v) if down else (r >= v), (v, payload, down, r)
                # at most two blocks of 256 floats away (ordered-integer distance)
                def key(x):
                    b = int(np.float32(x).view(np.uint32))
                    return (0x80000000 - (b & 0x7FFFFFFF)) if b & 0x80000000 else (0x80000000 + b)
                assert abs(key(r) - key(v)) < 512, (v, payload, down, r)


def test_packed_node_plane_without_a_finite_neighbour_is_refused():
    """planes at -inf (lower) / +inf (upper), NaN or in the last block of floats before +-FLT_MAX have no outward float carrying the
    payload; the packer reports that (NaN here, RT_E_STATE from rt_upload_scene) instead of returning a plane whose
    low byte is not the child ref."""
    L = _lib.load()
    fmax = float(np.finfo(np.float32).max)
    for v, down in ((-np.inf, 1), (np.inf, 0), (np.nan, 1), (np.nan, 0), (-fmax, 1), (fmax, 0)):
        for payload in (1, 77, 254):
            assert np.isnan(L.rt_debug_round_plane(float(v), payload, down)), (v, payload, down)
    # the inward side of +-FLT_MAX is fine
    for payload in (0, 1, 254):
        r = np.float32(L.rt_debug_round_plane(fmax, payload, 1))
        assert np.isfinite(r) and r <= fmax and int(r.view(np.uint32)) & 0xFF == payload
        r = np.float32(L.rt_debug_round_plane(-fmax, payload, 0))
        assert np.isfinite(r) and r >= -fmax and int(r.view(np.uint32)) & 0xFF == payload


def test_cu_mask_argument_check():
    """rt_set_cu_mask's guard: a mask that leaves one of the 8 XCDs of an MI355X without
    a CU (bit i lies on XCD i % 8) crashes the ROCm 7.2 runtime, so it is refused before any HIP call.  Pure host
    arithmetic: no GPU, and no run reproduces the crash."""
    L = _lib.load()

    def check(words, cus=256):
        m = (C.c_uint32 * len(words))(*words)
        return L.rt_check_cu_mask(m, len(words), cus)

    assert check([0xFFFFFFFF] * 8) == 256
    assert check([0xFFFFFFFF]) == 32                         # 4 CUs per XCD: the "@1/8 GPU" mask
    assert check([0xFF]) == 8                                # one CU per XCD
    assert check([0x01010101]) == _lib.RT_E_ARG              # XCD 0 only
    assert check([0x01010101] * 8) == _lib.RT_E_ARG          # all 32 CUs of XCD 0: "one XCD"
    assert check([0x7F]) == _lib.RT_E_ARG                    # first 7 CUs: XCD 7 empty
    assert check([0xFFFFFFFE & ~0x01010100]) == _lib.RT_E_ARG    # everything in the word but XCD 0
    assert check([0, 0xFF]) == 8
    assert check([0]) == _lib.RT_E_ARG
    assert check([0] * 8 + [0xFFFFFFFF]) == _lib.RT_E_ARG    # only bits beyond the device's CUs
    assert check([0xFF] + [0] * 7 + [0xFFFFFFFF]) == 8       # bits beyond the device's CUs are ignored
    assert L.rt_check_cu_mask(None, 1, 256) == _lib.RT_E_ARG
    assert check([0xFF] * 65) == _lib.RT_E_ARG
    assert check([0x3], cus=2) == 2                          # a device with fewer CUs than XCD classes: no class rule


def test_cameras_match_oracle_bitwise():
    assert np.array_equal(R.BasicCamera().as_array(), O.camera_basic().as_array())
    for nx, ny in ((1200, 675), (1920, 1080), (400, 225), (64, 36)):
        a = R.book_camera(nx, ny)
        b = O.book_camera(nx, ny)
        assert np.array_equal(a.as_array().view(np.uint32), b.as_array().view(np.uint32))
        assert a.to_abi().subtract_origin == 1
    # the GUI's odd call (main.zig:308-315): forward vector as look-at, pos + (0,1,0) as up
    pos, fwd = (3.0, 1.5, -2.0), (0.3, -0.1, 0.9)
    a = R.MovableCamera().SetPosition(pos, fwd, (3.0, 2.5, -2.0), 45.0, 1200 / 675)
    b = O.camera_set_position(pos, fwd, (3.0, 2.5, -2.0), 45.0, 1200 / 675)
    assert np.array_equal(a.as_array().view(np.uint32), b.as_array().view(np.uint32))


@pytest.mark.parametrize("half,seed", [(11, 0), (3, 0), (5, 123), (11, 246)])
def test_random_scene_matches_oracle_bitwise(half, seed):
    a = spheres_to_oracle(R.Random_Scene(half, seed))
    b = O.random_scene(half, seed)
    assert len(a) == len(b)
    assert a.tobytes() == b.tobytes()


def test_two_sphere_scene_matches_oracle():
    assert spheres_to_oracle(R.Two_Sphere_Scene()).tobytes() == O.two_sphere_scene().tobytes()


def test_random_scene_capacity_error():
    L = _lib.load()
    s = R.Spheres.empty(4)
    assert L.rt_random_scene(11, 0, 4, *s._out_ptrs()) == _lib.RT_E_CAP


@pytest.mark.parametrize("sl,st", [(12, 8), (0, 0)])
def test_tessellation_matches_oracle_bitwise(sl, st):
    src = R.Random_Scene(2, 0)
    a = triangles_to_oracle(R.Tessellate_Spheres(src, sl, st))
    b = O.tessellate(spheres_to_oracle(src), sl, st)
    assert len(a) == len(b) and a.tobytes() == b.tobytes()


def test_ppm_writer_matches_reference_format(tmp_path):
    """write_ppm_image, image.zig:28-34."""
    img = (np.arange(4 * 2 * 3) * 9 % 256).astype(np.uint8).reshape(8, 3)
    f = str(tmp_path / "a.ppm")
    g = str(tmp_path / "b.ppm")
    R.write_ppm_image(f, img, 4, 2)
    assert O.lib().orc_write_ppm(g.encode(), img.ctypes.data, 4, 2) == 0
    txt = open(f).read()
    assert txt == open(g).read()
    assert txt.startswith("P3\n4 2\n255\n") and txt.count("\n") == 3 + 8


def test_config_defaults_are_the_reference_constants():
    cfg = R.make_config(1200, 675, 25)
    assert cfg.max_depth == 16 and abs(cfg.t_min - 0.001) < 1e-9 and cfg.row_offset_q3 == 1
    assert (cfg.spp_begin, cfg.spp_end, cfg.spp_total) == (0, 25, 25)


def test_host_mirrors_refuse_runaway_arguments():
    """Arguments that would make a host-side mirror allocate without bound come back as RT_E_ARG at once
    (rt_random_scene visits (2 half_size)^2 cells whatever `cap` is; rt_tessellate_spheres makes 2 slices stacks
    triangles per sphere), and a NULL vector leaves rt_camera_set_position's camera untouched."""
    L = _lib.load()
    s = R.Spheres.empty(4)
    assert L.rt_random_scene(2049, 0, 4, *s._out_ptrs()) == _lib.RT_E_ARG
    assert L.rt_random_scene(0xFFFFFFFF, 0, 4, *s._out_ptrs()) == _lib.RT_E_ARG
    two = R.Two_Sphere_Scene()
    sv = two.view()
    assert L.rt_tessellate_spheres(C.byref(sv), 100000, 100000, 0, None, None, None, None) == _lib.RT_E_ARG
    assert L.rt_tessellate_spheres(C.byref(sv), 4100, 4100, 0, None, None, None, None) == _lib.RT_E_ARG   # 2 x 33.6 M >= 2^26
    assert L.rt_tessellate_spheres(C.byref(sv), 8, 8, 0, None, None, None, None) > 0
    cam = R.BasicCamera()
    before = bytes(cam._c)
    a3 = C.c_float * 3
    raw = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float)(("rt_camera_set_position", L))
    at, up = a3(0, 0, 0), a3(0, 1, 0)
    raw(C.addressof(cam._c), None, C.addressof(at), C.addressof(up), 20.0, 1.5)
    assert bytes(cam._c) == before
