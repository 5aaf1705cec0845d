"""rt_render_device_noalloc (include/rt_hip.h): the trace pass with its scratch -- work-queue head, counters,
partial-sum planes -- in the library's fixed device array instead of ctx-owned hipMalloc memory.

Checked through the C ABI against rt_render_device, the entry point it is a variant of, on the same inputs: ray
counts (integer work) bit-exact, the accumulation buffer bit-identical (same kernels, same chunking, same sum
order; the bar is 1e-3), on two calls in a row, the second with a smaller input, so that whatever the first call
left in the shared array must not leak into the second.
"""
import numpy as np
import pytest

import raytrace_amd as R
from raytrace_amd import _lib

pytestmark = pytest.mark.gpu

TOL_ACCUM = 1e-3


def _pair(ctx, cam, cfg):
    """Same cfg through rt_render_device and rt_render_device_noalloc; returns (accum, raycount) of each."""
    import torch
    n = cfg.nx * cfg.ny
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for noalloc in (False, True):
        acc = torch.zeros(n * 3, dtype=torch.float32, device="cuda")
        rc = torch.zeros(n, dtype=torch.int32, device="cuda")
        if noalloc:
            ctx.render_device_noalloc(cam, cfg, acc.data_ptr(), rc.data_ptr(), st)
        else:
            ctx.render_device(cam, cfg, acc.data_ptr(), rc.data_ptr(), st)
        torch.cuda.synchronize()
        out.append((acc.cpu().numpy(), rc.cpu().numpy().astype(np.uint32)))
    return out


def _check(ref, got, spp):
    (ra, rc), (ga, gc) = ref, got
    assert np.array_equal(gc, rc), "per-pixel ray counts differ"
    assert gc.min() >= spp
    assert float(np.abs(ga - ra).max()) <= TOL_ACCUM
    assert np.array_equal(ga.view(np.uint32), ra.view(np.uint32))      # in fact the same bits
    assert np.all(np.isfinite(ga)) and ga.min() >= 0


def test_noalloc_matches_render_device_twice_second_call_smaller():
    ctx = R.Context(0)
    try:
        book = R.Random_Scene(11, 0)
        ctx.upload_scene(book)
        # 1st call: 320x180, 12 spp in chunks of 4 (three partial planes in the scratch array, BVH kernel from LDS)
        nx, ny, spp = 320, 180, 12
        cfg = R.make_config(nx, ny, spp, spp_chunk=4)
        _check(*_pair(ctx, R.book_camera(nx, ny), cfg), spp)
        # 2nd call, smaller input: 96x54, 5 spp, chunks of 2 starting inside a chunk; planes overlap the 1st call's
        nx, ny, spp = 96, 54, 5
        cfg = R.make_config(nx, ny, spp, spp_begin=1, spp_end=5, spp_total=5, spp_chunk=2)
        _check(*_pair(ctx, R.book_camera(nx, ny), cfg), 4)
        # 3rd: another kernel (brute force, two spheres), no chunking, a tile: only queue head and counters are used
        ctx.upload_scene(R.Two_Sphere_Scene())
        cfg = R.make_config(64, 40, 3, tile=(8, 8, 40, 32))
        (ra, rc), (ga, gc) = _pair(ctx, R.BasicCamera(), cfg)
        assert np.array_equal(gc, rc) and np.array_equal(ga.view(np.uint32), ra.view(np.uint32))
    finally:
        ctx.close()


def test_noalloc_rejects_what_does_not_fit_and_auto_stays_within_the_scratch():
    ctx = R.Context(0)
    try:
        ctx.upload_scene(R.Two_Sphere_Scene())
        import torch
        nx, ny = 1920, 1080                                   # one plane = 24.9 MB: two fit in 64 MiB, three do not
        acc = torch.zeros(nx * ny * 3, dtype=torch.float32, device="cuda")
        cam = R.BasicCamera()
        with pytest.raises(R.RtError) as e:
            ctx.render_device_noalloc(cam, R.make_config(nx, ny, 6, spp_chunk=2), acc.data_ptr())
        assert e.value.code == _lib.RT_E_CAP
        assert float(acc.abs().max()) == 0.0                 # nothing was enqueued
        # AUTO takes no more planes than fit; two explicit planes fit; both give the image of rt_render_device
        ref = torch.zeros_like(acc)
        ctx.render_device(cam, R.make_config(nx, ny, 6, spp_chunk=3), ref.data_ptr())
        torch.cuda.synchronize()
        for chunk in (3, R.RT_SPP_CHUNK_AUTO):
            acc.zero_()
            ctx.render_device_noalloc(cam, R.make_config(nx, ny, 6, spp_chunk=chunk), acc.data_ptr())
            torch.cuda.synchronize()
            assert float((acc - ref).abs().max()) <= TOL_ACCUM
        # a second context on the same GPU shares the array (calls one after the other are fine)
        with R.Context(0) as other:
            other.upload_scene(R.Two_Sphere_Scene())
            acc.zero_()
            other.render_device_noalloc(cam, R.make_config(nx, ny, 6, spp_chunk=3), acc.data_ptr())
            torch.cuda.synchronize()
            assert torch.equal(acc, ref)
    finally:
        ctx.close()


def test_noalloc_allocates_nothing_and_can_be_captured_into_a_graph():
    """prove the claim, not just the result.  (a) free device memory is unchanged across a plain call;
    (b) the call is captured into a graph on a side stream (capture_error_mode "global": a hipMalloc anywhere in the
    process during capture would invalidate it), with spp_chunk 0 and with an explicit chunk, and two replays into a zeroed
    buffer each give rt_render_device's bits."""
    import torch
    ctx = R.Context(0)
    try:
        ctx.upload_scene(R.Random_Scene(11, 0))
        nx, ny, spp = 160, 90, 8
        cam = R.book_camera(nx, ny)
        n = nx * ny
        for chunk in (0, 4):
            cfg = R.make_config(nx, ny, spp, spp_chunk=chunk)
            ref = torch.zeros(n * 3, dtype=torch.float32, device="cuda")
            ref_rc = torch.zeros(n, dtype=torch.int32, device="cuda")
            ctx.render_device(cam, cfg, ref.data_ptr(), ref_rc.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            acc = torch.zeros_like(ref)
            rc = torch.zeros_like(ref_rc)
            # (a) a warm-up call, then one bracketed by hipMemGetInfo
            ctx.render_device_noalloc(cam, cfg, acc.data_ptr(), rc.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            free0, _ = torch.cuda.mem_get_info()
            ctx.render_device_noalloc(cam, cfg, acc.data_ptr(), rc.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            free1, _ = torch.cuda.mem_get_info()
            assert free1 == free0, (free0, free1)
            # (b) graph capture on a side stream
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side, capture_error_mode="global"):
                ctx.render_device_noalloc(cam, cfg, acc.data_ptr(), rc.data_ptr(), torch.cuda.current_stream().cuda_stream)
            for _ in range(2):
                acc.zero_(); rc.zero_()
                torch.cuda.synchronize()
                g.replay()
                torch.cuda.synchronize()
                assert torch.equal(rc, ref_rc)
                assert torch.equal(acc.view(torch.int32), ref.view(torch.int32))
            del g
    finally:
        ctx.close()
