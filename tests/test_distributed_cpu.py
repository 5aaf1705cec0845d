"""N>1 path on CPU: world_size-2/3 gloo ranks run the same sharding + reduce + tonemap driver the GPU
path uses (raytrace_amd/distributed.py) with the oracle standing in for the per-rank renderer (tests may
call the oracle; the product renderer is HipRangeRenderer).  The reduced image must equal the
single-process render: ray counts exactly, accumulation to float-sum reordering."""
import os
import socket

import numpy as np
import pytest

import oracle_lib as O
from raytrace_amd.distributed import render_sample_sharded, shard_rows, shard_spp, quantise

NX, NY, SPP = 48, 27, 10


def test_shard_spp_partitions_exactly():
    for total in (0, 1, 7, 100, 1024):
        for world in (1, 2, 3, 4, 8):
            ranges = [shard_spp(total, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            for (b0, e0), (b1, e1) in zip(ranges, ranges[1:]):
                assert e0 == b1
            sizes = [e - b for b, e in ranges]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    with pytest.raises(ValueError):
        shard_spp(10, 2, 2)


def test_shard_rows_partitions_exactly():
    for ny in (1, 18, 675, 1080, 2160):
        for world in (1, 2, 4, 8):
            bands = [shard_rows(ny, world, r) for r in range(world)]
            assert bands[0][0] == 0 and bands[-1][1] == ny
            covered = np.zeros(ny, int)
            for y0, y1 in bands:
                covered[y0:y1] += 1
                assert y0 % 8 == 0 or y0 == ny
            assert np.all(covered == 1)


class OracleRangeRenderer:
    """Stand-in for HipRangeRenderer on CPU ranks."""

    def __init__(self, nx, ny):
        import torch
        self.nx, self.ny = nx, ny
        self.world = O.World(spheres=O.random_scene(11, 0))
        self.cam = O.book_camera(nx, ny)
        self._torch = torch

    def device(self):
        return self._torch.device("cpu")

    def render_range(self, accum, b, e, total):
        cfg = O.make_config(self.nx, self.ny, total, spp_begin=b, spp_end=e, spp_total=total, threads=2)
        mean, _, _, cnt = O.render_counter(self.world, self.cam, cfg)
        accum += self._torch.from_numpy((mean * np.float32(total)).reshape(-1))     # the per-rank SUM buffer
        return cnt["rays"], 0.0

    def tonemap(self, accum, total):
        mean = accum / float(total)
        return mean, self._torch.from_numpy(quantise(mean.numpy()))


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = render_sample_sharded(OracleRangeRenderer(NX, NY), SPP, dist)
        rays = torch.tensor([res.rays], dtype=torch.int64)
        dist.all_reduce(rays)
        if rank == 0:
            q.put((res.accum_mean.numpy().copy(), res.rgb8.numpy().copy(), int(rays.item())))
        else:
            assert res.accum_mean is None and res.rgb8 is None
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_sample_sharded_render_matches_single_process(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    mean, rgb8, rays = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    w = O.World(spheres=O.random_scene(11, 0))
    ref_mean, ref_rgb8, _, cnt = O.render_counter(w, O.book_camera(NX, NY), O.make_config(NX, NY, SPP))
    assert rays == cnt["rays"]                                   # sample assignment is exact
    np.testing.assert_allclose(mean.reshape(-1, 3), ref_mean, rtol=0, atol=2e-6)   # float sums reorder across ranks
    assert (rgb8.reshape(-1, 3).astype(int) - ref_rgb8.astype(int)).__abs__().max() <= 1
