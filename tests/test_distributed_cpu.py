"""N>1 path on CPU: world_size-2/3 gloo ranks run the same sharding + reduce + tonemap driver the GPU
path uses (raytrace_amd/distributed.py) with the oracle standing in for the per-rank renderer (tests may
call the oracle; the product renderer is HipRangeRenderer).  The reduced image must equal the
single-process render: ray counts exactly, accumulation to float-sum reordering."""
import os
import socket

import numpy as np
import pytest

import oracle_lib as O
from raytrace_amd.distributed import render_sample_sharded, shard_rows, shard_spp, quantise, timed_sharded_steps

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


def _bench_worker(rank, world, port, q):
    """bench.py's timed region at world 2 (timed_sharded_steps is the function bench.py calls for the weak headline and
    for the c5_strong block): the STRONG split -- SPP samples divided by shard_spp -- on a tiny image."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        t = timed_sharded_steps(OracleRangeRenderer(NX, NY), SPP, dist, steps=2, warmup=1)
        q.put((rank, t))
    finally:
        dist.destroy_process_group()


def test_bench_timed_region_two_ranks_strong_split():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    w = O.World(spheres=O.random_scene(11, 0))
    _, _, _, cnt = O.render_counter(w, O.book_camera(NX, NY), O.make_config(NX, NY, SPP))
    a, b = got[0], got[1]
    # every rank holds the same whole-job figures: rays summed over ranks == the single-process render, MAX time
    assert a["rays_per_step"] == b["rays_per_step"] == float(cnt["rays"])
    assert a["rays_rank"] + b["rays_rank"] == cnt["rays"] and a["rays_rank"] > 0 and b["rays_rank"] > 0
    assert a["ms_per_step"] == b["ms_per_step"] > 0 and a["steps"] == 2 and a["world"] == 2 and a["backend"] == "gloo"


def test_bench_line_carries_c5_strong_only_for_unnamed_multi_gpu_runs():
    """bench.py's contract, checked on its text: the c5_strong block is built when world > 1 and no
    --workload was named, from WORKLOADS['c5'] = BASELINE configs[4]."""
    import ast
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "bench.py")).read()
    ast.parse(src)
    assert "if world > 1 and not workload_given:" in src and 'out["c5_strong"] = c5' in src
    assert '"c5": ("book", 3840, 2160, 1024, 16' in src and "timed_sharded_steps(r5, spp5, dist" in src
