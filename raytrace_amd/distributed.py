"""Multi-GPU driver: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI).

The path shards the way the reference's own thread fan-out does (raytracer.zig:594-617: every thread
renders the whole image with different samples, the images are then averaged): rank r renders the
global sample indices [begin_r, end_r) of every pixel into its own f32 accumulation SUM buffer, one
reduce (SUM, root 0) adds the buffers, and rank 0 divides by spp_total and tonemaps.  The counter RNG is
keyed by (pixel, global sample index), so the set of (pixel, sample) pairs is exactly
{0..P-1} x {0..spp-1} whatever the world size.  The reduce is the only collective: 24.9 MB at 1080p,
99.5 MB at 4K -- about a millisecond on xGMI next to a render of seconds, so it is a single blocking
call at the end, not a bucketed/overlapped scheme.

Image-tile sharding (row bands, no reduction, a gather instead) is provided for callers that want
it; sample sharding is the default because it balances perfectly (sky vs glass tiles do not).

`render_range` is injected so that the sharding/reduction logic can be exercised on CPU ranks (gloo)
with a stand-in renderer; the product renderer is HipRangeRenderer.
"""
from dataclasses import dataclass

import numpy as np


def shard_spp(spp_total, world, rank):
    """Contiguous sample-index range of `rank`: sizes differ by at most one, low ranks take the extra."""
    if world < 1 or not (0 <= rank < world) or spp_total < 0:
        raise ValueError("shard_spp(%r, %r, %r)" % (spp_total, world, rank))
    base, extra = divmod(spp_total, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_rows(ny, world, rank, align=8):
    """Row band [y0, y1) of `rank`, aligned to the kernel's 8x8 work tiles where possible."""
    if world < 1 or not (0 <= rank < world) or ny < 0:
        raise ValueError("shard_rows(%r, %r, %r)" % (ny, world, rank))
    units = (ny + align - 1) // align
    b, e = shard_spp(units, world, rank)
    return min(b * align, ny), min(e * align, ny)


@dataclass
class ShardedResult:
    accum_mean: object      # rank 0: [P*3] mean radiance (torch tensor); other ranks: None
    rgb8: object            # rank 0: [P*3] u8; other ranks: None
    rays: int               # this rank's rays
    kernel_ms: float
    reduce_ms: float = 0.0  # wall time of the exchange step on this rank (waiting for the slowest rank included)


class HipRangeRenderer:
    """The product renderer of one rank: rt_render_device on this rank's GPU."""

    def __init__(self, ctx, cam, nx, ny, max_depth=16, seed=1, t_min=0.001, spp_chunk=None):
        from ._lib import RT_SPP_CHUNK_AUTO
        self.spp_chunk = RT_SPP_CHUNK_AUTO if spp_chunk is None else spp_chunk
        self.ctx, self.cam = ctx, cam
        self.nx, self.ny, self.max_depth, self.seed, self.t_min = nx, ny, max_depth, seed, t_min

    def device(self):
        import torch
        return torch.device("cuda", self.ctx.device)

    def render_range(self, accum, spp_begin, spp_end, spp_total):
        """Adds the radiance sum of samples [spp_begin, spp_end) into `accum` (f32 [P*3] on this GPU)."""
        import torch
        from .api import make_config
        cfg = make_config(self.nx, self.ny, spp_total, max_depth=self.max_depth, seed=self.seed, t_min=self.t_min,
                          spp_begin=spp_begin, spp_end=spp_end, spp_total=spp_total, spp_chunk=self.spp_chunk)
        rays, _prim, ms, _ = self.ctx.render_device(self.cam, cfg, accum.data_ptr(), 0,
                                                    torch.cuda.current_stream().cuda_stream, want_stats=True)
        return rays, ms

    def tonemap(self, accum, spp_total):
        import torch
        n = self.nx * self.ny
        mean = torch.empty_like(accum)
        rgb8 = torch.empty(n * 3, dtype=torch.uint8, device=accum.device)
        self.ctx.tonemap_device(accum.data_ptr(), n, spp_total, mean.data_ptr(), rgb8.data_ptr(),
                                torch.cuda.current_stream().cuda_stream)
        return mean, rgb8


def render_sample_sharded(renderer, spp_total, dist=None, accum=None):
    """Render spp_total samples per pixel over all ranks of the default process group.

    renderer: object with .nx, .ny, .device(), .render_range(accum, b, e, total) -> (rays, ms) and
              .tonemap(accum, total) -> (mean, rgb8)
    dist:     torch.distributed (initialised) or None for a single process."""
    import torch
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    n = renderer.nx * renderer.ny * 3
    if accum is None:
        accum = torch.zeros(n, dtype=torch.float32, device=renderer.device())
    else:
        accum.zero_()
    begin, end = shard_spp(spp_total, world, rank)
    rays, ms = renderer.render_range(accum, begin, end, spp_total)
    reduce_ms = 0.0
    if world > 1:
        import time
        t0 = time.perf_counter()
        dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)     # RCCL over xGMI: the path's one exchange step
        if accum.is_cuda:
            torch.cuda.synchronize()
        reduce_ms = (time.perf_counter() - t0) * 1e3
    if rank == 0:
        mean, rgb8 = renderer.tonemap(accum, spp_total)
        return ShardedResult(mean, rgb8, rays, ms, reduce_ms)
    return ShardedResult(None, None, rays, ms, reduce_ms)


def timed_sharded_steps(renderer, spp_total, dist=None, steps=3, warmup=1, accum=None):
    """bench.py's timed region, for any workload: `warmup` untimed then exactly `steps` timed calls of
    render_sample_sharded, bracketed by synchronize + barrier on both sides; the time is the MAX over ranks, the rays
    the SUM over ranks.  Every rank returns the same dict:
        ms_per_step, rays_per_step (all ranks), rays_rank (this rank), kernel_ms (this rank's mean), reduce_ms (this
        rank's mean wall time of the exchange step), world, backend."""
    import time
    import torch
    world = dist.get_world_size() if dist is not None else 1
    dev = renderer.device()
    on_gpu = dev.type == "cuda"

    def fence():
        if on_gpu:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    if accum is None:
        accum = torch.zeros(renderer.nx * renderer.ny * 3, dtype=torch.float32, device=dev)
    kernel_ms, reduce_ms, rays = [], [], 0
    for _ in range(warmup):
        render_sample_sharded(renderer, spp_total, dist, accum)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = render_sample_sharded(renderer, spp_total, dist, accum)
        kernel_ms.append(res.kernel_ms)
        reduce_ms.append(res.reduce_ms)
        rays = res.rays
    fence()
    dt = time.perf_counter() - t0
    tot = torch.tensor([float(rays), 0.0], dtype=torch.float64, device=dev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    return {"ms_per_step": dt / max(1, steps) * 1e3, "rays_per_step": float(tot[0].item()), "rays_rank": rays,
            "kernel_ms": float(np.mean(kernel_ms)) if kernel_ms else 0.0,
            "reduce_ms": float(np.mean(reduce_ms)) if reduce_ms else 0.0,
            "world": world, "backend": dist.get_backend() if dist is not None else "none", "steps": steps, "seconds": dt}


def quantise(mean):
    """Host tonemap, trunc(255.99*sqrt(mean)) (raytracer.zig:527,545-549), for stand-in renderers."""
    return np.clip((np.float32(255.99) * np.sqrt(np.asarray(mean, dtype=np.float32))).astype(np.int32), 0, 255).astype(np.uint8)
