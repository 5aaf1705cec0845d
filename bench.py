#!/usr/bin/env python3
"""bench.py -- Mrays/s (all bounces) of the HIP trace path on the BASELINE.json workload.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full pass of the trace path over the workload with everything resident in HBM:
zero the accumulation buffer, trace every (pixel, sample) of this rank's sample range
(Render/trace/color, raytracer.zig:517-572), sum the per-GPU accumulation buffers onto rank 0 (RCCL
reduce; the analogue of the reference's thread join + average, raytracer.zig:605-617) and tonemap.
Workload at N=1: BASELINE.json configs[2] -- book final scene, 1920x1080, 100 spp, depth cap 16.
At N>1 every rank renders its own 100 global sample indices [100*rank, 100*(rank+1)) of a 100*N-spp
image (weak scaling: per-GPU work fixed).

Prints ONE JSON line on rank 0.  The CPU baseline leg (rank 0, N=1 only) times the oracle's
restatement of render_and_write on the host cores; it is the only use of oracle/ here.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3   # MI355X vector FP32 peak (MI355X_MICROARCH.md, chip-level parameters)

WORKLOADS = {
    # name: (scene, nx, ny, spp, max_depth, opcounts file)
    "c3": ("book", 1920, 1080, 100, 16, "opcounts_c3.json"),
    "c2": ("two", 1920, 1080, 100, 16, "opcounts_c2.json"),
    "c1": ("book", 400, 225, 10, 10, "opcounts_c1.json"),
    # BASELINE configs[4]: STRONG scaling -- the 1024 samples are split over the ranks.  flops/ray: the C3 figure
    # (same scene, camera and depth cap; only the pixel grid is finer)
    "c5": ("book", 3840, 2160, 1024, 16, "opcounts_c3.json"),
}
STRONG = {"c5"}
WORKLOAD_DESC = {
    "c3": "book final scene (Random_Scene(11), sfc64 seed 0) 1920x1080 100spp depth16",
    "c2": "single sphere + ground lambertian 1920x1080 100spp depth16",
    "c1": "book final scene 400x225 10spp depth10",
    "c5": "book final scene 3840x2160 1024spp depth16, samples split over the GPUs",
}


def cpu_baseline(workload, budget_threads):
    """Reference-style CPU render (oracle: threads x whole image x 1 spp each, sfc64 seeds i*123,
    reference BVH) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    scene, nx, ny, _spp, depth, _ = WORKLOADS[workload]
    sph = O.random_scene(11, 0) if scene == "book" else O.two_sphere_scene()
    cam = O.book_camera(nx, ny) if scene == "book" else O.camera_basic()
    w = O.World(spheres=sph)
    cfg = O.make_config(nx, ny, 1, max_depth=depth, traversal=O.TRAV_BVH_REF if scene == "book" else O.TRAV_LIST)
    _, cnt, sec = O.render_reference_style(w, cam, cfg, budget_threads, 1)
    return {"value": cnt["rays"] / sec / 1e6, "unit": "Mrays/s", "cores": budget_threads, "kind": "port",
            "sample": "%dx%d, 1 spp per thread x %d threads (render_and_write restatement, sfc64 seeds i*123, "
                      "reference BVH), %.1f s" % (nx, ny, budget_threads, sec)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()

    import numpy as np
    import torch
    import raytrace_amd as R

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # one rank per GPU; RT_BENCH_BACKEND=gloo with fewer GPUs than ranks is a rehearsal mode (ranks share GPUs)
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    device_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    scene, nx, ny, spp, depth, opfile = WORKLOADS[args.workload]
    spheres = R.Random_Scene(11, 0) if scene == "book" else R.Two_Sphere_Scene()
    cam = R.book_camera(nx, ny) if scene == "book" else R.BasicCamera()
    ctx = R.Context(device_index)
    ctx.upload_scene(spheres)
    from raytrace_amd.distributed import HipRangeRenderer, render_sample_sharded
    npix = nx * ny
    renderer = HipRangeRenderer(ctx, cam, nx, ny, max_depth=depth, seed=1)
    accum = torch.zeros(npix * 3, dtype=torch.float32, device="cuda")

    strong = args.workload in STRONG
    spp_total = spp if strong else spp * world      # weak: every GPU adds its own `spp` samples to the image
    kernel_ms = []
    reduce_ms = []
    rays_per_step = [0]

    def step(record):
        # zero accum -> trace this rank's samples [rank*spp, (rank+1)*spp) -> RCCL reduce -> tonemap on rank 0
        res = render_sample_sharded(renderer, spp_total, dist, accum)
        if record:
            kernel_ms.append(res.kernel_ms)
            reduce_ms.append(res.reduce_ms)
            rays_per_step[0] = res.rays

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    dt = time.perf_counter() - t0

    rays_total = torch.tensor([float(rays_per_step[0])], dtype=torch.float64, device="cuda")
    t_max = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(rays_total, op=dist.ReduceOp.SUM)
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    rays_all = float(rays_total.item())          # rays of ONE step over all ranks
    dt = float(t_max.item())

    if rank == 0:
        value = rays_all * args.steps / dt / 1e6
        with open(os.path.join(ROOT, "tests", "golden", opfile)) as f:
            op = json.load(f)
        fpr = op["flops_per_ray"]
        k_ms = float(np.mean(kernel_ms))
        # HBM bytes per launch of the trace kernel come from a separate rocprofv3 --pmc pass (FETCH_SIZE,
        # WRITE_SIZE; tools/profile_bench.sh), committed under profiles/: counters cannot be read in-process.
        traffic, traffic_src = None, None
        pmc_path = os.path.join(ROOT, "profiles", "latest_%s_pmc.json" % args.workload)
        if world == 1 and os.path.exists(pmc_path):
            with open(pmc_path) as f:
                d = json.load(f).get("derived", {})
            if "hbm_write_bytes" in d and "hbm_read_bytes_x2_correction" in d:
                # MI355X_MICROARCH.md, HBM: FETCH_SIZE (KiB) counts half the bytes on gfx950 -> doubled; WRITE_SIZE as is
                traffic = d["hbm_read_bytes_x2_correction"] + d["hbm_write_bytes"]
                traffic_src = "profiles/latest_%s_pmc.json (2 x FETCH_SIZE KiB + WRITE_SIZE KiB, separate --pmc passes)" % args.workload
        achieved = rays_per_step[0] * fpr / (k_ms * 1e-3) / 1e12
        # executed (hardware-counted) VALU work of the same launch, from the same committed PMC pass: active lanes
        # summed over all VALU instructions, against the chip's lane-issue peak (256 CU x 128 lanes x 2.4 GHz)
        executed = None
        if world == 1 and os.path.exists(pmc_path):
            with open(pmc_path) as f:
                d = json.load(f).get("derived", {})
            if "valu_lane_ops_per_launch" in d:
                lane_rate = d["valu_lane_ops_per_launch"] / (k_ms * 1e-3) / 1e12
                executed = {"valu_T_lane_ops_per_s": lane_rate, "peak_T_lane_ops_per_s": PEAK_FP32_TFLOPS / 2,
                            "frac": lane_rate / (PEAK_FP32_TFLOPS / 2),
                            "valu_wave_insts_per_64_rays": d.get("valu_wave_insts_per_64_rays"),
                            "lane_utilisation": d.get("valu_lane_utilisation"),
                            "source": "profiles/latest_%s_pmc.json (SQ_THREAD_CYCLES_VALU, SQ_INSTS_VALU)" % args.workload}
        out = {
            "metric": "Mrays/sec (all bounces) book scene 1080p 100spp; %FP32 VALU roofline @1/8 GPU",   # BASELINE.json, verbatim
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOAD_DESC[args.workload], "n_spheres": len(spheres), "nx": nx, "ny": ny,
                       "spp_per_gpu": spp_total / world, "spp_total": spp_total, "max_depth": depth, "seed": 1,
                       "rays_per_step": rays_all, "reduce_ms_rank0": float(np.mean(reduce_ms)) if world > 1 else 0.0,
                       "parallelism": "spp-range x%d + RCCL reduce" % world if world > 1 else "1 GPU"},
            "roofline": {"bound": "valu_fp32", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "executed": executed,
                         "kernel": ctx.last_kernel_name(), "kernel_ms": k_ms,
                         "algorithmic_flops_per_ray": fpr, "rays_per_launch": rays_per_step[0],
                         "kernel_mrays_per_s": rays_per_step[0] / (k_ms * 1e-3) / 1e6,
                         "note": "FP32 vector roofline (not HBM/MFMA): algorithmic flops per ray are those of the "
                                 "reference's own BVH traversal on this config (tests/golden/%s, SURVEY 8(d))" % opfile},
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = args.cpu_threads or min(16, os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(args.workload, threads)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
