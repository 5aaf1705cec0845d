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
image (weak scaling: per-GPU work fixed) -- the headline, metric-consistent with N=1 -- and, when no
--workload is named, the same line carries a `c5_strong` block: BASELINE.json configs[4], 3840x2160 with
1024 spp SPLIT over the ranks (the reference's thread fan-out, raytracer.zig:588-617), a few steps of its own.

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
    # BASELINE configs[3]: the book scene's spheres tessellated as the reference's raster preview does (main.zig:128-138)
    "c4": ("mesh", 1920, 1080, 256, 16, "opcounts_c4.json"),
    # BASELINE configs[4]: STRONG scaling -- the 1024 samples are split over the ranks.  flops/ray: the C3 figure
    # (same scene, camera and depth cap; only the pixel grid is finer)
    "c5": ("book", 3840, 2160, 1024, 16, "opcounts_c3.json"),
}
STRONG = {"c5"}
WORKLOAD_DESC = {
    "c3": "book final scene (Random_Scene(11), sfc64 seed 0) 1920x1080 100spp depth16",
    "c2": "single sphere + ground lambertian 1920x1080 100spp depth16",
    "c1": "book final scene 400x225 10spp depth10",
    "c4": "procedural triangle-mesh scene (book scene tessellated, 285 520 triangles) 1920x1080 256spp depth16",
    "c5": "book final scene 3840x2160 1024spp depth16, samples split over the GPUs",
}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_threads():
    """Hardware threads this process may use: the scheduler affinity, capped by the GPU box's CPU share per GPU
    (16: os.cpu_count() there reports the whole machine)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def build_native_oracle():
    """The timed build of the CPU port: -O3 -march=native, no op counters in the inner loops (oracle/Makefile,
    target native), compiled on THIS host (the checker's prebuilt generic -O2 library is the fallback)."""
    import subprocess
    out = os.path.join(ROOT, "oracle", "_native", "librt_oracle_native.so")
    try:
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "native"], check=True, timeout=120,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        if os.path.exists(out):
            return out, "-O3 -march=native, lean counters, built on this host"
    except Exception:
        pass
    return None, "prebuilt generic -O2 build with op counters (native build failed)"


def cpu_baseline(workload, threads):
    """Reference-style CPU render (oracle: threads x whole image x 1 spp each, sfc64 seeds i*123, reference BVH --
    the restatement of render_and_write, raytracer.zig:580-633) on a bounded sample of the same workload, timed
    wall-clock around spawn -> join like raytracer.zig:584-608."""
    lib, how = build_native_oracle()
    if lib:
        os.environ["RT_ORACLE_LIB"] = lib
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    scene, nx, ny, _spp, depth, _ = WORKLOADS[workload]
    if scene == "mesh":
        sph = O.random_scene(11, 0)
        w = O.World(triangles=O.tessellate(sph, 0, 0), materials=sph)
        cam, trav, what = O.book_camera(nx, ny), O.TRAV_LIST, "the checker's triangle BVH"
    elif scene == "book":
        w, cam, trav, what = O.World(spheres=O.random_scene(11, 0)), O.book_camera(nx, ny), O.TRAV_BVH_REF, "reference BVH"
    else:
        w, cam, trav, what = O.World(spheres=O.two_sphere_scene()), O.camera_basic(), O.TRAV_LIST, "list traversal"
    spp_thread = {"two": 40, "book": 3, "mesh": 2}[scene]        # 10-30 s of CPU work on 16 threads
    if nx * ny < 1000000:
        spp_thread *= 8
    cfg = O.make_config(nx, ny, spp_thread, max_depth=depth, traversal=trav)
    _, cnt, sec = O.render_reference_style(w, cam, cfg, threads, spp_thread)
    return {"value": cnt["rays"] / sec / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "cpu": cpu_model(), "build": how,
            "sample": "%dx%d, %d spp per thread x %d threads (render_and_write restatement, sfc64 seeds i*123, %s), "
                      "%.1f s" % (nx, ny, spp_thread, threads, what, sec)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default c3 (the headline); with --gpus N > 1 and no --workload the line also carries a "
                         "c5_strong block (BASELINE configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-one-eighth", action="store_true", help="skip the 32-CU (1/8 GPU) leg")
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()
    workload_given = args.workload is not None
    if not workload_given:
        args.workload = "c3"

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
    spheres = R.Two_Sphere_Scene() if scene == "two" else R.Random_Scene(11, 0)
    cam = R.BasicCamera() if scene == "two" else R.book_camera(nx, ny)
    ctx = R.Context(device_index)
    n_tri = 0
    if scene == "mesh":
        tris = R.Tessellate_Spheres(spheres, 0, 0)
        n_tri = len(tris)
        ctx.upload_scene(triangles=tris, materials=spheres)
    else:
        ctx.upload_scene(spheres)
    from raytrace_amd.distributed import HipRangeRenderer, timed_sharded_steps
    npix = nx * ny
    renderer = HipRangeRenderer(ctx, cam, nx, ny, max_depth=depth, seed=1)
    accum = torch.zeros(npix * 3, dtype=torch.float32, device="cuda")

    strong = args.workload in STRONG
    spp_total = spp if strong else spp * world      # weak: every GPU adds its own `spp` samples to the image
    # a step: zero accum -> trace this rank's samples [rank*spp, (rank+1)*spp) -> RCCL reduce -> tonemap on rank 0;
    # warmup untimed, then exactly --steps steps between fences (synchronize + barrier), MAX over ranks
    timed = timed_sharded_steps(renderer, spp_total, dist, steps=args.steps, warmup=args.warmup, accum=accum)
    kernel_ms, reduce_ms = [timed["kernel_ms"]], [timed["reduce_ms"]]
    rays_per_step = [timed["rays_rank"]]
    rays_all = timed["rays_per_step"]            # rays of ONE step over all ranks
    dt = timed["seconds"]

    # BASELINE configs[4] (C5: 3840x2160, 1024 spp SPLIT over the GPUs -- the reference's thread fan-out,
    # raytracer.zig:588-617): measured beside the weak-scaling headline whenever the driver runs --gpus N > 1
    # without naming a workload.  Its own short step count: one 4K step is ~620/N ms.
    c5 = None
    if world > 1 and not workload_given:
        _, nx5, ny5, spp5, depth5, op5 = WORKLOADS["c5"]
        r5 = HipRangeRenderer(ctx, R.book_camera(nx5, ny5), nx5, ny5, max_depth=depth5, seed=1)
        del accum
        t5 = timed_sharded_steps(r5, spp5, dist, steps=max(1, min(3, args.steps)), warmup=1)
        with open(os.path.join(ROOT, "tests", "golden", op5)) as f:
            fpr5 = json.load(f)["flops_per_ray"]
        mr5 = t5["rays_per_step"] / (t5["ms_per_step"] * 1e-3) / 1e6
        c5 = {"workload": WORKLOAD_DESC["c5"], "scaling": "strong", "nx": nx5, "ny": ny5, "spp_total": spp5,
              "spp_per_gpu": spp5 / world, "steps": t5["steps"], "warmup": 1, "ms_per_step": t5["ms_per_step"],
              "value": mr5, "unit": "Mrays/s", "rays_per_step": t5["rays_per_step"],
              "kernel_ms_rank0": t5["kernel_ms"], "reduce_ms_rank0": t5["reduce_ms"],
              "frac_of_n_x_157.3TF": mr5 * 1e6 * fpr5 / 1e12 / (PEAK_FP32_TFLOPS * world),
              "algorithmic_flops_per_ray": fpr5, "backend": t5["backend"],
              "rccl_ranks": t5["world"] if t5["backend"] == "nccl" else 0,
              "parallelism": "spp-range x%d (shard_spp) + one reduce(SUM, dst 0) of 99.5 MB + tonemap on rank 0" % world}

    if rank == 0:
        value = rays_all * args.steps / dt / 1e6
        with open(os.path.join(ROOT, "tests", "golden", opfile)) as f:
            op = json.load(f)
        fpr = op["flops_per_ray"]
        k_ms = float(np.mean(kernel_ms))
        # HBM bytes per launch of the trace kernel come from a separate rocprofv3 --pmc pass (FETCH_SIZE,
        # WRITE_SIZE; tools/profile_bench.sh), committed under profiles/: counters cannot be read in-process.
        traffic, traffic_src, pmc_git = None, None, None
        pmc_path = os.path.join(ROOT, "profiles", "latest_%s_pmc.json" % args.workload)
        if world == 1 and os.path.exists(pmc_path):
            with open(pmc_path) as f:
                pj = json.load(f)
            d, pmc_git = pj.get("derived", {}), pj.get("git", "unrecorded (profile older than round 3)")
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
        # what binds the global-memory tree walk (C4): the vector L1 serves ~1.93 divergent lane requests per clock and
        # CU whatever their width (tools/ubench/gather_rate.hip); requests per clock of the committed counter pass
        l1 = None
        if world == 1 and os.path.exists(pmc_path):
            with open(pmc_path) as f:
                d = json.load(f).get("derived", {})
            if "l1_requests_per_clk_per_cu" in d:
                l1 = {"requests_per_clk_per_cu": d["l1_requests_per_clk_per_cu"], "measured_gather_peak": 1.93,
                      "frac": d["l1_requests_per_clk_per_cu"] / 1.93, "requests_per_ray": d.get("l1_requests_per_ray"),
                      "l1_hit_rate": d.get("l1_hit_rate"),
                      "source": "profiles/latest_%s_pmc.json (TCP_TOTAL_CACHE_ACCESSES, TCP_TCC_READ_REQ)" % args.workload}
        out = {
            "metric": "Mrays/sec (all bounces) book scene 1080p 100spp; %FP32 VALU roofline @1/8 GPU",   # BASELINE.json, verbatim
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOAD_DESC[args.workload], "n_spheres": len(spheres), "n_triangles": n_tri,
                       "upload_ms": ctx.upload_ms(), "nx": nx, "ny": ny,
                       "spp_per_gpu": spp_total / world, "spp_total": spp_total, "max_depth": depth, "seed": 1,
                       "rays_per_step": rays_all, "reduce_ms_rank0": float(np.mean(reduce_ms)) if world > 1 else 0.0,
                       "parallelism": "spp-range x%d + RCCL reduce" % world if world > 1 else "1 GPU"},
            "roofline": {"bound": "valu_fp32", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "executed": executed, "l1_gather": l1,
                         # traffic / executed / l1_gather are read from the committed counter pass of the same command,
                         # not collected by this process (counters need rocprofv3 around it)
                         "counters_measured_in_run": False, "counters_profile_git": pmc_git,
                         "scope": "rank 0's GPU" if world > 1 else "the GPU",
                         "kernel": ctx.last_kernel_name(), "kernel_ms": k_ms,
                         "algorithmic_flops_per_ray": fpr, "rays_per_launch": rays_per_step[0],
                         "kernel_mrays_per_s": rays_per_step[0] / (k_ms * 1e-3) / 1e6,
                         "note": "FP32 vector roofline (not HBM/MFMA). achieved = ALGORITHMIC flops: per ray those of the "
                                 "reference's own traversal on this config (tests/golden/%s, SURVEY 8(d); C4: the checker's "
                                 "plain BVH, the reference has no triangle code) -- a throughput normalised to the reference's "
                                 "work, not executed instructions; the hardware-counted figure is `executed`" % opfile},
        }
        if world == 1 and not args.no_one_eighth:
            # the "@1/8 GPU" half of BASELINE.json's metric: the same step on 32 of the 256 CUs (a CU-masked stream,
            # 4 CUs on each XCD), against 1/8 of the FP32 vector peak
            ctx.set_cu_mask(32)
            ms8 = []
            for k in range(1 + min(3, args.steps)):
                acc8 = torch.zeros(npix * 3, dtype=torch.float32, device="cuda")
                cfg8 = R.make_config(nx, ny, spp, max_depth=depth, seed=1, spp_chunk=R.RT_SPP_CHUNK_AUTO)
                r8, _, m8, _ = ctx.render_device(cam, cfg8, acc8.data_ptr(), 0, 0, want_stats=True)
                if k:
                    ms8.append(m8)
            ctx.set_cu_mask(None)
            k8 = float(np.mean(ms8))
            a8 = r8 * fpr / (k8 * 1e-3) / 1e12
            out["roofline_one_eighth_gpu"] = {"cus": 32, "peak": PEAK_FP32_TFLOPS / 8, "unit": "TFLOP/s", "achieved": a8,
                                              "frac": a8 / (PEAK_FP32_TFLOPS / 8), "kernel_ms": k8,
                                              "mrays_per_s": r8 / (k8 * 1e-3) / 1e6,
                                              "note": "same workload and kernel on a CU-masked stream: 32 CUs, 4 on each XCD"}
        if c5 is not None:
            out["c5_strong"] = c5
        if world == 1 and not args.no_cpu_baseline:
            threads = args.cpu_threads or host_threads()
            out["cpu_baseline"] = cpu_baseline(args.workload, threads)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
