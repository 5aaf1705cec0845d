"""Phase-fill and shading-block counters of the shipped C3 and C4 kernels (rt_debug_wave_fill, rt_debug_block_counts;
RT_FLAG_DEPTH_HISTOGRAM build) as JSON: python tools/gpu_fill_json.py <out dir> -> <out dir>/{c3,c4}_fill.json.
VERDICT r2 item 7: the numbers instead of prose."""
import json, os, sys
sys.path.insert(0, ".")
import raytrace_amd as R
out_dir = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/fill"
os.makedirs(out_dir, exist_ok=True)
book = R.Random_Scene(11, 0)
cam = R.book_camera(1920, 1080)
ctx = R.Context(0)
for name, spp in (("c3", 100), ("c4", 64)):
    ctx.set_flags(0)
    if name == "c4":
        ctx.upload_scene(triangles=R.Tessellate_Spheres(book, 0, 0), materials=book)
    else:
        ctx.upload_scene(book)
    cfg = R.make_config(1920, 1080, spp, spp_chunk=R.RT_SPP_CHUNK_AUTO)
    for _ in range(2):
        fast = ctx.render(cam, cfg)
    ctx.set_flags(R.RT_FLAG_DEPTH_HISTOGRAM)
    res = ctx.render(cam, cfg)
    f = ctx.wave_fill()
    counts = ctx.block_counts()
    rays = res.rays
    passes = f["wave_shade_passes"] or 1
    d = {"workload": "%s: 1920x1080, %d spp, depth 16" % (name.upper(), spp), "kernel": ctx.last_kernel_name(), "scene": ctx.scene_info(),
         "rays": rays, "kernel_ms_production_build": fast.kernel_ms, "kernel_ms_counting_build": res.kernel_ms,
         "per_ray": {"lane_traversal_steps": f["lane_steps"] / rays, "lane_exact_tests": f["lane_exact_iters"] / rays},
         "per_64_rays": {"wave_traversal_steps": f["wave_steps"] * 64 / rays, "wave_exact_test_batches": f["wave_exact_iters"] * 64 / rays,
                         "wave_shading_passes": f["wave_shade_passes"] * 64 / rays},
         "fill": {"traversal": f["traverse_fill"], "exact_tests": f["exact_fill"], "shading": f["shade_fill"]},
         "shading_blocks": {k: {"runs_in_share_of_passes": w / passes, "lanes_per_run": (l / w if w else 0.0), "runs_per_64_rays": w * 64.0 / rays}
                            for k, (w, l) in counts.items()},
         "raw": {"wave_fill": f, "block_counts": {k: list(v) for k, v in counts.items()}}}
    json.dump(d, open(os.path.join(out_dir, "%s_fill.json" % name), "w"), indent=1)
    print(name, json.dumps(d["per_64_rays"]), json.dumps(d["fill"]), "%.2f ms" % fast.kernel_ms, flush=True)
