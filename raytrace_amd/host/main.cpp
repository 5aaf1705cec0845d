// Standalone program: the mirror of `pub fn main` in src/raytracer.zig:714-745 on top of the HIP backend.
// Same scene (Random_Scene(11) from sfc64 seed 0), same camera pose, same image size and output file as the reference's
// defaults; the per-pixel trace path runs on the GPU through the C ABI (rt::render_and_write, raytracer.hpp).
//
//     rt_main [nx ny num_samples [out.ppm [n_gpus]]]        defaults: 1200 675 1 Scene.ppm 1
//
// num_samples is the total per pixel (the reference renders num_samples on each CPU thread, raytracer.zig:594-608).
// With n_gpus > 1 the samples are split over that many GPUs (rt_render_multi, the role of the thread fan-out).
//
//     rt_main trace [num_samples [n_passes [out.ppm [nx ny [px py pz fx fy fz]]]]]
//
// the GUI's "Trace" button without the window (src/main.zig:307-318, 342-345): camera set the way start_trace sets it from
// the fly camera at (px,py,pz) looking along (fx,fy,fz) -- defaults: DemoState's initial values, main.zig:65-69 --, the
// render in n_passes polling passes (rt::TraceJob), the picture so far written to out.ppm after every pass (what the window
// would show), the finished Scene.ppm at the end.  Defaults: 25 samples (main.zig:344), 5 passes, Scene.ppm, 1200 x 675.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "raytracer.hpp"

static int trace_button(int argc, char **argv)
{
    const int num_samples = argc > 2 ? std::atoi(argv[2]) : 200 / 8;                                  // main.zig:344
    const int n_passes = argc > 3 ? std::atoi(argv[3]) : 5;
    const char *fname = argc > 4 ? argv[4] : "Scene.ppm";
    const long nx = argc > 6 ? std::atol(argv[5]) : 1200, ny = argc > 6 ? std::atol(argv[6]) : 675;
    float v[6] = { 0.0f, 4.0f, -4.0f, 0.0f, 0.0f, 1.0f };                                             // main.zig:66-67
    if (argc > 12) for (int k = 0; k < 6; ++k) v[k] = (float)std::atof(argv[7 + k]);
    if (num_samples < 1 || num_samples > 65535 || n_passes < 1 || nx < 1 || ny < 1) {
        std::fprintf(stderr, "usage: %s trace [num_samples [n_passes [out.ppm [nx ny [px py pz fx fy fz]]]]]\n", argv[0]);
        return 2;
    }
    rt::RndGen rng = rt::RndGen::init(0);                                                            // main.zig:570
    const rt::Hittable World = rt::Random_Scene(11, rng);                                            // main.zig:578
    rt_ctx *ctx = nullptr;
    int rc = rt_create(0, &ctx);
    if (rc != RT_OK) { std::fprintf(stderr, "rt_create: %s\n", rt_strerror(rc)); return 1; }
    rt::TraceJob job;
    rc = job.begin(ctx, World, { v[0], v[1], v[2] }, { v[3], v[4], v[5] }, (uint16_t)num_samples, (uint32_t)n_passes,
                   rt::shape_s{ (size_t)ny, (size_t)nx });
    while (rc == RT_OK && job.active) {                 // the window's frame loop: one pass per frame
        rc = job.poll();
        if (rc == RT_OK || rc == RT_PROGRESS_MORE) {
            std::printf("pass: %u of %d samples, %llu rays so far\n", job.spp_done, num_samples, (unsigned long long)job.stats.rays);
            const int wrc = job.finish(fname);          // the picture so far
            rc = wrc != RT_OK ? wrc : RT_OK;
        }
    }
    if (rc != RT_OK) { std::fprintf(stderr, "trace failed: %s (hip %d)\n", rt_strerror(rc), rt_last_hip_error(ctx)); rt_destroy(ctx); return 1; }
    std::printf("Traced %u samples per pixel in %.2f ms of kernels: %llu rays\n", job.spp_done, job.stats.kernel_ms,
                (unsigned long long)job.stats.rays);
    rt_destroy(ctx);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 1 && std::strcmp(argv[1], "trace") == 0) return trace_button(argc, argv);
    const size_t nx = argc > 1 ? (size_t)std::atol(argv[1]) : 1200, ny = argc > 2 ? (size_t)std::atol(argv[2]) : 675;
    const uint16_t num_samples = argc > 3 ? (uint16_t)std::atoi(argv[3]) : 1;                       // raytracer.zig:718
    const char *fname = argc > 4 ? argv[4] : "Scene.ppm";
    const int n_gpus = argc > 5 ? std::atoi(argv[5]) : 1;
    if (nx == 0 || ny == 0 || num_samples == 0 || n_gpus < 1 || n_gpus > 64) {
        std::fprintf(stderr, "usage: %s [nx ny num_samples [out.ppm [n_gpus]]]\n", argv[0]);
        return 2;
    }
    rt::cam.SetPosition({ 13.0f, 2.0f, 3.5f }, { 0.0f, 0.0f, 0.0f }, { 0.0f, 1.0f, 0.0f }, 20.0f, (float)nx / (float)ny);   // :730
    rt::RndGen rng = rt::RndGen::init(0);                                                            // :731
    const rt::Hittable World = rt::Random_Scene(11, rng);                                            // :737
    std::printf("World contains %zu spheres\n", World.objects.size());
    std::printf("Creating %zu x %zu image\n", nx, ny);                                                // :742

    std::vector<rt_ctx *> ctxs((size_t)n_gpus, nullptr);
    for (int g = 0; g < n_gpus; ++g) {
        const int rc = rt_create(g, &ctxs[(size_t)g]);
        if (rc != RT_OK) { std::fprintf(stderr, "rt_create(%d): %s\n", g, rt_strerror(rc)); return 1; }
    }
    const auto t0 = std::chrono::steady_clock::now();
    rt_stats stats{};
    int rc;
    if (n_gpus == 1) {
        rc = rt::render_and_write(ctxs[0], fname, num_samples, World, { ny, nx }, rt::img_io::ImgType::ppm, 1, &stats);   // :743
    } else {
        // every GPU gets the scene; the samples of every pixel are split over the GPUs and summed on GPU 0
        rt::SphereSoA soa(World.objects);
        const rt_spheres view = soa.view();
        rc = RT_OK;
        for (rt_ctx *c : ctxs) if (rc == RT_OK) rc = rt_upload_scene(c, &view, nullptr, nullptr);
        rt_config cfg{};
        cfg.nx = (uint32_t)nx; cfg.ny = (uint32_t)ny;
        cfg.spp_begin = 0; cfg.spp_end = num_samples; cfg.spp_total = num_samples;
        cfg.max_depth = 16; cfg.t_min = 0.001f; cfg.seed = 1; cfg.row_offset_q3 = 1; cfg.spp_chunk = RT_SPP_CHUNK_AUTO;
        const rt_camera c = rt::cam.to_abi();
        std::vector<uint8_t> rgb8(nx * ny * 3);
        if (rc == RT_OK) rc = rt_render_multi(ctxs.data(), (uint32_t)n_gpus, &c, &cfg, nullptr, rgb8.data(), nullptr, &stats);
        if (rc == RT_OK) rc = rt_write_ppm(fname, rgb8.data(), (uint32_t)nx, (uint32_t)ny);
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rc != RT_OK) { std::fprintf(stderr, "render failed: %s (hip %d)\n", rt_strerror(rc), rt_last_hip_error(ctxs[0])); return 1; }
    std::printf("Rendered in %.1f ms (trace kernel %.2f ms): %llu rays, %.1f Mrays/s\n", ms, stats.kernel_ms,
                (unsigned long long)stats.rays, (double)stats.rays / (stats.kernel_ms * 1e3));
    for (rt_ctx *c : ctxs) rt_destroy(c);
    return 0;
}
