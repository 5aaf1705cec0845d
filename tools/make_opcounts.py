"""Generate tests/golden/opcounts_<config>.json: the oracle's op counters on the exact BASELINE configs
and the algorithmic flops-per-ray figure of SURVEY 8(d).  CPU only; C3 takes minutes.

    python tools/make_opcounts.py c1 c2 c3
"""
import json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O

CONFIGS = {
    # name: (scene, nx, ny, spp, max_depth, traversal)
    "c1": ("book", 400, 225, 10, 10, O.TRAV_BVH_REF),
    "c2": ("two", 1920, 1080, 100, 16, O.TRAV_LIST),
    "c3": ("book", 1920, 1080, 100, 16, O.TRAV_BVH_REF),
    # C4: the reference has no ray-triangle code and no triangle traversal; the counts are those of the checker's own
    # plain BVH (median split, <= 4 triangles per leaf) on a random 1 % of the pixels at all 256 samples
    "c4": ("mesh", 1920, 1080, 256, 16, O.TRAV_LIST),
}
C4_SAMPLE_PIXELS = 20000


def flops_per_ray(c):
    """5 + 18*T + 10*B + 16*h + 31*s_lam + 54*s_met + 85*s_die + 21*m + 22*p."""
    rays = float(c["rays"])
    T = c["sphere_tests"] / rays
    B = c["aabb_tests"] / rays
    h = c["hits"] / rays
    m = c["misses"] / rays
    p = c["primary_rays"] / rays
    return (5 + 18 * T + 10 * B + 16 * h + 31 * c["scatter_lambertian"] / rays + 54 * c["scatter_metal"] / rays
            + 85 * c["scatter_dielectric"] / rays + 21 * m + 22 * p)


def flops_per_ray_triangles(c):
    """Build-defined (Roofline): 30 per triangle test (Moller-Trumbore through the u test: e1, e2 6, pvec 9,
    det 5, 1/det 1, tvec 3, u 6 -- the later stages are charged only to the accepted hit, +22), 12 per box test
    (3 axes x 2 planes x (sub, mul)), 24 per hit record (p 6, flat normal: cross 9, dot 5, sqrt 1, 3 div), the
    scatter / sky / camera terms of SURVEY 8(d)."""
    rays = float(c["rays"])
    return (30 * c["tri_tests"] / rays + 12 * c["aabb_tests"] / rays + (22 + 24) * c["hits"] / rays
            + 31 * c["scatter_lambertian"] / rays + 54 * c["scatter_metal"] / rays + 85 * c["scatter_dielectric"] / rays
            + 21 * c["misses"] / rays + 22 * c["primary_rays"] / rays)


def run(name):
    scene, nx, ny, spp, depth, trav = CONFIGS[name]
    pixels = None
    if scene == "mesh":
        sph, cam = O.random_scene(11, 0), O.book_camera(nx, ny)
        tris = O.tessellate(sph, 0, 0)
        w = O.World(triangles=tris, materials=sph)
        pixels = np.unique(np.random.default_rng(4).integers(0, nx * ny, C4_SAMPLE_PIXELS)).astype(np.uint32)
    elif scene == "book":
        sph, cam = O.random_scene(11, 0), O.book_camera(nx, ny)
        w = O.World(spheres=sph)
    else:
        sph, cam = O.two_sphere_scene(), O.camera_basic()
        w = O.World(spheres=sph)
    cfg = O.make_config(nx, ny, spp, max_depth=depth, seed=1, traversal=trav)
    t = time.time()
    _, _, _, cnt = O.render_counter(w, cam, cfg, pixels=pixels)
    dt = time.time() - t
    if scene == "mesh":
        out = {"config": name, "scene": scene, "nx": nx, "ny": ny, "spp": spp, "max_depth": depth, "seed": 1,
               "traversal": "checker's plain BVH (median split, <= 4 triangles per leaf); the reference has none",
               "n_triangles": int(len(tris)), "sampled_pixels": int(len(pixels)), "counters": cnt,
               "flops_per_ray": flops_per_ray_triangles(cnt),
               "tri_tests_per_ray": cnt["tri_tests"] / cnt["rays"], "aabb_tests_per_ray": cnt["aabb_tests"] / cnt["rays"],
               "rays_per_primary": cnt["rays"] / cnt["primary_rays"], "oracle_seconds": dt}
        with open(os.path.join(ROOT, "tests", "golden", "opcounts_%s.json" % name), "w") as f:
            json.dump(out, f, indent=1)
        print(name, "rays", cnt["rays"], "flops/ray %.1f" % out["flops_per_ray"], "%.1fs" % dt, flush=True)
        return
    out = {"config": name, "scene": scene, "nx": nx, "ny": ny, "spp": spp, "max_depth": depth, "seed": 1,
           "traversal": "bvh_ref" if trav == O.TRAV_BVH_REF else "list", "n_spheres": int(len(sph)),
           "counters": cnt, "flops_per_ray": flops_per_ray(cnt),
           "sphere_tests_per_ray": cnt["sphere_tests"] / cnt["rays"], "aabb_tests_per_ray": cnt["aabb_tests"] / cnt["rays"],
           "rays_per_primary": cnt["rays"] / cnt["primary_rays"], "oracle_seconds": dt}
    with open(os.path.join(ROOT, "tests", "golden", "opcounts_%s.json" % name), "w") as f:
        json.dump(out, f, indent=1)
    print(name, "rays", cnt["rays"], "flops/ray %.1f" % out["flops_per_ray"], "%.1fs" % dt, flush=True)


if __name__ == "__main__":
    for n in sys.argv[1:] or ["c1"]:
        run(n)
