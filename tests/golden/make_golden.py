"""Generate the committed golden vectors from the CPU oracle (oracle/rt_oracle.c).

The reference holds no fixtures for the trace path, and it is Zig with no toolchain here, so
these goldens pin the ORACLE (regression) and give the GPU tests a checker-independent data file:
per scene, 32x18 px, 4 spp, counter-RNG seed 1, depth caps 16 and 10: linear mean accumulation (f32),
per-pixel ray counts (u16) and the u8 image; plus the scene's first spheres and the RNG streams.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

NX, NY, SPP, SEED = 32, 18, 4, 1


def scenes():
    book = O.random_scene(11, 0)
    two = O.two_sphere_scene()
    # mesh scene: the two-sphere scene plus one glass and one metal ball, tessellated coarsely
    extra = np.zeros(2, dtype=O.SPHERE_DTYPE)
    extra[0] = ((-1.0, 0.0, -1.0), 0.5, O.DIELECTRIC, (0.95, 0.95, 0.95), 1.5)
    extra[1] = ((1.0, 0.0, -1.0), 0.5, O.METAL, (0.8, 0.6, 0.2), 0.15)
    mesh_src = np.concatenate([two, extra])
    return {"book": (book, None, O.book_camera(NX, NY)),
            "two": (two, None, O.camera_basic()),
            "mesh": (mesh_src, O.tessellate(mesh_src, 12, 8), O.camera_basic())}


def main():
    out = {}
    for name, (sph, tri, cam) in scenes().items():
        w = O.World(spheres=None if tri is not None else sph, triangles=tri, materials=sph if tri is not None else None)
        for depth in (16, 10):
            cfg = O.make_config(NX, NY, SPP, max_depth=depth, seed=SEED, threads=1)
            acc, rgb, rc, cnt = O.render_counter(w, cam, cfg)
            key = "%s_d%d" % (name, depth)
            out[key + "_accum"] = acc
            out[key + "_rgb8"] = rgb
            out[key + "_raycount"] = rc.astype(np.uint16)
            out[key + "_rays"] = np.uint64(cnt["rays"])
    book = O.random_scene(11, 0)
    out["book_n"] = np.uint32(len(book))
    out["book_first5_center"] = book["center"][:5]
    out["book_first5_albedo"] = book["albedo"][:5]
    out["book_kind"] = book["kind"].astype(np.uint8)
    out["book_radius"] = book["radius"]
    out["book_param"] = book["param"]
    np.savez_compressed(os.path.join(HERE, "oracle_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "oracle_golden.npz"), {k: getattr(v, "shape", ()) for k, v in out.items()})


if __name__ == "__main__":
    main()
