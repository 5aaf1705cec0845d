"""Shared test helpers (conversion between the product's SoA scene and the oracle's AoS records)."""
import numpy as np

import oracle_lib as O


def spheres_to_oracle(s):
    a = np.zeros(len(s), dtype=O.SPHERE_DTYPE)
    a["center"][:, 0], a["center"][:, 1], a["center"][:, 2] = s.cx, s.cy, s.cz
    a["radius"], a["kind"], a["param"] = s.radius, s.kind, s.param
    a["albedo"][:, 0], a["albedo"][:, 1], a["albedo"][:, 2] = s.r, s.g, s.b
    return a


def triangles_to_oracle(t):
    a = np.zeros(len(t), dtype=O.TRIANGLE_DTYPE)
    a["v0"], a["v1"], a["v2"], a["material"] = t.v0, t.v1, t.v2, t.material
    return a


def psnr_u8(a, b):
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    mse = np.mean((a - b) ** 2)
    return float("inf") if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
