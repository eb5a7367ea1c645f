"""Seeded ray batches for the closest-hit / any-hit parity tests."""
import numpy as np


def primary_rays(n, size=1024, seed=1):
    """Un-normalised camera rays as Draw builds them (main.cpp:128-133): origin (0,0,-3,1),
    direction (x - W/2 + jitter, y - H/2 + jitter, focal, 0), focal = W."""
    rng = np.random.default_rng(seed)
    o = np.tile(np.array([0, 0, -3, 1], dtype=np.float32), (n, 1))
    d = np.zeros((n, 4), dtype=np.float32)
    d[:, 0] = rng.integers(0, size, n) - size // 2 + rng.uniform(-0.1, 0.1, n)
    d[:, 1] = rng.integers(0, size, n) - size // 2 + rng.uniform(-0.1, 0.1, n)
    d[:, 2] = size
    return o, d


def interior_rays(n, seed=2):
    """Incoherent rays from points inside the box; direction normalised in 4-D with w = 1 before
    normalisation, like a diffuse bounce (pathtracer.cpp:13)."""
    rng = np.random.default_rng(seed)
    o = np.ones((n, 4), dtype=np.float32)
    o[:, :3] = rng.uniform(-0.95, 0.95, (n, 3))
    v = rng.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    d = np.ones((n, 4), dtype=np.float32)
    d[:, :3] = v
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return o, d


def surface_rays(tris, n, seed=3):
    """Origins exactly on triangle surfaces (no offset, as after a mirror or glass bounce,
    pathtracer.cpp:100): self-hits at t ~ 0 hinge on the last ulp."""
    rng = np.random.default_rng(seed)
    k = rng.integers(0, len(tris), n)
    u = rng.uniform(0, 1, n).astype(np.float32)
    v = (rng.uniform(0, 1, n) * (1 - u)).astype(np.float32)
    v0, v1, v2 = tris["v0"][k], tris["v1"][k], tris["v2"][k]
    o = (v0 + u[:, None] * (v1 - v0) + v[:, None] * (v2 - v0)).astype(np.float32)
    o[:, 3] = 1.0
    d3 = rng.normal(size=(n, 3))
    d3 /= np.linalg.norm(d3, axis=1, keepdims=True)
    d = np.zeros((n, 4), dtype=np.float32)
    d[:, :3] = d3
    d[:, 3] = rng.uniform(0, 0.7, n)
    return o, d


def degenerate_rays(tris, n, seed=4):
    """Rays parallel to triangle planes, along edges, with zero direction, and huge / tiny
    magnitudes: determinant ratios become inf, NaN or subnormal."""
    rng = np.random.default_rng(seed)
    o = np.ones((n, 4), dtype=np.float32)
    o[:, :3] = rng.uniform(-0.9, 0.9, (n, 3))
    d = np.zeros((n, 4), dtype=np.float32)
    axes = np.eye(3, dtype=np.float32)
    for i in range(n):
        kind = i % 6
        t = tris[rng.integers(0, len(tris))]
        e1 = t["v1"][:3] - t["v0"][:3]
        e2 = t["v2"][:3] - t["v0"][:3]
        if kind == 0:
            d[i, :3] = axes[i // 6 % 3] * (1 if i % 2 else -1)      # axis aligned: parallel to walls
        elif kind == 1:
            d[i, :3] = e1                                           # in the triangle's plane
            o[i, :3] = t["v0"][:3]
        elif kind == 2:
            d[i, :3] = e1 * np.float32(0.3) + e2 * np.float32(0.7)  # in plane, from outside
        elif kind == 3:
            d[i, :3] = 0.0                                          # zero direction
        elif kind == 4:
            d[i, :3] = rng.normal(size=3) * 1e-30                   # tiny: products go subnormal
        else:
            d[i, :3] = rng.normal(size=3) * 1e18                    # huge
    return o, d


def shadow_rays(tris, n, seed=5):
    """Rays toward a point near the glass-scene light with max_t = distance, as directLight casts."""
    rng = np.random.default_rng(seed)
    o, _ = interior_rays(n, seed)
    light = np.array([0.7, 0.95, -0.7, 1.0], dtype=np.float32) + \
        np.concatenate([rng.uniform(-0.1, 0.1, (n, 3)), np.zeros((n, 1))], axis=1).astype(np.float32)
    v = (light - o).astype(np.float32)
    r = np.sqrt((v * v).sum(axis=1)).astype(np.float32)
    d = (v / r[:, None]).astype(np.float32)
    return o, d, r
