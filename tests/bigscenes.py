"""Seeded synthetic scenes of 10^3..10^6 triangles, standing in for the reference's OBJ models
(wolf / bunny / dragon, main.cpp:55-66, which its repository does not ship): meshes of small
triangles inside the Cornell room, built as TRIANGLE_DTYPE arrays the way Triangle's constructor
does (normal = normalize(cross(v2 - v0, v1 - v0)), w = 1; TestModelH.h:40-48)."""
import numpy as np


def _finish(dtype, v0, v1, v2, color, material):
    n = len(v0)
    t = np.zeros(n, dtype=dtype)
    for name, v in (("v0", v0), ("v1", v1), ("v2", v2)):
        t[name][:, :3] = v.astype(np.float32)
        t[name][:, 3] = 1.0
    e1 = (t["v1"][:, :3] - t["v0"][:, :3]).astype(np.float32)
    e2 = (t["v2"][:, :3] - t["v0"][:, :3]).astype(np.float32)
    nrm = np.cross(e2, e1).astype(np.float32)
    ln = np.sqrt((nrm * nrm).sum(axis=1, keepdims=True)).astype(np.float32)
    ln[ln == 0] = 1
    t["normal"][:, :3] = nrm / ln
    t["normal"][:, 3] = 1.0
    t["color"] = color
    t["material"] = material
    return t


def heightfield(dtype, cells, seed=0, y0=-0.4, amp=0.25, materials=(0, 0, 0, 1, 2)):
    """2 * cells^2 triangles: a wavy sheet across the room; every cell's two triangles share
    their diagonal (tie rule) and neighbours share edges."""
    rng = np.random.default_rng(seed)
    g = np.linspace(-0.95, 0.95, cells + 1)
    X, Z = np.meshgrid(g, g, indexing="ij")
    Y = y0 + amp * np.sin(3.1 * X + 0.3) * np.cos(2.3 * Z - 0.2) + 0.02 * rng.standard_normal(X.shape)
    P = np.stack([X, Y, Z], axis=-1)
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    v0 = np.concatenate([a.reshape(-1, 3), a.reshape(-1, 3)])
    v1 = np.concatenate([b.reshape(-1, 3), c.reshape(-1, 3)])
    v2 = np.concatenate([c.reshape(-1, 3), d.reshape(-1, 3)])
    n = len(v0)
    order = rng.permutation(n)  # index order unrelated to position, like a mesh file's
    color = rng.uniform(0.2, 0.9, (n, 3)).astype(np.float32)
    material = np.asarray(materials, dtype=np.uint32)[rng.integers(0, len(materials), n)]
    return _finish(dtype, v0[order], v1[order], v2[order], color, material)


def torus(dtype, nu, nv, seed=1, R=0.45, r=0.18, center=(0.1, 0.15, -0.1), material=0):
    """2 * nu * nv triangles: a closed surface (rays enter and leave it)."""
    rng = np.random.default_rng(seed)
    u = np.linspace(0, 2 * np.pi, nu, endpoint=False)
    v = np.linspace(0, 2 * np.pi, nv, endpoint=False)
    U, V = np.meshgrid(u, v, indexing="ij")

    def pt(U, V):
        x = (R + r * np.cos(V)) * np.cos(U)
        z = (R + r * np.cos(V)) * np.sin(U)
        y = r * np.sin(V)
        return np.stack([x + center[0], y * 0.8 + 0.6 * x * 0.5 + center[1], z + center[2]], axis=-1)

    du, dv = 2 * np.pi / nu, 2 * np.pi / nv
    a, b, c, d = pt(U, V), pt(U + du, V), pt(U + du, V + dv), pt(U, V + dv)
    v0 = np.concatenate([a.reshape(-1, 3), a.reshape(-1, 3)])
    v1 = np.concatenate([b.reshape(-1, 3), c.reshape(-1, 3)])
    v2 = np.concatenate([c.reshape(-1, 3), d.reshape(-1, 3)])
    n = len(v0)
    color = np.tile(np.array([0.75, 0.6, 0.2], dtype=np.float32), (n, 1))
    return _finish(dtype, v0, v1, v2, color, np.full(n, material, dtype=np.uint32))


def soup(dtype, n, seed=2, size=0.06):
    """n small random triangles scattered through the room, some exact duplicates of each other
    (equal Morton keys, exact distance ties between different indices)."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-0.9, 0.9, (n, 3))
    v0 = c + rng.normal(scale=size, size=(n, 3))
    v1 = c + rng.normal(scale=size, size=(n, 3))
    v2 = c + rng.normal(scale=size, size=(n, 3))
    dup = rng.integers(0, n, n // 50)
    src = rng.integers(0, n, n // 50)
    v0[dup], v1[dup], v2[dup] = v0[src], v1[src], v2[src]
    color = rng.uniform(0.1, 1.0, (n, 3)).astype(np.float32)
    material = rng.integers(0, 4, n).astype(np.uint32)
    return _finish(dtype, v0, v1, v2, color, material)


def room(O):
    """The ten wall triangles of the Cornell box (TestModelH.h:87-105)."""
    return O.load_test_model(0)[:10].copy()


def mesh_in_room(O, dtype, cells=64, nu=96, nv=48, seed=0):
    """room + heightfield + torus: 10 + 2 cells^2 + 2 nu nv triangles."""
    return np.concatenate([room(O), heightfield(dtype, cells, seed), torus(dtype, nu, nv, seed + 1)])


def config4(O, dtype, nu=320, nv=160, seed=4, material=2):
    """BASELINE configs[3] stand-in: the empty room (LoadTestModelNone, 10 triangles) + a closed
    procedural mesh of 2 nu nv = 102 400 triangles (a torus displaced by a seeded wave field),
    vertices inside [-0.6, 0.6]^3, glass by default."""
    rng = np.random.default_rng(seed)
    ph = rng.uniform(0, 2 * np.pi, 4)
    u = np.linspace(0, 2 * np.pi, nu, endpoint=False)
    v = np.linspace(0, 2 * np.pi, nv, endpoint=False)
    du, dv = 2 * np.pi / nu, 2 * np.pi / nv

    def pt(U, V):
        rr = 0.16 * (1 + 0.18 * np.sin(5 * U + ph[0]) * np.cos(3 * V + ph[1]) + 0.05 * np.sin(17 * U + 11 * V + ph[2]))
        R = 0.38
        x = (R + rr * np.cos(V)) * np.cos(U)
        z = (R + rr * np.cos(V)) * np.sin(U)
        y = rr * np.sin(V) + 0.08 * np.sin(2 * U + ph[3])
        return np.stack([x, y - 0.2, z], axis=-1)

    U, V = np.meshgrid(u, v, indexing="ij")
    a, b, c, d = pt(U, V), pt(U + du, V), pt(U + du, V + dv), pt(U, V + dv)
    v0 = np.concatenate([a.reshape(-1, 3), a.reshape(-1, 3)])
    v1 = np.concatenate([b.reshape(-1, 3), c.reshape(-1, 3)])
    v2 = np.concatenate([c.reshape(-1, 3), d.reshape(-1, 3)])
    n = len(v0)
    assert np.abs(v0).max() <= 0.6
    color = np.tile(np.array([0.75, 0.75, 0.75], dtype=np.float32), (n, 1))
    mesh = _finish(dtype, v0, v1, v2, color, np.full(n, material, dtype=np.uint32))
    return np.concatenate([O.load_test_model(3), mesh])


def config5(dtype, n=1000000, seed=42):
    """BASELINE configs[4]: n triangles, centres uniform in [-1, 1]^3, edge vectors uniform in
    [-0.01, 0.01]^3, random diffuse colours."""
    rng = np.random.Generator(np.random.Philox(seed))
    c = rng.uniform(-1, 1, (n, 3))
    e1 = rng.uniform(-0.01, 0.01, (n, 3))
    e2 = rng.uniform(-0.01, 0.01, (n, 3))
    color = rng.uniform(0.1, 1.0, (n, 3)).astype(np.float32)
    return _finish(dtype, c, c + e1, c + e2, color, np.zeros(n, dtype=np.uint32))


def config4_vertices(nu=320, nv=160, seed=4):
    """The displaced torus of config4 as an indexed mesh: (nu * nv vertices, 2 nu nv faces of vertex indices)."""
    rng = np.random.default_rng(seed)
    ph = rng.uniform(0, 2 * np.pi, 4)
    u = np.linspace(0, 2 * np.pi, nu, endpoint=False)
    v = np.linspace(0, 2 * np.pi, nv, endpoint=False)
    U, V = np.meshgrid(u, v, indexing="ij")
    rr = 0.16 * (1 + 0.18 * np.sin(5 * U + ph[0]) * np.cos(3 * V + ph[1]) + 0.05 * np.sin(17 * U + 11 * V + ph[2]))
    R = 0.38
    P = np.stack([(R + rr * np.cos(V)) * np.cos(U), rr * np.sin(V) + 0.08 * np.sin(2 * U + ph[3]) - 0.2,
                  (R + rr * np.cos(V)) * np.sin(U)], axis=-1).reshape(-1, 3)
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    a = (i * nv + j).ravel()
    b = (((i + 1) % nu) * nv + j).ravel()
    c = (((i + 1) % nu) * nv + (j + 1) % nv).ravel()
    d = (i * nv + (j + 1) % nv).ravel()
    faces = np.concatenate([np.stack([a, b, c], axis=1), np.stack([a, c, d], axis=1)])
    return P, faces


def write_obj(path, vertices, faces, with_normals):
    """A Wavefront OBJ file of an indexed triangle mesh: `v` records, optionally one `vn` per vertex
    (the normalised sum of the adjacent face normals) with `f a//a b//b c//c` faces, else `f a b c`."""
    lines = ["# procedural mesh", "o mesh"]
    lines += ["v %.9g %.9g %.9g" % tuple(p) for p in vertices]
    if with_normals:
        fn = np.cross(vertices[faces[:, 1]] - vertices[faces[:, 0]], vertices[faces[:, 2]] - vertices[faces[:, 0]])
        vn = np.zeros_like(vertices)
        for k in range(3):
            np.add.at(vn, faces[:, k], fn)
        vn /= np.maximum(np.linalg.norm(vn, axis=1, keepdims=True), 1e-30)
        lines += ["vn %.9g %.9g %.9g" % tuple(n) for n in vn]
        lines += ["f %d//%d %d//%d %d//%d" % (a, a, b, b, c, c) for a, b, c in faces + 1]
    else:
        lines += ["f %d %d %d" % (a, b, c) for a, b, c in faces + 1]
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
