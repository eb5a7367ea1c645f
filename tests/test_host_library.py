"""CPU-side checks of libmipt.so: it loads, exports every symbol include/mipt.h declares,
and its host-side logic (scene builders, defaults, key function) is right. No kernel runs."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def lib(pkg):
    if not os.path.exists(pkg.abi.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return pkg.abi.load_library()


def test_header_symbols_all_exported(pkg, lib):
    header = open(os.path.join(ROOT, "include", "mipt.h")).read()
    declared = set(re.findall(r"\b(mipt_[a-z_0-9]+)\s*\(", header))
    assert declared == set(pkg.abi.EXPORTS), declared ^ set(pkg.abi.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mipt_version().decode().startswith("mipt ")


def test_struct_layouts(pkg):
    abi = pkg.abi
    assert C.sizeof(abi.MiptTriangle) == 80 and C.sizeof(abi.MiptSphere) == 36
    assert C.sizeof(abi.MiptCamera) == 4 * (1 + 4 + 16 + 16)
    assert C.sizeof(abi.MiptLight) == 28
    assert C.sizeof(abi.MiptParams) == 52
    assert C.sizeof(abi.MiptStats) == 120


@pytest.mark.parametrize("which,name", [(0, "cornell"), (1, "colorbleeding"), (2, "glasscube"), (3, "none")])
def test_library_scene_builders_match_reference_dump(pkg, lib, O, which, name):
    ours = pkg.api.load_test_model(which)
    ref = O.golden_scene(name)
    assert len(ours) == len(ref)
    for field in ("v0", "v1", "v2", "color", "normal"):
        assert np.array_equal(bits(ours[field]), bits(ref[field])), field
    assert np.array_equal(ours["material"], ref["material"])


def test_scene_builder_rejects_bad_arguments(pkg, lib):
    buf = np.zeros(30, dtype=pkg.abi.TRIANGLE_DTYPE)
    assert lib.mipt_load_test_model(9, buf.ctypes.data, 30) == pkg.abi.MIPT_E_INVALID
    assert lib.mipt_load_test_model(0, buf.ctypes.data, 29) == pkg.abi.MIPT_E_INVALID
    assert lib.mipt_load_test_model(0, None, 30) == pkg.abi.MIPT_E_INVALID


def test_defaults_match_oracle(pkg, lib, O):
    abi, api = pkg.abi, pkg.api
    for w in (500, 1024):
        a, b = api.default_camera(w), O.default_camera(w)
        assert bytes(a) == bytes(b)
    for glass in (False, True):
        assert bytes(api.default_light(glass)) == bytes(O.default_light(glass))
    assert bytes(api.default_params(640, 480)) == bytes(O.default_params(640, 480))
    p = api.default_params(8, 8)
    assert (p.max_depth, p.rng_mode) == (10, abi.RNG_MINSTD_PATH)
    assert np.float32(p.rr_prob) == np.float32(0.15) and np.float32(p.glass_weight) == np.float32(1.15)


def test_path_key_matches_oracle(pkg, lib, O):
    o = O.oracle()
    rng = np.random.default_rng(0)
    for seed, pixel, sample in rng.integers(0, 2**32, (2000, 3), dtype=np.uint64):
        assert lib.mipt_path_key(int(seed), int(pixel), int(sample)) == o.or_path_key(int(seed), int(pixel), int(sample))
    keys = {lib.mipt_path_key(0, p, s) for p in range(64) for s in range(64)}
    assert len(keys) == 64 * 64


def test_argument_checks_need_no_gpu(pkg, lib):
    abi = pkg.abi
    h = C.c_void_p()
    assert lib.mipt_scene_create(None, 3, None, 0, 0, C.byref(h)) == abi.MIPT_E_INVALID
    assert b"null primitive array" in lib.mipt_last_error()
    assert lib.mipt_scene_create(None, 0, None, 0, 0, None) == abi.MIPT_E_INVALID
    assert lib.mipt_closest_hit(None, None, None, 0, None, None, None) == abi.MIPT_E_INVALID
    st = abi.MiptStats()
    assert lib.mipt_get_stats(None, C.byref(st)) == abi.MIPT_E_INVALID
    # the entry points added beside them: option bits, the call that allocates nothing, frames
    assert lib.mipt_scene_create_opts(None, 0, None, 0, 0, 0x80, C.byref(h)) == abi.MIPT_E_INVALID
    assert b"option bits" in lib.mipt_last_error()
    cam, light, par = abi.MiptCamera(), abi.MiptLight(), abi.MiptParams()
    assert lib.mipt_render_accumulate_noalloc_dev(None, C.byref(cam), C.byref(light), C.byref(par), 0, 1, None, None) == abi.MIPT_E_INVALID
    assert lib.mipt_frame_poll(None) == abi.MIPT_E_INVALID
    assert lib.mipt_frame_create(None, 8, 8, C.byref(h)) == abi.MIPT_E_INVALID
    lib.mipt_frame_destroy(None)


def _box(r, F, faces=((0, 0), (0, 1), (1, 0), (1, 1), (2, 0), (2, 1))):
    """Triangles of the listed faces (axis, side) of the parallelepiped r + F [0,1]^3, two per face."""
    v = []
    for w, side in faces:
        ip, iq = (w + 1) % 3, (w + 2) % 3

        def at(p, q):
            c = np.zeros(3)
            c[w], c[ip], c[iq] = side, p, q
            return np.float32(r + F @ c)
        P = [at(0, 0), at(1, 0), at(1, 1), at(0, 1)]
        v += [(P[0], P[1], P[2]), (P[0], P[2], P[3])]
    return v


def test_scan_unit_planner_groups_walls_and_blocks(pkg, lib):
    # host logic of the linear scan's candidate filter, no GPU: which triangles become coplanar pairs,
    # which pairs lie in axis planes, which pairs are faces of one parallelepiped (box units)
    api, abi = pkg.api, pkg.abi
    # LoadTestModelGlassCube (TestModelH.h:351-482): 5 walls, 5 faces of the block, the mirror = 11 pairs; the
    # walls and the block's top in axis planes; the walls one box, the block's sides two two-face boxes
    assert api.plan_scan_units(api.load_test_model(abi.SCENE_GLASS_CUBE)) == (11, 6, 3)
    # LoadTestModel (TestModelH.h:55-201): 5 walls + 2 x (4 sides + top) = 15 pairs, 7 in axis planes, 5 boxes
    assert api.plan_scan_units(api.load_test_model(abi.SCENE_CORNELL)) == (15, 7, 5)
    assert api.plan_scan_units(np.zeros(0, dtype=abi.TRIANGLE_DTYPE)) == (0, 0, 0)

    def scene(tri_list):
        t = np.zeros(len(tri_list), dtype=abi.TRIANGLE_DTYPE)
        for k, (a, b, c) in enumerate(tri_list):
            t["v0"][k, :3], t["v1"][k, :3], t["v2"][k, :3] = a, b, c
            t["v0"][k, 3] = t["v1"][k, 3] = t["v2"][k, 3] = 1.0
        return t
    rng = np.random.default_rng(7)
    Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    F = Q * np.array([0.5, 0.3, 0.7])
    r = np.array([0.1, -0.2, 0.05])
    # a turned box with all six faces: 6 pairs, none in an axis plane, one box
    assert api.plan_scan_units(scene(_box(r, F))) == (6, 0, 1)
    # two turned faces that meet in an edge make a box, two axis-plane faces do not (their own form is as cheap)
    assert api.plan_scan_units(scene(_box(r, F, ((0, 0), (1, 0))))) == (2, 0, 1)
    assert api.plan_scan_units(scene(_box(r, np.diag([0.5, 0.3, 0.7]), ((0, 0), (1, 0))))) == (2, 2, 0)
    # two parallel faces share no edge: pairs, no box
    assert api.plan_scan_units(scene(_box(r, F, ((0, 0), (0, 1))))) == (2, 0, 0)
    # one corner moved by 10^-3 of the size: the faces at that corner are no parallelograms any more, the
    # three others still make a box
    moved = _box(r, F)
    far = np.float32(r + F @ np.ones(3))
    moved = [tuple(np.float32(p + F[:, 0] * 1e-3) if np.array_equal(p, far) else p for p in tri) for tri in moved]
    pairs, axis, boxes = api.plan_scan_units(scene(moved))
    assert boxes == 1 and axis == 0 and pairs <= 6
    assert lib.mipt_plan_scan_units(None, 3, None, None, None) == abi.MIPT_E_INVALID


def l_shapes(abi, n, seed=3):
    """n L-shaped pairs of turned quads (two faces of a turned box that meet in an edge), 4 triangles each,
    apart from one another: every one of them qualifies as a two-face box unit."""
    rng = np.random.default_rng(seed)
    tri_list = []
    for k in range(n):
        Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        F = Q * np.array([0.16, 0.11, 0.13])
        r = np.array([-0.8 + 0.42 * (k % 4), -0.6 + 0.5 * (k // 4), 0.1 * (k % 3)])
        tri_list += _box(r, F, ((0, 0), (1, 0)))
    t = np.zeros(len(tri_list), dtype=abi.TRIANGLE_DTYPE)
    for k, (a, b, c) in enumerate(tri_list):
        t["v0"][k, :3], t["v1"][k, :3], t["v2"][k, :3] = a, b, c
        t["v0"][k, 3] = t["v1"][k, 3] = t["v2"][k, 3] = 1.0
        e1, e2 = np.float32(b) - np.float32(a), np.float32(c) - np.float32(a)
        nrm = np.cross(e2, e1)
        t["normal"][k, :3] = nrm / max(np.linalg.norm(nrm), 1e-30)
        t["normal"][k, 3] = 1.0
    t["color"] = 0.7
    return t


def test_a_chunk_holds_at_most_seven_box_units(pkg, lib):
    # a chunk's record carries the number of its box units in three bits: eight two-face boxes in one
    # chunk of 32 triangles must come out as seven boxes + two pair units, not as a count that wraps to 0
    api, abi = pkg.api, pkg.abi
    for n, want_boxes in ((7, 7), (8, 7), (16, 14)):       # 16 L-shapes = two chunks of eight
        pairs, axis, boxes = api.plan_scan_units(l_shapes(abi, n))
        assert (pairs, axis, boxes) == (2 * n, 0, want_boxes), (n, pairs, axis, boxes)


def test_missing_library_fails_loudly(pkg, tmp_path):
    with pytest.raises(pkg.abi.MiptLibraryMissing):
        pkg.abi.load_library(str(tmp_path / "libmipt.so"))


def test_save_bmp_writes_what_sdl_wrote_for_the_reference(pkg, lib, tmp_path):
    # the reference ships a screenshot.bmp written by its SDL_SaveImage (500 x 500): same 122 header
    # bytes for the same size, rows bottom-up, ARGB words untouched
    rng = np.random.default_rng(3)
    screen = (rng.integers(0, 1 << 24, (500, 500), dtype=np.uint32) | np.uint32(128 << 24))
    path = tmp_path / "shot.bmp"
    pkg.api.save_bmp(path, screen)
    blob = path.read_bytes()
    golden = open(os.path.join(ROOT, "tests", "golden", "screenshot_bmp_header_500x500.bin"), "rb").read()
    assert blob[:122] == golden and len(blob) == 122 + 500 * 500 * 4
    rows = np.frombuffer(blob, dtype="<u4", offset=122).reshape(500, 500)
    assert np.array_equal(rows[::-1], screen)
    assert lib.mipt_save_bmp(None, None, 4, 4) == pkg.abi.MIPT_E_INVALID
    small = np.zeros((3, 5), dtype=np.uint32)
    pkg.api.save_bmp(tmp_path / "s.bmp", small)
    assert (tmp_path / "s.bmp").stat().st_size == 122 + 60


def test_host_entry_points_run_clean_under_the_host_sanitizers():
    # AddressSanitizer + UndefinedBehaviorSanitizer over the library's host-only entry points: the scene
    # builders, the OBJ loader on the reference's wolf and on a dozen malformed files (indices out of
    # range, degenerate polygons, garbage tokens, a 100 000-character line, NUL bytes), normaliseTriangles
    # and the .bmp writer (computer-graphics_amd/host/san_host.cpp; no GPU code is linked); and over the
    # scan-unit planner -- test models and all their prefixes, random boxes with faces missing, eight
    # two-face boxes in one chunk, NaN / infinite / enormous / zero-area triangles -- plus the entry
    # points that only compute on the host, in the library's own translation unit built with the
    # sanitizers on its host side (san_plan.cpp; no kernel is launched)
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-C", os.path.join(root, "computer-graphics_amd", "host"), "sanitize"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "host sanitizer run ok" in r.stdout and "planner sanitizer run ok" in r.stdout
