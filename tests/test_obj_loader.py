"""OBJ ingestion and placement (LoadObject, parser.cpp:8-46; normaliseTriangles, utils.cpp:52-140)
against golden vectors produced by the reference's OWN loader code: oracle/ref_harness/
ref_obj_dump.cpp compiles OBJ_Loader.h, TestModelH.h and GLM from the reference tree and dumps the
triangles (`make -C oracle golden`). wolf.obj is the reference's asset (CW/raytracer/Objects/),
kept as a data fixture; cases.obj is synthetic (every vertex form, negative indices, a quad,
a convex and a concave polygon, CRLF). Bit-exact on every field. CPU only: host code of libmipt.so."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "obj")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_same_triangles(ours, ref):
    assert len(ours) == len(ref)
    for field in ("v0", "v1", "v2", "color", "normal"):
        bad = np.nonzero((bits(ours[field]) != bits(ref[field])).any(axis=1))[0]
        assert len(bad) == 0, (field, bad[:5], ours[field][bad[:2]], ref[field][bad[:2]])
    assert np.array_equal(ours["material"], ref["material"])


@pytest.fixture(scope="module")
def api(pkg):
    if not os.path.exists(pkg.abi.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    pkg.abi.load_library()
    return pkg.api


def golden(pkg, name):
    return np.fromfile(os.path.join(GOLD, name), dtype=pkg.abi.TRIANGLE_DTYPE)


def glasswolf(pkg, api):
    """main.cpp:78-89: the empty room, then the wolf placed as glass."""
    wolf = api.load_obj(os.path.join(GOLD, "wolf.obj"))
    placed = api.normalise_triangles(wolf, 1.0, (-0.2, -1, -0.5), (3.15, -2.5, 0), (-1, 1, 1), pkg.abi.MAT_GLASS)
    return np.concatenate([api.load_test_model(3), placed])


def test_wolf_loads_like_the_reference(pkg, api):
    wolf = api.load_obj(os.path.join(GOLD, "wolf.obj"))
    assert len(wolf) == 1784
    assert_same_triangles(wolf, golden(pkg, "wolf_raw.bin"))
    # normals are unnormalised cross products (no vn records in the file)
    assert np.allclose(wolf["normal"][0], [3.53081, -12.5208, -22.5578, 1], rtol=1e-5)


def test_glasswolf_scene_matches_the_reference(pkg, api):
    scene = glasswolf(pkg, api)
    assert len(scene) == 1794
    assert_same_triangles(scene, golden(pkg, "glasswolf_scene.bin"))
    # placement divides w too: w = 1 / (scale * L)
    assert 0 < scene["v0"][10][3] < 0.01 and (scene["material"][10:] == pkg.abi.MAT_GLASS).all()


def test_every_face_form_and_polygon(pkg, api):
    raw = api.load_obj(os.path.join(GOLD, "cases.obj"))
    assert_same_triangles(raw, golden(pkg, "cases_raw.bin"))
    placed = api.normalise_triangles(raw, 0.5, (0.1, -0.2, 0.3), (0.4, 1.1, -0.7), (1, -1, 0.5), pkg.abi.MAT_MIRROR)
    assert_same_triangles(placed, golden(pkg, "cases_placed.bin"))


def test_loader_errors(pkg, api, tmp_path):
    with pytest.raises(api.MiptError):
        api.load_obj(tmp_path / "missing.obj")
    (tmp_path / "a.txt").write_text("v 0 0 0\n")
    with pytest.raises(api.MiptError):
        api.load_obj(tmp_path / "a.txt")
    (tmp_path / "bad.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 7\n")
    with pytest.raises(api.MiptError) as e:
        api.load_obj(tmp_path / "bad.obj")
    assert "line 4" in str(e.value)
    (tmp_path / "empty.obj").write_text("# nothing\n")
    assert len(api.load_obj(tmp_path / "empty.obj")) == 0
