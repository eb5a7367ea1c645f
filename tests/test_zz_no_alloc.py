"""The entry point that allocates nothing (mipt_render_accumulate_noalloc_dev) against
mipt_render_accumulate_dev: bit-equal images on two calls in a row, the second smaller, and a
clear error beyond the fixed scratch size. Through the C ABI (ctypes)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_no_alloc_entry_point_equals_render_accumulate_dev(pkg, O):
    import torch
    assert torch.cuda.is_available(), "this test needs the MI355X"
    abi, api = pkg.abi, pkg.api
    tris = O.load_test_model(2)
    sph = api.make_sphere(0.3, [-0.45, 0.7, -0.55, 1], [1, 0, 0])
    dev = torch.device("cuda", 0)
    for spheres in (None, sph):
        sc = api.Scene(tris, spheres)
        # first call: 40 samples (three sample blocks, the last ragged); second call: a smaller image
        # and fewer samples, which leaves stale sums of the first call in the shared array
        for (W, H, begin, count, seed, fog) in ((320, 200, 0, 40, 5, 0.0), (96, 72, 7, 9, 6, 0.2)):
            cam, light = O.default_camera(W), O.default_light(True)
            p = O.default_params(W, H, seed=seed, fog_sigma_t=fog)
            ref = torch.zeros(W * H * 3, dtype=torch.float32, device=dev)
            out = torch.zeros_like(ref)
            sc.render_accumulate_dev(cam, light, p, begin, count, ref)
            sc.render_accumulate_noalloc_dev(cam, light, p, begin, count, out)
            torch.cuda.synchronize()
            assert ref.abs().sum().item() > 0
            assert torch.equal(ref.view(torch.int32), out.view(torch.int32)), (W, H, count)
            # samples are ADDED, as in the existing entry point
            sc.render_accumulate_dev(cam, light, p, begin + count, 3, ref)
            sc.render_accumulate_noalloc_dev(cam, light, p, begin + count, 3, out)
            torch.cuda.synchronize()
            assert torch.equal(ref.view(torch.int32), out.view(torch.int32))
        sc.close()


def test_no_alloc_entry_point_rejects_oversized_calls(pkg, O):
    import torch
    abi, api = pkg.abi, pkg.api
    sc = api.Scene(O.load_test_model(0))
    W = H = 1024
    cam, light = O.default_camera(W), O.default_light(False)
    p = O.default_params(W, H)
    buf = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda:0")
    per_block = W * H * 12
    fits = abi.NOALLOC_MAX_BYTES // per_block * 16          # 32 samples
    sc.render_accumulate_noalloc_dev(cam, light, p, 0, fits, buf)     # exactly at the limit
    with pytest.raises(api.MiptError) as e:
        sc.render_accumulate_noalloc_dev(cam, light, p, 0, fits + 1, buf)
    assert e.value.code == abi.MIPT_E_INVALID and "MIPT_NOALLOC_MAX_BYTES" in str(e.value)
    torch.cuda.synchronize()
    ref = torch.zeros_like(buf)
    sc.render_accumulate_dev(cam, light, p, 0, fits, ref)
    torch.cuda.synchronize()
    assert torch.equal(ref.view(torch.int32), buf.view(torch.int32))
    sc.close()


def test_no_alloc_entry_point_under_graph_capture(pkg, O):
    # the callers the entry point is for: its three operations (memset of the queue head, path kernel,
    # ordered sum) are captured into a HIP graph on a side stream and replayed; every replay adds the
    # captured samples again, exactly as the plain entry point called that often
    import torch
    abi, api = pkg.abi, pkg.api
    tris = O.load_test_model(2)
    sc = api.Scene(tris)
    W, H = 160, 96
    cam, light = O.default_camera(W), O.default_light(True)
    p = O.default_params(W, H, seed=12, fog_sigma_t=0.1)
    dev = torch.device("cuda", 0)
    out = torch.zeros(W * H * 3, dtype=torch.float32, device=dev)
    ref = torch.zeros_like(out)
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            sc.render_accumulate_noalloc_dev(cam, light, p, 0, 32, out)
    torch.cuda.synchronize()
    assert not out.any().item()          # capture records, it does not run
    for _ in range(3):
        graph.replay()
        sc.render_accumulate_dev(cam, light, p, 0, 32, ref)
    torch.cuda.synchronize()
    assert ref.abs().sum().item() > 0
    assert torch.equal(ref.view(torch.int32), out.view(torch.int32))
    with pytest.raises(api.MiptError) as e:     # the counters are scene scratch: not through this entry point
        ps = O.default_params(W, H, seed=12, flags=abi.FLAG_STATS)
        sc.render_accumulate_noalloc_dev(cam, light, ps, 0, 16, out)
    assert e.value.code == abi.MIPT_E_INVALID
    del graph
    sc.close()
