"""ctypes wrapper around oracle/_build/libpt_oracle.so (the CPU checker).

Only tests, smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _pkg():
    import __graft_entry__ as g
    return g.load_package()


def build_oracle():
    subprocess.run(["make", "-C", ORACLE_DIR, "all"], check=True, capture_output=True)


_libs = {}


def oracle(fast=False):
    name = "libpt_oracle_fast.so" if fast else "libpt_oracle.so"
    if name in _libs:
        return _libs[name]
    path = os.path.join(ORACLE_DIR, "_build", name)
    if not os.path.exists(path):
        build_oracle()
    abi = _pkg().abi
    lib = C.CDLL(path)
    vp = C.c_void_p
    P = C.POINTER
    lib.or_load_test_model.argtypes = [C.c_int, vp]
    lib.or_default_camera.argtypes = [P(abi.MiptCamera), C.c_int]
    lib.or_default_camera.restype = None
    lib.or_default_light.argtypes = [P(abi.MiptLight)]
    lib.or_default_light.restype = None
    lib.or_params_default.argtypes = [P(abi.MiptParams), C.c_int, C.c_int]
    lib.or_params_default.restype = None
    lib.or_rng_canonical.argtypes = [P(C.c_uint32)]
    lib.or_rng_canonical.restype = C.c_float
    lib.or_path_key.argtypes = [C.c_uint32] * 3
    lib.or_path_key.restype = C.c_uint32
    lib.or_counter_uniform.argtypes = [C.c_uint32, C.c_uint32]
    lib.or_counter_uniform.restype = C.c_float
    lib.or_closest_hit.argtypes = [vp, C.c_size_t, vp, C.c_size_t, vp, vp, C.c_size_t, vp, vp, vp]
    lib.or_any_hit.argtypes = [vp, C.c_size_t, vp, vp, vp, C.c_size_t, vp]
    lib.or_trace_paths.argtypes = [vp, C.c_size_t, vp, C.c_size_t, P(abi.MiptCamera),
                                   P(abi.MiptLight), P(abi.MiptParams), vp, vp, vp, C.c_size_t, vp, vp]
    lib.or_render_accumulate.argtypes = [vp, C.c_size_t, vp, C.c_size_t, P(abi.MiptCamera),
                                         P(abi.MiptLight), P(abi.MiptParams), C.c_uint32, C.c_uint32,
                                         vp, C.c_int, P(abi.MiptStats)]
    lib.or_resolve_argb.argtypes = [vp, C.c_int, C.c_int, C.c_uint32, vp]
    lib.or_resolve_argb.restype = None
    lib.or_fog_expf.argtypes = [C.c_float]
    lib.or_fog_expf.restype = C.c_float
    lib.or_fog_logf.argtypes = [C.c_float]
    lib.or_fog_logf.restype = C.c_float
    lib.or_det3.argtypes = [vp, vp, vp]
    lib.or_det3.restype = C.c_float
    lib.or_inverse3_mul.argtypes = [vp] * 5
    lib.or_inverse3_mul.restype = None
    lib.or_dot3.argtypes = [vp, vp]
    lib.or_dot3.restype = C.c_float
    lib.or_dot4.argtypes = [vp, vp]
    lib.or_dot4.restype = C.c_float
    lib.or_cross3.argtypes = [vp] * 3
    lib.or_cross3.restype = None
    lib.or_normalize3.argtypes = [vp, vp]
    lib.or_normalize3.restype = None
    lib.or_normalize4.argtypes = [vp, vp]
    lib.or_normalize4.restype = None
    lib.or_length4.argtypes = [vp]
    lib.or_length4.restype = C.c_float
    lib.or_mat4_mul.argtypes = [vp] * 3
    lib.or_mat4_mul.restype = None
    lib.or_mat4_mul_vec4.argtypes = [vp] * 3
    lib.or_mat4_mul_vec4.restype = None
    _libs[name] = lib
    return lib


def ptr(a):
    return None if a is None else a.ctypes.data


def load_test_model(which):
    abi = _pkg().abi
    tris = np.zeros(30, dtype=abi.TRIANGLE_DTYPE)
    n = oracle().or_load_test_model(which, ptr(tris))
    assert n > 0
    return tris[:n].copy()


def golden_scene(name):
    abi = _pkg().abi
    return np.fromfile(os.path.join(GOLDEN, f"scene_{name}.bin"), dtype=abi.TRIANGLE_DTYPE)


def default_camera(width):
    abi = _pkg().abi
    cam = abi.MiptCamera()
    oracle().or_default_camera(C.byref(cam), width)
    return cam


def default_light(glass=False):
    abi = _pkg().abi
    light = abi.MiptLight()
    oracle().or_default_light(C.byref(light))
    if glass:  # main.cpp:68-69
        light.pos[:] = [np.float32(0.7), np.float32(0.95), np.float32(-0.7), 1.0]
    return light


def default_params(width, height, **kw):
    abi = _pkg().abi
    p = abi.MiptParams()
    oracle().or_params_default(C.byref(p), width, height)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def closest_hit(tris, spheres, origin4, dir4, fast=False):
    n = origin4.shape[0]
    idx = np.full(n, -7, dtype=np.int32)
    t = np.zeros(n, dtype=np.float32)
    pos = np.zeros((n, 4), dtype=np.float32)
    ns = 0 if spheres is None else len(spheres)
    oracle(fast).or_closest_hit(ptr(tris), len(tris), ptr(spheres) if ns else None, ns,
                                ptr(origin4), ptr(dir4), n, ptr(idx), ptr(t), ptr(pos))
    return idx, t, pos


def any_hit(tris, origin4, dir4, max_t):
    n = origin4.shape[0]
    occ = np.zeros(n, dtype=np.uint8)
    oracle().or_any_hit(ptr(tris), len(tris), ptr(origin4), ptr(dir4), ptr(max_t), n, ptr(occ))
    return occ


def trace_paths(tris, spheres, cam, light, params, px, py, sample):
    n = len(px)
    rgb = np.zeros((n, 3), dtype=np.float32)
    counts = np.zeros((n, 2), dtype=np.uint32)
    ns = 0 if spheres is None else len(spheres)
    rc = oracle().or_trace_paths(ptr(tris), len(tris), ptr(spheres) if ns else None, ns,
                                 C.byref(cam), C.byref(light), C.byref(params),
                                 ptr(px), ptr(py), ptr(sample), n, ptr(rgb), ptr(counts))
    assert rc == 0
    return rgb, counts


def render_accumulate(tris, spheres, cam, light, params, sample_begin, sample_count, accum=None,
                      threads=1, fast=False):
    abi = _pkg().abi
    W, H = params.width, params.height
    if accum is None:
        accum = np.zeros((W, H, 3), dtype=np.float32)
    stats = abi.MiptStats()
    ns = 0 if spheres is None else len(spheres)
    rc = oracle(fast).or_render_accumulate(ptr(tris), len(tris), ptr(spheres) if ns else None, ns,
                                           C.byref(cam), C.byref(light), C.byref(params),
                                           sample_begin, sample_count, ptr(accum), threads,
                                           C.byref(stats))
    assert rc == 0
    return accum, stats


def resolve_argb(accum, n_samples):
    W, H = accum.shape[0], accum.shape[1]
    out = np.zeros((H, W), dtype=np.uint32)
    oracle().or_resolve_argb(ptr(accum), W, H, n_samples, ptr(out))
    return out
