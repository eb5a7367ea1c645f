"""ctypes view of include/mipt.h: POD structs, error codes and the library loader.

The structs mirror the reference's scene model one to one:
``Triangle`` (CW/raytracer/Source/TestModelH.h:11-49), ``Sphere`` (sphere.h:9-16),
``Camera`` / ``Light`` (definitions.h:16-37); the material string of
material.h:5-11 becomes the ``MAT_*`` code.
"""
import ctypes as C
import os

import numpy as np

MIPT_OK = 0
MIPT_E_INVALID = -1
MIPT_E_HIP = -2
MIPT_E_NOMEM = -3
MIPT_E_UNSUPPORTED = -4
MIPT_E_RCCL = -5

MAT_DIFFUSE, MAT_MIRROR, MAT_GLASS, MAT_SPECULAR, MAT_NONE = range(5)
MATERIAL_NAMES = {"diffuse": MAT_DIFFUSE, "mirror": MAT_MIRROR, "glass": MAT_GLASS,
                  "specular": MAT_SPECULAR}

RNG_MINSTD_PATH, RNG_COUNTER, RNG_MINSTD_GLOBAL = 0, 1, 2
FLAG_STATS = 1
SCENE_LINEAR_SCAN = 1  # MIPT_SCENE_LINEAR_SCAN
MULTI_ALLOW_SHARED_DEVICE = 1  # MIPT_MULTI_ALLOW_SHARED_DEVICE
MULTI_FRAME_REDUCE, MULTI_FRAME_EXACT = 0, 1  # MIPT_MULTI_FRAME_*
MAX_DEPTH_LIMIT = 16
NOALLOC_MAX_BYTES = 32 << 20  # MIPT_NOALLOC_MAX_BYTES

SCENE_CORNELL, SCENE_COLOR_BLEEDING, SCENE_GLASS_CUBE, SCENE_NONE = range(4)


class MiptTriangle(C.Structure):
    _fields_ = [("v0", C.c_float * 4), ("v1", C.c_float * 4), ("v2", C.c_float * 4),
                ("color", C.c_float * 3), ("normal", C.c_float * 4), ("material", C.c_uint32)]


class MiptSphere(C.Structure):
    _fields_ = [("radius", C.c_float), ("center", C.c_float * 4), ("color", C.c_float * 3),
                ("material", C.c_uint32)]


class MiptCamera(C.Structure):
    _fields_ = [("focal", C.c_float), ("pos", C.c_float * 4), ("rot_x", C.c_float * 16),
                ("rot_y", C.c_float * 16)]


class MiptLight(C.Structure):
    _fields_ = [("pos", C.c_float * 4), ("color", C.c_float * 3)]


class MiptParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("max_depth", C.c_int32),
                ("rr_prob", C.c_float), ("diffuse_weight", C.c_float), ("glass_weight", C.c_float),
                ("ior_glass", C.c_float), ("ior_air", C.c_float), ("rng_mode", C.c_uint32),
                ("seed", C.c_uint32), ("fog_sigma_t", C.c_float), ("fog_albedo", C.c_float),
                ("flags", C.c_uint32)]


class MiptStats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("tri_tests", C.c_uint64), ("gate_passes", C.c_uint64), ("sphere_tests", C.c_uint64),
                ("fog_events", C.c_uint64), ("diffuse_bounces", C.c_uint64),
                ("mirror_bounces", C.c_uint64), ("glass_bounces", C.c_uint64), ("bvh_redo", C.c_uint64),
                ("bvh_overflow", C.c_uint64), ("bvh_nodes", C.c_uint64), ("kernel_ms", C.c_double),
                ("specular_bounces", C.c_uint64)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


# numpy dtype with the exact memory layout of MiptTriangle (80 bytes) / MiptSphere (36 bytes)
TRIANGLE_DTYPE = np.dtype([("v0", "<f4", 4), ("v1", "<f4", 4), ("v2", "<f4", 4),
                           ("color", "<f4", 3), ("normal", "<f4", 4), ("material", "<u4")])
SPHERE_DTYPE = np.dtype([("radius", "<f4"), ("center", "<f4", 4), ("color", "<f4", 3),
                         ("material", "<u4")])
assert TRIANGLE_DTYPE.itemsize == C.sizeof(MiptTriangle) == 80
assert SPHERE_DTYPE.itemsize == C.sizeof(MiptSphere) == 36

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libmipt.so")

# every symbol include/mipt.h declares; tests check the built library exports all of them
EXPORTS = [
    "mipt_params_default", "mipt_last_error", "mipt_version", "mipt_scene_create",
    "mipt_scene_destroy", "mipt_closest_hit", "mipt_any_hit", "mipt_render_accumulate",
    "mipt_resolve_argb", "mipt_trace_paths", "mipt_closest_hit_dev", "mipt_any_hit_dev",
    "mipt_render_accumulate_dev", "mipt_resolve_argb_dev", "mipt_get_stats", "mipt_path_key",
    "mipt_load_test_model", "mipt_default_camera", "mipt_default_light", "mipt_debug_selfcheck", "mipt_scene_has_bvh", "mipt_scene_tree_info", "mipt_scene_pair_units", "mipt_scene_axis_pair_units", "mipt_scene_box_units", "mipt_plan_scan_units", "mipt_load_obj", "mipt_normalise_triangles", "mipt_save_bmp", "mipt_frame_create", "mipt_frame_destroy", "mipt_frame_reset",
    "mipt_frame_add_samples", "mipt_frame_present", "mipt_frame_read_pixels", "mipt_frame_samples",
    "mipt_render_accumulate_noalloc_dev", "mipt_scene_create_opts",
    "mipt_multi_create", "mipt_multi_destroy", "mipt_multi_device_count", "mipt_multi_render_accumulate",
    "mipt_multi_sample_range", "mipt_multi_scene", "mipt_frame_poll",
    "mipt_render_items_dev", "mipt_fold_items_dev", "mipt_multi_render_accumulate_chained", "mipt_multi_item_range",
    "mipt_multi_create_opts", "mipt_multi_frame_create", "mipt_multi_frame_destroy", "mipt_multi_frame_reset",
    "mipt_multi_frame_add_samples", "mipt_multi_frame_present", "mipt_multi_frame_read_pixels",
    "mipt_multi_frame_samples", "mipt_multi_frame_device", "mipt_multi_frame_poll",
]


class MiptLibraryMissing(RuntimeError):
    pass


_lib = None


def load_library(path=None):
    """dlopen libmipt.so (built in-tree by __graft_entry__.build()). There is no
    fallback: without the HIP library the product path does not exist."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise MiptLibraryMissing(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # PyTorch ships a HIP runtime of its own (torch/lib/libamdhip64.so) and libmipt.so names the
    # system's by the same soname: whichever the process loads first serves both. Loaded second,
    # torch finds no GPU through the other one; so torch goes first wherever it is installed (the
    # *_dev methods take its tensors anyway). A C++ host has no such concern.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(p)
    P = C.POINTER
    vp = C.c_void_p
    f32p, i32p, u32p, u8p = vp, vp, vp, vp  # raw addresses; callers pass numpy/torch data pointers
    lib.mipt_params_default.argtypes = [P(MiptParams), C.c_int, C.c_int]
    lib.mipt_params_default.restype = None
    lib.mipt_last_error.restype = C.c_char_p
    lib.mipt_version.restype = C.c_char_p
    lib.mipt_scene_create.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.c_int, P(vp)]
    lib.mipt_scene_create_opts.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.c_int, C.c_uint32, P(vp)]
    lib.mipt_scene_destroy.argtypes = [vp]
    lib.mipt_scene_destroy.restype = None
    lib.mipt_closest_hit.argtypes = [vp, f32p, f32p, C.c_size_t, i32p, f32p, f32p]
    lib.mipt_any_hit.argtypes = [vp, f32p, f32p, f32p, C.c_size_t, u8p]
    lib.mipt_render_accumulate.argtypes = [vp, P(MiptCamera), P(MiptLight), P(MiptParams),
                                           C.c_uint32, C.c_uint32, f32p]
    lib.mipt_resolve_argb.argtypes = [vp, f32p, C.c_int, C.c_int, C.c_uint32, u32p]
    lib.mipt_trace_paths.argtypes = [vp, P(MiptCamera), P(MiptLight), P(MiptParams), i32p, i32p,
                                     u32p, C.c_size_t, f32p, u32p]
    lib.mipt_closest_hit_dev.argtypes = [vp, f32p, f32p, C.c_size_t, i32p, f32p, f32p, vp]
    lib.mipt_any_hit_dev.argtypes = [vp, f32p, f32p, f32p, C.c_size_t, u8p, vp]
    lib.mipt_render_accumulate_dev.argtypes = [vp, P(MiptCamera), P(MiptLight), P(MiptParams),
                                               C.c_uint32, C.c_uint32, f32p, vp]
    lib.mipt_resolve_argb_dev.argtypes = [vp, f32p, C.c_int, C.c_int, C.c_uint32, u32p, vp]
    lib.mipt_render_accumulate_noalloc_dev.argtypes = lib.mipt_render_accumulate_dev.argtypes
    lib.mipt_render_items_dev.argtypes = [vp, P(MiptCamera), P(MiptLight), P(MiptParams), C.c_uint32, C.c_uint32, vp]
    lib.mipt_fold_items_dev.argtypes = [vp, f32p, vp]
    lib.mipt_get_stats.argtypes = [vp, P(MiptStats)]
    lib.mipt_multi_create.argtypes = [vp, C.c_size_t, vp, C.c_size_t, P(C.c_int), C.c_int, P(vp)]
    lib.mipt_multi_create_opts.argtypes = [vp, C.c_size_t, vp, C.c_size_t, P(C.c_int), C.c_int, C.c_uint32, P(vp)]
    lib.mipt_multi_destroy.argtypes = [vp]
    lib.mipt_multi_destroy.restype = None
    lib.mipt_multi_frame_create.argtypes = [vp, C.c_int, C.c_int, C.c_uint32, P(vp)]
    lib.mipt_multi_frame_destroy.argtypes = [vp]
    lib.mipt_multi_frame_destroy.restype = None
    lib.mipt_multi_frame_reset.argtypes = [vp]
    lib.mipt_multi_frame_add_samples.argtypes = [vp, P(MiptCamera), P(MiptLight), P(MiptParams), C.c_uint32]
    lib.mipt_multi_frame_present.argtypes = [vp, vp]
    lib.mipt_multi_frame_read_pixels.argtypes = [vp, vp]
    lib.mipt_multi_frame_samples.argtypes = [vp]
    lib.mipt_multi_frame_device.argtypes = [vp]
    lib.mipt_multi_frame_poll.argtypes = [vp]
    lib.mipt_multi_device_count.argtypes = [vp]
    lib.mipt_multi_scene.argtypes = [vp, C.c_int]
    lib.mipt_multi_scene.restype = vp
    lib.mipt_multi_render_accumulate.argtypes = [vp, P(MiptCamera), P(MiptLight), P(MiptParams),
                                                 C.c_uint32, C.c_uint32, f32p]
    lib.mipt_multi_sample_range.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int, P(C.c_uint32), P(C.c_uint32)]
    lib.mipt_multi_sample_range.restype = None
    lib.mipt_multi_item_range.argtypes = lib.mipt_multi_sample_range.argtypes
    lib.mipt_multi_item_range.restype = None
    lib.mipt_multi_render_accumulate_chained.argtypes = lib.mipt_multi_render_accumulate.argtypes
    lib.mipt_path_key.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    lib.mipt_path_key.restype = C.c_uint32
    lib.mipt_debug_selfcheck.argtypes = [vp, C.c_uint64, C.c_uint32, P(C.c_uint64 * 8)]
    lib.mipt_scene_has_bvh.argtypes = [vp, P(C.c_uint64), P(C.c_double)]
    lib.mipt_scene_tree_info.argtypes = [vp, P(C.c_int), P(C.c_int), P(C.c_float)]
    lib.mipt_scene_pair_units.argtypes = [vp]
    lib.mipt_scene_axis_pair_units.argtypes = [vp]
    lib.mipt_scene_box_units.argtypes = [vp]
    lib.mipt_plan_scan_units.argtypes = [vp, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.mipt_save_bmp.argtypes = [C.c_char_p, vp, C.c_int, C.c_int]
    lib.mipt_frame_create.argtypes = [vp, C.c_int, C.c_int, P(vp)]
    lib.mipt_frame_destroy.argtypes = [vp]
    lib.mipt_frame_destroy.restype = None
    lib.mipt_frame_reset.argtypes = [vp]
    lib.mipt_frame_add_samples.argtypes = [vp, P(MiptCamera), P(MiptLight), P(MiptParams), C.c_uint32]
    lib.mipt_frame_present.argtypes = [vp, vp]
    lib.mipt_frame_read_pixels.argtypes = [vp, vp]
    lib.mipt_frame_samples.argtypes = [vp]
    lib.mipt_frame_poll.argtypes = [vp]
    lib.mipt_load_obj.argtypes = [C.c_char_p, vp, C.c_size_t]
    lib.mipt_normalise_triangles.argtypes = [vp, C.c_size_t, C.c_float, P(C.c_float * 3), P(C.c_float * 3),
                                             P(C.c_float * 3), C.c_uint32]
    lib.mipt_load_test_model.argtypes = [C.c_int, vp, C.c_size_t]
    lib.mipt_default_camera.argtypes = [P(MiptCamera), C.c_int]
    lib.mipt_default_camera.restype = None
    lib.mipt_default_light.argtypes = [P(MiptLight), C.c_int]
    lib.mipt_default_light.restype = None
    for name in EXPORTS:
        getattr(lib, name)  # AttributeError here means the library is stale
    if path is None:
        _lib = lib
    return lib
