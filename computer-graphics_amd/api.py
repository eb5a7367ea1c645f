"""Host-side mirror of the reference's scene API over the C ABI.

Names follow the reference: ``load_test_model`` = LoadTestModel* (TestModelH.h),
``Scene.closest_hit`` = closestIntersection, ``Scene.any_hit`` = existsIntersection
(intersections.h:7-13), ``Scene.render_accumulate`` = the pixel loop of Draw
(main.cpp:121-139), ``Scene.resolve_argb`` = its tonemap loop (main.cpp:141-145).
numpy arrays are host buffers (the library copies them); torch CUDA tensors are
passed by device pointer and run on torch's current stream.
"""
import ctypes as C

import numpy as np

from . import abi


class MiptError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mipt error {code}: {msg}")
        self.code = code


def _check(rc):
    if rc != abi.MIPT_OK:
        raise MiptError(rc, abi.load_library().mipt_last_error().decode())


def _ptr(a):
    return None if a is None else a.ctypes.data


def _f32(a, shape_last=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape_last is not None and (a.ndim != 2 or a.shape[1] != shape_last):
        raise ValueError(f"expected an (n, {shape_last}) float32 array, got {a.shape}")
    return a


def load_test_model(which):
    """LoadTestModel / ...ColorBleeding / ...GlassCube / ...None as a TRIANGLE_DTYPE array."""
    lib = abi.load_library()
    tris = np.zeros(30, dtype=abi.TRIANGLE_DTYPE)
    n = lib.mipt_load_test_model(which, _ptr(tris), 30)
    if n < 0:
        raise MiptError(n, lib.mipt_last_error().decode())
    return tris[:n].copy()


def plan_scan_units(tris):
    """(coplanar pairs, pairs in axis planes, boxes) the linear scan's candidate filter would form over
    these triangles; computed on the host, no GPU needed."""
    lib = abi.load_library()
    tris = np.ascontiguousarray(tris)
    pairs, axis, boxes = C.c_int(), C.c_int(), C.c_int()
    _check(lib.mipt_plan_scan_units(_ptr(tris), len(tris), C.byref(pairs), C.byref(axis), C.byref(boxes)))
    return pairs.value, axis.value, boxes.value


def load_obj(path):
    """LoadObject (parser.cpp:8-46): the triangles of an OBJ file as a TRIANGLE_DTYPE array."""
    lib = abi.load_library()
    n = lib.mipt_load_obj(str(path).encode(), None, 0)
    if n < 0:
        raise MiptError(n, lib.mipt_last_error().decode())
    tris = np.zeros(n, dtype=abi.TRIANGLE_DTYPE)
    if n:
        n2 = lib.mipt_load_obj(str(path).encode(), _ptr(tris), n)
        if n2 != n:
            raise MiptError(n2, lib.mipt_last_error().decode())
    return tris


def normalise_triangles(tris, scale=1.0, displacement=(0, 0, 0), rotate=(0, 0, 0), adjust_normal=(1, 1, 1),
                        material=abi.MAT_DIFFUSE):
    """normaliseTriangles (utils.cpp:52-140); returns a placed copy."""
    out = np.ascontiguousarray(tris, dtype=abi.TRIANGLE_DTYPE).copy()
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    _check(abi.load_library().mipt_normalise_triangles(_ptr(out) if len(out) else None, len(out), float(scale),
                                                       C.byref(f3(displacement)), C.byref(f3(rotate)),
                                                       C.byref(f3(adjust_normal)), int(material)))
    return out


def save_bmp(path, screen):
    """SDL_SaveImage (SDLauxiliary.h:25-54): screen is the (H, W) uint32 ARGB buffer of resolve_argb."""
    screen = np.ascontiguousarray(screen, dtype=np.uint32)
    _check(abi.load_library().mipt_save_bmp(str(path).encode(), _ptr(screen), screen.shape[1], screen.shape[0]))


def default_camera(width):
    cam = abi.MiptCamera()
    abi.load_library().mipt_default_camera(C.byref(cam), width)
    return cam


def default_light(glass=False):
    """reset_light (light.cpp:6-9); glass=True gives the light of the glass scenes (main.cpp:68-69)."""
    light = abi.MiptLight()
    abi.load_library().mipt_default_light(C.byref(light), 1 if glass else 0)
    return light


def default_params(width, height, **overrides):
    p = abi.MiptParams()
    abi.load_library().mipt_params_default(C.byref(p), width, height)
    for k, v in overrides.items():
        setattr(p, k, v)
    return p


def make_sphere(radius, center, color, material=abi.MAT_GLASS):
    s = np.zeros(1, dtype=abi.SPHERE_DTYPE)
    s["radius"] = radius
    s["center"] = center
    s["color"] = color
    s["material"] = material
    return s


def path_key(seed, pixel, sample):
    return abi.load_library().mipt_path_key(seed, pixel, sample)


class Scene:
    """Owns a MiptScene handle (device copy of the triangles / spheres)."""

    def __init__(self, tris, spheres=None, device=0, linear_scan=False):
        self._lib = abi.load_library()
        tris = np.ascontiguousarray(tris, dtype=abi.TRIANGLE_DTYPE)
        ns = 0 if spheres is None else len(spheres)
        if ns:
            spheres = np.ascontiguousarray(spheres, dtype=abi.SPHERE_DTYPE)
        self.n_tris, self.n_spheres, self.device = len(tris), ns, device
        h = C.c_void_p()
        _check(self._lib.mipt_scene_create_opts(_ptr(tris) if len(tris) else None, len(tris),
                                                _ptr(spheres) if ns else None, ns, device,
                                                abi.SCENE_LINEAR_SCAN if linear_scan else 0, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mipt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host-pointer entry points (numpy) -------------------------------------------------
    def closest_hit(self, origin4, dir4, want_position=True):
        o, d = _f32(origin4, 4), _f32(dir4, 4)
        if o.shape != d.shape:
            raise ValueError("origin4 and dir4 differ in shape")
        n = o.shape[0]
        idx = np.full(n, -7, dtype=np.int32)
        t = np.zeros(n, dtype=np.float32)
        pos = np.zeros((n, 4), dtype=np.float32) if want_position else None
        _check(self._lib.mipt_closest_hit(self._h, _ptr(o), _ptr(d), n, _ptr(idx), _ptr(t), _ptr(pos)))
        return idx, t, pos

    def any_hit(self, origin4, dir4, max_t):
        o, d = _f32(origin4, 4), _f32(dir4, 4)
        m = np.ascontiguousarray(max_t, dtype=np.float32)
        n = o.shape[0]
        if d.shape[0] != n or m.shape != (n,):
            raise ValueError("origin4, dir4 and max_t differ in length")
        occ = np.zeros(n, dtype=np.uint8)
        _check(self._lib.mipt_any_hit(self._h, _ptr(o), _ptr(d), _ptr(m), n, _ptr(occ)))
        return occ

    def render_accumulate(self, cam, light, params, sample_begin, sample_count, accum=None):
        W, H = params.width, params.height
        if accum is None:
            accum = np.zeros((W, H, 3), dtype=np.float32)
        if accum.dtype != np.float32 or accum.shape != (W, H, 3) or not accum.flags.c_contiguous:
            raise ValueError("accum must be a C-contiguous float32 (W, H, 3) array")
        _check(self._lib.mipt_render_accumulate(self._h, C.byref(cam), C.byref(light), C.byref(params),
                                                sample_begin, sample_count, _ptr(accum)))
        return accum

    def resolve_argb(self, accum, n_samples):
        W, H = accum.shape[0], accum.shape[1]
        out = np.zeros((H, W), dtype=np.uint32)
        _check(self._lib.mipt_resolve_argb(self._h, _ptr(accum), W, H, n_samples, _ptr(out)))
        return out

    def trace_paths(self, cam, light, params, px, py, sample):
        px = np.ascontiguousarray(px, dtype=np.int32)
        py = np.ascontiguousarray(py, dtype=np.int32)
        sample = np.ascontiguousarray(sample, dtype=np.uint32)
        n = len(px)
        rgb = np.zeros((n, 3), dtype=np.float32)
        counts = np.zeros((n, 2), dtype=np.uint32)
        _check(self._lib.mipt_trace_paths(self._h, C.byref(cam), C.byref(light), C.byref(params),
                                          _ptr(px), _ptr(py), _ptr(sample), n, _ptr(rgb), _ptr(counts)))
        return rgb, counts

    def selfcheck(self, n_rays, seed=0):
        """Device-side comparison of the fast arithmetic with the literal reference arithmetic:
        returns (reciprocal mismatches over all 2^32 floats, worst hardware-reciprocal error in ulp,
        closest-hit mismatches, any-hit mismatches) for n_rays generated rays."""
        out = (C.c_uint64 * 8)()
        _check(self._lib.mipt_debug_selfcheck(self._h, n_rays, seed, C.byref(out)))
        self.last_selfcheck_bvh = (int(out[4]), int(out[5]))  # (exhaustive re-runs, overflows)
        self.last_selfcheck_sincos = int(out[6])  # floats in [0, 2 pi] with a fused/plain sincos mismatch
        self.last_selfcheck_grazing = int(out[7])  # tree scenes: plane-grazing rays that differ from the linear scan
        return int(out[0]), out[1] / 1024.0, int(out[2]), int(out[3])

    def bvh_info(self):
        """(has an LBVH, node count, build milliseconds)"""
        n, ms = C.c_uint64(), C.c_double()
        has = self._lib.mipt_scene_has_bvh(self._h, C.byref(n), C.byref(ms))
        return bool(has), int(n.value), float(ms.value)

    def tree_info(self):
        """(the tree came from the clustering rather than the radix tree, the clustered tree's depth,
        its surface-area cost over the radix tree's); None without a tree"""
        c, d, r = C.c_int(), C.c_int(), C.c_float()
        if not self._lib.mipt_scene_tree_info(self._h, C.byref(c), C.byref(d), C.byref(r)):
            return None
        return bool(c.value), int(d.value), float(r.value)

    def pair_units(self):
        """pairs of coplanar triangles the linear scan's candidate filter merged"""
        return int(self._lib.mipt_scene_pair_units(self._h))

    def axis_pair_units(self):
        """Pairs among pair_units() whose plane is normal to a coordinate axis."""
        return int(self._lib.mipt_scene_axis_pair_units(self._h))

    def box_units(self):
        """Groups of pair_units() that are faces of one parallelepiped and are filtered in its frame."""
        return int(self._lib.mipt_scene_box_units(self._h))

    def stats(self):
        s = abi.MiptStats()
        _check(self._lib.mipt_get_stats(self._h, C.byref(s)))
        return s

    # ---- device-pointer entry points (torch CUDA tensors, current stream) -------------------
    @staticmethod
    def _stream():
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def closest_hit_dev(self, origin4, dir4, tri_idx, t, position4=None):
        n = origin4.shape[0]
        _check(self._lib.mipt_closest_hit_dev(self._h, origin4.data_ptr(), dir4.data_ptr(), n,
                                              tri_idx.data_ptr(), t.data_ptr(),
                                              position4.data_ptr() if position4 is not None else None,
                                              self._stream()))

    def any_hit_dev(self, origin4, dir4, max_t, occluded):
        n = origin4.shape[0]
        _check(self._lib.mipt_any_hit_dev(self._h, origin4.data_ptr(), dir4.data_ptr(),
                                          max_t.data_ptr(), n, occluded.data_ptr(), self._stream()))

    def render_accumulate_dev(self, cam, light, params, sample_begin, sample_count, accum):
        """accum: float32 CUDA tensor of W*H*3 elements on the scene's device; samples are ADDED."""
        if accum.numel() != params.width * params.height * 3 or not accum.is_contiguous():
            raise ValueError("accum must be a contiguous float32 tensor of W*H*3 elements")
        _check(self._lib.mipt_render_accumulate_dev(self._h, C.byref(cam), C.byref(light),
                                                    C.byref(params), sample_begin, sample_count,
                                                    accum.data_ptr(), self._stream()))

    def render_items_dev(self, cam, light, params, sample_begin, sample_count):
        """First half of render_accumulate_dev: the work items' sums stay in the scene's scratch."""
        _check(self._lib.mipt_render_items_dev(self._h, C.byref(cam), C.byref(light), C.byref(params),
                                               sample_begin, sample_count, self._stream()))

    def fold_items_dev(self, accum):
        """Second half: the pending item sums join accum's running values in sample order."""
        if not accum.is_contiguous():
            raise ValueError("accum must be a contiguous float32 tensor of W*H*3 elements")
        _check(self._lib.mipt_fold_items_dev(self._h, accum.data_ptr(), self._stream()))

    def render_accumulate_noalloc_dev(self, cam, light, params, sample_begin, sample_count, accum):
        """render_accumulate_dev through the entry point that allocates nothing (fixed scratch in the
        library, at most abi.NOALLOC_MAX_BYTES of item sums per call)."""
        if accum.numel() != params.width * params.height * 3 or not accum.is_contiguous():
            raise ValueError("accum must be a contiguous float32 tensor of W*H*3 elements")
        _check(self._lib.mipt_render_accumulate_noalloc_dev(self._h, C.byref(cam), C.byref(light),
                                                            C.byref(params), sample_begin, sample_count,
                                                            accum.data_ptr(), self._stream()))

    def resolve_argb_dev(self, accum, width, height, n_samples, screen):
        _check(self._lib.mipt_resolve_argb_dev(self._h, accum.data_ptr(), width, height, n_samples,
                                               screen.data_ptr(), self._stream()))


class Frame:
    """`pixels` + `samplesSeenSoFar` of the reference's interactive loop (main.cpp:50-56, 98-103),
    resident on the scene's device: add_samples = Draw's pixel loop, present = its tonemap loop,
    reset = reset_evolutionModel (utils.cpp:24-32)."""

    def __init__(self, scene, width, height):
        self._lib = abi.load_library()
        self.scene, self.width, self.height = scene, width, height
        h = C.c_void_p()
        _check(self._lib.mipt_frame_create(scene._h, width, height, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mipt_frame_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def samples(self):
        return int(self._lib.mipt_frame_samples(self._h))

    def poll(self):
        """True when every sample added so far has been rendered (non-blocking)."""
        rc = self._lib.mipt_frame_poll(self._h)
        if rc < 0:
            _check(rc)
        return rc == 1

    def reset(self):
        _check(self._lib.mipt_frame_reset(self._h))

    def add_samples(self, cam, light, params, n):
        _check(self._lib.mipt_frame_add_samples(self._h, C.byref(cam), C.byref(light), C.byref(params), n))

    def present(self):
        screen = np.zeros((self.height, self.width), dtype=np.uint32)
        _check(self._lib.mipt_frame_present(self._h, _ptr(screen)))
        return screen

    def read_pixels(self):
        acc = np.zeros((self.width, self.height, 3), dtype=np.float32)
        _check(self._lib.mipt_frame_read_pixels(self._h, _ptr(acc)))
        return acc


class MultiScene:
    """The scene replicated on several GPUs of one node, driven from this one process: render_accumulate
    splits the sample range over the devices and sums the per-device images with one RCCL reduce
    (mipt_multi_*; replaces the OpenMP loop of main.cpp:121)."""

    def __init__(self, tris, spheres=None, devices=(0,), allow_shared_device=False):
        """allow_shared_device: a device may be listed more than once (rehearsal of the several-device
        code on one GPU; relay calls only, RCCL takes one rank per device)."""
        self._lib = abi.load_library()
        tris = np.ascontiguousarray(tris, dtype=abi.TRIANGLE_DTYPE)
        ns = 0 if spheres is None else len(spheres)
        if ns:
            spheres = np.ascontiguousarray(spheres, dtype=abi.SPHERE_DTYPE)
        devs = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        _check(self._lib.mipt_multi_create_opts(_ptr(tris) if len(tris) else None, len(tris),
                                                _ptr(spheres) if ns else None, ns, devs, len(devices),
                                                abi.MULTI_ALLOW_SHARED_DEVICE if allow_shared_device else 0, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mipt_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_count(self):
        return int(self._lib.mipt_multi_device_count(self._h))

    def render_accumulate(self, cam, light, params, sample_begin, sample_count, accum=None):
        W, H = params.width, params.height
        if accum is None:
            accum = np.zeros((W, H, 3), dtype=np.float32)
        if accum.dtype != np.float32 or accum.shape != (W, H, 3) or not accum.flags.c_contiguous:
            raise ValueError("accum must be a C-contiguous float32 (W, H, 3) array")
        _check(self._lib.mipt_multi_render_accumulate(self._h, C.byref(cam), C.byref(light), C.byref(params),
                                                      sample_begin, sample_count, _ptr(accum)))
        return accum

    def render_accumulate_chained(self, cam, light, params, sample_begin, sample_count, accum=None):
        """The one-GPU image bit for bit: whole work items per device, the image handed from device to device."""
        W, H = params.width, params.height
        if accum is None:
            accum = np.zeros((W, H, 3), dtype=np.float32)
        if accum.dtype != np.float32 or accum.shape != (W, H, 3) or not accum.flags.c_contiguous:
            raise ValueError("accum must be a C-contiguous float32 (W, H, 3) array")
        _check(self._lib.mipt_multi_render_accumulate_chained(self._h, C.byref(cam), C.byref(light), C.byref(params),
                                                              sample_begin, sample_count, _ptr(accum)))
        return accum


class MultiFrame:
    """Frame on the devices of a MultiScene (mipt_multi_frame_*): exact=False sums the devices' images
    with one RCCL reduce per add_samples, exact=True relays the image from device to device (the
    one-GPU frame bit for bit). Close it before its MultiScene."""

    def __init__(self, multi, width, height, exact=False):
        self._lib = abi.load_library()
        self.multi, self.width, self.height = multi, width, height
        h = C.c_void_p()
        _check(self._lib.mipt_multi_frame_create(multi._h, width, height,
                                                 abi.MULTI_FRAME_EXACT if exact else abi.MULTI_FRAME_REDUCE, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mipt_multi_frame_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def samples(self):
        return int(self._lib.mipt_multi_frame_samples(self._h))

    @property
    def device_index(self):
        """index into the MultiScene's device list of the device that holds the image"""
        return int(self._lib.mipt_multi_frame_device(self._h))

    def poll(self):
        rc = self._lib.mipt_multi_frame_poll(self._h)
        if rc < 0:
            _check(rc)
        return rc == 1

    def reset(self):
        _check(self._lib.mipt_multi_frame_reset(self._h))

    def add_samples(self, cam, light, params, n):
        _check(self._lib.mipt_multi_frame_add_samples(self._h, C.byref(cam), C.byref(light), C.byref(params), n))

    def present(self):
        screen = np.zeros((self.height, self.width), dtype=np.uint32)
        _check(self._lib.mipt_multi_frame_present(self._h, _ptr(screen)))
        return screen

    def read_pixels(self):
        acc = np.zeros((self.width, self.height, 3), dtype=np.float32)
        _check(self._lib.mipt_multi_frame_read_pixels(self._h, _ptr(acc)))
        return acc


def multi_sample_range(begin, count, rank, n):
    b, c = C.c_uint32(), C.c_uint32()
    abi.load_library().mipt_multi_sample_range(begin, count, rank, n, C.byref(b), C.byref(c))
    return int(b.value), int(c.value)


def multi_item_range(begin, count, rank, n):
    b, c = C.c_uint32(), C.c_uint32()
    abi.load_library().mipt_multi_item_range(begin, count, rank, n, C.byref(b), C.byref(c))
    return int(b.value), int(c.value)
