// mipt_multi.cpp -- one host process driving several GPUs: the sample split + sum-reduce that
// replaces the reference's only parallel construct, the OpenMP loop over image columns
// (CW/raytracer/Source/main.cpp:121), behind one render call.
//
// Every listed device holds a replica of the scene (a scene is at most ~100 MB). A render call
// splits the sample range [begin, begin + count) into contiguous parts, one per device -- a path's
// random stream depends on (seed, pixel, global sample index) only, so the parts together are exactly
// the paths one GPU would trace -- launches every part on its device's own stream, and sums the
// per-device images on the first device with ONE collective: ncclReduce(sum, float32, 3 W H) over
// RCCL (xGMI between the GPUs of a node), inside a group call since all ranks live in this thread.
// The first device renders straight into the image the reduce lands in (in-place at the root), so
// nothing is added on the host. librccl.so is loaded on first use (dlopen), so the library has no
// load-time dependency on it.
//
// The relay ("chained") variant needs no collective: the devices trace whole work items at once and
// the image then travels from device to device by peer copies, each folding its items in -- the
// additions happen in the order ONE device performs them, so the bits do not depend on the number of
// devices.
//
// MiptMultiFrame keeps the progressive image (`pixels` + `samplesSeenSoFar`, main.cpp:50-60) on the
// devices: adding samples is asynchronous, and presenting moves only the ARGB screen to the host.
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/mipt.h"

extern "C" int mipt_internal_fail(int code, const char* msg);
extern "C" void mipt_internal_discard_items(MiptScene* s);

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;

int fail(int code, const char* fmt, const char* a = "", const char* b = "") {
  char msg[400];
  snprintf(msg, sizeof msg, fmt, a, b);
  return mipt_internal_fail(code, msg);
}

int load_rccl() {
  if (g_rccl.lib) return MIPT_OK;
  void* lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!lib) return fail(MIPT_E_RCCL, "mipt_multi: cannot load librccl.so: %s", dlerror());
  Rccl r;
  r.lib = lib;
  r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
  r.Reduce = reinterpret_cast<decltype(r.Reduce)>(dlsym(lib, "ncclReduce"));
  r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(lib, "ncclGroupStart"));
  r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
  if (!r.CommInitAll || !r.CommDestroy || !r.Reduce || !r.GroupStart || !r.GroupEnd || !r.GetErrorString) {
    dlclose(lib);
    return fail(MIPT_E_RCCL, "mipt_multi: librccl.so lacks a collective entry point");
  }
  g_rccl = r;
  return MIPT_OK;
}

#define HIP_TRY(call)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      (void)hipGetLastError();                                                               \
      return fail(e_ == hipErrorOutOfMemory ? MIPT_E_NOMEM : MIPT_E_HIP, "%s failed: %s", #call, \
                  hipGetErrorString(e_));                                                    \
    }                                                                                        \
  } while (0)
#define RCCL_TRY(call)                                                                       \
  do {                                                                                       \
    ncclResult_t r_ = (call);                                                                \
    if (r_ != ncclSuccess) return fail(MIPT_E_RCCL, "%s failed: %s", #call, g_rccl.GetErrorString(r_)); \
  } while (0)

struct Part {
  int device = 0;
  MiptScene* scene = nullptr;
  hipStream_t stream = nullptr;
  float* d_image = nullptr;  // this device's samples of the current call (the first device: the image the reduce lands in)
  ncclComm_t comm = nullptr;
  hipEvent_t folded = nullptr;  // relay: this device's items are in the image it holds
};

}  // namespace

struct MiptMulti {
  std::vector<Part> parts;
  float* h_total = nullptr;  // pinned staging buffer of the host-pointer calls
  size_t image_floats = 0;
  bool shared = false;  // some device is listed more than once (rehearsal): no communicator, relay only
};

// rank r of n gets a contiguous share; the first count % n ranks one sample more
extern "C" void mipt_multi_sample_range(uint32_t begin, uint32_t count, int rank, int n, uint32_t* part_begin,
                                        uint32_t* part_count) {
  const uint32_t base = count / (uint32_t)n, extra = count % (uint32_t)n;
  const uint32_t r = (uint32_t)rank;
  *part_begin = begin + r * base + (r < extra ? r : extra);
  *part_count = base + (r < extra ? 1u : 0u);
}

// the same split in whole work items (16 samples): every share a multiple of 16 but the last one in use
extern "C" void mipt_multi_item_range(uint32_t begin, uint32_t count, int rank, int n, uint32_t* part_begin,
                                      uint32_t* part_count) {
  const uint32_t item = 16u;
  const uint32_t n_items = count / item + (count % item ? 1u : 0u);
  uint32_t first = 0, mine = 0;
  mipt_multi_sample_range(0, n_items, rank, n, &first, &mine);
  const uint64_t lo = (uint64_t)first * item, hi = (uint64_t)(first + mine) * item;
  const uint32_t lo_c = lo < count ? (uint32_t)lo : count, hi_c = hi < count ? (uint32_t)hi : count;
  *part_begin = begin + lo_c;
  *part_count = hi_c - lo_c;
}

extern "C" void mipt_multi_destroy(MiptMulti* m) {
  if (!m) return;
  for (Part& p : m->parts) {
    if (!p.scene) continue;  // creation stopped before this device
    (void)hipSetDevice(p.device);
    if (p.stream) (void)hipStreamSynchronize(p.stream);
    if (p.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(p.comm);
    if (p.d_image) (void)hipFree(p.d_image);
    if (p.folded) (void)hipEventDestroy(p.folded);
    if (p.stream) (void)hipStreamDestroy(p.stream);
    if (p.scene) mipt_scene_destroy(p.scene);
  }
  if (m->h_total) (void)hipHostFree(m->h_total);
  (void)hipGetLastError();  // a failure while tearing down is not the next call's error
  delete m;
}

static int multi_create_impl(const MiptTriangle* tris, size_t n_tris, const MiptSphere* spheres, size_t n_spheres,
                             const int* devices, int n_devices, MiptMulti* m) {
  if (!m->shared) {
    const int rc = load_rccl();
    if (rc != MIPT_OK) return rc;
  }
  m->parts.resize((size_t)n_devices);
  for (int r = 0; r < n_devices; ++r) {
    Part& p = m->parts[(size_t)r];
    p.device = devices[r];
    const int rc = mipt_scene_create(tris, n_tris, spheres, n_spheres, p.device, &p.scene);
    if (rc != MIPT_OK) return rc;
    HIP_TRY(hipSetDevice(p.device));
    HIP_TRY(hipStreamCreate(&p.stream));
    HIP_TRY(hipEventCreateWithFlags(&p.folded, hipEventDisableTiming));
  }
  // the relay's copies go straight over xGMI where the devices can reach each other's memory (they
  // are staged through the host otherwise: slower, still correct)
  for (int a = 0; a < n_devices; ++a)
    for (int b = 0; b < n_devices; ++b) {
      if (devices[a] == devices[b]) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, devices[a], devices[b]) == hipSuccess && can) {
        (void)hipSetDevice(devices[a]);
        (void)hipDeviceEnablePeerAccess(devices[b], 0);  // "already enabled" is fine
      }
      (void)hipGetLastError();
    }
  if (m->shared) return MIPT_OK;
  std::vector<ncclComm_t> comms((size_t)n_devices);
  RCCL_TRY(g_rccl.CommInitAll(comms.data(), n_devices, devices));
  for (int r = 0; r < n_devices; ++r) m->parts[(size_t)r].comm = comms[(size_t)r];
  return MIPT_OK;
}

extern "C" int mipt_multi_create_opts(const MiptTriangle* tris, size_t n_tris, const MiptSphere* spheres,
                                      size_t n_spheres, const int* devices, int n_devices, uint32_t options,
                                      MiptMulti** out) {
  if (!out) return fail(MIPT_E_INVALID, "mipt_multi_create: out is null");
  *out = nullptr;
  if (options & ~(uint32_t)MIPT_MULTI_ALLOW_SHARED_DEVICE) return fail(MIPT_E_INVALID, "mipt_multi_create_opts: unknown option bits");
  if (!devices || n_devices < 1 || n_devices > 64) return fail(MIPT_E_INVALID, "mipt_multi_create: need 1..64 devices");
  bool shared = false;
  for (int a = 0; a < n_devices; ++a)
    for (int b = a + 1; b < n_devices; ++b)
      if (devices[a] == devices[b]) shared = true;
  if (shared && !(options & MIPT_MULTI_ALLOW_SHARED_DEVICE))
    return fail(MIPT_E_INVALID, "mipt_multi_create: a device is listed twice");
  MiptMulti* m = new (std::nothrow) MiptMulti();
  if (!m) return fail(MIPT_E_NOMEM, "mipt_multi_create: host allocation failed");
  m->shared = shared;
  int rc;
  try {
    rc = multi_create_impl(tris, n_tris, spheres, n_spheres, devices, n_devices, m);
  } catch (const std::bad_alloc&) {
    rc = fail(MIPT_E_NOMEM, "mipt_multi_create: host allocation failed");
  }
  if (rc != MIPT_OK) { mipt_multi_destroy(m); return rc; }
  *out = m;
  return MIPT_OK;
}

extern "C" int mipt_multi_create(const MiptTriangle* tris, size_t n_tris, const MiptSphere* spheres, size_t n_spheres,
                                 const int* devices, int n_devices, MiptMulti** out) {
  return mipt_multi_create_opts(tris, n_tris, spheres, n_spheres, devices, n_devices, 0u, out);
}

extern "C" int mipt_multi_device_count(const MiptMulti* m) { return m ? (int)m->parts.size() : 0; }
extern "C" MiptScene* mipt_multi_scene(MiptMulti* m, int rank) {
  return (m && rank >= 0 && (size_t)rank < m->parts.size()) ? m->parts[(size_t)rank].scene : nullptr;
}

static int ensure_images(MiptMulti* m, size_t n) {
  if (n == m->image_floats) return MIPT_OK;
  // no size is valid until every buffer below exists: an allocation that fails part-way must not
  // leave the old size standing over freed or differently sized buffers
  m->image_floats = 0;
  for (Part& p : m->parts) {
    HIP_TRY(hipSetDevice(p.device));
    HIP_TRY(hipStreamSynchronize(p.stream));
    if (p.d_image) { (void)hipFree(p.d_image); p.d_image = nullptr; }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p.d_image), n * sizeof(float)));
  }
  if (m->h_total) { (void)hipHostFree(m->h_total); m->h_total = nullptr; }
  HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&m->h_total), n * sizeof(float), hipHostMallocDefault));
  m->image_floats = n;
  return MIPT_OK;
}

// After a failed step: every stream drained, every replica ready for the next render call.
static void settle(MiptMulti* m) {
  for (Part& p : m->parts) {
    (void)hipSetDevice(p.device);
    (void)hipStreamSynchronize(p.stream);
    mipt_internal_discard_items(p.scene);
  }
  (void)hipGetLastError();
}

// Samples [begin, begin + count) of every pixel, split over the devices, summed into `root_image` on the
// first device (which renders straight into it; the others into images[r]): one grouped ncclReduce,
// in place at the root. Asynchronous.
static int reduce_render(MiptMulti* m, const MiptCamera* cam, const MiptLight* light, const MiptParams* params,
                         uint32_t begin, uint32_t count, float* root_image, const std::vector<float*>& images) {
  const size_t n = (size_t)params->width * params->height * 3;
  const int world = (int)m->parts.size();
  for (int r = 0; r < world; ++r) {  // every device renders its share, asynchronously
    Part& p = m->parts[(size_t)r];
    uint32_t b = 0, c = 0;
    mipt_multi_sample_range(begin, count, r, world, &b, &c);
    HIP_TRY(hipSetDevice(p.device));
    float* image = r == 0 ? root_image : images[(size_t)r];
    if (r != 0) HIP_TRY(hipMemsetAsync(image, 0, n * sizeof(float), p.stream));
    if (c) {
      const int rc = mipt_render_accumulate_dev(p.scene, cam, light, params, b, c, image, p.stream);
      if (rc != MIPT_OK) return rc;
    }
  }
  // the one exchange step: sum of the per-device images on the first device (a communicator of one
  // rank goes through the same call)
  RCCL_TRY(g_rccl.GroupStart());
  for (int r = 0; r < world; ++r) {
    Part& p = m->parts[(size_t)r];
    float* image = r == 0 ? root_image : images[(size_t)r];
    const ncclResult_t res = g_rccl.Reduce(image, image, n, ncclFloat32, ncclSum, 0, p.comm, p.stream);
    if (res != ncclSuccess) {
      (void)g_rccl.GroupEnd();
      return fail(MIPT_E_RCCL, "ncclReduce failed: %s", g_rccl.GetErrorString(res));
    }
  }
  RCCL_TRY(g_rccl.GroupEnd());
  return MIPT_OK;
}

// The relay: the devices order[0 .. n_order) trace the whole work items of their shares at once; the
// image, which sits in images[order[0]], is folded there and then handed on by peer copies, each
// device folding its items in. Returns in *last the device that holds the image afterwards.
// Asynchronous.
static int relay_render(MiptMulti* m, const MiptCamera* cam, const MiptLight* light, const MiptParams* params,
                        uint32_t begin, uint32_t count, const std::vector<int>& order, const std::vector<float*>& images,
                        int* last) {
  const size_t n = (size_t)params->width * params->height * 3;
  const int world = (int)order.size();
  std::vector<uint32_t> counts((size_t)world);
  for (int k = 0; k < world; ++k) {  // all devices trace at once
    Part& p = m->parts[(size_t)order[(size_t)k]];
    uint32_t b = 0, c = 0;
    mipt_multi_item_range(begin, count, k, world, &b, &c);
    counts[(size_t)k] = c;
    if (!c) continue;
    const int rc = mipt_render_items_dev(p.scene, cam, light, params, b, c, p.stream);
    if (rc != MIPT_OK) return rc;
  }
  int holder = order[0];
  for (int k = 0; k < world; ++k) {  // the image's round: peer copy, fold, mark
    if (!counts[(size_t)k]) continue;  // shares are handed out from the front: nothing further down either
    const int r = order[(size_t)k];
    Part& p = m->parts[(size_t)r];
    HIP_TRY(hipSetDevice(p.device));
    if (r != holder) {
      Part& q = m->parts[(size_t)holder];
      HIP_TRY(hipStreamWaitEvent(p.stream, q.folded, 0));
      HIP_TRY(hipMemcpyPeerAsync(images[(size_t)r], p.device, images[(size_t)holder], q.device, n * sizeof(float), p.stream));
      holder = r;
    }
    const int rc = mipt_fold_items_dev(p.scene, images[(size_t)r], p.stream);
    if (rc != MIPT_OK) return rc;
    HIP_TRY(hipEventRecord(p.folded, p.stream));
  }
  *last = holder;
  return MIPT_OK;
}

static int check_call(const char* who, const MiptMulti* m, const MiptParams* params, const void* buffer) {
  if (!m || !params || !buffer) return fail(MIPT_E_INVALID, "%s: null argument", who);
  if (params->width <= 0 || params->height <= 0 || params->width > 32768 || params->height > 32768)
    return fail(MIPT_E_INVALID, "%s: image size out of range", who);
  return MIPT_OK;
}

static int sync_all(MiptMulti* m) {
  for (Part& p : m->parts) {
    HIP_TRY(hipSetDevice(p.device));
    HIP_TRY(hipStreamSynchronize(p.stream));
  }
  return MIPT_OK;
}

extern "C" int mipt_multi_render_accumulate(MiptMulti* m, const MiptCamera* cam, const MiptLight* light,
                                            const MiptParams* params, uint32_t sample_begin, uint32_t sample_count,
                                            float* accum_rgb) {
  int rc = check_call("mipt_multi_render_accumulate", m, params, accum_rgb);
  if (rc != MIPT_OK) return rc;
  if (m->shared)
    return fail(MIPT_E_UNSUPPORTED, "mipt_multi_render_accumulate: RCCL takes one rank per device; a scene with a shared "
                                    "device renders through mipt_multi_render_accumulate_chained");
  if (sample_count == 0) return MIPT_OK;
  const size_t n = (size_t)params->width * params->height * 3;
  rc = ensure_images(m, n);
  if (rc != MIPT_OK) return rc;
  // the caller's running sums go to the first device, which adds its own samples to them (main.cpp:137)
  // and then receives the other devices' in the reduce
  Part& root = m->parts[0];
  memcpy(m->h_total, accum_rgb, n * sizeof(float));
  HIP_TRY(hipSetDevice(root.device));
  HIP_TRY(hipMemcpyAsync(root.d_image, m->h_total, n * sizeof(float), hipMemcpyHostToDevice, root.stream));
  std::vector<float*> images;
  for (Part& p : m->parts) images.push_back(p.d_image);
  rc = reduce_render(m, cam, light, params, sample_begin, sample_count, root.d_image, images);
  if (rc != MIPT_OK) { settle(m); return rc; }
  HIP_TRY(hipSetDevice(root.device));
  HIP_TRY(hipMemcpyAsync(m->h_total, root.d_image, n * sizeof(float), hipMemcpyDeviceToHost, root.stream));
  rc = sync_all(m);
  if (rc != MIPT_OK) return rc;
  memcpy(accum_rgb, m->h_total, n * sizeof(float));
  return MIPT_OK;
}

// The image ONE device would produce, from all of them: every device traces the whole work items of
// its share at once; the caller's image goes to the first device and then from device to device
// (peer copies), each folding its items into it in sample order; the last one hands it back.
// No collective and no RCCL call: the order of the additions is the one-device order.
extern "C" int mipt_multi_render_accumulate_chained(MiptMulti* m, const MiptCamera* cam, const MiptLight* light,
                                                    const MiptParams* params, uint32_t sample_begin,
                                                    uint32_t sample_count, float* accum_rgb) {
  int rc = check_call("mipt_multi_render_accumulate_chained", m, params, accum_rgb);
  if (rc != MIPT_OK) return rc;
  if (sample_count == 0) return MIPT_OK;
  const size_t n = (size_t)params->width * params->height * 3;
  rc = ensure_images(m, n);
  if (rc != MIPT_OK) return rc;
  const int world = (int)m->parts.size();
  Part& first = m->parts[0];
  memcpy(m->h_total, accum_rgb, n * sizeof(float));
  HIP_TRY(hipSetDevice(first.device));
  HIP_TRY(hipMemcpyAsync(first.d_image, m->h_total, n * sizeof(float), hipMemcpyHostToDevice, first.stream));
  std::vector<int> order;
  std::vector<float*> images;
  for (int r = 0; r < world; ++r) { order.push_back(r); images.push_back(m->parts[(size_t)r].d_image); }
  int holder = 0;
  rc = relay_render(m, cam, light, params, sample_begin, sample_count, order, images, &holder);
  if (rc != MIPT_OK) { settle(m); return rc; }
  Part& last = m->parts[(size_t)holder];
  HIP_TRY(hipSetDevice(last.device));
  HIP_TRY(hipMemcpyAsync(m->h_total, last.d_image, n * sizeof(float), hipMemcpyDeviceToHost, last.stream));
  rc = sync_all(m);
  if (rc != MIPT_OK) return rc;
  memcpy(accum_rgb, m->h_total, n * sizeof(float));
  return MIPT_OK;
}

// ---------------------------------------------------------------------------------------
// progressive frame on several devices

struct MiptMultiFrame {
  MiptMulti* multi = nullptr;
  int width = 0, height = 0;
  uint32_t mode = 0;
  uint32_t samples = 0;
  int home = 0;               // the device that holds the image (always 0 in reduce mode)
  bool poisoned = false;      // a call failed half-way: the image is undefined until the next reset
  std::vector<float*> images;    // one image per device: the frame's where it is `home`, staging elsewhere
  std::vector<uint32_t*> screens;  // ARGB screen on every device that can come to hold the image
};

extern "C" void mipt_multi_frame_destroy(MiptMultiFrame* f) {
  if (!f) return;
  MiptMulti* m = f->multi;
  for (size_t r = 0; r < m->parts.size(); ++r) {
    (void)hipSetDevice(m->parts[r].device);
    (void)hipStreamSynchronize(m->parts[r].stream);
    if (r < f->images.size() && f->images[r]) (void)hipFree(f->images[r]);
    if (r < f->screens.size() && f->screens[r]) (void)hipFree(f->screens[r]);
  }
  (void)hipGetLastError();
  delete f;
}

static int multi_frame_create_impl(MiptMultiFrame* f) {
  MiptMulti* m = f->multi;
  const size_t n = (size_t)f->width * f->height;
  f->images.assign(m->parts.size(), nullptr);
  f->screens.assign(m->parts.size(), nullptr);
  for (size_t r = 0; r < m->parts.size(); ++r) {
    HIP_TRY(hipSetDevice(m->parts[r].device));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&f->images[r]), n * 12));
    // the sum-reduce always lands on the first device; the relay leaves the image where its round ended
    if (r == 0 || f->mode == MIPT_MULTI_FRAME_EXACT) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&f->screens[r]), n * 4));
  }
  HIP_TRY(hipSetDevice(m->parts[0].device));
  HIP_TRY(hipMemsetAsync(f->images[0], 0, n * 12, m->parts[0].stream));
  return MIPT_OK;
}

extern "C" int mipt_multi_frame_create(MiptMulti* m, int width, int height, uint32_t mode, MiptMultiFrame** out) {
  if (!out) return fail(MIPT_E_INVALID, "mipt_multi_frame_create: out is null");
  *out = nullptr;
  if (!m) return fail(MIPT_E_INVALID, "mipt_multi_frame_create: multi is null");
  if (width <= 0 || height <= 0 || width > 32768 || height > 32768)
    return fail(MIPT_E_INVALID, "mipt_multi_frame_create: image size out of range");
  if (mode != MIPT_MULTI_FRAME_REDUCE && mode != MIPT_MULTI_FRAME_EXACT)
    return fail(MIPT_E_INVALID, "mipt_multi_frame_create: unknown mode");
  if (mode == MIPT_MULTI_FRAME_REDUCE && m->shared)
    return fail(MIPT_E_UNSUPPORTED, "mipt_multi_frame_create: RCCL takes one rank per device; use MIPT_MULTI_FRAME_EXACT "
                                    "on a scene with a shared device");
  MiptMultiFrame* f = new (std::nothrow) MiptMultiFrame();
  if (!f) return fail(MIPT_E_NOMEM, "mipt_multi_frame_create: host allocation failed");
  f->multi = m; f->width = width; f->height = height; f->mode = mode;
  int rc;
  try {
    rc = multi_frame_create_impl(f);
  } catch (const std::bad_alloc&) {
    rc = fail(MIPT_E_NOMEM, "mipt_multi_frame_create: host allocation failed");
  }
  if (rc != MIPT_OK) { mipt_multi_frame_destroy(f); return rc; }
  *out = f;
  return MIPT_OK;
}

extern "C" int mipt_multi_frame_reset(MiptMultiFrame* f) {
  if (!f) return fail(MIPT_E_INVALID, "mipt_multi_frame_reset: frame is null");
  Part& p = f->multi->parts[(size_t)f->home];
  HIP_TRY(hipSetDevice(p.device));
  // the last operation on the image ran on this device's stream
  HIP_TRY(hipMemsetAsync(f->images[(size_t)f->home], 0, (size_t)f->width * f->height * 12, p.stream));
  f->samples = 0;
  f->poisoned = false;
  return MIPT_OK;
}

extern "C" int mipt_multi_frame_add_samples(MiptMultiFrame* f, const MiptCamera* cam, const MiptLight* light,
                                            const MiptParams* params, uint32_t n_samples) {
  if (!f || !params) return fail(MIPT_E_INVALID, "mipt_multi_frame_add_samples: null argument");
  if (f->poisoned) return fail(MIPT_E_INVALID, "mipt_multi_frame_add_samples: an earlier call failed; reset the frame first");
  if (n_samples == 0) return MIPT_OK;
  if ((uint64_t)f->samples + n_samples > 0x7FFFFFFFull)
    return fail(MIPT_E_INVALID, "mipt_multi_frame_add_samples: sample count exceeds int range");
  MiptMulti* m = f->multi;
  MiptParams p = *params;
  p.width = f->width; p.height = f->height;
  const int world = (int)m->parts.size();
  int rc;
  try {
    if (f->mode == MIPT_MULTI_FRAME_REDUCE) {
      rc = reduce_render(m, cam, light, &p, f->samples, n_samples, f->images[0], f->images);
    } else {
      // the round starts where the image is and ends where it will be found next time
      std::vector<int> order;
      for (int k = 0; k < world; ++k) order.push_back((f->home + k) % world);
      int holder = f->home;
      rc = relay_render(m, cam, light, &p, f->samples, n_samples, order, f->images, &holder);
      if (rc == MIPT_OK) f->home = holder;
    }
  } catch (const std::bad_alloc&) {
    rc = fail(MIPT_E_NOMEM, "mipt_multi_frame_add_samples: host allocation failed");
  }
  if (rc != MIPT_OK) { settle(m); f->poisoned = true; return rc; }
  f->samples += n_samples;
  return MIPT_OK;
}

extern "C" int mipt_multi_frame_present(MiptMultiFrame* f, uint32_t* screen_buffer) {
  if (!f || !screen_buffer) return fail(MIPT_E_INVALID, "mipt_multi_frame_present: null argument");
  if (f->poisoned) return fail(MIPT_E_INVALID, "mipt_multi_frame_present: an earlier call failed; reset the frame first");
  Part& p = f->multi->parts[(size_t)f->home];
  uint32_t* d_screen = f->screens[(size_t)f->home];
  const int rc = mipt_resolve_argb_dev(p.scene, f->images[(size_t)f->home], f->width, f->height, f->samples, d_screen, p.stream);
  if (rc != MIPT_OK) return rc;
  HIP_TRY(hipMemcpyAsync(screen_buffer, d_screen, (size_t)f->width * f->height * 4, hipMemcpyDeviceToHost, p.stream));
  HIP_TRY(hipStreamSynchronize(p.stream));
  return MIPT_OK;
}

extern "C" int mipt_multi_frame_read_pixels(MiptMultiFrame* f, float* accum_rgb) {
  if (!f || !accum_rgb) return fail(MIPT_E_INVALID, "mipt_multi_frame_read_pixels: null argument");
  Part& p = f->multi->parts[(size_t)f->home];
  HIP_TRY(hipSetDevice(p.device));
  HIP_TRY(hipMemcpyAsync(accum_rgb, f->images[(size_t)f->home], (size_t)f->width * f->height * 12, hipMemcpyDeviceToHost, p.stream));
  HIP_TRY(hipStreamSynchronize(p.stream));
  return MIPT_OK;
}

extern "C" int mipt_multi_frame_samples(const MiptMultiFrame* f) { return f ? (int)f->samples : 0; }

extern "C" int mipt_multi_frame_device(const MiptMultiFrame* f) { return f ? f->home : -1; }

extern "C" int mipt_multi_frame_poll(MiptMultiFrame* f) {
  if (!f) return fail(MIPT_E_INVALID, "mipt_multi_frame_poll: frame is null");
  for (Part& p : f->multi->parts) {
    HIP_TRY(hipSetDevice(p.device));
    const hipError_t e = hipStreamQuery(p.stream);
    if (e == hipSuccess) continue;
    (void)hipGetLastError();
    if (e == hipErrorNotReady) return 0;
    return fail(MIPT_E_HIP, "hipStreamQuery failed: %s", hipGetErrorString(e));
  }
  return 1;
}
