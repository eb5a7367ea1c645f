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
// librccl.so is loaded on first use (dlopen), so the library has no load-time dependency on it.
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
  float* d_image = nullptr;  // this device's samples of the current call
  ncclComm_t comm = nullptr;
  hipEvent_t folded = nullptr;  // chained rendering: this device's items are in its image
};

}  // namespace

struct MiptMulti {
  std::vector<Part> parts;
  float* d_total = nullptr;  // on parts[0].device: the reduced image
  float* h_total = nullptr;  // pinned staging buffer
  size_t image_floats = 0;
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
  if (!m->parts.empty()) (void)hipSetDevice(m->parts[0].device);
  if (m->d_total) (void)hipFree(m->d_total);
  if (m->h_total) (void)hipHostFree(m->h_total);
  (void)hipGetLastError();  // a failure while tearing down is not the next call's error
  delete m;
}

static int multi_create_impl(const MiptTriangle* tris, size_t n_tris, const MiptSphere* spheres, size_t n_spheres,
                             const int* devices, int n_devices, MiptMulti* m) {
  int rc = load_rccl();
  if (rc != MIPT_OK) return rc;
  m->parts.resize((size_t)n_devices);
  for (int r = 0; r < n_devices; ++r) {
    Part& p = m->parts[(size_t)r];
    p.device = devices[r];
    rc = mipt_scene_create(tris, n_tris, spheres, n_spheres, p.device, &p.scene);
    if (rc != MIPT_OK) return rc;
    HIP_TRY(hipSetDevice(p.device));
    HIP_TRY(hipStreamCreate(&p.stream));
    HIP_TRY(hipEventCreateWithFlags(&p.folded, hipEventDisableTiming));
  }
  std::vector<ncclComm_t> comms((size_t)n_devices);
  RCCL_TRY(g_rccl.CommInitAll(comms.data(), n_devices, devices));
  for (int r = 0; r < n_devices; ++r) m->parts[(size_t)r].comm = comms[(size_t)r];
  return MIPT_OK;
}

extern "C" int mipt_multi_create(const MiptTriangle* tris, size_t n_tris, const MiptSphere* spheres, size_t n_spheres,
                                 const int* devices, int n_devices, MiptMulti** out) {
  if (!out) return fail(MIPT_E_INVALID, "mipt_multi_create: out is null");
  *out = nullptr;
  if (!devices || n_devices < 1 || n_devices > 64) return fail(MIPT_E_INVALID, "mipt_multi_create: need 1..64 devices");
  for (int a = 0; a < n_devices; ++a)
    for (int b = a + 1; b < n_devices; ++b)
      if (devices[a] == devices[b]) return fail(MIPT_E_INVALID, "mipt_multi_create: a device is listed twice");
  MiptMulti* m = new (std::nothrow) MiptMulti();
  if (!m) return fail(MIPT_E_NOMEM, "mipt_multi_create: host allocation failed");
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

extern "C" int mipt_multi_device_count(const MiptMulti* m) { return m ? (int)m->parts.size() : 0; }
extern "C" MiptScene* mipt_multi_scene(MiptMulti* m, int rank) {
  return (m && rank >= 0 && (size_t)rank < m->parts.size()) ? m->parts[(size_t)rank].scene : nullptr;
}

static int ensure_images(MiptMulti* m, size_t n) {
  if (n == m->image_floats) return MIPT_OK;
  for (Part& p : m->parts) {
    HIP_TRY(hipSetDevice(p.device));
    HIP_TRY(hipStreamSynchronize(p.stream));
    if (p.d_image) { (void)hipFree(p.d_image); p.d_image = nullptr; }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p.d_image), n * sizeof(float)));
  }
  HIP_TRY(hipSetDevice(m->parts[0].device));
  if (m->d_total) { (void)hipFree(m->d_total); m->d_total = nullptr; }
  if (m->h_total) { (void)hipHostFree(m->h_total); m->h_total = nullptr; }
  m->image_floats = 0;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m->d_total), n * sizeof(float)));
  HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&m->h_total), n * sizeof(float), hipHostMallocDefault));
  m->image_floats = n;
  return MIPT_OK;
}

extern "C" int mipt_multi_render_accumulate(MiptMulti* m, const MiptCamera* cam, const MiptLight* light,
                                            const MiptParams* params, uint32_t sample_begin, uint32_t sample_count,
                                            float* accum_rgb) {
  if (!m || !params || !accum_rgb) return fail(MIPT_E_INVALID, "mipt_multi_render_accumulate: null argument");
  if (params->width <= 0 || params->height <= 0 || params->width > 32768 || params->height > 32768)
    return fail(MIPT_E_INVALID, "mipt_multi_render_accumulate: image size out of range");
  if (sample_count == 0) return MIPT_OK;
  const size_t n = (size_t)params->width * params->height * 3;
  int rc = ensure_images(m, n);
  if (rc != MIPT_OK) return rc;
  const int world = (int)m->parts.size();
  for (int r = 0; r < world; ++r) {  // every device renders its share, asynchronously
    Part& p = m->parts[(size_t)r];
    uint32_t b = 0, c = 0;
    mipt_multi_sample_range(sample_begin, sample_count, r, world, &b, &c);
    HIP_TRY(hipSetDevice(p.device));
    HIP_TRY(hipMemsetAsync(p.d_image, 0, n * sizeof(float), p.stream));
    if (c) {
      rc = mipt_render_accumulate_dev(p.scene, cam, light, params, b, c, p.d_image, p.stream);
      if (rc != MIPT_OK) return rc;
    }
  }
  // the one exchange step: sum of the per-device images on the first device
  RCCL_TRY(g_rccl.GroupStart());
  for (int r = 0; r < world; ++r) {
    Part& p = m->parts[(size_t)r];
    const ncclResult_t res = g_rccl.Reduce(p.d_image, r == 0 ? m->d_total : p.d_image, n, ncclFloat32, ncclSum, 0,
                                           p.comm, p.stream);
    if (res != ncclSuccess) {
      (void)g_rccl.GroupEnd();
      return fail(MIPT_E_RCCL, "ncclReduce failed: %s", g_rccl.GetErrorString(res));
    }
  }
  RCCL_TRY(g_rccl.GroupEnd());
  Part& root = m->parts[0];
  HIP_TRY(hipSetDevice(root.device));
  HIP_TRY(hipMemcpyAsync(m->h_total, m->d_total, n * sizeof(float), hipMemcpyDeviceToHost, root.stream));
  for (int r = 0; r < world; ++r) {
    HIP_TRY(hipSetDevice(m->parts[(size_t)r].device));
    HIP_TRY(hipStreamSynchronize(m->parts[(size_t)r].stream));
  }
  for (size_t i = 0; i < n; ++i) accum_rgb[i] += m->h_total[i];  // main.cpp:137, for the whole range at once
  return MIPT_OK;
}

// The image ONE device would produce, from all of them: every device traces the whole work items of
// its share at once; the caller's image goes to the first device and then from device to device
// (peer copies), each folding its items into it in sample order; the last one hands it back.
// No collective and no RCCL call: the order of the additions is the one-device order.
extern "C" int mipt_multi_render_accumulate_chained(MiptMulti* m, const MiptCamera* cam, const MiptLight* light,
                                                    const MiptParams* params, uint32_t sample_begin,
                                                    uint32_t sample_count, float* accum_rgb) {
  if (!m || !params || !accum_rgb) return fail(MIPT_E_INVALID, "mipt_multi_render_accumulate_chained: null argument");
  if (params->width <= 0 || params->height <= 0 || params->width > 32768 || params->height > 32768)
    return fail(MIPT_E_INVALID, "mipt_multi_render_accumulate_chained: image size out of range");
  if (sample_count == 0) return MIPT_OK;
  const size_t n = (size_t)params->width * params->height * 3;
  int rc = ensure_images(m, n);
  if (rc != MIPT_OK) return rc;
  const int world = (int)m->parts.size();
  std::vector<uint32_t> counts((size_t)world);
  for (int r = 0; r < world; ++r) {  // all devices trace at once
    Part& p = m->parts[(size_t)r];
    uint32_t b = 0, c = 0;
    mipt_multi_item_range(sample_begin, sample_count, r, world, &b, &c);
    counts[(size_t)r] = c;
    if (!c) continue;
    rc = mipt_render_items_dev(p.scene, cam, light, params, b, c, p.stream);
    if (rc != MIPT_OK) return rc;
  }
  memcpy(m->h_total, accum_rgb, n * sizeof(float));
  for (int r = 0; r < world; ++r) {  // the image's round: upload / peer copy, fold, mark
    Part& p = m->parts[(size_t)r];
    HIP_TRY(hipSetDevice(p.device));
    if (r == 0) {
      HIP_TRY(hipMemcpyAsync(p.d_image, m->h_total, n * sizeof(float), hipMemcpyHostToDevice, p.stream));
    } else {
      Part& q = m->parts[(size_t)r - 1];
      HIP_TRY(hipStreamWaitEvent(p.stream, q.folded, 0));
      HIP_TRY(hipMemcpyPeerAsync(p.d_image, p.device, q.d_image, q.device, n * sizeof(float), p.stream));
    }
    if (counts[(size_t)r]) {
      rc = mipt_fold_items_dev(p.scene, p.d_image, p.stream);
      if (rc != MIPT_OK) return rc;
    }
    HIP_TRY(hipEventRecord(p.folded, p.stream));
  }
  Part& last = m->parts[(size_t)world - 1];
  HIP_TRY(hipSetDevice(last.device));
  HIP_TRY(hipMemcpyAsync(m->h_total, last.d_image, n * sizeof(float), hipMemcpyDeviceToHost, last.stream));
  for (int r = 0; r < world; ++r) {
    HIP_TRY(hipSetDevice(m->parts[(size_t)r].device));
    HIP_TRY(hipStreamSynchronize(m->parts[(size_t)r].stream));
  }
  memcpy(accum_rgb, m->h_total, n * sizeof(float));
  return MIPT_OK;
}
