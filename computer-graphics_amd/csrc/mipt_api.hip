// mipt_api.hip -- the C ABI of include/mipt.h over the kernels of mipt_kernels.hpp.
// Host side: validates shapes before any launch, owns the scene's device memory, never throws.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <iterator>
#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "mipt_kernels.hpp"

using namespace mipt;

// ---------------------------------------------------------------------------------------
// errors

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(call)                                                                           \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      (void)hipGetLastError(); /* reported here: must not surface again after a later launch */ \
      return fail(e_ == hipErrorOutOfMemory ? MIPT_E_NOMEM : MIPT_E_HIP, "%s failed: %s", #call, \
                  hipGetErrorString(e_));                                                       \
    }                                                                                           \
  } while (0)

extern "C" const char* mipt_last_error(void) { return g_err; }
// for the host-only translation units of the library
extern "C" int mipt_internal_fail(int code, const char* msg) { return fail(code, "%s", msg); }
extern "C" const char* mipt_version(void) { return "mipt 0.1 gfx950 hip"; }

// ---------------------------------------------------------------------------------------
// Scratch of the entry point that allocates nothing (mipt_render_accumulate_noalloc_dev): the path
// kernel's work-queue head (first 256 bytes) and its per-item radiance sums, in a file-scope device
// array. Every GPU the library runs on holds its own copy at its own address until the process
// ends. In host code the array's NAME is only a stand-in for the symbol -- never a device address:
// noalloc_setup() asks HIP for the address once per device (with that device current) and host
// code uses that pointer from then on.
constexpr size_t NOALLOC_QUEUE_BYTES = 256;
constexpr size_t NOALLOC_SUMS_BYTES = (size_t)MIPT_NOALLOC_MAX_BYTES;
static __device__ __align__(256) unsigned char scratch[NOALLOC_QUEUE_BYTES + NOALLOC_SUMS_BYTES];

constexpr int NOALLOC_MAX_DEVICES = 64;
static std::mutex g_noalloc_mutex;
static unsigned char* g_noalloc_ptr[NOALLOC_MAX_DEVICES] = {};

// Called with `device` current (mipt_scene_create): the address of this device's copy of `scratch`.
static int noalloc_setup(int device) {
  if (device < 0 || device >= NOALLOC_MAX_DEVICES) return MIPT_OK;  // such a device simply has no no-alloc path
  std::lock_guard<std::mutex> lock(g_noalloc_mutex);
  if (g_noalloc_ptr[device]) return MIPT_OK;
  void* ptr = nullptr;
  HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
  if (!ptr) return fail(MIPT_E_HIP, "hipGetSymbolAddress(scratch) returned a null address on device %d", device);
  g_noalloc_ptr[device] = static_cast<unsigned char*>(ptr);
  return MIPT_OK;
}

// ---------------------------------------------------------------------------------------
// scene

struct MiptScene {
  int device = 0;
  DeviceScene d{};
  TriRec* d_tris = nullptr;
  ScanUnit* d_units = nullptr;
  uint32_t n_pair_units = 0;
  uint32_t n_axis_units = 0;
  uint32_t n_box_units = 0;
  ShadeRec* d_shade = nullptr;
  SphereRec* d_spheres = nullptr;
  unsigned long long* d_stats = nullptr;
  uint32_t* d_queue = nullptr;
  float* d_partial = nullptr;
  size_t partial_bytes = 0;
  // scenes whose path kernel keeps its vertex stacks in global memory (vstack_kind): the stacks of
  // VSTACK_BLOCKS workgroups at the deepest allowed path, allocated with the scene
  float4* d_vstack = nullptr;
  // staging memory of the entry points that take HOST pointers (device copies of their arguments):
  // kept and grown on demand, so that a loop over such calls does not allocate and free every time
  void* d_stage = nullptr;
  size_t stage_bytes = 0;
  BvhNode* d_bvh = nullptr;
  // ray binning of the large tree batches: (cell key, ray index) pairs, sorted copy, rocPRIM's
  // temporary storage, and the coherence probe's counter; grown on demand, kept with the scene
  uint32_t* d_sort = nullptr;  // 4 arrays of sort_cap entries: keys, keys sorted, indices, indices sorted
  float4* d_sorted_rays = nullptr;  // origins, directions (sort_cap each), then sort_cap limits: the batch in walking order
  void* d_sort_tmp = nullptr;
  unsigned int* d_breaks = nullptr;
  unsigned int* h_breaks = nullptr;  // pinned: the probe's count arrives here while the batch runs
  hipEvent_t ev_probe = nullptr;
  bool probe_pending = false;
  size_t probe_rays = 0;
  int scattered = -1;   // what the last finished probe said about this scene's batches: -1 nothing yet
  uint32_t calls_since_probe = 0;
  size_t sort_cap = 0, sort_tmp_bytes = 0;
  int* d_big = nullptr;
  int n_tree_tris = 0;
  int n_wide_nodes = 0;
  bool bvh_radix_tree = false;  // the binary tree came from the radix-tree fallback, not from the clustering
  int bvh_depth = 0, bvh_rounds = 0;  // of the clustering (diagnostics)
  float bvh_cost_ratio = 0.f;         // surface-area cost of the clustered tree / the radix tree's
  double bvh_build_ms = 0.0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // recorded after every launch that uses the scene's scratch memory (queue heads, item sums,
  // counters); the next such launch waits for it, so calls on different streams do not overlap
  hipEvent_t ev_busy = nullptr;
  bool busy_valid = false;
  // item sums left in d_partial by mipt_render_items_dev, waiting for mipt_fold_items_dev
  bool items_pending = false;
  uint32_t items_sblocks = 0;
  size_t items_n_img = 0;
  bool timed = false;
  bool stats_valid = false;
  uint64_t last_paths = 0;
};

static void scene_free(MiptScene* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->d_tris) (void)hipFree(s->d_tris);
  if (s->d_units) (void)hipFree(s->d_units);
  if (s->d_shade) (void)hipFree(s->d_shade);
  if (s->d_spheres) (void)hipFree(s->d_spheres);
  if (s->d_stats) (void)hipFree(s->d_stats);
  if (s->d_queue) (void)hipFree(s->d_queue);
  if (s->d_partial) (void)hipFree(s->d_partial);
  if (s->d_vstack) (void)hipFree(s->d_vstack);
  if (s->d_stage) (void)hipFree(s->d_stage);
  if (s->d_bvh) (void)hipFree(s->d_bvh);
  if (s->d_big) (void)hipFree(s->d_big);
  if (s->d_sort) (void)hipFree(s->d_sort);
  if (s->d_sorted_rays) (void)hipFree(s->d_sorted_rays);
  if (s->d_sort_tmp) (void)hipFree(s->d_sort_tmp);
  if (s->d_breaks) (void)hipFree(s->d_breaks);
  if (s->h_breaks) (void)hipHostFree(s->h_breaks);
  if (s->ev_probe) (void)hipEventDestroy(s->ev_probe);
  if (s->ev0) (void)hipEventDestroy(s->ev0);
  if (s->ev1) (void)hipEventDestroy(s->ev1);
  if (s->ev_busy) (void)hipEventDestroy(s->ev_busy);
  (void)hipGetLastError();  // a failure while tearing down is not the next call's error
  delete s;
}

extern "C" void mipt_scene_destroy(MiptScene* s) { scene_free(s); }

// A launch that uses the scene's scratch memory waits for the previous one, whichever stream that
// ran on, and leaves its own mark behind.
static int scratch_acquire(MiptScene* s, hipStream_t st) {
  if (s->busy_valid) HIP_TRY(hipStreamWaitEvent(st, s->ev_busy, 0));
  return MIPT_OK;
}
static int scratch_release(MiptScene* s, hipStream_t st) {
  HIP_TRY(hipEventRecord(s->ev_busy, st));
  s->busy_valid = true;
  return MIPT_OK;
}


// RAII device buffer for temporaries and the host-pointer entry points
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
  template <typename T> T* as() { return static_cast<T*>(p); }
};

static inline unsigned grid_for(size_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }

// The scene's staging memory, at least `bytes` large (the host-pointer entry points synchronise before
// they return, so the memory is free again whenever one of them starts), handed out in 256-byte steps.
static inline size_t stage_round(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
static int stage_reserve(MiptScene* s, size_t bytes) {
  if (bytes <= s->stage_bytes) return MIPT_OK;
  if (s->d_stage) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipFree(s->d_stage));
    s->d_stage = nullptr;
    s->stage_bytes = 0;
  }
  HIP_TRY(hipMalloc(&s->d_stage, bytes));
  s->stage_bytes = bytes;
  return MIPT_OK;
}
struct StageCursor {
  char* at;
  template <typename T> T* take(size_t bytes) {
    T* p = reinterpret_cast<T*>(at);
    at += stage_round(bytes);
    return p;
  }
};

// The binary tree by locally-ordered clustering over the sorted triangles (k_ploc_* in mipt_bvh.hpp).
// On success *root is the root's index in `bin`; *root = -1 (and MIPT_OK) when the tree came out deeper
// than PLOC_MAX_DEPTH, i.e. deeper than the walk's stack is sized for, or when the clustering was given up
// after PLOC_MAX_ROUNDS rounds (10^6 triangles take 58; a scene that needs more merges a pair or two a round
// and ends as a chain anyway): the caller builds the radix tree instead. *depth (0 if given up), *rounds:
// what the clustering did (diagnostics).
static int build_ploc(const Aabb* tri_box, const int* sorted_ids, int n, BvhBin* bin, int* root, int* depth, int* rounds) {
  *root = -1;
  DevBuf cbox0, cbox1, cid0, cid1, nn, keep, pos, counter, tmp;
  HIP_TRY(cbox0.alloc((size_t)n * sizeof(Aabb))); HIP_TRY(cbox1.alloc((size_t)n * sizeof(Aabb)));
  HIP_TRY(cid0.alloc((size_t)n * 8)); HIP_TRY(cid1.alloc((size_t)n * 8));
  HIP_TRY(nn.alloc((size_t)n * 4)); HIP_TRY(keep.alloc((size_t)n * 4)); HIP_TRY(pos.alloc((size_t)n * 4));
  HIP_TRY(counter.alloc(4));
  HIP_TRY(hipMemset(counter.p, 0, 4));
  size_t tmp_bytes = 0;
  HIP_TRY(rocprim::exclusive_scan(nullptr, tmp_bytes, keep.as<uint32_t>(), pos.as<uint32_t>(), 0u, (size_t)n,
                                  rocprim::plus<uint32_t>(), nullptr));
  HIP_TRY(tmp.alloc(tmp_bytes));
  Aabb* cb = cbox0.as<Aabb>();
  Aabb* cb2 = cbox1.as<Aabb>();
  int2* ci = cid0.as<int2>();
  int2* ci2 = cid1.as<int2>();
  hipLaunchKernelGGL(k_ploc_init, dim3(grid_for(n)), dim3(BLOCK), 0, nullptr, tri_box, sorted_ids, n, cb, ci);
  HIP_TRY(hipGetLastError());
  int m = n;
  int round = 0;
  for (; m > 1; ++round) {
    if (round >= PLOC_MAX_ROUNDS) { *depth = 0; *rounds = round; return MIPT_OK; }  // *root stays -1
    hipLaunchKernelGGL(k_ploc_nearest, dim3(grid_for(m)), dim3(BLOCK), 0, nullptr, cb, m, nn.as<int>());
    hipLaunchKernelGGL(k_ploc_merge, dim3(grid_for(m)), dim3(BLOCK), 0, nullptr, cb, ci, nn.as<int>(), m, bin,
                       counter.as<int>(), keep.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    size_t bytes = tmp_bytes;
    HIP_TRY(rocprim::exclusive_scan(tmp.p, bytes, keep.as<uint32_t>(), pos.as<uint32_t>(), 0u, (size_t)m,
                                    rocprim::plus<uint32_t>(), nullptr));
    hipLaunchKernelGGL(k_ploc_compact, dim3(grid_for(m)), dim3(BLOCK), 0, nullptr, cb, ci, keep.as<uint32_t>(),
                       pos.as<uint32_t>(), m, cb2, ci2);
    HIP_TRY(hipGetLastError());
    uint32_t last_pos = 0, last_keep = 0;
    HIP_TRY(hipMemcpy(&last_pos, pos.as<uint32_t>() + (m - 1), 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&last_keep, keep.as<uint32_t>() + (m - 1), 4, hipMemcpyDeviceToHost));
    const int next = (int)(last_pos + last_keep);
    if (next < 1 || next >= m) return fail(MIPT_E_HIP, "tree clustering stalled at %d clusters (round %d)", m, round);
    m = next;
    std::swap(cb, cb2);
    std::swap(ci, ci2);
  }
  int n_nodes = 0;
  int2 top = make_int2(-1, 0);
  HIP_TRY(hipMemcpy(&n_nodes, counter.p, 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(&top, ci, 8, hipMemcpyDeviceToHost));
  if (n_nodes != n - 1 || top.x < 0 || top.x >= n - 1)
    return fail(MIPT_E_HIP, "tree clustering ended with %d nodes of %d, root %d", n_nodes, n - 1, top.x);
  *depth = top.y;
  *rounds = round;
  if (top.y <= PLOC_MAX_DEPTH) *root = top.x;
  return MIPT_OK;
}

// The radix tree over the sorted keys (Karras 2012) with its boxes: topology in one launch, then refit
// waves from the leaves up. The root is node 0.
static int build_radix_tree(const unsigned long long* keys, const int* sorted_ids, const Aabb* tri_box, int n, BvhBin* bin) {
  DevBuf node_box, ready, done;
  HIP_TRY(node_box.alloc((size_t)n * sizeof(Aabb)));
  HIP_TRY(ready.alloc((size_t)n * 2));
  HIP_TRY(done.alloc(4));
  hipLaunchKernelGGL(k_bvh_topology, dim3(grid_for(n - 1)), dim3(BLOCK), 0, nullptr, keys, sorted_ids, n, bin);
  HIP_TRY(hipGetLastError());
  // refit: ready flags ping-pong between the two halves of `ready`; the count of finished nodes
  // is read back once per wave (a tree over sorted Morton keys is ~log2(n) + a few levels deep)
  HIP_TRY(hipMemset(ready.p, 0, (size_t)n * 2));
  HIP_TRY(hipMemset(done.p, 0, 4));
  unsigned char* r0 = ready.as<unsigned char>();
  unsigned char* r1 = r0 + n;
  int n_done = 0, prev = -1, waves = 0;
  while (n_done < n - 1) {
    if (n_done == prev || ++waves > 4096)
      return fail(MIPT_E_HIP, "LBVH refit stalled at %d of %d nodes after %d waves", n_done, n - 1, waves);
    prev = n_done;
    hipLaunchKernelGGL(k_bvh_refit, dim3(grid_for(n - 1)), dim3(BLOCK), 0, nullptr, bin, n, tri_box,
                       node_box.as<Aabb>(), r1, r0, done.as<int>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(&n_done, done.p, 4, hipMemcpyDeviceToHost));
    std::swap(r0, r1);
  }
  return MIPT_OK;
}

// Surface-area cost of a binary tree of n - 1 nodes (k_bin_cost)
static int tree_cost(const BvhBin* bin, int n, double* cost) {
  const int blocks = grid_for(n - 1);
  DevBuf partial;
  HIP_TRY(partial.alloc((size_t)blocks * 8));
  hipLaunchKernelGGL(k_bin_cost, dim3(blocks), dim3(BLOCK), 0, nullptr, bin, n - 1, partial.as<double>());
  HIP_TRY(hipGetLastError());
  std::vector<double> h(blocks);
  HIP_TRY(hipMemcpy(h.data(), partial.p, (size_t)blocks * 8, hipMemcpyDeviceToHost));
  double c = 0.0;
  for (double v : h) c += v;
  *cost = c;
  return MIPT_OK;
}

// Tree over the triangles listed in `subset` (n >= 2 of them): bounds -> Morton keys -> radix sort ->
// binary tree (locally-ordered clustering; the radix tree over the keys + refit waves as the fallback)
// -> collapse into four-wide nodes, level by level. Everything runs on the null stream of the scene's
// device; the tree is final on return.
static int build_bvh(MiptScene* s, const std::vector<int>& subset) {
  const int n = (int)subset.size();
  DevBuf box, cen, scene, keys, ids, keys2, ids2, tmp, sub, bin, bin_b, wide, front;
  HIP_TRY(sub.alloc((size_t)n * 4));
  HIP_TRY(hipMemcpy(sub.p, subset.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  HIP_TRY(box.alloc((size_t)s->d.n_tris * sizeof(Aabb)));
  HIP_TRY(cen.alloc((size_t)n * 12));
  HIP_TRY(scene.alloc(6 * 4));
  HIP_TRY(keys.alloc((size_t)n * 8)); HIP_TRY(keys2.alloc((size_t)n * 8));
  HIP_TRY(ids.alloc((size_t)n * 4)); HIP_TRY(ids2.alloc((size_t)n * 4));
  HIP_TRY(bin.alloc((size_t)(n - 1) * sizeof(BvhBin)));
  HIP_TRY(wide.alloc((size_t)(n - 1) * sizeof(BvhNode)));  // one wide node per binary node at most
  HIP_TRY(front.alloc((size_t)n * 2 * sizeof(int2)));
  struct Events {  // destroyed on every return path
    hipEvent_t a = nullptr, b = nullptr;
    ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
  } ev;
  HIP_TRY(hipEventCreate(&ev.a)); HIP_TRY(hipEventCreate(&ev.b));
  hipEvent_t e0 = ev.a, e1 = ev.b;
  HIP_TRY(hipEventRecord(e0, nullptr));
  const uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
  HIP_TRY(hipMemcpy(scene.p, init, sizeof init, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_bvh_bounds, dim3(grid_for(n)), dim3(BLOCK), 0, nullptr, s->d_tris, sub.as<int>(), n,
                     box.as<Aabb>(), cen.as<float>(), scene.as<uint32_t>());
  hipLaunchKernelGGL(k_bvh_morton, dim3(grid_for(n)), dim3(BLOCK), 0, nullptr, cen.as<float>(), n,
                     scene.as<uint32_t>(), sub.as<int>(), keys.as<unsigned long long>(), ids.as<int>());
  HIP_TRY(hipGetLastError());
  size_t tmp_bytes = 0;
  HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys.as<unsigned long long>(), keys2.as<unsigned long long>(),
                                    ids.as<int>(), ids2.as<int>(), (size_t)n, 0, 63, nullptr));
  HIP_TRY(tmp.alloc(tmp_bytes));
  HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, keys.as<unsigned long long>(), keys2.as<unsigned long long>(),
                                    ids.as<int>(), ids2.as<int>(), (size_t)n, 0, 63, nullptr));
  int bin_root = -1;
  bool radix = !MIPT_BVH_PLOC;
#ifdef MIPT_DEV_SWITCHES
  { const char* e = getenv("MIPT_RADIX_TREE"); if (e && e[0] == '1') radix = true; }  // developer build only
#endif
  if (!radix) {
    const int rc = build_ploc(box.as<Aabb>(), ids2.as<int>(), n, bin.as<BvhBin>(), &bin_root, &s->bvh_depth, &s->bvh_rounds);
    if (rc != MIPT_OK) return rc;
    radix = bin_root < 0;
    if (!radix) {
      // both trees; the clustered one is kept when its surface-area cost is clearly lower (PLOC_KEEP_RATIO):
      // clustering wins on meshes, the radix tree on evenly scattered small triangles, where splitting
      // space in the middle is already the best there is
      HIP_TRY(bin_b.alloc((size_t)(n - 1) * sizeof(BvhBin)));
      const int rr = build_radix_tree(keys2.as<unsigned long long>(), ids2.as<int>(), box.as<Aabb>(), n, bin_b.as<BvhBin>());
      if (rr != MIPT_OK) return rr;
      double c_ploc = 0.0, c_radix = 0.0;
      const int r1 = tree_cost(bin.as<BvhBin>(), n, &c_ploc), r2 = tree_cost(bin_b.as<BvhBin>(), n, &c_radix);
      if (r1 != MIPT_OK) return r1;
      if (r2 != MIPT_OK) return r2;
      s->bvh_cost_ratio = c_radix > 0.0 ? (float)(c_ploc / c_radix) : 0.f;
      if (!(c_ploc < PLOC_KEEP_RATIO * c_radix)) {
        std::swap(bin.p, bin_b.p);
        bin_root = 0;
        s->bvh_radix_tree = true;
      }
    }
#ifdef MIPT_DEV_SWITCHES
    fprintf(stderr, "[mipt tree] %d triangles: clustering %d rounds, depth %d, cost %.3f of the radix tree's%s\n", n, s->bvh_rounds,
            s->bvh_depth, s->bvh_cost_ratio, radix || s->bvh_radix_tree ? " -> radix tree" : "");
#endif
  }
  if (radix) {
    s->bvh_radix_tree = true;
    bin_root = 0;
    const int rc = build_radix_tree(keys2.as<unsigned long long>(), ids2.as<int>(), box.as<Aabb>(), n, bin.as<BvhBin>());
    if (rc != MIPT_OK) return rc;
  }
  // collapse: the frontier holds (binary node, wide slot) pairs of one level; `done` is reused as
  // {size of the next frontier, wide nodes allocated}
  DevBuf counters;
  HIP_TRY(counters.alloc(8));
  int2* f0 = front.as<int2>();
  int2* f1 = f0 + n;
  const int2 root_entry = make_int2(bin_root, 0);
  HIP_TRY(hipMemcpy(f0, &root_entry, sizeof root_entry, hipMemcpyHostToDevice));
  int n_front = 1, n_wide = 1, levels = 0;
  while (n_front > 0) {
    if (++levels > 128) return fail(MIPT_E_HIP, "LBVH collapse did not finish in %d levels", levels);
    const int c[2] = {0, n_wide};
    HIP_TRY(hipMemcpy(counters.p, c, 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_bvh_collapse, dim3(grid_for(n_front)), dim3(BLOCK), 0, nullptr, bin.as<BvhBin>(), f0, n_front,
                       f1, counters.as<int>(), wide.as<BvhNode>(), counters.as<int>() + 1);
    HIP_TRY(hipGetLastError());
    int back[2];
    HIP_TRY(hipMemcpy(back, counters.p, 8, hipMemcpyDeviceToHost));
    n_front = back[0]; n_wide = back[1];
    if (n_front > n - 1 || n_wide > n - 1) return fail(MIPT_E_HIP, "LBVH collapse overran: %d entries, %d nodes of %d", n_front, n_wide, n - 1);
    std::swap(f0, f1);
  }
  HIP_TRY(hipMalloc(&s->d_bvh, (size_t)n_wide * sizeof(BvhNode)));
  HIP_TRY(hipMemcpy(s->d_bvh, wide.p, (size_t)n_wide * sizeof(BvhNode), hipMemcpyDeviceToDevice));
  s->n_wide_nodes = n_wide;
  HIP_TRY(hipEventRecord(e1, nullptr));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  s->bvh_build_ms = ms;
  s->d.bvh = s->d_bvh;
  s->d.bvh_root = 0;
  return MIPT_OK;
}

// Tree scenes: which triangles stay out of the tree. A triangle whose extent exceeds 1/8 of the
// scene's (a wall of the room around a mesh) would put its box around every node above it;
// up to MAX_BIG such triangles are tested one by one before the walk instead.
constexpr size_t MAX_BIG = 64;
static void build_scan_units(const MiptTriangle* tris, const std::vector<TriRec>& tr, const std::vector<int>& ids,
                             std::vector<ScanUnit>& units, uint32_t& n_pair_units, uint32_t& n_axis_units,
                             uint32_t& n_box_units);
static int upload_scan_units(MiptScene* s, const std::vector<ScanUnit>& units);
static int build_tree_scene(MiptScene* s, const MiptTriangle* tris, const std::vector<TriRec>& tr, size_t n_tris) {
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  std::vector<float> extent(n_tris);
  // A triangle with a NaN, infinite or enormous coordinate (beyond 2^40, where the filters' error
  // bounds stop holding) must not shape the tree: its box would poison every node above it. Such
  // triangles are tested one by one like the big ones -- the filter passes them unexamined and the
  // reference's own arithmetic decides -- and the scene's extent is taken over the others.
  std::vector<char> odd(n_tris, 0);
  for (size_t i = 0; i < n_tris; ++i) {
    const float* v[3] = {tris[i].v0, tris[i].v1, tris[i].v2};
    for (int j = 0; j < 3; ++j)
      for (int k = 0; k < 3; ++k)
        if (!(std::fabs(v[j][k]) <= 1.0995116e12f)) odd[i] = 1;
    if (odd[i]) continue;
    float e = 0.f;
    for (int k = 0; k < 3; ++k) {
      const float a = std::fmin(v[0][k], std::fmin(v[1][k], v[2][k])), b = std::fmax(v[0][k], std::fmax(v[1][k], v[2][k]));
      if (a < lo[k]) lo[k] = a;
      if (b > hi[k]) hi[k] = b;
      e = std::fmax(e, b - a);
    }
    extent[i] = e;
  }
  const float scene_extent = std::fmax(hi[0] - lo[0], std::fmax(hi[1] - lo[1], hi[2] - lo[2]));
  std::vector<int> big, small, odd_ids;
  for (size_t i = 0; i < n_tris; ++i) {
    if (odd[i]) odd_ids.push_back((int)i);
    else (extent[i] > 0.125f * scene_extent ? big : small).push_back((int)i);
  }
  if (big.size() > MAX_BIG) {  // keep the largest, in index order; the rest go into the tree after all
    std::vector<int> by_size = big;
    std::sort(by_size.begin(), by_size.end(), [&](int a, int b) { return extent[a] > extent[b] || (extent[a] == extent[b] && a < b); });
    by_size.resize(MAX_BIG);
    std::sort(by_size.begin(), by_size.end());
    std::vector<int> rest;
    std::set_difference(big.begin(), big.end(), by_size.begin(), by_size.end(), std::back_inserter(rest));
    small.insert(small.end(), rest.begin(), rest.end());
    std::sort(small.begin(), small.end());
    big = by_size;
  }
  if (!odd_ids.empty()) {  // however many there are
    big.insert(big.end(), odd_ids.begin(), odd_ids.end());
    std::sort(big.begin(), big.end());
  }
#ifdef MIPT_DEV_SWITCHES
  const char* no_big = getenv("MIPT_NO_BIG_LIST");  // developer build only: every triangle in the tree
  if (no_big && no_big[0] == '1') { small.insert(small.end(), big.begin(), big.end()); std::sort(small.begin(), small.end()); big.clear(); }
#endif
  s->n_tree_tris = (int)small.size();
  if (!big.empty()) {
    HIP_TRY(hipMalloc(&s->d_big, big.size() * 4));
    HIP_TRY(hipMemcpy(s->d_big, big.data(), big.size() * 4, hipMemcpyHostToDevice));
  }
  s->d.big_ids = s->d_big;
  s->d.n_big = (int)big.size();
  if (!big.empty()) {  // the candidate filter's units over the big list (walls come in coplanar pairs)
    std::vector<ScanUnit> units;
    build_scan_units(tris, tr, big, units, s->n_pair_units, s->n_axis_units, s->n_box_units);
    const int rc = upload_scan_units(s, units);
    if (rc != MIPT_OK) return rc;
  }
  if (small.size() >= 2) return build_bvh(s, small);
  // zero or one triangle left: no inner node, the scene still counts as a tree scene
  HIP_TRY(hipMalloc(&s->d_bvh, sizeof(BvhNode)));
  HIP_TRY(hipMemset(s->d_bvh, 0, sizeof(BvhNode)));
  s->d.bvh = s->d_bvh;
  s->d.bvh_root = small.empty() ? BVH_DONE : -1 - small[0];
  return MIPT_OK;
}


// ---------------------------------------------------------------------------------------
// scan units of the linear scan's candidate filter (ScanUnit in mipt_device.hpp)

static inline bool same_vertex(const float* a, const float* b) { return memcmp(a, b, 12) == 0; }

// Can triangles A, B (consecutive indices) be filtered as one unit? They must share an edge
// bitwise, lie in one plane to 10^-6 of their size, and B's far vertex q must sit beyond the shared
// edge with (qu, qv) in a range that keeps the error amplification of unit_candidates <= 9.
static bool make_pair_unit(const MiptTriangle& A, const MiptTriangle& B, ScanUnit& u) {
  const float* av[3] = {A.v0, A.v1, A.v2};
  const float* bv[3] = {B.v0, B.v1, B.v2};
  int a_of_b[3] = {-1, -1, -1}, n_shared = 0;
  bool a_used[3] = {false, false, false};
  for (int j = 0; j < 3; ++j)
    for (int i = 0; i < 3; ++i)
      if (!a_used[i] && a_of_b[j] < 0 && same_vertex(av[i], bv[j])) { a_of_b[j] = i; a_used[i] = true; ++n_shared; }
  if (n_shared != 2) return false;
  int ir = 0, jq = 0;
  for (int i = 0; i < 3; ++i) if (!a_used[i]) ir = i;
  for (int j = 0; j < 3; ++j) if (a_of_b[j] < 0) jq = j;
  const float* r = av[ir];
  const float* s1 = av[(ir + 1) % 3];
  const float* s2 = av[(ir + 2) % 3];
  const float* q = bv[jq];
  volatile float f1[3], f2[3];
  double F1[3], F2[3], Q[3];
  float emax = 0.f;
  for (int k = 0; k < 3; ++k) {
    f1[k] = s1[k] - r[k]; f2[k] = s2[k] - r[k];
    F1[k] = f1[k]; F2[k] = f2[k]; Q[k] = (double)q[k] - (double)r[k];
    // every edge of either triangle: the frame's, the shared one, and the two to q
    const float e[5] = {f1[k], f2[k], s2[k] - s1[k], q[k] - s1[k], q[k] - s2[k]};
    for (float x : e) emax = std::fmax(emax, std::fabs(x));
  }
  float vmax = 0.f;
  for (int k = 0; k < 3; ++k) vmax = std::fmax(vmax, std::fmax(std::fabs(r[k]), std::fabs(q[k])));
  if (!(emax >= 9.094947e-13f && emax <= 1.0995116e12f && vmax <= 1.0995116e12f)) return false;
  const double N[3] = {F1[1] * F2[2] - F1[2] * F2[1], F1[2] * F2[0] - F1[0] * F2[2], F1[0] * F2[1] - F1[1] * F2[0]};
  const double nn = N[0] * N[0] + N[1] * N[1] + N[2] * N[2];
  // the frame must be well shaped: area at least 10^-3 of emax^2
  if (!(nn > 1e-6 * (double)emax * emax * (double)emax * emax)) return false;
  const double QxF2[3] = {Q[1] * F2[2] - Q[2] * F2[1], Q[2] * F2[0] - Q[0] * F2[2], Q[0] * F2[1] - Q[1] * F2[0]};
  const double F1xQ[3] = {F1[1] * Q[2] - F1[2] * Q[1], F1[2] * Q[0] - F1[0] * Q[2], F1[0] * Q[1] - F1[1] * Q[0]};
  const double qu = (QxF2[0] * N[0] + QxF2[1] * N[1] + QxF2[2] * N[2]) / nn;
  const double qv = (F1xQ[0] * N[0] + F1xQ[1] * N[1] + F1xQ[2] * N[2]) / nn;
  const double h = (Q[0] * N[0] + Q[1] * N[1] + Q[2] * N[2]) / std::sqrt(nn);  // distance of q from the plane
  if (!(std::fabs(h) <= 1e-6 * emax)) return false;
  const double area_ratio = qu + qv - 1.0;  // area(B) / area(A)
  if (!(area_ratio >= 0.5 && std::fabs(qu) <= 2.0 && std::fabs(qv) <= 2.0)) return false;
  memset(&u, 0, sizeof u);
  u.v0x = r[0]; u.v0y = r[1]; u.v0z = r[2];
  u.e1x = f1[0]; u.e1y = f1[1]; u.e1z = f1[2];
  u.e2x = f2[0]; u.e2y = f2[1]; u.e2z = f2[2];
  // tri_candidate's 2^-12 E1 E2 and 2^-17 max(E1, E2) with E = emax; x2 for the frame not being the
  // reference's labelling, x2 (kB) for S = 2 max(|s|, emax), x4 (kA) for the coplanarity tolerance,
  // / area_ratio where the second triangle's own determinant is the smaller one
  u.kA = (float)(0.000244140625 * 8.0 * FILTER_TIGHT * (double)emax * emax / std::fmin(1.0, area_ratio));
  u.kB = 7.62939453125e-06f * 4.0f * emax;
  u.emax = emax;
  u.qc = (float)(1.0 / area_ratio); u.qu = (float)qu; u.qv = (float)qv;
  // A plane normal to a coordinate axis (both frame edges and the far vertex have an exactly zero
  // component there) is stored in the axis form of mipt_device.hpp: no cross products in the filter.
#ifdef MIPT_DEV_SWITCHES
  const char* no_axis = getenv("MIPT_NO_AXIS_UNITS");  // developer build only
  if (no_axis && no_axis[0] == '1') return true;
#endif
  for (int k = 0; k < 3; ++k) {
    if (!(f1[k] == 0.0f && f2[k] == 0.0f && q[k] - r[k] == 0.0f)) continue;
    const int i = (k + 1) % 3, j = (k + 2) % 3;
    const double D = F1[i] * F2[j] - F1[j] * F2[i];
    const double up = 1.0 + 9.5367431640625e-07;  // the bound factors are rounded up
    const float kA = (float)((double)u.kA / std::fabs(D) * up), kB = (float)((double)u.kB / std::fabs(D) * up);
    const float kT = (float)((double)u.kA * 0.0625 / std::fabs(D) * up);
    if (!(std::isfinite(kA) && std::isfinite(kB) && std::isfinite(kT) && kA > 0.f && kT > 0.f)) break;
    u.e1x = (float)(F2[j] / D); u.e1y = (float)(-F2[i] / D);
    u.e1z = (float)(-F1[j] / D); u.e2x = (float)(F1[i] / D);
    u.e2y = kT;
    const uint32_t axis = (uint32_t)k;
    memcpy(&u.e2z, &axis, 4);
    u.v0x = r[k]; u.v0y = r[i]; u.v0z = r[j];  // permuted like the ray will be
    u.kA = kA; u.kB = kB;
    u.meta = 0x80000000u;  // marks the axis form for build_scan_units (which sets meta afterwards)
    break;
  }
  return true;
}

static void make_single_unit(const TriRec& t, ScanUnit& u) {
  memset(&u, 0, sizeof u);
  u.v0x = t.v0x; u.v0y = t.v0y; u.v0z = t.v0z;
  u.e1x = t.e1x; u.e1y = t.e1y; u.e1z = t.e1z;
  u.e2x = t.e2x; u.e2y = t.e2y; u.e2z = t.e2z;
  u.kA = t.kA; u.kB = t.kB;
}

// ---- box units (BoxUnit in mipt_device.hpp): coplanar pairs that are faces of one parallelepiped

// The four corners of the parallelogram two triangles with a shared edge form, in cyclic order
// r, s1, q, s2 (A = (r, s1, s2), B = (s1, s2, q)); false when they share no edge or q is not
// s1 + s2 - r to 10^-6 of the size.
struct QuadRef {
  const float* c[4];
  uint32_t bit_a, bit_b;  // chunk-relative mask bits of A and B
  bool axis_form;         // the pair on its own would take the axis form of the filter
  bool used;
};
static bool quad_of_pair(const MiptTriangle& A, const MiptTriangle& B, QuadRef& qd) {
  const float* av[3] = {A.v0, A.v1, A.v2};
  const float* bv[3] = {B.v0, B.v1, B.v2};
  int a_of_b[3] = {-1, -1, -1}, n_shared = 0;
  bool a_used[3] = {false, false, false};
  for (int j = 0; j < 3; ++j)
    for (int i = 0; i < 3; ++i)
      if (!a_used[i] && a_of_b[j] < 0 && same_vertex(av[i], bv[j])) { a_of_b[j] = i; a_used[i] = true; ++n_shared; }
  if (n_shared != 2) return false;
  int ir = 0, jq = 0;
  for (int i = 0; i < 3; ++i) if (!a_used[i]) ir = i;
  for (int j = 0; j < 3; ++j) if (a_of_b[j] < 0) jq = j;
  const float* r = av[ir];
  const float* s1 = av[(ir + 1) % 3];
  const float* s2 = av[(ir + 2) % 3];
  const float* q = bv[jq];
  double size = 0.0, dev = 0.0;
  for (int k = 0; k < 3; ++k) {
    size = std::fmax(size, std::fmax(std::fabs((double)s1[k] - r[k]), std::fabs((double)s2[k] - r[k])));
    dev = std::fmax(dev, std::fabs((double)q[k] - ((double)s1[k] + (double)s2[k] - (double)r[k])));
  }
  if (!(size > 0.0 && dev <= 1e-6 * size)) return false;
  qd.c[0] = r; qd.c[1] = s1; qd.c[2] = q; qd.c[3] = s2;
  qd.used = false;
  return true;
}

static bool invert3(const double M[3][3], double inv[3][3], double& det) {
  const double c00 = M[1][1] * M[2][2] - M[1][2] * M[2][1], c01 = M[1][2] * M[2][0] - M[1][0] * M[2][2],
               c02 = M[1][0] * M[2][1] - M[1][1] * M[2][0];
  det = M[0][0] * c00 + M[0][1] * c01 + M[0][2] * c02;
  if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return false;
  inv[0][0] = c00 / det; inv[1][0] = c01 / det; inv[2][0] = c02 / det;
  inv[0][1] = (M[0][2] * M[2][1] - M[0][1] * M[2][2]) / det;
  inv[1][1] = (M[0][0] * M[2][2] - M[0][2] * M[2][0]) / det;
  inv[2][1] = (M[0][1] * M[2][0] - M[0][0] * M[2][1]) / det;
  inv[0][2] = (M[0][1] * M[1][2] - M[0][2] * M[1][1]) / det;
  inv[1][2] = (M[0][2] * M[1][0] - M[0][0] * M[1][2]) / det;
  inv[2][2] = (M[0][0] * M[1][1] - M[0][1] * M[1][0]) / det;
  return true;
}

// Tries to grow a box from quad `seed`: a corner r of it with its two neighbours a, b, and a third
// edge c taken from another quad that meets the seed along an edge at r. Every unused quad whose
// corners sit on the corners of one face of the parallelepiped (r; a - r, b - r, c - r) to 10^-6 of
// the size joins. Three faces make a box, and so do two when one of them would otherwise need the
// general (cross product) form of the filter: the vertical sides of a block whose base is no
// parallelogram pair up around two opposite vertical edges.
static bool make_box_unit(std::vector<QuadRef>& quads, size_t seed, BoxUnit& box, std::vector<size_t>& members) {
  const QuadRef& Q0 = quads[seed];
  for (int i0 = 0; i0 < 4; ++i0) {
    const float* r = Q0.c[i0];
    const float* a = Q0.c[(i0 + 1) % 4];
    const float* b = Q0.c[(i0 + 3) % 4];
    for (size_t j = 0; j < quads.size(); ++j) {
      if (j == seed || quads[j].used) continue;
      const QuadRef& Qj = quads[j];
      const float* c = nullptr;
      for (int i = 0; i < 4 && !c; ++i) {
        if (!same_vertex(Qj.c[i], r)) continue;
        const float* n1 = Qj.c[(i + 1) % 4];
        const float* n2 = Qj.c[(i + 3) % 4];
        const bool n1_shared = same_vertex(n1, a) || same_vertex(n1, b);
        const bool n2_shared = same_vertex(n2, a) || same_vertex(n2, b);
        if (n1_shared && !n2_shared) c = n2;
        else if (n2_shared && !n1_shared) c = n1;
      }
      if (!c) continue;
      // frame: columns A, B, C
      double M[3][3], inv[3][3], vol;
      for (int k = 0; k < 3; ++k) {
        M[k][0] = (double)a[k] - r[k]; M[k][1] = (double)b[k] - r[k]; M[k][2] = (double)c[k] - r[k];
      }
      if (!invert3(M, inv, vol)) continue;
      double edge = 0.0;
      for (int k = 0; k < 3; ++k) for (int l = 0; l < 3; ++l) edge = std::fmax(edge, std::fabs(M[k][l]));
      if (!(std::fabs(vol) >= 1e-3 * edge * edge * edge)) continue;  // well shaped
      // which unused quads are faces of this parallelepiped
      struct Face { int quad; bool flip, a_is_lo; };
      Face face[3][2];
      for (int w = 0; w < 3; ++w) for (int sd = 0; sd < 2; ++sd) face[w][sd].quad = -1;
      int n_faces = 0;
      double emax = edge;
      for (size_t m = 0; m < quads.size(); ++m) {
        if (quads[m].used) continue;
        const QuadRef& Qm = quads[m];
        int fc[4][3];
        bool ok = true;
        for (int v = 0; v < 4 && ok; ++v) {
          double x[3] = {(double)Qm.c[v][0] - r[0], (double)Qm.c[v][1] - r[1], (double)Qm.c[v][2] - r[2]};
          double ideal[3] = {0, 0, 0};
          for (int i = 0; i < 3; ++i) {
            const double f = inv[i][0] * x[0] + inv[i][1] * x[1] + inv[i][2] * x[2];
            const double fr = std::nearbyint(f);
            if (!(fr == 0.0 || fr == 1.0)) { ok = false; break; }
            fc[v][i] = (int)fr;
            for (int k = 0; k < 3; ++k) ideal[k] += fr * M[k][i];
          }
          for (int k = 0; k < 3 && ok; ++k) if (!(std::fabs(ideal[k] - x[k]) <= 1e-6 * edge)) ok = false;
        }
        if (!ok) continue;
        // one coordinate constant over the four corners, the other two running over the unit square
        int w = -1;
        for (int i = 0; i < 3; ++i)
          if (fc[0][i] == fc[1][i] && fc[1][i] == fc[2][i] && fc[2][i] == fc[3][i]) { w = (w < 0) ? i : 3; }
        if (w < 0 || w > 2) continue;
        const int ip = (w + 1) % 3, iq = (w + 2) % 3;
        // cyclic order r, s1, q, s2: opposite corners differ in both coordinates, neighbours in one
        if (fc[0][ip] == fc[2][ip] || fc[0][iq] == fc[2][iq] || fc[1][ip] == fc[3][ip] || fc[1][iq] == fc[3][iq]) continue;
        const int sd = fc[0][w];
        if (face[w][sd].quad >= 0) continue;
        Face F;
        F.quad = (int)m;
        // the pair's shared edge s1 - s2 is the diagonal; it must run (1,0) - (0,1) after the flip
        F.flip = (fc[1][ip] + fc[1][iq]) != 1;
        const int pr = F.flip ? 1 - fc[0][ip] : fc[0][ip];
        F.a_is_lo = (pr == 0 && fc[0][iq] == 0);
        if (!F.a_is_lo && !(pr == 1 && fc[0][iq] == 1)) continue;
        face[w][sd] = F;
        ++n_faces;
        for (int u = 0; u < 4; ++u) for (int v = u + 1; v < 4; ++v) for (int k = 0; k < 3; ++k)
          emax = std::fmax(emax, std::fabs((double)Qm.c[u][k] - (double)Qm.c[v][k]));
      }
      if (n_faces < 2) continue;
      if (n_faces == 2) {
        bool general = false;
        for (int w = 0; w < 3; ++w) for (int sd = 0; sd < 2; ++sd)
          if (face[w][sd].quad >= 0 && !quads[(size_t)face[w][sd].quad].axis_form) general = true;
        if (!general) continue;
      }
      double vmax = 0.0;
      for (int k = 0; k < 3; ++k) vmax = std::fmax(vmax, std::fabs((double)r[k]));
      if (!(emax >= 9.094947e-13 && emax <= 1.0995116e12 && vmax + 3.0 * emax <= 1.0995116e12)) continue;
      // the record
      memset(&box, 0, sizeof box);
      box.rx = r[0]; box.ry = r[1]; box.rz = r[2];
      double R[3], rmax = 0.0, rmin = INFINITY;
      for (int i = 0; i < 3; ++i) {
        R[i] = 0.0;
        for (int k = 0; k < 3; ++k) { box.m[3 * i + k] = (float)inv[i][k]; R[i] += std::fabs((double)box.m[3 * i + k]); }
        R[i] *= 1.000001;
        rmax = std::fmax(rmax, R[i]); rmin = std::fmin(rmin, R[i]);
      }
      const double eps = 5.9604644775390625e-08, av = std::fabs(vol);
      const double kappa = rmax / rmin, lambda = std::fmax(1.0, emax * rmax);
      if (!(kappa <= 64.0 && lambda <= 64.0)) continue;
      const double up = 1.000001;
      box.rmax = (float)(rmax * up);
      box.c0 = (float)(0.0211 + 1e-5 * lambda);
      box.k6 = (float)(2.0 * 12.0 * eps * kappa * up);
      for (int w = 0; w < 3; ++w) {
        box.ax[w].kA = (float)(std::fmax(0.001953125 * emax * emax / av, 0.001953125 * R[w]) * FILTER_TIGHT * 1.001);
        const double kR = 3.0517578125e-05 * lambda * emax / (rmax * R[w] * av);
        box.ax[w].kh = (float)((2.0 * kR + 8e-6 * lambda * kappa + 2.0 * 60.0 * eps * kappa) * up);
        box.ax[w].kT = (float)((3.0517578125e-05 * lambda * emax * emax / (rmax * av) + 10.0 * eps) * 1.01);
        box.ax[w].rw = (float)(R[w] * up);
        if (!(std::isfinite(box.ax[w].kA) && std::isfinite(box.ax[w].kh) && std::isfinite(box.ax[w].kT) && box.ax[w].kA > 0.f)) { n_faces = 0; break; }
        for (int sd = 0; sd < 2; ++sd) {
          const Face& F = face[w][sd];
          if (F.quad < 0) continue;
          const QuadRef& Qm = quads[(size_t)F.quad];
          const uint32_t lo = F.a_is_lo ? Qm.bit_a : Qm.bit_b, hi = F.a_is_lo ? Qm.bit_b : Qm.bit_a;
          box.ax[w].faces |= (lo | (hi << 5) | (F.flip ? 0x400u : 0u) | 0x800u) << (12 * sd);
        }
      }
      if (n_faces < 2) continue;
      members.clear();
      for (int w = 0; w < 3; ++w) for (int sd = 0; sd < 2; ++sd)
        if (face[w][sd].quad >= 0) members.push_back((size_t)face[w][sd].quad);
      return true;
    }
  }
  return false;
}

constexpr size_t MAX_BOXES_PER_CHUNK = 7;  // bits 26-28 of a chunk's meta word
// per chunk of 32 triangles: pairs (greedy over consecutive indices) first, then singles
// `ids` lists the triangles to cover, ascending (all of them for a linear-scan scene, the big ones
// of a tree scene); mask bit p of chunk c stands for triangle ids[32 c + p].
static void build_scan_units(const MiptTriangle* tris, const std::vector<TriRec>& tr, const std::vector<int>& ids,
                             std::vector<ScanUnit>& units, uint32_t& n_pair_units, uint32_t& n_axis_units,
                             uint32_t& n_box_units) {
  const size_t n = ids.size();
  const size_t n_chunks = (n + 31) / 32;
  units.assign(n_chunks * 32 + 1, ScanUnit{});
  n_pair_units = 0;
  n_axis_units = 0;
  n_box_units = 0;
  bool pairing = true, boxing = true;
#ifdef MIPT_DEV_SWITCHES
  const char* no_pairs = getenv("MIPT_NO_PAIRS");  // developer build only: every triangle its own unit
  pairing = !(no_pairs && no_pairs[0] == '1');
  const char* no_boxes = getenv("MIPT_NO_BOXES");  // developer build only: no box units
  boxing = !(no_boxes && no_boxes[0] == '1');
#endif
  for (size_t c = 0; c < n_chunks; ++c) {
    const size_t lo = c * 32, hi = std::min(n, lo + 32);
    struct PairEntry { ScanUnit u; bool axis_form; int quad; };
    std::vector<PairEntry> pair_list;
    std::vector<QuadRef> quads;
    std::vector<ScanUnit> axis_pairs, pairs, singles;
    for (size_t p = lo; p < hi;) {
      ScanUnit u;
      const int i = ids[p];
      if (pairing && p + 1 < hi && ids[p + 1] == i + 1 && std::isfinite(tr[i].kA) && std::isfinite(tr[i + 1].kA) &&
          make_pair_unit(tris[i], tris[i + 1], u)) {
        PairEntry e;
        e.axis_form = u.meta == 0x80000000u;
        u.meta = (uint32_t)(p - lo) | ((uint32_t)(p + 1 - lo) << 5);
        e.u = u;
        e.quad = -1;
        QuadRef qd;
        if (boxing && quad_of_pair(tris[i], tris[i + 1], qd)) {
          qd.bit_a = (uint32_t)(p - lo); qd.bit_b = (uint32_t)(p + 1 - lo);
          qd.axis_form = e.axis_form;
          e.quad = (int)quads.size();
          quads.push_back(qd);
        }
        pair_list.push_back(e);
        p += 2;
      } else {
        make_single_unit(tr[i], u);
        u.meta = (uint32_t)(p - lo);
        singles.push_back(u);
        p += 1;
      }
    }
    // parallelograms that are faces of one parallelepiped leave the pair list for a box unit
    std::vector<BoxUnit> boxes;
    for (size_t qi = 0; qi < quads.size(); ++qi) {
      if (quads[qi].used) continue;
      BoxUnit box;
      std::vector<size_t> members;
      // the chunk's meta word has three bits for the number of boxes: quads beyond the seventh
      // box stay pair units
      if (boxes.size() == MAX_BOXES_PER_CHUNK) break;
      if (!make_box_unit(quads, qi, box, members)) continue;
      for (size_t m : members) quads[m].used = true;
      boxes.push_back(box);
    }
    size_t n_axis_here = 0;
    for (const PairEntry& e : pair_list) {
      if (e.axis_form) ++n_axis_here;
      if (e.quad >= 0 && quads[(size_t)e.quad].used) continue;
      (e.axis_form ? axis_pairs : pairs).push_back(e.u);
    }
    size_t k = lo;
    for (const BoxUnit& b : boxes) { memcpy(&units[k], &b, sizeof b); k += 2; }
    // a stable insertion sort in place (at most 16 units): std::stable_sort moves the elements through
    // a temporary buffer that the library allocates without regard for the records' 64-byte alignment
    {
      auto key = [](const ScanUnit& x) { uint32_t a; memcpy(&a, &x.e2z, 4); return a; };
      for (size_t i = 1; i < axis_pairs.size(); ++i) {
        const ScanUnit u = axis_pairs[i];
        size_t j = i;
        for (; j > 0 && key(axis_pairs[j - 1]) > key(u); --j) axis_pairs[j] = axis_pairs[j - 1];
        axis_pairs[j] = u;
      }
    }
    for (const ScanUnit& u : axis_pairs) units[k++] = u;
    for (const ScanUnit& u : pairs) units[k++] = u;
    for (const ScanUnit& u : singles) units[k++] = u;
    const size_t all_pairs = axis_pairs.size() + pairs.size();
    // what the fields of the meta word can hold (unit_mask): 5 + 5 + 6 + 3 bits
    if (axis_pairs.size() > 31 || all_pairs > 31 || all_pairs + singles.size() > 63 || boxes.size() > MAX_BOXES_PER_CHUNK)
      throw std::length_error("scan-unit counts of a chunk do not fit its meta word");
    // the counts of the chunk (unit_mask); a box keeps its meta where a scan unit does
    const uint32_t counts = ((uint32_t)axis_pairs.size() << 10) | ((uint32_t)all_pairs << 15) |
                            ((uint32_t)(all_pairs + singles.size()) << 20) | ((uint32_t)boxes.size() << 26);
    if (boxes.empty()) units[lo].meta |= counts;
    else reinterpret_cast<BoxUnit*>(&units[lo])->meta = counts;
    n_pair_units += (uint32_t)pair_list.size();
    n_axis_units += (uint32_t)n_axis_here;
    n_box_units += (uint32_t)boxes.size();
  }
}

static int upload_scan_units(MiptScene* s, const std::vector<ScanUnit>& units) {
  HIP_TRY(hipMalloc(&s->d_units, units.size() * sizeof(ScanUnit)));
  HIP_TRY(hipMemcpy(s->d_units, units.data(), units.size() * sizeof(ScanUnit), hipMemcpyHostToDevice));
  s->d.units = s->d_units;
  return MIPT_OK;
}

// The per-triangle records of the scan (TriRec in mipt_device.hpp) and the largest |coordinate|.
static void make_tri_recs(const MiptTriangle* tris, size_t n_tris, std::vector<TriRec>& tr, float& scale) {
  tr.assign(n_tris + 2, TriRec{});  // two records of padding: the scan prefetches ahead
  memset(tr.data(), 0, tr.size() * sizeof(TriRec));
  scale = 0.f;
  for (size_t i = 0; i < n_tris; ++i) {
    const MiptTriangle& t = tris[i];
    TriRec& r = tr[i];
    memset(&r, 0, sizeof r);
    r.v0x = t.v0[0]; r.v0y = t.v0[1]; r.v0z = t.v0[2]; r.v0w = t.v0[3];
    // intersections.cpp:17-18; plain float subtraction, same on host and device
    volatile float e1x = t.v1[0] - t.v0[0], e1y = t.v1[1] - t.v0[1], e1z = t.v1[2] - t.v0[2];
    volatile float e2x = t.v2[0] - t.v0[0], e2y = t.v2[1] - t.v0[1], e2z = t.v2[2] - t.v0[2];
    for (int k = 0; k < 3; ++k) {
      const float m = std::fmax(std::fabs(t.v0[k]), std::fmax(std::fabs(t.v1[k]), std::fabs(t.v2[k])));
      if (m > scale && m <= 1.0995116e12f) scale = m;
    }
    r.e1x = e1x; r.e1y = e1y; r.e1z = e1z;
    r.e2x = e2x; r.e2y = e2y; r.e2z = e2z;
    // cofactors; each product is stored before the subtraction so that no FMA can form
    volatile float a0 = e1y * e2z, b0 = e2y * e1z;
    volatile float a1 = e1x * e2z, b1 = e2x * e1z;
    volatile float a2 = e1x * e2y, b2 = e2x * e1y;
    r.c0 = a0 - b0; r.c1 = a1 - b1; r.c2 = a2 - b2;
    // error-bound factors of the candidate filter (tri_candidate); a triangle whose magnitudes
    // leave the range the bounds were derived for is marked "never reject"
    const float E1 = std::fmax(std::fmax(std::fabs(r.e1x), std::fabs(r.e1y)), std::fabs(r.e1z));
    const float E2 = std::fmax(std::fmax(std::fabs(r.e2x), std::fabs(r.e2y)), std::fabs(r.e2z));
    const float V0 = std::fmax(std::fmax(std::fabs(r.v0x), std::fabs(r.v0y)), std::fabs(r.v0z));
    const bool sane = E1 >= 9.094947e-13f && E1 <= 1.0995116e12f && E2 >= 9.094947e-13f && E2 <= 1.0995116e12f &&
                      V0 <= 1.0995116e12f;  // [2^-40, 2^40]; NaN fails
    r.kA = sane ? (0.000244140625f * FILTER_TIGHT) * (E1 * E2) : INFINITY;  // 2^-12
    r.kB = sane ? 7.62939453125e-06f * std::fmax(E1, E2) : 0.f;  // 2^-17
    r.kT = sane ? 7.62939453125e-06f * (E1 * E2) : 0.f;
  }
}

static int scene_kind(const MiptScene* s) {
  if (s->d.bvh) return SCENE_BVH;
  const int n_prims = s->d.n_tris + s->d.n_spheres;
  if (n_prims <= LDS_TABLE_MAX) return SCENE_LINEAR_LDS;
  // a primitive index must fit the vertex stack's tag below the two marks at the top of its range
  return n_prims < TAG16_MAX_PRIMS ? SCENE_LINEAR : SCENE_LINEAR_BIG;
}

// Persistent grid of the path kernel for scenes with global vertex stacks: what fits the chip at the
// kernel's occupancy (256 CUs x 5 workgroups at most), so that the stacks have a fixed, modest size:
// 2 planes x 2 buffers x (MIPT_MAX_DEPTH_LIMIT + 1) levels x 16 bytes per lane = 357 MB.
constexpr unsigned VSTACK_BLOCKS = 1280;
static inline size_t vstack_bytes(size_t lanes, size_t max_depth) { return 2 * 2 * (max_depth + 1) * lanes * sizeof(float4); }

static int scene_create_scan(const MiptTriangle* tris, size_t n_tris, const MiptSphere* spheres,
                             size_t n_spheres, int device, uint32_t options, MiptScene* s);
static int scene_create_impl(const MiptTriangle* tris, size_t n_tris, const MiptSphere* spheres,
                             size_t n_spheres, int device, uint32_t options, MiptScene* s) {
  const int rc = scene_create_scan(tris, n_tris, spheres, n_spheres, device, options, s);
  if (rc != MIPT_OK) return rc;
  if (vstack_kind(scene_kind(s)))
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_vstack),
                      vstack_bytes((size_t)VSTACK_BLOCKS * BLOCK, MIPT_MAX_DEPTH_LIMIT)));
  return MIPT_OK;
}

static int scene_create_scan(const MiptTriangle* tris, size_t n_tris, const MiptSphere* spheres,
                             size_t n_spheres, int device, uint32_t options, MiptScene* s) {
  s->device = device;
  HIP_TRY(hipSetDevice(device));
  std::vector<TriRec> tr;
  float scale = 0.f;
  make_tri_recs(tris, n_tris, tr, scale);
  std::vector<ShadeRec> sh(n_tris + n_spheres ? n_tris + n_spheres : 1);
  std::vector<SphereRec> sp(n_spheres ? n_spheres : 1);
  for (size_t i = 0; i < n_tris; ++i) {
    const MiptTriangle& t = tris[i];
    ShadeRec& q = sh[i];
    q.nx = t.normal[0]; q.ny = t.normal[1]; q.nz = t.normal[2]; q.nw = t.normal[3];
    q.r = t.color[0]; q.g = t.color[1]; q.b = t.color[2];
    q.material = t.material;
    // constants of the normal (this file is compiled with -ffp-contract=off: same bits as on the device)
    const V4 n4 = {q.nx, q.ny, q.nz, q.nw};
    const V4 nn = normalize4(n4);
    V3 nt, nb;
    orthonormal_system(n4, nt, nb);
    q.nnx = nn.x; q.nny = nn.y; q.nnz = nn.z; q.nnw = nn.w;
    q.ntx = nt.x; q.nty = nt.y; q.ntz = nt.z;
    q.nbx = nb.x; q.nby = nb.y; q.nbz = nb.z;
    q.pad0 = q.pad1 = 0.f;
  }
  for (size_t j = 0; j < n_spheres; ++j) {
    const MiptSphere& u = spheres[j];
    SphereRec& r = sp[j];
    memset(&r, 0, sizeof r);
    r.cx = u.center[0]; r.cy = u.center[1]; r.cz = u.center[2]; r.cw = u.center[3];
    r.radius = u.radius; r.r = u.color[0]; r.g = u.color[1]; r.b = u.color[2];
    r.material = u.material;
    ShadeRec& q = sh[n_tris + j];
    memset(&q, 0, sizeof q);
    q.r = u.color[0]; q.g = u.color[1]; q.b = u.color[2];
    q.material = u.material;
  }
  HIP_TRY(hipMalloc(&s->d_tris, tr.size() * sizeof(TriRec)));
  HIP_TRY(hipMalloc(&s->d_shade, sh.size() * sizeof(ShadeRec)));
  HIP_TRY(hipMalloc(&s->d_spheres, sp.size() * sizeof(SphereRec)));
  HIP_TRY(hipMalloc(&s->d_stats, (ST_COUNT + PF_COUNT) * sizeof(unsigned long long)));
  // 256 bytes for the path kernel's queue head, then BATCH_HEADS heads of the batch kernels' queue,
  // one 128-byte line each
  HIP_TRY(hipMalloc(&s->d_queue, 256 + BATCH_HEADS * 128));
  HIP_TRY(hipMemcpy(s->d_tris, tr.data(), tr.size() * sizeof(TriRec), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(s->d_shade, sh.data(), sh.size() * sizeof(ShadeRec), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(s->d_spheres, sp.data(), sp.size() * sizeof(SphereRec), hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(s->d_stats, 0, (ST_COUNT + PF_COUNT) * sizeof(unsigned long long)));
  HIP_TRY(hipEventCreate(&s->ev0));
  HIP_TRY(hipEventCreate(&s->ev1));
  HIP_TRY(hipEventCreateWithFlags(&s->ev_busy, hipEventDisableTiming));
  {
    const int rc = noalloc_setup(device);
    if (rc != MIPT_OK) return rc;
  }
  s->d.tris = s->d_tris;
  s->d.shade = s->d_shade;
  s->d.spheres = s->d_spheres;
  s->d.n_tris = (int)n_tris;
  s->d.n_spheres = (int)n_spheres;
  s->d.scale = scale;
  if (n_tris >= MIPT_BVH_MIN_TRIS && !(options & MIPT_SCENE_LINEAR_SCAN)) return build_tree_scene(s, tris, tr, n_tris);
  std::vector<ScanUnit> units;
  std::vector<int> all(n_tris);
  for (size_t i = 0; i < n_tris; ++i) all[i] = (int)i;
  build_scan_units(tris, tr, all, units, s->n_pair_units, s->n_axis_units, s->n_box_units);
  return upload_scan_units(s, units);
}

extern "C" int mipt_scene_create(const MiptTriangle* tris, size_t n_tris, const MiptSphere* spheres,
                                 size_t n_spheres, int device, MiptScene** out) {
  return mipt_scene_create_opts(tris, n_tris, spheres, n_spheres, device, 0u, out);
}

extern "C" int mipt_scene_create_opts(const MiptTriangle* tris, size_t n_tris, const MiptSphere* spheres,
                                      size_t n_spheres, int device, uint32_t options, MiptScene** out) {
  if (!out) return fail(MIPT_E_INVALID, "mipt_scene_create: out is null");
  *out = nullptr;
  if (options & ~(uint32_t)MIPT_SCENE_LINEAR_SCAN)
    return fail(MIPT_E_INVALID, "mipt_scene_create_opts: unknown option bits 0x%x", options);
  if ((n_tris && !tris) || (n_spheres && !spheres))
    return fail(MIPT_E_INVALID, "mipt_scene_create: null primitive array with non-zero count");
  if (n_tris > (size_t)1 << 24 || n_spheres > 4096)
    return fail(MIPT_E_INVALID, "mipt_scene_create: too many primitives (%zu triangles, %zu spheres)",
                n_tris, n_spheres);
  for (size_t i = 0; i < n_tris; ++i)
    if (tris[i].material > MIPT_MAT_NONE)
      return fail(MIPT_E_INVALID, "mipt_scene_create: triangle %zu has material code %u", i, tris[i].material);
  for (size_t j = 0; j < n_spheres; ++j)
    if (spheres[j].material > MIPT_MAT_NONE)
      return fail(MIPT_E_INVALID, "mipt_scene_create: sphere %zu has material code %u", j, spheres[j].material);
  int n_dev = 0;
  HIP_TRY(hipGetDeviceCount(&n_dev));
  if (device < 0 || device >= n_dev)
    return fail(MIPT_E_INVALID, "mipt_scene_create: device %d out of range (%d devices)", device, n_dev);
  MiptScene* s = new (std::nothrow) MiptScene();
  if (!s) return fail(MIPT_E_NOMEM, "mipt_scene_create: host allocation failed");
  int rc;
  try {
    rc = scene_create_impl(tris, n_tris, spheres, n_spheres, device, options, s);
  } catch (const std::bad_alloc&) {
    rc = fail(MIPT_E_NOMEM, "mipt_scene_create: host allocation failed");
  } catch (...) {
    rc = fail(MIPT_E_INVALID, "mipt_scene_create: unexpected exception");
  }
  if (rc != MIPT_OK) { scene_free(s); return rc; }
  *out = s;
  return MIPT_OK;
}

// ---------------------------------------------------------------------------------------
// batch queries

extern "C" int mipt_scene_has_bvh(MiptScene* s, uint64_t* n_nodes, double* build_ms) {
  if (n_nodes) *n_nodes = 0;
  if (build_ms) *build_ms = 0.0;
  if (!s || !s->d.bvh) return 0;
  if (n_nodes) *n_nodes = (uint64_t)s->n_wide_nodes;
  if (build_ms) *build_ms = s->bvh_build_ms;
  return 1;
}

extern "C" int mipt_scene_tree_info(MiptScene* s, int* clustered, int* depth, float* cost_ratio) {
  if (clustered) *clustered = 0;
  if (depth) *depth = 0;
  if (cost_ratio) *cost_ratio = 0.f;
  if (!s || !s->d.bvh) return 0;
  if (clustered) *clustered = s->bvh_radix_tree ? 0 : 1;
  if (depth) *depth = s->bvh_depth;
  if (cost_ratio) *cost_ratio = s->bvh_cost_ratio;
  return 1;
}

// persistent grid of the tree batch kernels: enough workgroups to fill the chip, no more than the rays need
static inline unsigned batch_grid(size_t n_rays) {
  const size_t need = (n_rays + BLOCK - 1) / BLOCK;
  return (unsigned)(need < PERSISTENT_BLOCKS ? need : PERSISTENT_BLOCKS);
}

extern "C" int mipt_scene_pair_units(MiptScene* s) { return s ? (int)s->n_pair_units : 0; }
extern "C" int mipt_scene_axis_pair_units(MiptScene* s) { return s ? (int)s->n_axis_units : 0; }
extern "C" int mipt_scene_box_units(MiptScene* s) { return s ? (int)s->n_box_units : 0; }

extern "C" int mipt_plan_scan_units(const MiptTriangle* tris, size_t n_tris, int* pair_units, int* axis_pair_units,
                                    int* box_units) {
  if (n_tris && !tris) return fail(MIPT_E_INVALID, "mipt_plan_scan_units: null triangle array");
  if (n_tris > (size_t)1 << 24) return fail(MIPT_E_INVALID, "mipt_plan_scan_units: too many triangles (%zu)", n_tris);
  try {
    std::vector<TriRec> tr;
    float scale = 0.f;
    make_tri_recs(tris, n_tris, tr, scale);
    std::vector<int> all(n_tris);
    for (size_t i = 0; i < n_tris; ++i) all[i] = (int)i;
    std::vector<ScanUnit> units;
    uint32_t np = 0, na = 0, nb = 0;
    build_scan_units(tris, tr, all, units, np, na, nb);
    if (pair_units) *pair_units = (int)np;
    if (axis_pair_units) *axis_pair_units = (int)na;
    if (box_units) *box_units = (int)nb;
  } catch (const std::bad_alloc&) {
    return fail(MIPT_E_NOMEM, "mipt_plan_scan_units: host allocation failed");
  } catch (...) {
    return fail(MIPT_E_INVALID, "mipt_plan_scan_units: unexpected exception");
  }
  return MIPT_OK;
}

// Tree batches of SORT_MIN_RAYS rays or more: bin the rays by origin cell unless they come coherent.
// On return b.order is null (walk the rays as they come) or b holds the batch in walking order and
// the index each result goes to. Whether a scene's batches come scattered is learnt without waiting:
// every such call may launch a probe whose count travels to pinned host memory while the batch runs,
// and the NEXT call reads it (batches of one caller resemble each other; the order only affects
// speed, never results). The first big batch of a scene is therefore walked as it comes.
constexpr size_t SORT_MIN_RAYS = (size_t)1 << 20;
struct OrderedBatch {
  const float4* origin;
  const float4* dir;
  const float* max_t;
  const uint32_t* order;
};
static int order_rays(MiptScene* s, const float* d_origin4, const float* d_dir4, const float* d_max_t, size_t n_rays,
                      hipStream_t st, OrderedBatch& b) {
  b = OrderedBatch{(const float4*)d_origin4, (const float4*)d_dir4, d_max_t, nullptr};
  if (n_rays < SORT_MIN_RAYS || n_rays > 0xFFFFFFFFull || !(s->d.scale > 0.f)) return MIPT_OK;
  if (n_rays > s->sort_cap) {
    if (s->busy_valid) HIP_TRY(hipEventSynchronize(s->ev_busy));
    HIP_TRY(hipStreamSynchronize(st));
    if (s->d_sort) { (void)hipFree(s->d_sort); s->d_sort = nullptr; }
    if (s->d_sorted_rays) { (void)hipFree(s->d_sorted_rays); s->d_sorted_rays = nullptr; }
    if (s->d_sort_tmp) { (void)hipFree(s->d_sort_tmp); s->d_sort_tmp = nullptr; s->sort_tmp_bytes = 0; }
    s->sort_cap = 0;
    HIP_TRY(hipMalloc(&s->d_sort, n_rays * 4 * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&s->d_sorted_rays, n_rays * (2 * sizeof(float4) + sizeof(float))));
    size_t tmp = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp, s->d_sort, s->d_sort + n_rays, s->d_sort + 2 * n_rays,
                                      s->d_sort + 3 * n_rays, n_rays, 0, 12, st));
    HIP_TRY(hipMalloc(&s->d_sort_tmp, tmp ? tmp : 1));
    s->sort_tmp_bytes = tmp;
    s->sort_cap = n_rays;
    if (!s->d_breaks) HIP_TRY(hipMalloc(&s->d_breaks, sizeof(unsigned int)));
    if (!s->h_breaks) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_breaks), sizeof(unsigned int), hipHostMallocDefault));
    if (!s->ev_probe) HIP_TRY(hipEventCreateWithFlags(&s->ev_probe, hipEventDisableTiming));
  }
  uint32_t* keys = s->d_sort;
  uint32_t* keys2 = keys + s->sort_cap;
  uint32_t* vals = keys2 + s->sort_cap;
  uint32_t* vals2 = vals + s->sort_cap;
  if (s->probe_pending && hipEventQuery(s->ev_probe) == hipSuccess) {
    // the probe saw one workgroup in 64; fewer than one ray in eight changing cell is coherent enough
    s->scattered = (size_t)*s->h_breaks * 64 * 8 >= s->probe_rays ? 1 : 0;
    s->probe_pending = false;
    s->calls_since_probe = 0;
  }
  (void)hipGetLastError();  // hipErrorNotReady from the query is not an error
  // keys are needed when sorting, and for a fresh probe: always while nothing is known, else now and then
  const bool sort = s->scattered == 1;
  const bool probe = !s->probe_pending && (s->scattered < 0 || sort || (++s->calls_since_probe & 15u) == 0);
  if (!sort && !probe) return MIPT_OK;
  if (probe) HIP_TRY(hipMemsetAsync(s->d_breaks, 0, sizeof(unsigned int), st));
  hipLaunchKernelGGL(k_ray_cell_keys, dim3(grid_for(n_rays)), dim3(BLOCK), 0, st, (const float4*)d_origin4,
                     (unsigned long long)n_rays, s->d.scale, keys, vals, probe ? s->d_breaks : nullptr);
  HIP_TRY(hipGetLastError());
  if (probe) {
    HIP_TRY(hipMemcpyAsync(s->h_breaks, s->d_breaks, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(s->ev_probe, st));
    s->probe_pending = true;
    s->probe_rays = n_rays;
  }
  if (!sort) return MIPT_OK;
  size_t tmp = s->sort_tmp_bytes;
  HIP_TRY(rocprim::radix_sort_pairs(s->d_sort_tmp, tmp, keys, keys2, vals, vals2, n_rays, 0, 12, st));
  float4* so = s->d_sorted_rays;
  float4* sd = so + s->sort_cap;
  float* sm = reinterpret_cast<float*>(sd + s->sort_cap);
  hipLaunchKernelGGL(k_gather_rays, dim3(grid_for(n_rays)), dim3(BLOCK), 0, st, (const float4*)d_origin4,
                     (const float4*)d_dir4, d_max_t, vals2, (unsigned long long)n_rays, so, sd, sm);
  HIP_TRY(hipGetLastError());
  b = OrderedBatch{so, sd, d_max_t ? sm : nullptr, vals2};
  return MIPT_OK;
}

extern "C" int mipt_closest_hit_dev(MiptScene* s, const float* d_origin4, const float* d_dir4, size_t n_rays,
                                    int32_t* d_tri_idx, float* d_t, float* d_position4, void* stream) {
  if (!s) return fail(MIPT_E_INVALID, "mipt_closest_hit: scene is null");
  if (n_rays == 0) return MIPT_OK;
  if (!d_origin4 || !d_dir4 || !d_tri_idx || !d_t) return fail(MIPT_E_INVALID, "mipt_closest_hit: null buffer");
  if (n_rays > ((size_t)1 << 31) * BLOCK) return fail(MIPT_E_INVALID, "mipt_closest_hit: too many rays");
  HIP_TRY(hipSetDevice(s->device));
  if (s->d.bvh) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(s->d_queue) + 32;
    int rc = scratch_acquire(s, (hipStream_t)stream);
    if (rc != MIPT_OK) return rc;
    OrderedBatch b;
    rc = order_rays(s, d_origin4, d_dir4, nullptr, n_rays, (hipStream_t)stream, b);
    if (rc != MIPT_OK) return rc;
    HIP_TRY(hipMemsetAsync(q, 0, BATCH_HEADS * 128, (hipStream_t)stream));
    hipLaunchKernelGGL(k_bvh_batch<false>, dim3(batch_grid(n_rays)), dim3(BLOCK), 0, (hipStream_t)stream, s->d,
                       b.origin, b.dir, (const float*)nullptr,
                       (unsigned long long)n_rays, q, d_tri_idx, d_t, (float4*)d_position4, (uint8_t*)nullptr, b.order);
    HIP_TRY(hipGetLastError());
    return scratch_release(s, (hipStream_t)stream);
  } else
    hipLaunchKernelGGL(k_closest_hit<false>, dim3(grid_for(n_rays)), dim3(BLOCK), 0, (hipStream_t)stream, s->d,
                       (const float4*)d_origin4, (const float4*)d_dir4, n_rays, d_tri_idx, d_t,
                       (float4*)d_position4);
  HIP_TRY(hipGetLastError());
  return MIPT_OK;
}

extern "C" int mipt_any_hit_dev(MiptScene* s, const float* d_origin4, const float* d_dir4, const float* d_max_t,
                                size_t n_rays, uint8_t* d_occluded, void* stream) {
  if (!s) return fail(MIPT_E_INVALID, "mipt_any_hit: scene is null");
  if (n_rays == 0) return MIPT_OK;
  if (!d_origin4 || !d_dir4 || !d_max_t || !d_occluded) return fail(MIPT_E_INVALID, "mipt_any_hit: null buffer");
  if (n_rays > ((size_t)1 << 31) * BLOCK) return fail(MIPT_E_INVALID, "mipt_any_hit: too many rays");
  HIP_TRY(hipSetDevice(s->device));
  if (s->d.bvh) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(s->d_queue) + 32;
    int rc = scratch_acquire(s, (hipStream_t)stream);
    if (rc != MIPT_OK) return rc;
    OrderedBatch b;
    rc = order_rays(s, d_origin4, d_dir4, d_max_t, n_rays, (hipStream_t)stream, b);
    if (rc != MIPT_OK) return rc;
    HIP_TRY(hipMemsetAsync(q, 0, BATCH_HEADS * 128, (hipStream_t)stream));
    hipLaunchKernelGGL(k_bvh_batch<true>, dim3(batch_grid(n_rays)), dim3(BLOCK), 0, (hipStream_t)stream, s->d,
                       b.origin, b.dir, b.max_t, (unsigned long long)n_rays, q,
                       (int32_t*)nullptr, (float*)nullptr, (float4*)nullptr, d_occluded, b.order);
    HIP_TRY(hipGetLastError());
    return scratch_release(s, (hipStream_t)stream);
  } else
    hipLaunchKernelGGL(k_any_hit<false>, dim3(grid_for(n_rays)), dim3(BLOCK), 0, (hipStream_t)stream, s->d,
                       (const float4*)d_origin4, (const float4*)d_dir4, d_max_t, n_rays, d_occluded);
  HIP_TRY(hipGetLastError());
  return MIPT_OK;
}

extern "C" int mipt_closest_hit(MiptScene* s, const float* origin4, const float* dir4, size_t n_rays,
                                int32_t* tri_idx, float* t, float* position4) {
  if (!s) return fail(MIPT_E_INVALID, "mipt_closest_hit: scene is null");
  if (n_rays == 0) return MIPT_OK;
  if (!origin4 || !dir4 || !tri_idx || !t) return fail(MIPT_E_INVALID, "mipt_closest_hit: null buffer");
  HIP_TRY(hipSetDevice(s->device));
  int rc = stage_reserve(s, 3 * stage_round(n_rays * 16) + 2 * stage_round(n_rays * 4));
  if (rc != MIPT_OK) return rc;
  StageCursor st{static_cast<char*>(s->d_stage)};
  float* o = st.take<float>(n_rays * 16);
  float* d = st.take<float>(n_rays * 16);
  float* pos = st.take<float>(n_rays * 16);
  int32_t* idx = st.take<int32_t>(n_rays * 4);
  float* tt = st.take<float>(n_rays * 4);
  HIP_TRY(hipMemcpy(o, origin4, n_rays * 16, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d, dir4, n_rays * 16, hipMemcpyHostToDevice));
  // outputs keep the caller's values where the kernel does not write (t / position on a miss)
  HIP_TRY(hipMemcpy(tt, t, n_rays * 4, hipMemcpyHostToDevice));
  if (position4) HIP_TRY(hipMemcpy(pos, position4, n_rays * 16, hipMemcpyHostToDevice));
  rc = mipt_closest_hit_dev(s, o, d, n_rays, idx, tt, position4 ? pos : nullptr, nullptr);
  if (rc != MIPT_OK) return rc;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(tri_idx, idx, n_rays * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(t, tt, n_rays * 4, hipMemcpyDeviceToHost));
  if (position4) HIP_TRY(hipMemcpy(position4, pos, n_rays * 16, hipMemcpyDeviceToHost));
  return MIPT_OK;
}

extern "C" int mipt_any_hit(MiptScene* s, const float* origin4, const float* dir4, const float* max_t,
                            size_t n_rays, uint8_t* occluded) {
  if (!s) return fail(MIPT_E_INVALID, "mipt_any_hit: scene is null");
  if (n_rays == 0) return MIPT_OK;
  if (!origin4 || !dir4 || !max_t || !occluded) return fail(MIPT_E_INVALID, "mipt_any_hit: null buffer");
  HIP_TRY(hipSetDevice(s->device));
  int rc = stage_reserve(s, 2 * stage_round(n_rays * 16) + stage_round(n_rays * 4) + stage_round(n_rays));
  if (rc != MIPT_OK) return rc;
  StageCursor st{static_cast<char*>(s->d_stage)};
  float* o = st.take<float>(n_rays * 16);
  float* d = st.take<float>(n_rays * 16);
  float* m = st.take<float>(n_rays * 4);
  uint8_t* occ = st.take<uint8_t>(n_rays);
  HIP_TRY(hipMemcpy(o, origin4, n_rays * 16, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d, dir4, n_rays * 16, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(m, max_t, n_rays * 4, hipMemcpyHostToDevice));
  rc = mipt_any_hit_dev(s, o, d, m, n_rays, occ, nullptr);
  if (rc != MIPT_OK) return rc;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(occluded, occ, n_rays, hipMemcpyDeviceToHost));
  return MIPT_OK;
}

// ---------------------------------------------------------------------------------------
// path kernel launch

static int check_params(const char* who, const MiptCamera* cam, const MiptLight* light, const MiptParams* p) {
  if (!cam || !light || !p) return fail(MIPT_E_INVALID, "%s: null camera / light / params", who);
  if (p->width <= 0 || p->height <= 0 || p->width > 32768 || p->height > 32768)
    return fail(MIPT_E_INVALID, "%s: image size %d x %d out of range", who, p->width, p->height);
  if (p->max_depth < 0 || p->max_depth > MIPT_MAX_DEPTH_LIMIT)
    return fail(MIPT_E_INVALID, "%s: max_depth %d outside 0..%d", who, p->max_depth, MIPT_MAX_DEPTH_LIMIT);
  if (p->rng_mode == MIPT_RNG_MINSTD_GLOBAL)
    return fail(MIPT_E_UNSUPPORTED,
                "%s: MIPT_RNG_MINSTD_GLOBAL is one sequential stream over all pixels; only the CPU oracle has it", who);
  if (p->rng_mode != MIPT_RNG_MINSTD_PATH && p->rng_mode != MIPT_RNG_COUNTER)
    return fail(MIPT_E_INVALID, "%s: unknown rng_mode %u", who, p->rng_mode);
  if (!(p->fog_sigma_t >= 0.f)) return fail(MIPT_E_INVALID, "%s: fog_sigma_t must be >= 0", who);
  return MIPT_OK;
}

// mat4 * mat4 in GLM's order (type_mat4x4.inl:722-727), each operation rounded to float
static void mat4_mul(const float a[16], const float b[16], float out[16]) {
  for (int j = 0; j < 4; ++j)
    for (int k = 0; k < 4; ++k) {
      volatile float s0 = a[0 * 4 + k] * b[j * 4 + 0];
      volatile float s1 = a[1 * 4 + k] * b[j * 4 + 1];
      volatile float s2 = a[2 * 4 + k] * b[j * 4 + 2];
      volatile float s3 = a[3 * 4 + k] * b[j * 4 + 3];
      volatile float t0 = s0 + s1;
      volatile float t1 = t0 + s2;
      out[j * 4 + k] = t1 + s3;
    }
}

static void fill_args(PathArgs& a, const MiptScene* s, const MiptCamera* cam, const MiptLight* light,
                      const MiptParams* p) {
  memset(&a, 0, sizeof a);
  a.sc = s->d;
  mat4_mul(cam->rot_x, cam->rot_y, a.M);
  memcpy(a.cam_pos, cam->pos, sizeof a.cam_pos);
  a.focal = cam->focal;
  memcpy(a.light_pos, light->pos, sizeof a.light_pos);
  memcpy(a.light_col, light->color, sizeof a.light_col);
  a.light_gray = (light->color[0] == light->color[1] && light->color[1] == light->color[2]) ? 1u : 0u;
  {
    volatile float num = p->ior_air - p->ior_glass, den = p->ior_air + p->ior_glass;
    volatile float r0 = num / den;
    a.glass_R0 = r0 * r0;
    volatile float inv = 1.0f / p->ior_glass;
    a.eta_enter = inv;
    a.eta_exit = 1.0f / inv;
  }
  a.p = *p;
  a.stats = s->d_stats;
}

// dynamic LDS of the path kernel: the per-lane vertex stack, then (small scenes) the record tables
static size_t lds_bytes(const MiptScene* s, const MiptParams* p, PathArgs& a) {
  // the vertex stack(s) of tags, then the sphere normals beside them, then the record tables
  const int kind = scene_kind(s);
  if (vstack_kind(kind)) {  // vertex stacks in global memory: no dynamic LDS at all
    a.lds_sph_off = a.lds_table_off = 0;
    return 0;
  }
  const size_t tag_size = kind == SCENE_LINEAR_LDS ? 1 : kind == SCENE_LINEAR ? 2 : 4;
  const size_t n_stacks = paths_deferred(kind, s->d.n_spheres) ? 2 : 1;
  size_t bytes = n_stacks * (size_t)p->max_depth * BLOCK * tag_size;
  bytes = (bytes + 63) & ~(size_t)63;
  a.lds_sph_off = (uint32_t)(bytes / 4);
  if (s->d.n_spheres > 0) bytes += (size_t)p->max_depth * BLOCK * 16;
  a.lds_table_off = (uint32_t)(bytes / 4);
  if (kind == SCENE_LINEAR_LDS)
    bytes += (size_t)s->d.n_tris * sizeof(TriRec) + (size_t)(s->d.n_tris + s->d.n_spheres) * sizeof(ShadeRec);
  return bytes;
}

template <bool LIST, int KIND>
static void launch_paths_kind(const PathArgs& a, dim3 grid, size_t lds, hipStream_t st) {
  const bool stats = (a.p.flags & MIPT_FLAG_STATS) != 0;
  const bool counter = a.p.rng_mode == MIPT_RNG_COUNTER;
  if (counter) {
    if (stats) hipLaunchKernelGGL((k_paths<MIPT_RNG_COUNTER, true, LIST, KIND>), grid, dim3(BLOCK), lds, st, a);
    else hipLaunchKernelGGL((k_paths<MIPT_RNG_COUNTER, false, LIST, KIND>), grid, dim3(BLOCK), lds, st, a);
  } else {
    if (stats) hipLaunchKernelGGL((k_paths<MIPT_RNG_MINSTD_PATH, true, LIST, KIND>), grid, dim3(BLOCK), lds, st, a);
    else hipLaunchKernelGGL((k_paths<MIPT_RNG_MINSTD_PATH, false, LIST, KIND>), grid, dim3(BLOCK), lds, st, a);
  }
}

template <bool LIST>
static void launch_paths(const MiptScene* s, const PathArgs& a, dim3 grid, size_t lds, hipStream_t st) {
  switch (scene_kind(s)) {
    case SCENE_BVH: launch_paths_kind<LIST, SCENE_BVH>(a, grid, lds, st); break;
    case SCENE_LINEAR: launch_paths_kind<LIST, SCENE_LINEAR>(a, grid, lds, st); break;
    case SCENE_LINEAR_BIG: launch_paths_kind<LIST, SCENE_LINEAR_BIG>(a, grid, lds, st); break;
    default: launch_paths_kind<LIST, SCENE_LINEAR_LDS>(a, grid, lds, st); break;
  }
}

// Scratch of one launch of the path kernel: where the queue head and the per-item sums live.
struct RenderScratch {
  uint32_t* queue;
  float* sums;
};

// Work items of one call: ITEM_SAMPLES consecutive samples of one pixel, enumerated tile by tile
// (8x8 pixels) and sample block by sample block; one scratch image per sample block.
struct RenderShape {
  uint32_t tiles_x, item_samples, n_sblocks;
  uint64_t n_slots, n_items;
  size_t n_img, sums_bytes;
};

static int render_shape(const MiptParams* p, uint32_t sample_count, RenderShape& r) {
  const uint32_t tiles_y = (uint32_t)((p->height + 7) / 8);
  r.tiles_x = (uint32_t)((p->width + 7) / 8);
  r.n_slots = (uint64_t)r.tiles_x * tiles_y * 64u;
  r.item_samples = sample_count < ITEM_SAMPLES ? sample_count : ITEM_SAMPLES;
  r.n_sblocks = (sample_count + r.item_samples - 1) / r.item_samples;
  r.n_items = r.n_slots * r.n_sblocks;
  if (r.n_items >= 0xFF000000ull)
    return fail(MIPT_E_INVALID, "mipt_render_accumulate: %llu work items in one call; render fewer samples per call",
                (unsigned long long)r.n_items);
  r.n_img = (size_t)p->width * p->height * 3;
  r.sums_bytes = r.n_img * sizeof(float) * r.n_sblocks;
  return MIPT_OK;
}

static int render_check(const MiptScene* s, const MiptCamera* cam, const MiptLight* light, const MiptParams* p,
                        uint32_t sample_begin, uint32_t sample_count, const float* d_accum_rgb) {
  if (!s) return fail(MIPT_E_INVALID, "mipt_render_accumulate: scene is null");
  int rc = check_params("mipt_render_accumulate", cam, light, p);
  if (rc != MIPT_OK) return rc;
  if (!d_accum_rgb) return fail(MIPT_E_INVALID, "mipt_render_accumulate: accumulation buffer is null");
  if (sample_begin + sample_count < sample_begin)
    return fail(MIPT_E_INVALID, "mipt_render_accumulate: sample range wraps around 2^32");
  return MIPT_OK;
}

// The launches of one render call: queue head cleared, path kernel, ordered sum of the item sums
// into the caller's image. Allocates nothing; `timed` = bracket the path kernel with the scene's events.
static int render_launch(MiptScene* s, const MiptCamera* cam, const MiptLight* light, const MiptParams* p,
                         uint32_t sample_begin, uint32_t sample_count, float* d_accum_rgb, hipStream_t st,
                         const RenderShape& r, RenderScratch scr, bool timed) {
  PathArgs a;
  fill_args(a, s, cam, light, p);
  a.sample_begin = sample_begin;
  a.sample_count = sample_count;
  a.item_samples = r.item_samples;
  a.tiles_x = r.tiles_x;
  a.n_slots = (uint32_t)r.n_slots;
  a.n_items = (uint32_t)r.n_items;
  a.queue = scr.queue;
  a.partial = scr.sums;
  // enough resident waves to fill the chip at any occupancy; every lane loops over the queue
  const uint64_t batches = r.n_items / 64u;
  const bool gstack = vstack_kind(scene_kind(s));
  const unsigned max_grid = gstack ? VSTACK_BLOCKS : PERSISTENT_BLOCKS;
  const unsigned grid = (unsigned)(batches / 4u + 1u < max_grid ? batches / 4u + 1u : max_grid);
  if (gstack) {
    a.vstack = s->d_vstack;
    a.vstride = VSTACK_BLOCKS * BLOCK;
    a.vplane = 2u * (uint32_t)(p->max_depth + 1) * a.vstride;
  }
  HIP_TRY(hipMemsetAsync(scr.queue, 0, sizeof(uint32_t), st));
  const bool stats = (p->flags & MIPT_FLAG_STATS) != 0;
  if (stats) HIP_TRY(hipMemsetAsync(s->d_stats, 0, (ST_COUNT + PF_COUNT) * sizeof(unsigned long long), st));
#ifdef MIPT_PROFILE_SECTIONS
  HIP_TRY(hipMemsetAsync(s->d_stats + ST_COUNT, 0, PF_COUNT * sizeof(unsigned long long), st));
#endif
  if (timed) HIP_TRY(hipEventRecord(s->ev0, st));
  const size_t lds = lds_bytes(s, p, a);
  launch_paths<false>(s, a, dim3(grid), lds, st);
  HIP_TRY(hipGetLastError());
  if (timed) HIP_TRY(hipEventRecord(s->ev1, st));
  if (d_accum_rgb) {  // null: the item sums stay in the scratch for mipt_fold_items_dev
    hipLaunchKernelGGL(k_reduce_chunks, dim3(grid_for(r.n_img)), dim3(BLOCK), 0, st, scr.sums, r.n_sblocks, r.n_img,
                       d_accum_rgb);
    HIP_TRY(hipGetLastError());
  }
  if (timed) {  // the entry point that allocates nothing leaves what mipt_get_stats reports alone
    s->timed = true;
    s->stats_valid = stats;
    s->last_paths = (uint64_t)p->width * p->height * sample_count;
  }
  return MIPT_OK;
}

// mipt_render_accumulate_dev, or (d_accum_rgb null) its first half: the item sums stay in the scene's scratch
static int render_dev(MiptScene* s, const MiptCamera* cam, const MiptLight* light, const MiptParams* p,
                      uint32_t sample_begin, uint32_t sample_count, float* d_accum_rgb, bool fold, void* stream) {
  static float not_null;  // render_check wants a buffer; the first half has none
  int rc = render_check(s, cam, light, p, sample_begin, sample_count, fold ? d_accum_rgb : &not_null);
  if (rc != MIPT_OK) return rc;
  if (s->items_pending)
    return fail(MIPT_E_INVALID, "mipt_render_*: the items of the last mipt_render_items_dev have not been folded");
  if (sample_count == 0) return MIPT_OK;
  HIP_TRY(hipSetDevice(s->device));
  hipStream_t st = (hipStream_t)stream;
  RenderShape r;
  rc = render_shape(p, sample_count, r);
  if (rc != MIPT_OK) return rc;
  if (r.sums_bytes > s->partial_bytes) {
    // the scratch image stack is kept with the scene and only ever grows; a launch still reading
    // the old one (on any stream) must have finished before it is freed
    if (s->busy_valid) HIP_TRY(hipEventSynchronize(s->ev_busy));
    HIP_TRY(hipStreamSynchronize(st));
    if (s->d_partial) { (void)hipFree(s->d_partial); s->d_partial = nullptr; s->partial_bytes = 0; }
    HIP_TRY(hipMalloc(&s->d_partial, r.sums_bytes));
    s->partial_bytes = r.sums_bytes;
  }
  rc = scratch_acquire(s, st);
  if (rc != MIPT_OK) return rc;
  rc = render_launch(s, cam, light, p, sample_begin, sample_count, fold ? d_accum_rgb : nullptr, st, r,
                     RenderScratch{s->d_queue, s->d_partial}, true);
  if (rc != MIPT_OK) return rc;
  if (!fold) { s->items_pending = true; s->items_sblocks = r.n_sblocks; s->items_n_img = r.n_img; }
  return scratch_release(s, st);
}

extern "C" int mipt_render_accumulate_dev(MiptScene* s, const MiptCamera* cam, const MiptLight* light,
                                          const MiptParams* p, uint32_t sample_begin, uint32_t sample_count,
                                          float* d_accum_rgb, void* stream) {
  return render_dev(s, cam, light, p, sample_begin, sample_count, d_accum_rgb, true, stream);
}

// The two halves of mipt_render_accumulate_dev. Between them the item sums sit in the scene's
// scratch; the fold may run on another stream (it waits for the render through the scene's event)
// and into a buffer that did not exist when the render began -- e.g. the image another GPU has just
// sent, which is how several GPUs produce the bits one GPU produces.
extern "C" int mipt_render_items_dev(MiptScene* s, const MiptCamera* cam, const MiptLight* light,
                                     const MiptParams* p, uint32_t sample_begin, uint32_t sample_count,
                                     void* stream) {
  return render_dev(s, cam, light, p, sample_begin, sample_count, nullptr, false, stream);
}

extern "C" int mipt_fold_items_dev(MiptScene* s, float* d_accum_rgb, void* stream) {
  if (!s) return fail(MIPT_E_INVALID, "mipt_fold_items_dev: scene is null");
  if (!d_accum_rgb) return fail(MIPT_E_INVALID, "mipt_fold_items_dev: accumulation buffer is null");
  if (!s->items_pending) return MIPT_OK;  // nothing rendered (a range of zero samples): the image stays as it is
  HIP_TRY(hipSetDevice(s->device));
  hipStream_t st = (hipStream_t)stream;
  int rc = scratch_acquire(s, st);
  if (rc != MIPT_OK) return rc;
  hipLaunchKernelGGL(k_reduce_chunks, dim3(grid_for(s->items_n_img)), dim3(BLOCK), 0, st, s->d_partial, s->items_sblocks,
                     s->items_n_img, d_accum_rgb);
  HIP_TRY(hipGetLastError());
  s->items_pending = false;
  return scratch_release(s, st);
}

// For the library's multi-GPU driver: forget the item sums of a mipt_render_items_dev that will not be
// folded (a later step of the same call failed), so that the scene accepts render calls again.
extern "C" void mipt_internal_discard_items(MiptScene* s) {
  if (s) s->items_pending = false;
}

// The same call for callers that must not allocate (code under HIP graph capture, tight serving
// loops): queue head and item sums live in the file-scope array `scratch` of this device, whose
// address noalloc_setup() fetched at scene creation. No hipMalloc / hipFree, no events.
extern "C" int mipt_render_accumulate_noalloc_dev(MiptScene* s, const MiptCamera* cam, const MiptLight* light,
                                                  const MiptParams* p, uint32_t sample_begin,
                                                  uint32_t sample_count, float* d_accum_rgb, void* stream) {
  int rc = render_check(s, cam, light, p, sample_begin, sample_count, d_accum_rgb);
  if (rc != MIPT_OK) return rc;
  if (p->flags & MIPT_FLAG_STATS)  // the counters are scene scratch, ordered by events this call does not use
    return fail(MIPT_E_INVALID, "mipt_render_accumulate_noalloc_dev: MIPT_FLAG_STATS is not available here");
  if (sample_count == 0) return MIPT_OK;
  RenderShape r;
  rc = render_shape(p, sample_count, r);
  if (rc != MIPT_OK) return rc;
  if (r.sums_bytes > NOALLOC_SUMS_BYTES)
    return fail(MIPT_E_INVALID,
                "mipt_render_accumulate_noalloc_dev: %d x %d pixels x %u samples need %zu bytes of item sums, the "
                "fixed scratch holds %zu (MIPT_NOALLOC_MAX_BYTES); render fewer samples per call",
                p->width, p->height, sample_count, r.sums_bytes, NOALLOC_SUMS_BYTES);
  unsigned char* base = (s->device >= 0 && s->device < NOALLOC_MAX_DEVICES) ? g_noalloc_ptr[s->device] : nullptr;
  if (!base) return fail(MIPT_E_UNSUPPORTED, "mipt_render_accumulate_noalloc_dev: no scratch address for device %d", s->device);
  HIP_TRY(hipSetDevice(s->device));
  return render_launch(s, cam, light, p, sample_begin, sample_count, d_accum_rgb, (hipStream_t)stream, r,
                       RenderScratch{reinterpret_cast<uint32_t*>(base),
                                     reinterpret_cast<float*>(base + NOALLOC_QUEUE_BYTES)},
                       false);
}

extern "C" int mipt_render_accumulate(MiptScene* s, const MiptCamera* cam, const MiptLight* light,
                                      const MiptParams* p, uint32_t sample_begin, uint32_t sample_count,
                                      float* accum_rgb) {
  if (!s) return fail(MIPT_E_INVALID, "mipt_render_accumulate: scene is null");
  int rc = check_params("mipt_render_accumulate", cam, light, p);
  if (rc != MIPT_OK) return rc;
  if (!accum_rgb) return fail(MIPT_E_INVALID, "mipt_render_accumulate: accumulation buffer is null");
  HIP_TRY(hipSetDevice(s->device));
  const size_t bytes = (size_t)p->width * p->height * 3 * sizeof(float);
  rc = stage_reserve(s, bytes);
  if (rc != MIPT_OK) return rc;
  float* acc = static_cast<float*>(s->d_stage);
  HIP_TRY(hipMemcpy(acc, accum_rgb, bytes, hipMemcpyHostToDevice));
  rc = mipt_render_accumulate_dev(s, cam, light, p, sample_begin, sample_count, acc, nullptr);
  if (rc != MIPT_OK) return rc;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(accum_rgb, acc, bytes, hipMemcpyDeviceToHost));
  return MIPT_OK;
}

extern "C" int mipt_trace_paths(MiptScene* s, const MiptCamera* cam, const MiptLight* light, const MiptParams* p,
                                const int32_t* px, const int32_t* py, const uint32_t* sample, size_t n_paths,
                                float* rgb, uint32_t* ray_counts) {
  if (!s) return fail(MIPT_E_INVALID, "mipt_trace_paths: scene is null");
  int rc = check_params("mipt_trace_paths", cam, light, p);
  if (rc != MIPT_OK) return rc;
  if (n_paths == 0) return MIPT_OK;
  if (!px || !py || !sample || !rgb) return fail(MIPT_E_INVALID, "mipt_trace_paths: null buffer");
  if (n_paths > 0x7FFFFFFFu) return fail(MIPT_E_INVALID, "mipt_trace_paths: too many paths");
  for (size_t i = 0; i < n_paths; ++i)
    if (px[i] < 0 || px[i] >= p->width || py[i] < 0 || py[i] >= p->height)
      return fail(MIPT_E_INVALID, "mipt_trace_paths: path %zu pixel (%d, %d) outside the image", i, px[i], py[i]);
  HIP_TRY(hipSetDevice(s->device));
  DevBuf dx, dy, ds, drgb, dcnt, dvs;
  HIP_TRY(dx.alloc(n_paths * 4)); HIP_TRY(dy.alloc(n_paths * 4)); HIP_TRY(ds.alloc(n_paths * 4));
  HIP_TRY(drgb.alloc(n_paths * 12)); HIP_TRY(dcnt.alloc(n_paths * 8));
  HIP_TRY(hipMemcpy(dx.p, px, n_paths * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dy.p, py, n_paths * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(ds.p, sample, n_paths * 4, hipMemcpyHostToDevice));
  PathArgs a;
  fill_args(a, s, cam, light, p);
  a.px = dx.as<int32_t>(); a.py = dy.as<int32_t>(); a.samp = ds.as<uint32_t>();
  a.n_paths = (uint32_t)n_paths;
  a.rgb = drgb.as<float>();
  a.counts = dcnt.as<uint32_t>();
  if (vstack_kind(scene_kind(s))) {
    // one lane per listed path: the scene's vertex stacks if they are large enough, else a temporary
    const size_t lanes = (size_t)grid_for(n_paths) * BLOCK;
    if (lanes > 0x7FFFFFFFull / (2u * (MIPT_MAX_DEPTH_LIMIT + 1))) return fail(MIPT_E_INVALID, "mipt_trace_paths: too many paths");
    a.vstack = s->d_vstack;
    if (lanes > (size_t)VSTACK_BLOCKS * BLOCK) {
      HIP_TRY(dvs.alloc(vstack_bytes(lanes, (size_t)p->max_depth)));
      a.vstack = dvs.as<float4>();
    }
    a.vstride = (uint32_t)lanes;
    a.vplane = 2u * (uint32_t)(p->max_depth + 1) * a.vstride;
  }
  const bool stats = (p->flags & MIPT_FLAG_STATS) != 0;
  if (s->busy_valid) HIP_TRY(hipEventSynchronize(s->ev_busy));  // the counters are scene scratch too
  if (stats) HIP_TRY(hipMemset(s->d_stats, 0, (ST_COUNT + PF_COUNT) * sizeof(unsigned long long)));
  HIP_TRY(hipEventRecord(s->ev0, nullptr));
  const size_t lds = lds_bytes(s, p, a);
  launch_paths<true>(s, a, dim3(grid_for(n_paths)), lds, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(s->ev1, nullptr));
  HIP_TRY(hipDeviceSynchronize());
  s->timed = true;
  s->stats_valid = stats;
  s->last_paths = n_paths;
  HIP_TRY(hipMemcpy(rgb, drgb.p, n_paths * 12, hipMemcpyDeviceToHost));
  if (ray_counts) HIP_TRY(hipMemcpy(ray_counts, dcnt.p, n_paths * 8, hipMemcpyDeviceToHost));
  return MIPT_OK;
}

extern "C" int mipt_get_stats(MiptScene* s, MiptStats* out) {
  if (!s || !out) return fail(MIPT_E_INVALID, "mipt_get_stats: null argument");
  memset(out, 0, sizeof *out);
  if (!s->timed) return fail(MIPT_E_INVALID, "mipt_get_stats: no render or trace call has run on this scene");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipEventSynchronize(s->ev1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
  out->kernel_ms = ms;
  out->paths = s->last_paths;
#ifdef MIPT_PROFILE_SECTIONS
  {
    unsigned long long pf[PF_COUNT];
    HIP_TRY(hipMemcpy(pf, s->d_stats + ST_COUNT, sizeof pf, hipMemcpyDeviceToHost));
    static const char* names[PF_N_QUERIES] = {"queue", "raygen", "phase1", "phase2", "hit", "terminal", "scatter",
                                          "diffuse", "mirror", "glass", "renorm", "unwind"};
    double tot = 0;
    for (int k = 0; k < PF_N_QUERIES; ++k) tot += (double)pf[k];
    fprintf(stderr, "[mipt sections, %% of wave time]");
    for (int k = 0; k < PF_N_QUERIES; ++k) fprintf(stderr, " %s %.1f", names[k], 100.0 * (double)pf[k] / (tot > 0 ? tot : 1));
    fprintf(stderr, " (total %.3g ticks); candidates per query %.2f, phase-2 rounds per wave query %.2f\n", tot,
            (double)pf[PF_N_CANDS] / (double)(pf[PF_N_QUERIES] ? pf[PF_N_QUERIES] : 1),
            64.0 * (double)pf[PF_N_ROUNDS] / (double)(pf[PF_N_QUERIES] ? pf[PF_N_QUERIES] : 1));
    fprintf(stderr, "[mipt sections] completion: %.3f flushes per iteration, %.2f jobs per flush, %.2f vertices per job\n",
            (double)pf[PF_N_FLUSH] / (double)(pf[PF_N_ITERS] ? pf[PF_N_ITERS] : 1),
            (double)pf[PF_N_JOBS] / (double)(pf[PF_N_FLUSH] ? pf[PF_N_FLUSH] : 1),
            (double)pf[PF_N_JOB_LEVELS] / (double)(pf[PF_N_JOBS] ? pf[PF_N_JOBS] : 1));
    if (pf[PF_W_BURSTS])
      fprintf(stderr, "[mipt sections] tree walk: %.2f bursts per iteration, %.1f lanes entering; per burst %.2f node rounds at %.1f lanes, "
                      "%.2f leaf rounds at %.1f lanes; per iteration %.1f lanes begin a ray, %.1f are shaded\n",
              (double)pf[PF_W_BURSTS] / (double)(pf[PF_N_ITERS] ? pf[PF_N_ITERS] : 1),
              (double)pf[PF_W_ENTERED] / (double)pf[PF_W_BURSTS],
              (double)pf[PF_W_NODE_ROUNDS] / (double)pf[PF_W_BURSTS],
              (double)pf[PF_W_NODE_LANES] / (double)(pf[PF_W_NODE_ROUNDS] ? pf[PF_W_NODE_ROUNDS] : 1),
              (double)pf[PF_W_LEAF_ROUNDS] / (double)pf[PF_W_BURSTS],
              (double)pf[PF_W_LEAF_LANES] / (double)(pf[PF_W_LEAF_ROUNDS] ? pf[PF_W_LEAF_ROUNDS] : 1),
              (double)pf[PF_N_FRESH] / (double)(pf[PF_N_ITERS] ? pf[PF_N_ITERS] : 1),
              (double)pf[PF_N_READY] / (double)(pf[PF_N_ITERS] ? pf[PF_N_ITERS] : 1));
    fprintf(stderr, "[mipt sections] queries with >= 3 / 4 / 6 candidates: %.2f %% / %.2f %% / %.3f %%\n",
            100.0 * (double)pf[PF_N_GE3] / (double)(pf[PF_N_QUERIES] ? pf[PF_N_QUERIES] : 1),
            100.0 * (double)pf[PF_N_GE4] / (double)(pf[PF_N_QUERIES] ? pf[PF_N_QUERIES] : 1),
            100.0 * (double)pf[PF_N_GE6] / (double)(pf[PF_N_QUERIES] ? pf[PF_N_QUERIES] : 1));
  }
#endif
  if (s->stats_valid) {
    unsigned long long h[ST_COUNT];
    HIP_TRY(hipMemcpy(h, s->d_stats, sizeof h, hipMemcpyDeviceToHost));
    out->paths = h[ST_PATHS];
    out->closest_rays = h[ST_CLOSEST]; out->shadow_rays = h[ST_SHADOW];
    out->tri_tests = h[ST_TESTS]; out->gate_passes = h[ST_GATE];
    out->sphere_tests = h[ST_SPHERE]; out->fog_events = h[ST_FOG];
    out->diffuse_bounces = h[ST_DIFFUSE]; out->mirror_bounces = h[ST_MIRROR]; out->glass_bounces = h[ST_GLASS];
    out->bvh_redo = h[ST_BVH_REDO]; out->bvh_overflow = h[ST_BVH_OVERFLOW]; out->bvh_nodes = h[ST_BVH_NODES];
    out->specular_bounces = h[ST_SPECULAR];
  }
  return MIPT_OK;
}

// ---------------------------------------------------------------------------------------
// resolve

extern "C" int mipt_resolve_argb_dev(MiptScene* s, const float* d_accum_rgb, int width, int height,
                                     uint32_t n_samples, uint32_t* d_screen, void* stream) {
  if (!s) return fail(MIPT_E_INVALID, "mipt_resolve_argb: scene is null");
  if (!d_accum_rgb || !d_screen) return fail(MIPT_E_INVALID, "mipt_resolve_argb: null buffer");
  if (width <= 0 || height <= 0 || width > 32768 || height > 32768)
    return fail(MIPT_E_INVALID, "mipt_resolve_argb: image size %d x %d out of range", width, height);
  if (n_samples > 0x7FFFFFFFu) return fail(MIPT_E_INVALID, "mipt_resolve_argb: sample count exceeds int range");
  HIP_TRY(hipSetDevice(s->device));
  hipLaunchKernelGGL(k_resolve_argb, dim3(grid_for((size_t)width * height)), dim3(BLOCK), 0, (hipStream_t)stream,
                     d_accum_rgb, width, height, n_samples, d_screen);
  HIP_TRY(hipGetLastError());
  return MIPT_OK;
}

extern "C" int mipt_resolve_argb(MiptScene* s, const float* accum_rgb, int width, int height, uint32_t n_samples,
                                 uint32_t* screen_buffer) {
  if (!s) return fail(MIPT_E_INVALID, "mipt_resolve_argb: scene is null");
  if (!accum_rgb || !screen_buffer) return fail(MIPT_E_INVALID, "mipt_resolve_argb: null buffer");
  if (width <= 0 || height <= 0 || width > 32768 || height > 32768)
    return fail(MIPT_E_INVALID, "mipt_resolve_argb: image size %d x %d out of range", width, height);
  HIP_TRY(hipSetDevice(s->device));
  const size_t n = (size_t)width * height;
  DevBuf acc, scr;
  HIP_TRY(acc.alloc(n * 12)); HIP_TRY(scr.alloc(n * 4));
  HIP_TRY(hipMemcpy(acc.p, accum_rgb, n * 12, hipMemcpyHostToDevice));
  int rc = mipt_resolve_argb_dev(s, acc.as<float>(), width, height, n_samples, scr.as<uint32_t>(), nullptr);
  if (rc != MIPT_OK) return rc;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(screen_buffer, scr.p, n * 4, hipMemcpyDeviceToHost));
  return MIPT_OK;
}

// ---------------------------------------------------------------------------------------
// progressive frame (device-resident pixels + samplesSeenSoFar)

struct MiptFrame {
  MiptScene* scene = nullptr;
  int device = 0;  // the scene's device, kept here so that destroying the frame never reads the scene
  int width = 0, height = 0;
  uint32_t samples = 0;
  float* d_accum = nullptr;
  uint32_t* d_screen = nullptr;
  hipStream_t stream = nullptr;
};

extern "C" void mipt_frame_destroy(MiptFrame* f) {
  if (!f) return;
  (void)hipSetDevice(f->device);
  if (f->stream) { (void)hipStreamSynchronize(f->stream); (void)hipStreamDestroy(f->stream); }
  if (f->d_accum) (void)hipFree(f->d_accum);
  if (f->d_screen) (void)hipFree(f->d_screen);
  (void)hipGetLastError();
  delete f;
}

static int frame_create_impl(MiptFrame* f) {
  const size_t n = (size_t)f->width * f->height;
  HIP_TRY(hipSetDevice(f->scene->device));
  HIP_TRY(hipStreamCreate(&f->stream));
  HIP_TRY(hipMalloc(&f->d_accum, n * 12));
  HIP_TRY(hipMalloc(&f->d_screen, n * 4));
  HIP_TRY(hipMemsetAsync(f->d_accum, 0, n * 12, f->stream));
  return MIPT_OK;
}

extern "C" int mipt_frame_create(MiptScene* scene, int width, int height, MiptFrame** out) {
  if (!out) return fail(MIPT_E_INVALID, "mipt_frame_create: out is null");
  *out = nullptr;
  if (!scene) return fail(MIPT_E_INVALID, "mipt_frame_create: scene is null");
  if (width <= 0 || height <= 0 || width > 32768 || height > 32768)
    return fail(MIPT_E_INVALID, "mipt_frame_create: image size %d x %d out of range", width, height);
  MiptFrame* f = new (std::nothrow) MiptFrame();
  if (!f) return fail(MIPT_E_NOMEM, "mipt_frame_create: host allocation failed");
  f->scene = scene; f->device = scene->device; f->width = width; f->height = height;
  const int rc = frame_create_impl(f);
  if (rc != MIPT_OK) { mipt_frame_destroy(f); return rc; }
  *out = f;
  return MIPT_OK;
}

extern "C" int mipt_frame_reset(MiptFrame* f) {
  if (!f) return fail(MIPT_E_INVALID, "mipt_frame_reset: frame is null");
  HIP_TRY(hipSetDevice(f->scene->device));
  HIP_TRY(hipMemsetAsync(f->d_accum, 0, (size_t)f->width * f->height * 12, f->stream));
  f->samples = 0;
  return MIPT_OK;
}

extern "C" int mipt_frame_add_samples(MiptFrame* f, const MiptCamera* cam, const MiptLight* light,
                                      const MiptParams* params, uint32_t n_samples) {
  if (!f || !params) return fail(MIPT_E_INVALID, "mipt_frame_add_samples: null argument");
  if (n_samples == 0) return MIPT_OK;
  if ((uint64_t)f->samples + n_samples > 0x7FFFFFFFull)
    return fail(MIPT_E_INVALID, "mipt_frame_add_samples: sample count exceeds int range");
  MiptParams p = *params;
  p.width = f->width; p.height = f->height;
  const int rc = mipt_render_accumulate_dev(f->scene, cam, light, &p, f->samples, n_samples, f->d_accum, f->stream);
  if (rc != MIPT_OK) return rc;
  f->samples += n_samples;
  return MIPT_OK;
}

extern "C" int mipt_frame_present(MiptFrame* f, uint32_t* screen_buffer) {
  if (!f || !screen_buffer) return fail(MIPT_E_INVALID, "mipt_frame_present: null argument");
  const int rc = mipt_resolve_argb_dev(f->scene, f->d_accum, f->width, f->height, f->samples, f->d_screen, f->stream);
  if (rc != MIPT_OK) return rc;
  HIP_TRY(hipMemcpyAsync(screen_buffer, f->d_screen, (size_t)f->width * f->height * 4, hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(hipStreamSynchronize(f->stream));
  return MIPT_OK;
}

extern "C" int mipt_frame_read_pixels(MiptFrame* f, float* accum_rgb) {
  if (!f || !accum_rgb) return fail(MIPT_E_INVALID, "mipt_frame_read_pixels: null argument");
  HIP_TRY(hipSetDevice(f->scene->device));
  HIP_TRY(hipMemcpyAsync(accum_rgb, f->d_accum, (size_t)f->width * f->height * 12, hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(hipStreamSynchronize(f->stream));
  return MIPT_OK;
}

extern "C" int mipt_frame_samples(const MiptFrame* f) { return f ? (int)f->samples : 0; }

extern "C" int mipt_frame_poll(MiptFrame* f) {
  if (!f) return fail(MIPT_E_INVALID, "mipt_frame_poll: frame is null");
  HIP_TRY(hipSetDevice(f->device));
  const hipError_t e = hipStreamQuery(f->stream);
  if (e == hipSuccess) return 1;
  if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
  (void)hipGetLastError();
  return fail(MIPT_E_HIP, "hipStreamQuery failed: %s", hipGetErrorString(e));
}

extern "C" int mipt_debug_selfcheck(MiptScene* s, uint64_t n_rays, uint32_t seed, uint64_t out[8]) {
  if (!s || !out) return fail(MIPT_E_INVALID, "mipt_debug_selfcheck: null argument");
  if (n_rays > ((uint64_t)1 << 36)) return fail(MIPT_E_INVALID, "mipt_debug_selfcheck: at most 2^36 rays per call");
  HIP_TRY(hipSetDevice(s->device));
  DevBuf d;
  HIP_TRY(d.alloc(8 * sizeof(unsigned long long)));
  HIP_TRY(hipMemset(d.p, 0, 8 * sizeof(unsigned long long)));
  hipLaunchKernelGGL(k_selfcheck_rcp, dim3(4096 / BLOCK), dim3(BLOCK), 0, nullptr, d.as<unsigned long long>());
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_selfcheck_sincos, dim3(4096), dim3(BLOCK), 0, nullptr, d.as<unsigned long long>());
  HIP_TRY(hipGetLastError());
  // a launch holds fewer than 2^32 threads: 2^30 rays per launch, each with a seed of its own
  const uint64_t per_launch = (uint64_t)1 << 30;
  uint32_t launch_seed = seed;
  for (uint64_t done = 0; done < n_rays; done += per_launch, launch_seed += 0x9E3779B9u) {
    const uint64_t n = n_rays - done < per_launch ? n_rays - done : per_launch;
    const dim3 grid((unsigned)((n + BLOCK - 1) / BLOCK));
    if (s->d.bvh)
      hipLaunchKernelGGL(k_selfcheck_scan<true>, grid, dim3(BLOCK), 0, nullptr, s->d, (unsigned long long)n,
                         launch_seed, d.as<unsigned long long>());
    else
      hipLaunchKernelGGL(k_selfcheck_scan<false>, grid, dim3(BLOCK), 0, nullptr, s->d, (unsigned long long)n,
                         launch_seed, d.as<unsigned long long>());
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipDeviceSynchronize());
  unsigned long long h[8];
  HIP_TRY(hipMemcpy(h, d.p, sizeof h, hipMemcpyDeviceToHost));
  for (int i = 0; i < 8; ++i) out[i] = h[i];
  return MIPT_OK;
}

extern "C" uint32_t mipt_path_key(uint32_t seed, uint32_t pixel, uint32_t sample) {
  return mipt::path_key(seed, pixel, sample);
}
