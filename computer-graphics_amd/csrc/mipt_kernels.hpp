// mipt_kernels.hpp -- the gfx950 kernels of the path-tracing hot path.
//
//   k_closest_hit   batch closestIntersection      (intersections.cpp:6-55)
//   k_any_hit       batch existsIntersection       (intersections.cpp:57-85)
//   k_paths         Draw's pixel loop + trace      (main.cpp:121-139, pathtracer.cpp:5-111,
//                                                   light.cpp:13-31, utils.cpp:10-22)
//   k_reduce_chunks ordered sum of per-chunk partial images into the accumulation buffer
//   k_resolve_argb  Draw's tonemap loop            (main.cpp:141-145, SDLauxiliary.h:149-161)
//
// All of them are plain f32 VALU work on wave-uniform triangle data; nothing here is a dense
// contraction, so there is no MFMA. Compiled with -ffp-contract=off (see mipt_device.hpp).
#pragma once

#include "mipt_bvh.hpp"

namespace mipt {

// samples per work item of the path kernel, and the size of its persistent grid (256 CUs x 8)
#ifndef MIPT_ITEM_SAMPLES
#define MIPT_ITEM_SAMPLES 16
#endif
constexpr uint32_t ITEM_SAMPLES = MIPT_ITEM_SAMPLES;
constexpr uint32_t PERSISTENT_BLOCKS = 2048;

// Tags of the per-lane vertex stack: a primitive index (diffuse vertex), or one of two marks at the
// top of the tag type's range.
constexpr uint32_t TAG_GLASS = 0xFFFFFFFEu;
constexpr uint32_t TAG_FOG = 0xFFFFFFFDu;
constexpr uint32_t TERMINAL_IN_MEDIUM = 0xFFFFFFFFu;  // terminal record: the path ended on a fog event

struct PathArgs {
  DeviceScene sc;
  float M[16];  // cameraRotationX * cameraRotationY (main.cpp:133), column-major
  float cam_pos[4];
  float focal;
  float light_pos[4];
  float light_col[3];
  uint32_t light_gray;  // the three light channels are equal: one division instead of three
  // bounceGlass constants (pathtracer.cpp:30-39), evaluated once on the host with the same IEEE
  // operations: R0 = ((air - glass) / (air + glass))^2, 1/glass, 1/(1/glass)
  float glass_R0, eta_enter, eta_exit;
  MiptParams p;
  uint32_t sample_begin;
  uint32_t sample_count;
  uint32_t item_samples;  // samples per work item
  uint32_t tiles_x;       // 8x8 pixel tiles per image row
  uint32_t n_slots;       // tiles * 64 (pixel slots incl. padding of ragged image sizes)
  uint32_t n_items;       // n_slots * sample blocks
  uint32_t* queue;        // head of the work queue, zeroed before the launch
  float* partial;         // [sample block][W*H*3] per-item radiance sums
  // list mode (mipt_trace_paths)
  const int32_t* px;
  const int32_t* py;
  const uint32_t* samp;
  uint32_t n_paths;
  float* rgb;
  uint32_t* counts;
  // Scenes whose vertex stacks live in global memory (vstack_kind): two planes of 16-byte entries,
  // entry (buffer b, level k) of lane g at ((b * (max_depth + 1) + k) * vstride + g); plane 0 the
  // vertex's normal, plane 1 (vplane entries further) its colour and kind. Level max_depth of plane 1
  // holds the colour and material of the path's terminal vertex.
  float4* vstack;
  uint32_t vstride;        // lanes of the launch (grid * BLOCK)
  uint32_t vplane;         // 2 * (max_depth + 1) * vstride
  uint32_t lds_sph_off;    // dword offset of the sphere normals beside the vertex stack in dynamic LDS
  uint32_t lds_table_off;  // SCENE_LINEAR_LDS: dword offset of the record tables in dynamic LDS
  unsigned long long* stats;  // ST_COUNT device counters, STATS builds only
};

// ---------------------------------------------------------------------------------------

// Developer build (-DMIPT_PROFILE_SECTIONS): wall time of the path kernel's sections, summed over
// waves in units of the constant 100 MHz shader clock. One active lane per wave adds the time since
// the previous mark to the section's LDS counter; mipt_get_stats prints the totals.
enum ProfSection {
  PF_QUEUE = 0, PF_RAYGEN, PF_PHASE1, PF_PHASE2, PF_HIT, PF_TERMINAL, PF_SCATTER, PF_DIFFUSE, PF_MIRROR,
  PF_GLASS, PF_RENORM, PF_UNWIND, PF_N_QUERIES, PF_N_CANDS, PF_N_ROUNDS, PF_N_GE3, PF_N_GE4, PF_N_GE6,
  PF_N_ITERS, PF_N_FLUSH, PF_N_JOBS, PF_N_JOB_LEVELS,
  PF_W_BURSTS, PF_W_ENTERED, PF_W_NODE_ROUNDS, PF_W_NODE_LANES, PF_W_LEAF_ROUNDS, PF_W_LEAF_LANES,  // tree walk, in this order
  PF_N_READY, PF_N_FRESH, PF_COUNT
};
#ifdef MIPT_PROFILE_SECTIONS
struct Prof {
  uint32_t last;  // wave-uniform (kept in a scalar register): the time of the last mark the WAVE passed,
                  // whichever of its lanes were active there
  uint32_t* lds;
  __device__ __forceinline__ void start() { last = __builtin_amdgcn_readfirstlane((uint32_t)__builtin_readcyclecounter()); }
  __device__ __forceinline__ void mark(int k) {
    const uint32_t now = __builtin_amdgcn_readfirstlane((uint32_t)__builtin_readcyclecounter());
    if (lds && (int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) atomicAdd(lds + k, now - last);
    last = now;
  }
  __device__ __forceinline__ void count(int k, uint32_t n) { if (lds && n) atomicAdd(lds + k, n); }
  __device__ __forceinline__ void count_wave(int k) {
    if (lds && (int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) atomicAdd(lds + k, 1u);
  }
};
#elif defined(MIPT_ASM_SECTIONS)
// Developer build: the section boundaries as comments in the ISA listing (`make asm EXTRA=-DMIPT_ASM_SECTIONS`)
struct Prof {
  uint32_t* lds;
  __device__ __forceinline__ void start() {}
  __device__ __forceinline__ void mark(int k) { asm volatile("; ==SEC %0" ::"i"(k) : "memory"); }
  __device__ __forceinline__ void count(int, uint32_t) {}
  __device__ __forceinline__ void count_wave(int) {}
};
#else
struct Prof {
  uint32_t* lds;
  __device__ __forceinline__ void start() {}
  __device__ __forceinline__ void mark(int) {}
  __device__ __forceinline__ void count(int, uint32_t) {}
  __device__ __forceinline__ void count_wave(int) {}
};
#endif

// The linear scan of closestIntersection / existsIntersection, three ways with identical results:
//   SCAN_LITERAL    the reference's arithmetic for every triangle, in index order;
//   SCAN_FAST       the same with the estimate-based gate and rcp_exact (tri_test<FAST>);
//   SCAN_TWO_PHASE  phase 1 runs the cheap candidate filter over all triangles (wave-uniform
//                   records in SGPRs, every lane busy) and leaves each lane a bit mask of the few
//                   triangles its ray can possibly hit; phase 2 lets every lane run the full test
//                   on ITS OWN candidates, in index order. Triangles the filter rejects could not
//                   have been accepted, and rejected tests never change the scan's state, so the
//                   sequence of accepted triangles -- tie rule included -- is unchanged.
enum ScanMode { SCAN_LITERAL = 0, SCAN_FAST = 1, SCAN_TWO_PHASE = 2 };

// Kinds of scene the path kernel is instantiated for: linear scan with the triangle and shading
// records staged in LDS (scenes of at most LDS_TABLE_MAX primitives), linear scan from global
// memory, LBVH.
enum SceneKind { SCENE_LINEAR_LDS = 0, SCENE_LINEAR = 1, SCENE_BVH = 2, SCENE_LINEAR_BIG = 3 };
constexpr int LDS_TABLE_MAX = 64;
// The vertex stack holds primitive indices: a byte each for the scenes whose tables fit LDS (at most
// 64 primitives), 16 bits for linear scenes of fewer than TAG16_MAX_PRIMS primitives (the two marks
// TAG_GLASS / TAG_FOG sit at the top of the 16-bit range), 32 bits for everything else: tree
// scenes, and the linear scan forced onto a large scene (SCENE_LINEAR_BIG).
constexpr int TAG16_MAX_PRIMS = 0xFFFD;
template <int KIND> struct TagType { typedef uint32_t T; };
template <> struct TagType<SCENE_LINEAR_LDS> { typedef uint8_t T; };
template <> struct TagType<SCENE_LINEAR> { typedef uint16_t T; };
// Linear-scan scenes without spheres finish their lit paths in batches (see k_paths): two vertex
// stacks, two terminal records and two radiance sums per lane.
// The linear scan forced onto a scene too large for 16-bit tags (SCENE_LINEAR_BIG) keeps its vertex
// stacks in global memory as RECORDS -- the vertex's normal, colour and kind, 32 bytes -- so the
// completion needs no index into the scene and no second stack of sphere normals, and the LDS holds
// two terminal records and two sums per lane whatever the scene's size: such scenes always finish in
// batches. Tree scenes can be built the same way (-DMIPT_TREE_VSTACK=1) and were measured so: the
// batched completion saves them at most 5 % (the bound: a build that skips the completion altogether)
// and the stores and loads of the records cost as much -- 0 ... -8 % paths/s on four scenes against
// 32-bit tags in LDS with every lit path completed at once, which is what they keep.
#ifndef MIPT_TREE_VSTACK
#define MIPT_TREE_VSTACK 0
#endif
__host__ __device__ inline bool vstack_kind(int kind) { return (MIPT_TREE_VSTACK && kind == SCENE_BVH) || kind == SCENE_LINEAR_BIG; }
__host__ __device__ inline bool paths_deferred(int kind, int n_spheres) {
  return vstack_kind(kind) || (kind != SCENE_BVH && n_spheres == 0);
}
constexpr uint32_t VK_DIFFUSE = 0u, VK_GLASS = 1u, VK_FOG = 2u;  // kinds of a vertex record

#ifndef MIPT_SCAN_MODE
#define MIPT_SCAN_MODE 2
#endif

template <bool STATS, int MODE = (STATS ? SCAN_FAST : MIPT_SCAN_MODE)>
__device__ __forceinline__ void scan_triangles(const DeviceScene& sc, V4 o, V4 d, bool shadow, HitAcc& h,
                                               uint32_t& n_tests, uint32_t& n_gate, Prof& prof,
                                               const TriRec* lane_tris) {
  const TriRec MIPT_CONST* tris = as_constant(sc.tris);
  const float px = -d.x, py = -d.y, pz = -d.z;  // vec3(-ray.direction), intersections.cpp:20
  const int n = sc.n_tris;
  if (MODE != SCAN_TWO_PHASE) {
    for (int i = 0; i < n; ++i)
      tri_test<STATS, MODE == SCAN_FAST>(tris + i, i, o.x, o.y, o.z, px, py, pz, shadow, h, n_tests, n_gate);
    return;
  }
  float Dm = fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fabsf(d.z));
  const float Om = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
  if (!(Dm >= 9.094947e-13f && Dm <= 1.0995116e12f && Om <= 1.0995116e12f)) Dm = __builtin_inff();
  for (int base = 0; base < n; base += 32) {
    uint32_t mask = unit_mask(sc.units + base, o, d, Dm);
#ifdef MIPT_EXP_ONE_CAND
    mask &= 0u - mask;  // timing experiment only: wrong results
#endif
    prof.mark(PF_PHASE1);
    prof.count(PF_N_QUERIES, 1u);
    prof.count(PF_N_CANDS, (uint32_t)__popc(mask));
    prof.count(PF_N_GE3, __popc(mask) >= 3 ? 1u : 0u);
    prof.count(PF_N_GE4, __popc(mask) >= 4 ? 1u : 0u);
    prof.count(PF_N_GE6, __popc(mask) >= 6 ? 1u : 0u);
    // (a do-while: as `while (__any(...))` the compiler also copied the state at the loop's head)
    if (__any(mask != 0)) do {
      prof.count_wave(PF_N_ROUNDS);
      // The scan state passes through an empty asm statement at the head of every round: without
      // it the compiler keeps two copies of the six state registers across the divergent test
      // and moves them back and forth at every join (31 moves per round, 3 with it and the
      // do-while: +4 % paths/s).
      asm volatile("" : "+v"(h.lim), "+v"(h.lim_lo), "+v"(h.lim_hi), "+v"(h.idx), "+v"(h.by), "+v"(h.bz), "+v"(mask));
      if (mask != 0) {
        const int i = base + __ffs((int)mask) - 1;
        mask &= mask - 1u;
        // each lane fetches ITS candidate record (from the caller's table: global memory, or LDS)
        const TriRec T = lane_tris[i];
        // existsIntersection is an OR: an accepted shadow ray clears the lane's mask
        tri_test<false, true>(&T, i, o.x, o.y, o.z, px, py, pz, shadow, h, n_tests, n_gate, &mask);
      }
    } while (__any(mask != 0));
    prof.mark(PF_PHASE2);
  }
}

// One closestIntersection / existsIntersection query against the scene: the linear scan for
// small scenes, the LBVH traversal (mipt_bvh.hpp) when the scene carries a tree. `redo` counts
// rays sent through the exhaustive traversal, `overflow` rays whose stack or hit list was too
// small even there (must stay 0).
template <bool STATS, bool BVH>
__device__ __forceinline__ void scene_query(const DeviceScene& sc, V4 o, V4 d, bool shadow, HitAcc& h,
                                            uint32_t& n_tests, uint32_t& n_gate, uint32_t& redo,
                                            uint32_t& overflow, Prof& prof, const TriRec* lane_tris,
                                            uint32_t* n_nodes = nullptr) {
  if (!BVH) {  // kernels are instantiated per kind of scene: the linear ones carry no traversal stack
    scan_triangles<STATS>(sc, o, d, shadow, h, n_tests, n_gate, prof, lane_tris);
    return;
  }
  // blocking form of the resumable walk: every lane stays until its own ray is done
  BvhRay r = bvh_ray(sc.scale, o, d);
  BvhTrav T;
  BvhMem M;
  bvh_begin(T, M, &sc, r, shadow, h.lim);
  while (T.cur != BVH_DONE) bvh_burst(T, M, StackScratch{M.stack}, &sc, sc.bvh, r, shadow, h.lim, overflow);
  if (T.redo) {
    redo++;
    bvh_closest_exhaustive(&sc, sc.bvh, r, h, overflow);
  } else {
    bvh_finish(T, M, shadow, h);
  }
  if (STATS) {  // tree scenes: tests = leaves whose triangle was looked at, nodes = inner nodes fetched
    n_tests += r.n_leaves;
    if (n_nodes) *n_nodes += r.n_nodes;
  }
}

template <bool BVH>
__global__ __launch_bounds__(BLOCK) void k_closest_hit(DeviceScene sc, const float4* __restrict__ origin,
                                                       const float4* __restrict__ dir, size_t n_rays,
                                                       int32_t* __restrict__ tri_idx, float* __restrict__ t_out,
                                                       float4* __restrict__ pos_out) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n_rays) return;
  const float4 of = origin[i], df = dir[i];
  const V4 o = {of.x, of.y, of.z, of.w}, d = {df.x, df.y, df.z, df.w};
  HitAcc h;
  h.set_lim(3.402823466e+38f);
  h.idx = -1; h.by = 0.f; h.bz = 0.f;
  uint32_t n_tests = 0, n_gate = 0, redo = 0, overflow = 0;
  Prof prof{};
  scene_query<false, BVH>(sc, o, d, false, h, n_tests, n_gate, redo, overflow, prof, sc.tris);
  V4 pos = {0, 0, 0, 0}, nrm;
  bool sphere_hit = false;
  for (int s = 0; s < sc.n_spheres; ++s) {
    if (sphere_test(sc.spheres[s], o, d, h.lim, pos, nrm)) {
      h.idx = sc.n_tris + s;
      sphere_hit = true;
    }
  }
  tri_idx[i] = h.idx;
  if (h.idx < 0) return;
  t_out[i] = h.lim;
  if (pos_out) {
    if (!sphere_hit) {
      // position = v0 + x.y*e1 + x.z*e2 (intersections.cpp:29), w = v0.w + x.y*0 + x.z*0
      const TriRec T = sc.tris[h.idx];
      pos.x = (T.v0x + h.by * T.e1x) + h.bz * T.e2x;
      pos.y = (T.v0y + h.by * T.e1y) + h.bz * T.e2y;
      pos.z = (T.v0z + h.by * T.e1z) + h.bz * T.e2z;
      pos.w = (T.v0w + h.by * 0.0f) + h.bz * 0.0f;
    }
    pos_out[i] = make_float4(pos.x, pos.y, pos.z, pos.w);
  }
}

template <bool BVH>
__global__ __launch_bounds__(BLOCK) void k_any_hit(DeviceScene sc, const float4* __restrict__ origin,
                                                   const float4* __restrict__ dir,
                                                   const float* __restrict__ max_t, size_t n_rays,
                                                   uint8_t* __restrict__ occluded) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n_rays) return;
  const float4 of = origin[i], df = dir[i];
  const V4 o = {of.x, of.y, of.z, of.w}, d = {df.x, df.y, df.z, df.w};
  HitAcc h;
  h.set_lim(max_t[i]);
  h.idx = 0; h.by = 0.f; h.bz = 0.f;
  uint32_t n_tests = 0, n_gate = 0, redo = 0, overflow = 0;
  Prof prof{};
  scene_query<false, BVH>(sc, o, d, true, h, n_tests, n_gate, redo, overflow, prof, sc.tris);
  occluded[i] = (uint8_t)h.idx;
}

// Batch queries against a tree scene: persistent lanes that draw rays from a queue, so a lane whose
// walk is over takes the next ray while its neighbours finish theirs (bvh_burst returns once
// most of the wave is done). SHADOW = existsIntersection, else closestIntersection incl. the spheres.
#ifndef MIPT_BATCH_RESERVE
#define MIPT_BATCH_RESERVE 128
#endif
constexpr unsigned long long BATCH_RESERVE = MIPT_BATCH_RESERVE;
#ifndef MIPT_BATCH_HEADS
#define MIPT_BATCH_HEADS 16
#endif
constexpr int BATCH_HEADS = MIPT_BATCH_HEADS;  // queue heads of the batch kernels, 128 bytes apart

// Waves per SIMD of the batch kernels, measured: the any-hit walk is best at 7 (8 spills); the
// closest-hit walk carries its nearest hit in registers and is best at 6 (at 7 it spills: -15 %).
#ifndef MIPT_BATCH_WAVES
#define MIPT_BATCH_WAVES 7
#endif
#ifndef MIPT_BATCH_WAVES_CLOSEST
#define MIPT_BATCH_WAVES_CLOSEST 6
#endif
#define MIPT_BATCH_OCC                                                                                    \
  __attribute__((amdgpu_waves_per_eu(SHADOW ? MIPT_BATCH_WAVES : MIPT_BATCH_WAVES_CLOSEST, \
                                     SHADOW ? MIPT_BATCH_WAVES : MIPT_BATCH_WAVES_CLOSEST)))

// Large incoherent batches walk the tree faster when rays that start in the same part of the scene
// travel in the same wave (the top of the tree is fetched once per wave instead of once per lane:
// 1.43 -> 1.81 Grays/s on 3840 x 2160 random rays against 10^6 triangles). The batch entry points
// bin the rays by the cell of their origin on a 16^3 grid over the scene (a 12-bit Morton key, one
// radix sort of (key, ray index) pairs) and hand the walk the sorted indices; every ray's result
// still lands at its own index. n_breaks counts neighbouring rays whose keys differ: a batch
// that is coherent as it comes (camera rays) is left alone.
__global__ __launch_bounds__(BLOCK) void k_ray_cell_keys(const float4* __restrict__ origin, unsigned long long n_rays,
                                                         float scale, uint32_t* __restrict__ keys,
                                                         uint32_t* __restrict__ vals, unsigned int* __restrict__ n_breaks) {
  const unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
  uint32_t key = 0;
  if (i < n_rays) {
    const float4 o = origin[i];
    const float k = scale > 0.f ? 8.0f / scale : 0.f;
    const float c[3] = {(o.x + scale) * k, (o.y + scale) * k, (o.z + scale) * k};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float f = fminf(fmaxf(c[a], 0.f), 15.f);  // a NaN coordinate lands in cell 0
      uint32_t q = (uint32_t)f;
      q = (q | (q << 4)) & 0x0C3u;   // 4 bits -> every third bit
      q = (q | (q << 2)) & 0x249u;
      key |= q << a;
    }
    keys[i] = key;
    vals[i] = (uint32_t)i;
  }
  // the probe (when asked for) looks at one workgroup in 64 (an atomic per wave of the whole batch
  // would cost more than the sort)
  if (!n_breaks || (blockIdx.x & 63u) != 0u) return;
  const uint32_t prev = (uint32_t)__shfl_up((int)key, 1, 64);
  const bool brk = i < n_rays && (threadIdx.x & 63) != 0 && prev != key;
  const unsigned long long b = __ballot(brk);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_breaks, (unsigned int)__popcll(b));
}

// the rays in walking order: a streaming pass (the walk itself would wait twice per ray for the
// indirection)
__global__ __launch_bounds__(BLOCK) void k_gather_rays(const float4* __restrict__ origin, const float4* __restrict__ dir,
                                                       const float* __restrict__ max_t,
                                                       const uint32_t* __restrict__ order, unsigned long long n_rays,
                                                       float4* __restrict__ s_origin, float4* __restrict__ s_dir,
                                                       float* __restrict__ s_max_t) {
  const unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n_rays) return;
  const uint32_t j = order[i];
  s_origin[i] = origin[j];
  s_dir[i] = dir[j];
  if (max_t) s_max_t[i] = max_t[j];
}

template <bool SHADOW>
__global__ __launch_bounds__(BLOCK) MIPT_BATCH_OCC void k_bvh_batch(DeviceScene sc, const float4* __restrict__ origin,
                                                     const float4* __restrict__ dir,
                                                     const float* __restrict__ max_t, unsigned long long n_rays,
                                                     unsigned long long* __restrict__ queue,
                                                     int32_t* __restrict__ tri_idx, float* __restrict__ t_out,
                                                     float4* __restrict__ pos_out, uint8_t* __restrict__ occluded,
                                                     const uint32_t* __restrict__ order) {
  // order: null, or origin / dir / max_t are in walking order and order[k] is the index ray k's result goes to
  const int lane = threadIdx.x & 63;
  __shared__ int lds_stack[(BVH_LDS_STACK ? BVH_LDS_STACK : 1) * BLOCK];
  BvhTrav T;
  BvhMem M;
  const StackLds stk{lds_stack + threadIdx.x, M.stack};
  T.cur = BVH_DONE;
  BvhRay r = {};
  unsigned long long my = 0;
  float lim0 = 0.f;
  bool have = false, drained = false;
  uint32_t overflow = 0;
  unsigned long long res_next = 0, res_end = 0;  // wave-uniform: the wave's reserved run of rays
  const unsigned long long per_head = (n_rays + BATCH_HEADS - 1) / BATCH_HEADS;
  int head = (int)(blockIdx.x % BATCH_HEADS), heads_done = 0;  // wave-uniform
  for (;;) {
    const unsigned long long need = __ballot(!have);
    if (need && !drained) {
      // The wave reserves BATCH_RESERVE rays with one atomic and hands them to its lanes as they
      // run dry. The rays are split into BATCH_HEADS contiguous parts, each with its own queue head (one
      // head took 50 M same-address atomics per second on fast scenes and was the bottleneck); a
      // wave starts at the part its workgroup index picks and moves on when a part has run dry.
      while (res_next == res_end && !drained) {
        const unsigned long long lo = (unsigned long long)head * per_head;
        const unsigned long long hi = lo + per_head < n_rays ? lo + per_head : n_rays;
        unsigned long long base = 0;
        if (lane == 0 && lo < hi) base = atomicAdd(queue + head * 16, (unsigned long long)BATCH_RESERVE);
        base = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
               (unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)base);
        base += lo;
        if (lo >= hi || base >= hi) {
          head = head + 1 == BATCH_HEADS ? 0 : head + 1;
          if (++heads_done == BATCH_HEADS) drained = true;
        } else {
          res_next = base;
          res_end = base + BATCH_RESERVE < hi ? base + BATCH_RESERVE : hi;
        }
      }
      if (!drained) {
        const unsigned long long avail = res_end - res_next;
        const unsigned long long rank = (unsigned long long)__popcll(need & ((1ull << lane) - 1ull));
        if (!have && rank < avail) {
          my = res_next + rank;
          const float4 of = origin[my], df = dir[my];
          r = bvh_ray(sc.scale, {of.x, of.y, of.z, of.w}, {df.x, df.y, df.z, df.w});
          lim0 = SHADOW ? max_t[my] : 3.402823466e+38f;
          if (order) my = order[my];  // from here on `my` is only where the result goes
          bvh_begin(T, M, &sc, r, SHADOW, lim0);
          have = true;
        }
        const unsigned long long n_need = (unsigned long long)__popcll(need);
        res_next += n_need < avail ? n_need : avail;
      }
    }
    if (!__any(have)) break;  // every lane idle: the queue has run dry
    if (have) {
      // refills are cheap here (two loads and the big-triangle filter): leave the walk when half is done
      bvh_burst<1, 2, !SHADOW>(T, M, stk, &sc, sc.bvh, r, SHADOW, lim0, overflow);
      if (T.cur == BVH_DONE) {
        HitAcc h;
        h.set_lim(lim0);
        h.idx = SHADOW ? 0 : -1; h.by = 0.f; h.bz = 0.f;
        if (T.redo) bvh_closest_exhaustive(&sc, sc.bvh, r, h, overflow);
        else bvh_finish(T, M, SHADOW, h);
        if (SHADOW) {
          occluded[my] = (uint8_t)h.idx;
        } else {
          V4 pos = {0, 0, 0, 0}, nrm;
          bool sphere_hit = false;
          for (int s = 0; s < sc.n_spheres; ++s) {
            if (sphere_test(sc.spheres[s], r.o, r.d, h.lim, pos, nrm)) {
              h.idx = sc.n_tris + s;
              sphere_hit = true;
            }
          }
          tri_idx[my] = h.idx;
          if (h.idx >= 0) {
            t_out[my] = h.lim;
            if (pos_out) {
              if (!sphere_hit) {  // intersections.cpp:29
                const TriRec Tr = sc.tris[h.idx];
                pos.x = (Tr.v0x + h.by * Tr.e1x) + h.bz * Tr.e2x;
                pos.y = (Tr.v0y + h.by * Tr.e1y) + h.bz * Tr.e2y;
                pos.z = (Tr.v0z + h.by * Tr.e1z) + h.bz * Tr.e2z;
                pos.w = (Tr.v0w + h.by * 0.0f) + h.bz * 0.0f;
              }
              pos_out[my] = make_float4(pos.x, pos.y, pos.z, pos.w);
            }
          }
        }
        have = false;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// The path kernel.
//
// Persistent lanes over a global work queue. A work item is a run of ITEM_SAMPLES consecutive
// samples of one pixel; a lane adds the item's path radiances in sample order and stores the sum
// to the item's own slot of a scratch image stack, so the image is deterministic (no float
// atomics) whichever lane ran which item; k_reduce_chunks then adds the item sums of a pixel
// in sample order. Path lengths vary from 1 ray (camera ray misses the box) to 20, so lanes
// that finish an item draw the next one from the queue instead of waiting for their neighbours.
// The reference's recursion (trace -> bounce* -> trace ...) is single-child, so it is run as a
// loop with one ray/scene query per iteration. A lane whose path ended starts its next sample
// in the same iteration ("regeneration"), and a terminal vertex turns the lane's next query
// into the shadow ray of directLight, handled by the same triangle scan with the any-hit
// rule -- the scan, most of the work, therefore runs with every lane of the wave active.
//
// pathtracer.cpp passes `Ray&` down the recursion, and bounceDiffuse reads ray.direction AFTER
// the recursive call returns (pathtracer.cpp:17-18): the cosine factor of every diffuse vertex
// is taken with the direction of the path's LAST segment. The per-vertex factors are therefore
// applied when the path ends, deepest vertex first, from a small per-lane stack of primitive
// tags kept in LDS.

// Register budget of the path kernel: 5 waves per SIMD (<= 102 VGPRs) measured fastest on the
// fog-on Cornell workload (4 waves: -3.5 %, 6: -3 %, 3: -16 %).
#ifndef MIPT_WAVES_PER_EU
#define MIPT_WAVES_PER_EU 5
#endif
// tree scenes: the walk's stack takes 12 KB of LDS per workgroup on top of the path state, which
// fits four workgroups per CU; 128 registers per lane then leave the kernel without spills
// (measured best of 3, 4, 5)
#ifndef MIPT_TREE_WAVES_PER_EU
#define MIPT_TREE_WAVES_PER_EU 4
#endif
// how many big triangles of a tree scene have their records staged in LDS by the path kernel (0: none)
#ifndef MIPT_BIG_LDS
#define MIPT_BIG_LDS 16
#endif
#define MIPT_PATHS_OCC                                                                              \
  __attribute__((amdgpu_waves_per_eu(KIND == SCENE_BVH ? MIPT_TREE_WAVES_PER_EU : MIPT_WAVES_PER_EU, \
                                     KIND == SCENE_BVH ? MIPT_TREE_WAVES_PER_EU : MIPT_WAVES_PER_EU)))

// The kernel's argument block, read where a value is needed instead of being held in scalar
// registers from the kernel's entry on: the path kernel's loop is short of scalar registers (23 were
// spilled to lanes of a vector register and constants were re-made before every use), and the camera,
// the light, the glass constants, the queue's geometry and the weights of the completion are each
// needed in ONE branch of it. The empty asm statement hides the address from the compiler, which
// would otherwise hoist the loads to the entry again. PathArgs is the kernel's only argument, so it
// sits at offset 0 of the argument segment.
__device__ __forceinline__ const PathArgs MIPT_CONST* late_args() {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  const PathArgs MIPT_CONST* k = (const PathArgs MIPT_CONST*)__builtin_amdgcn_kernarg_segment_ptr();
#pragma clang diagnostic pop
  asm volatile("" : "+s"(k));
  return k;
}

template <int RNG_MODE, bool STATS, bool LIST, int KIND>
__global__ __launch_bounds__(BLOCK) MIPT_PATHS_OCC void k_paths(const PathArgs a) {
  extern __shared__ uint32_t lds_stack[];  // vertex stack(s) of tags, sphere normals, record tables
  const int tid = threadIdx.x;
  const MiptParams& P = a.p;
  const int W = P.width, H = P.height;

  // work of the lane: a run of samples [s, s_end) of pixel (x, y). LIST: the one listed path.
  // Otherwise an item drawn from the global queue (see the loop head).
  const int lane = tid & 63;
  int x = 0, y = 0;
  uint32_t s = 0, s_end = 0, pixel = 0, pkey = 0;
  size_t list_i = 0;
  // State a lane touches only when an item or a path begins or ends is parked in LDS, not in
  // registers. Rows of `park`, for context c = 0, 1 of the lane's radiance sum and buffer b = 0, 1
  // of its vertex stack:
  //   11 c + 0..2   radiance sum of the work item
  //   11 c + 3      the item's output slot (index into a.partial, divided by 3)
  //   11 b + 4..7   direction of the path's last segment
  //   11 b + 8..10  terminal vertex: clamped cosine to the light, distance to the light, primitive
  // Linear-scan scenes without spheres, and scenes with global vertex stacks, use both sets (deferred
  // completion, below), the others one.
  typedef typename TagType<KIND>::T TagT;
  constexpr uint32_t TAGV_GLASS = (uint32_t)(TagT)TAG_GLASS, TAGV_FOG = (uint32_t)(TagT)TAG_FOG;
  constexpr bool GSTACK = (MIPT_TREE_VSTACK && KIND == SCENE_BVH) || KIND == SCENE_LINEAR_BIG;  // vstack_kind(KIND)
  constexpr bool DEFER_KIND = GSTACK || KIND != SCENE_BVH;
  const bool defer = paths_deferred(KIND, a.sc.n_spheres);  // wave-uniform
  __shared__ float park[DEFER_KIND ? 22 : 11][BLOCK];
  const size_t gtid = (size_t)blockIdx.x * BLOCK + tid;
  // entry of the vertex stack the running path writes next (GSTACK): buffer js & JS_CUR, level = stack height
#define MIPT_VTOP() (((size_t)(((js & JS_CUR) ? P.max_depth + 1 : 0) + (ds & 0xFFu))) * a.vstride + gtid)
  bool have = false;
  if (LIST) {
    list_i = (size_t)blockIdx.x * BLOCK + tid;
    if (list_i < a.n_paths) {
      x = a.px[list_i]; y = a.py[list_i];
      s = a.samp[list_i]; s_end = s + 1;
      pixel = (uint32_t)(x * H + y);
      pkey = pixel_key(P.seed, pixel);
      have = true;
    }
  }
  // wave-uniform queue state: the items of the last batch not handed out yet, that batch's first
  // id, tile and sample block, and whether the queue has run dry
  uint32_t res_next = 0, res_end = 0, res_base = 0, res_sb = 0, res_tx = 0, res_ty = 0;
  bool drained = LIST;

  park[0][tid] = 0.f; park[1][tid] = 0.f; park[2][tid] = 0.f;
  uint32_t st[ST_COUNT];
  if (STATS || LIST) {
#pragma unroll
    for (int k = 0; k < ST_COUNT; ++k) st[k] = 0;
  }

  // path state
  V4 o = {0, 0, 0, 0}, d = {0, 0, 0, 0};
  Rng rng = {1u, 0u};
  // the path's depth (bits 8..) and the height of its vertex stack (bits 0-7) share a register: one
  // register more and the kernel spills at five waves per SIMD
  uint32_t ds = 0;
  bool shadow = false, fresh = true;
  float shadow_max = 0.f;
  // Completion state: which stack buffer the running path writes (cur), which sum context its item
  // uses (ctx), and the lane's deferred job, if any: a path that ended lit, waiting for its
  // radiance to be computed and added (its stack depth, its sum context, whether it was the last
  // path of its item).
  // Packed into one register: bit 0 cur, bit 1 ctx, bit 2 job pending, bit 3 the job's context,
  // bit 4 the job is the last path of its item, bits 8.. the job's stack depth.
  uint32_t js = 0;
  constexpr uint32_t JS_CUR = 1u, JS_CTX = 2u, JS_PEND = 4u, JS_JCTX = 8u, JS_LAST = 16u;
  TagT* tags = reinterpret_cast<TagT*>(lds_stack);
  float* sph_n = reinterpret_cast<float*>(lds_stack + a.lds_sph_off);
  const bool fog = P.fog_sigma_t > 0.f;
  constexpr bool BVH = KIND == SCENE_BVH;
  BvhTrav trav;      // tree scenes: the lane's walk, resumable across iterations of the main loop
  BvhMem trav_mem;
  __shared__ int bvh_lds_stack[(BVH && BVH_LDS_STACK ? BVH_LDS_STACK : 0) * BLOCK + 1];
  trav.cur = BVH_DONE;
  bool walking = false;
  // tree scenes: the records of the big triangles (the room's walls, tested before every walk) in LDS,
  // when there are few enough of them: the candidates' tests then start without a trip to memory
  __shared__ TriRec big_lds[BVH && MIPT_BIG_LDS > 0 ? MIPT_BIG_LDS : 1];
  const TriRec* big_recs = nullptr;
  if (BVH && MIPT_BIG_LDS > 0 && a.sc.n_big > 0 && a.sc.n_big <= MIPT_BIG_LDS) {
    uint32_t* dst = reinterpret_cast<uint32_t*>(big_lds);
    for (int i = tid; i < a.sc.n_big * 16; i += BLOCK)
      dst[i] = reinterpret_cast<const uint32_t*>(a.sc.tris + a.sc.big_ids[i >> 4])[i & 15];
    __syncthreads();
    big_recs = big_lds;
  }
  // per-primitive records the shading reads by hit index: from LDS for small scenes
  const TriRec* tri_tab = a.sc.tris;
  const ShadeRec* shade_tab = a.sc.shade;
  if (KIND == SCENE_LINEAR_LDS) {
    uint32_t* tab = lds_stack + a.lds_table_off;
    const int n_tri_dw = a.sc.n_tris * (int)(sizeof(TriRec) / 4);
    const int n_shade_dw = (a.sc.n_tris + a.sc.n_spheres) * (int)(sizeof(ShadeRec) / 4);
    const uint32_t* gt = reinterpret_cast<const uint32_t*>(a.sc.tris);
    const uint32_t* gs = reinterpret_cast<const uint32_t*>(a.sc.shade);
    for (int i = tid; i < n_shade_dw; i += BLOCK) tab[i] = gs[i];
    for (int i = tid; i < n_tri_dw; i += BLOCK) tab[n_shade_dw + i] = gt[i];
    __syncthreads();
    shade_tab = reinterpret_cast<const ShadeRec*>(tab);
    tri_tab = reinterpret_cast<const TriRec*>(tab + n_shade_dw);
  }
  // A field of the scene: tree scenes fetch it from the argument block at the point of use (see late_args)
#define MIPT_SC(f) (BVH ? late_args()->sc.f : a.sc.f)
  Prof prof{};
#ifdef MIPT_PROFILE_SECTIONS
  __shared__ uint32_t prof_lds[PF_COUNT];
  if (tid < PF_COUNT) prof_lds[tid] = 0;
  __syncthreads();
  prof.lds = prof_lds;
#endif
  prof.start();

  for (;;) {
    if (!LIST) {
      // Work distribution. An item is ITEM_SAMPLES consecutive samples of one pixel; item ids run
      // over 8x8 pixel tiles, sample block by sample block. A wave draws 64 ids (one tile) at a
      // time with ONE atomic on the queue head and hands them to its lanes as they run dry, so
      // every lane stays busy until the whole frame is done -- however uneven the path lengths.
      const unsigned long long need = __ballot(!have);
      if (need && !drained) {
        if (res_next == res_end) {
          const PathArgs MIPT_CONST* kq = late_args();  // the queue's geometry: once per 64 items
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(kq->queue, 64u);
          base = __builtin_amdgcn_readfirstlane(base);
          if (base >= kq->n_items) {
            drained = true;
          } else {
            const uint32_t n_slots = kq->n_slots, tiles_x = kq->tiles_x;
            res_next = base;
            res_end = base + 64u;  // n_items is a multiple of 64: a batch is one tile of one sample block
            res_base = base;
            res_sb = base / n_slots;
            const uint32_t tile = (base - res_sb * n_slots) >> 6;
            res_ty = tile / tiles_x;
            res_tx = tile - res_ty * tiles_x;
          }
        }
        if (!drained) {
          const uint32_t avail = res_end - res_next;
          const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
          if (!have && rank < avail) {
            const uint32_t l = res_next + rank - res_base, sb = res_sb;
            x = (int)(res_tx * 8u + (l >> 3));
            y = (int)(res_ty * 8u + (l & 7u));
            if (x < W && y < H) {
              pixel = (uint32_t)(x * H + y);
              pkey = pixel_key(P.seed, pixel);
              s = a.sample_begin + sb * a.item_samples;
              const uint32_t last = a.sample_begin + a.sample_count;
              s_end = (s + a.item_samples < last) ? s + a.item_samples : last;
              // a new item takes the lane's other sum context: the previous item's may still wait
              // for a deferred job (that job is flushed before a third item could need it again)
              if (defer) js ^= JS_CTX;
              const int c0 = (js & JS_CTX) ? 11 : 0;
              park[c0 + 3][tid] = __uint_as_float(sb * (uint32_t)(W * H) + pixel);  // < n_items: fits 32 bits
              park[c0 + 0][tid] = 0.f; park[c0 + 1][tid] = 0.f; park[c0 + 2][tid] = 0.f;
              fresh = true;
              have = true;
            }
          }
          const uint32_t n_need = (uint32_t)__popcll(need);
          res_next += (n_need < avail) ? n_need : avail;
        }
      }
    }
    bool final_round = false;
    if (!__any(have)) {
      if (!drained) continue;
      if (!__any((js & JS_PEND) != 0)) break;
      final_round = true;  // no work left: one more pass for the deferred jobs
    }
    prof.mark(PF_QUEUE);
    // A lane is `ready` for shading when its scene query has an answer. The linear scan answers
    // within the iteration; a walk through the tree may span several (bvh_burst returns when most
    // of the walking lanes are done, so finished lanes shade and start their next ray instead of
    // waiting for the longest walk of the wave).
    bool ready = false;
    HitAcc h;
    h.set_lim(3.402823466e+38f);
    h.idx = -1; h.by = 0.f; h.bz = 0.f;
    uint32_t n_tests = 0, n_gate = 0, redo = 0, overflow = 0, n_nodes = 0;
    if (have) {
    if (fresh && !walking) {
      // main.cpp:124-133: two draws jitter the pixel, direction = Rx*Ry*d, not normalised
      rng_begin<RNG_MODE>(rng, pkey, s);
      const float u1 = rng_next<RNG_MODE>(rng);
      const float u2 = rng_next<RNG_MODE>(rng);
      const float dvx = (float)((double)(x - W / 2) + (double)(u1 * 2.0f - 1.0f) * 0.1);
      const float dvy = (float)((double)(y - H / 2) + (double)(u2 * 2.0f - 1.0f) * 0.1);
      const PathArgs MIPT_CONST* ka = late_args();  // the camera: needed only where a path begins
      const float MIPT_CONST* M = ka->M;
      const float MIPT_CONST* cam_pos = ka->cam_pos;
      const float dvz = ka->focal, dvw = 0.0f;
      // mat4 * vec4 (type_mat4x4.inl:641-652): (m0*v0 + m1*v1) + (m2*v2 + m3*v3)
      d.x = (M[0] * dvx + M[4] * dvy) + (M[8] * dvz + M[12] * dvw);
      d.y = (M[1] * dvx + M[5] * dvy) + (M[9] * dvz + M[13] * dvw);
      d.z = (M[2] * dvx + M[6] * dvy) + (M[10] * dvz + M[14] * dvw);
      d.w = (M[3] * dvx + M[7] * dvy) + (M[11] * dvz + M[15] * dvw);
      o = {cam_pos[0], cam_pos[1], cam_pos[2], cam_pos[3]};
      ds = 0; shadow = false; fresh = false;
      if (STATS || LIST) st[ST_PATHS]++;
    }

    prof.mark(PF_RAYGEN);
    // ---- one ray/scene query: closestIntersection, or existsIntersection for a shadow ray
    const float lim0 = shadow ? shadow_max : 3.402823466e+38f;
    if (BVH) {
      // (the scene's fields are fetched from the argument block where the walk uses them: held in scalar
      // registers from the kernel's entry on they crowd out the loop's own and are spilled)
      BvhRay r = bvh_ray(late_args()->sc.scale, o, d);  // o, d stay put while the lane walks
      if (!walking) { prof.count(PF_N_FRESH, 1u); bvh_begin(trav, trav_mem, &late_args()->sc, r, shadow, lim0, big_recs); walking = true; }
      prof.mark(PF_PHASE1);
      {
        const PathArgs MIPT_CONST* kb = late_args();
        bvh_burst(trav, trav_mem, StackLds{bvh_lds_stack + tid, trav_mem.stack}, &kb->sc, kb->sc.bvh, r, shadow, lim0, overflow,
                  prof.lds ? prof.lds + PF_W_BURSTS : nullptr);
      }
      n_tests = r.n_leaves; n_nodes = r.n_nodes;
      prof.mark(PF_PHASE2);
      if (trav.cur == BVH_DONE) {
        walking = false;
        ready = true;
        h.set_lim(lim0);
        h.idx = shadow ? 0 : -1;
        if (trav.redo) {
          const PathArgs MIPT_CONST* kx = late_args();
          redo = 1; bvh_closest_exhaustive(&kx->sc, kx->sc.bvh, r, h, overflow); n_tests = r.n_leaves; n_nodes = r.n_nodes;
        }
        else bvh_finish(trav, trav_mem, shadow, h);
      }
      if (STATS) { st[ST_TESTS] += n_tests; st[ST_BVH_REDO] += redo; st[ST_BVH_OVERFLOW] += overflow; st[ST_BVH_NODES] += n_nodes; }
    } else {
      h.set_lim(lim0);
      h.idx = shadow ? 0 : -1;
      // phase 2 gathers each lane's candidate record from the LDS copy when there is one (+2 %)
      scene_query<STATS, false>(a.sc, o, d, shadow, h, n_tests, n_gate, redo, overflow, prof, tri_tab);
      if (STATS) { st[ST_TESTS] += n_tests; st[ST_GATE] += n_gate; }
      ready = true;
    }
    }  // if (have)
    prof.count(PF_N_READY, ready ? 1u : 0u);
    bool path_done = false;
    bool lit = false;  // the path ended at a terminal vertex the light reaches: a job for the completion below
    if (ready) {
    if (STATS || LIST) st[shadow ? ST_SHADOW : ST_CLOSEST]++;

    if (shadow) {
      // light.cpp:26-29: occluded -> zero
      lit = h.idx == 0;
      path_done = true;
    } else {
      V4 pos = {0, 0, 0, 0}, nrm = {0, 0, 0, 0};
      bool sphere_hit = false;
      const int n_sph = MIPT_SC(n_spheres);
      for (int j = 0; j < n_sph; ++j) {  // intersections.cpp:39-52
        if (STATS) st[ST_SPHERE]++;
        if (sphere_test(MIPT_SC(spheres)[j], o, d, h.lim, pos, nrm)) {
          h.idx = MIPT_SC(n_tris) + j;
          sphere_hit = true;
        }
      }
      if (h.idx < 0) {
        path_done = true;  // pathtracer.cpp:74-75: miss, colour stays 0
      } else {
        const ShadeRec sh = (BVH ? MIPT_SC(shade) : shade_tab)[h.idx];
        if (!sphere_hit) {
          const TriRec T = (BVH ? MIPT_SC(tris) : tri_tab)[h.idx];
          pos.x = (T.v0x + h.by * T.e1x) + h.bz * T.e2x;  // intersections.cpp:29
          pos.y = (T.v0y + h.by * T.e1y) + h.bz * T.e2y;
          pos.z = (T.v0z + h.by * T.e1z) + h.bz * T.e2z;
          pos.w = (T.v0w + h.by * 0.0f) + h.bz * 0.0f;
          nrm = {sh.nx, sh.ny, sh.nz, sh.nw};
        }

        bool medium = false;
        if (fog) {
          // free flight in the homogeneous medium against the world-space hit distance
          const float len = sqrtf(dot3({d.x, d.y, d.z}, {d.x, d.y, d.z}));
          const float dist = h.lim * len;
          const float uf = rng_next<RNG_MODE>(rng);
          const float fs = -fog_logf(1.0f - uf) / P.fog_sigma_t;
          if (fs < dist) {
            const float k = fs / len;
            o.x = o.x + d.x * k; o.y = o.y + d.y * k; o.z = o.z + d.z * k; o.w = 1.0f;
            medium = true;
            if (STATS) st[ST_FOG]++;
          }
        }

        // Every branch below ends in normalize(new direction); the branches only prepare the
        // vector and ONE normalisation runs for all lanes after they rejoin.
        V4 nd = d;
        bool renorm = false, offset_origin = false;
        // the fog's scattering and bounceDiffuse both end in sincos(phi) of a sampled direction:
        // the branches prepare (phi, radius, axial component), one evaluation serves both
        bool need_sc = false, scatter = false;
        float sc_phi = 0.f, sc_r = 0.f, sc_y = 0.f;
        V3 Nt = {sh.ntx, sh.nty, sh.ntz}, Nb = {sh.nbx, sh.nby, sh.nbz};
        const bool terminal = ds >= ((uint32_t)P.max_depth << 8) || rng_next<RNG_MODE>(rng) <= P.rr_prob;  // :80
        prof.mark(PF_HIT);
        if (terminal) {
          // pathtracer.cpp:81-93 + directLight (light.cpp:13-31). Only what the shadow ray needs
          // is computed here: the light's radiance at this vertex (inverse square law, fog
          // transmittance, albedo) is evaluated when the path is completed, and only if the shadow
          // ray came through.
          V4 lp;
          const float MIPT_CONST* light_pos = late_args()->light_pos;
          lp.x = light_pos[0] + (rng_next<RNG_MODE>(rng) * 2.0f - 1.0f) * 0.1f;
          lp.y = light_pos[1];
          lp.z = light_pos[2] + (rng_next<RNG_MODE>(rng) * 2.0f - 1.0f) * 0.1f;
          lp.w = light_pos[3];
          const V4 at = medium ? o : pos;
          const V4 v = sub4(lp, at);
          const V4 vecr = normalize4(v);
          const float r = sqrtf(dot4(v, v));
          float m = 1.0f;
          if (!medium) {
            // normalize(inter.normal) (light.cpp:20) is a per-triangle constant
            const V4 nn = sphere_hit ? normalize4(nrm) : V4{sh.nnx, sh.nny, sh.nnz, sh.nnw};
            const float c = dot4(vecr, nn);
            m = (c < 0.f) ? 0.f : c;  // std::max(c, 0.f)
            o = add4(pos, scale4(vecr, 0.001f));  // light.cpp:25
          }
          const int t0 = (js & JS_CUR) ? 15 : 4;
          park[t0 + 0][tid] = d.x; park[t0 + 1][tid] = d.y; park[t0 + 2][tid] = d.z; park[t0 + 3][tid] = d.w;  // last segment
          park[t0 + 4][tid] = m; park[t0 + 5][tid] = r;
          park[t0 + 6][tid] = __uint_as_float(medium ? TERMINAL_IN_MEDIUM : (uint32_t)h.idx);
          if (GSTACK && !medium)  // what the completion needs of the primitive (pathtracer.cpp:88-93)
            a.vstack[((size_t)(((js & JS_CUR) ? P.max_depth + 1 : 0) + P.max_depth)) * a.vstride + gtid + a.vplane] =
                make_float4(sh.r, sh.g, sh.b, __uint_as_float(sh.material));
          nd = vecr; renorm = true;  // Ray ctor normalises again (ray.h:11-12)
          shadow_max = r;
          shadow = true;
          // A terminal vertex the light does not reach (cosine <= 0) has L = 0 whatever the
          // shadow ray finds, and existsIntersection draws no random numbers: the path ends here
          // with the radiance the reference computes. The counting variants still cast the ray.
          if (!STATS && !LIST && !medium && m == 0.f) { path_done = true; shadow = false; }
          prof.mark(PF_TERMINAL);
        } else if (medium) {
          // isotropic scattering, weight = single-scattering albedo
          const float u1 = rng_next<RNG_MODE>(rng);
          const float u2 = rng_next<RNG_MODE>(rng);
          sc_y = 1.0f - 2.0f * u1;
          const float s2 = 1.0f - sc_y * sc_y;
          sc_r = sqrtf(s2 < 0.f ? 0.f : s2);
          sc_phi = 6.28318530718f * u2;
          need_sc = true; scatter = true;
          if (GSTACK)
            a.vstack[MIPT_VTOP() + a.vplane] = make_float4(0.f, 0.f, 0.f, __uint_as_float(VK_FOG));
          else
            tags[((size_t)(((js & JS_CUR) ? P.max_depth : 0) + (ds & 0xFFu))) * BLOCK + tid] = (TagT)TAG_FOG;
          ds += 0x101u;  // sp++, depth++
          prof.mark(PF_SCATTER);
        } else {
          o = pos;  // pathtracer.cpp:100
          const uint32_t mat = sh.material;
          if (mat == MIPT_MAT_DIFFUSE) {
            // bounceDiffuse (pathtracer.cpp:5-19), orthonormalSystem / diffuseHemisphere (utils.cpp:10-22)
            if (STATS) st[ST_DIFFUSE]++;
            // orthonormalSystem(normal) is a per-triangle constant, stored with the triangle
            if (sphere_hit) orthonormal_system(nrm, Nt, Nb);
            // g++ evaluates the two argument draws right to left: first draw -> b, second -> a
            const float ub = rng_next<RNG_MODE>(rng);
            sc_y = rng_next<RNG_MODE>(rng);
            sc_r = sqrtf(1.0f - sc_y * sc_y);
            sc_phi = (float)((2 * 3.14) * (double)ub);
            need_sc = true;
            if (GSTACK) {
              // the record: the normal and the colour the unwind multiplies by (pathtracer.cpp:17-18)
              const size_t vi = MIPT_VTOP();
              a.vstack[vi] = make_float4(nrm.x, nrm.y, nrm.z, nrm.w);
              a.vstack[vi + a.vplane] = make_float4(sh.r, sh.g, sh.b, __uint_as_float(VK_DIFFUSE));
            } else {
              tags[((size_t)(((js & JS_CUR) ? P.max_depth : 0) + (ds & 0xFFu))) * BLOCK + tid] = (TagT)h.idx;
            }
            if (!GSTACK && sphere_hit) {
              float* q = sph_n + ((size_t)(ds & 0xFFu) * BLOCK + tid) * 4;
              q[0] = nrm.x; q[1] = nrm.y; q[2] = nrm.z; q[3] = nrm.w;
            }
            ds += 0x101u;
            prof.mark(PF_DIFFUSE);
          } else if (mat == MIPT_MAT_MIRROR) {
            // bounceMirror (pathtracer.cpp:21-27); weight 1, nothing to unwind
            if (STATS) st[ST_MIRROR]++;
            const float ct = dot4(nrm, d) * -1.0f;
            nd = add4(d, scale4(nrm, ct * 2.0f)); renorm = true;
            ds += 0x100u;  // depth++
            prof.mark(PF_MIRROR);
          } else if (mat == MIPT_MAT_GLASS) {
            // bounceGlass (pathtracer.cpp:29-63)
            if (STATS) st[ST_GLASS]++;
            // (read here like the camera and the light; the empty asm statement pins the two indices in
            // scalar registers -- without it the compiler selects between their ADDRESSES per lane and
            // fetches the chosen one with a vector load the branch then waits for)
            const PathArgs MIPT_CONST* kg = late_args();
            const float R0 = kg->glass_R0;
            float eta_enter = kg->eta_enter, eta_exit = kg->eta_exit;
            asm volatile("" : "+s"(eta_enter), "+s"(eta_exit));
            float eta = eta_enter;
            V4 n = nrm;
            if (dot4(n, d) > 0.f) {  // inside the glass
              n = scale4(n, -1.0f);
              eta = eta_exit;
            }
            const float c = dot4(n, d) * -1.0f;
            const float cost1 = (c < 0.f) ? 0.f : c;
            const float cost2 = 1.0f - (eta * eta) * (1.0f - cost1 * cost1);
            const float Rprob = R0 + (1.0f - R0) * pow5f(1.0f - cost1);
            if (cost2 > 0.f && rng_next<RNG_MODE>(rng) > Rprob) {
              const float kk = eta * cost1 - sqrtf(cost2);
              nd = add4(scale4(d, eta), scale4(n, kk));
            } else {
              nd = add4(d, scale4(n, cost1 * 2.0f));
            }
            renorm = true;
            if (GSTACK)
              a.vstack[MIPT_VTOP() + a.vplane] = make_float4(0.f, 0.f, 0.f, __uint_as_float(VK_GLASS));
            else
              tags[((size_t)(((js & JS_CUR) ? P.max_depth : 0) + (ds & 0xFFu))) * BLOCK + tid] = (TagT)TAG_GLASS;
            ds += 0x101u;
            prof.mark(PF_GLASS);
          } else if (mat == MIPT_MAT_SPECULAR) {
            // bounceSpecular (pathtracer.cpp:65-70); weight 1
            if (STATS) st[ST_SPECULAR]++;
            const float k2 = dot4(d, nrm) * 2.f;
            nd = sub4(d, scale4(nrm, k2)); renorm = true;
            ds += 0x100u;
          } else {
            path_done = true;  // no branch of pathtracer.cpp:103-110 taken: colour stays 0
          }
        }
        if (need_sc) {
          float sn, cs;
          sincosf_glibc(sc_phi, sn, cs);
          const float sx = cs * sc_r, sz = sn * sc_r;
          if (scatter) {
            d = {sx, sz, sc_y, 0.0f};
          } else {  // utils.cpp:20-22 and pathtracer.cpp:9-13
            nd.x = (sx * Nb.x + sc_y * nrm.x) + sz * Nt.x;
            nd.y = (sx * Nb.y + sc_y * nrm.y) + sz * Nt.y;
            nd.z = (sx * Nb.z + sc_y * nrm.z) + sz * Nt.z;
            nd.w = 1.0f;
            renorm = true; offset_origin = true;  // origin += dir * 0.001 after normalising
          }
        }
        if (renorm) {
          d = normalize4(nd);
          if (offset_origin) o = add4(o, scale4(d, 0.001f));  // pathtracer.cpp:14
        }
        prof.mark(PF_RENORM);
      }
    }
    }  // if (ready)

    // ---- completion of paths
    // A path that ends lit still owes its radiance: the light at the terminal vertex, multiplied
    // back through the vertices (the recursion unwound, deepest first) and added to the item's sum.
    // Nothing the lane does next depends on that value, and at any moment only a few lanes of a
    // wave have one to compute. So a lane leaves it behind as a JOB -- vertex stack and terminal
    // record stay in the buffers it just used, the next path writes the other pair -- and starts
    // its next path; the wave computes all its jobs together when some lane needs its second
    // buffer back: it ends another lit path, or it is about to start an item on the sum context
    // its job still adds to. A lane's jobs run in the order of its paths, so the item's sum is
    // formed in sample order as before. Scenes with spheres or a tree keep one buffer and compute
    // the job at once (their stacks are too large to double in LDS; see vstack_kind for what moving
    // them to global memory measured).
    const bool want_job = path_done && lit;
    const bool item_end = path_done && s + 1 >= s_end;
    const bool pend = (js & JS_PEND) != 0;
    // the job's sum context differs from the running item's (bits 3 and 1 of js)
    const bool other_ctx = (((js >> 3) ^ (js >> 1)) & 1u) != 0;
    bool flush;
    if (defer) {
      flush = __any(pend && path_done && (want_job || (item_end && other_ctx))) || final_round;
    } else {
      if (want_job) js = JS_PEND | ((ds & 0xFFu) << 8);  // one buffer, one context: all other bits stay 0
      flush = __any((js & JS_PEND) != 0);
    }
    prof.count_wave(PF_N_ITERS);
    if (flush) {
      prof.count_wave(PF_N_FLUSH);
      if (js & JS_PEND) {
        prof.count(PF_N_JOBS, 1u);
        prof.count(PF_N_JOB_LEVELS, js >> 8);
        const int b = (defer && !(js & JS_CUR)) ? 1 : 0;  // the buffer the running path does not write
        const int t0 = 11 * b + 4, c0 = (js & JS_JCTX) ? 11 : 0;
        const int job_sp = (int)(js >> 8);
        const V4 dfin = {park[t0 + 0][tid], park[t0 + 1][tid], park[t0 + 2][tid], park[t0 + 3][tid]};
        const float m = park[t0 + 4][tid], r = park[t0 + 5][tid];
        const uint32_t tprim = __float_as_uint(park[t0 + 6][tid]);
        // directLight (light.cpp:20-24) and pathtracer.cpp:88-93
        const float den = (float)(((4.0 * 3.14159) * (double)r) * (double)r);  // light.cpp:22
        const float tr = fog ? fog_expf(-(P.fog_sigma_t * r)) : 1.0f;  // transmittance to the light
        float cr, cg, cb;
        const PathArgs MIPT_CONST* kc = late_args();
        const float light_col[3] = {kc->light_col[0], kc->light_col[1], kc->light_col[2]};
        const uint32_t light_gray = kc->light_gray;
        const float glass_weight = kc->p.glass_weight, diffuse_weight = kc->p.diffuse_weight, fog_albedo = kc->p.fog_albedo;
        // Global vertex stacks: the loads of the terminal vertex' colour and of the four deepest vertex
        // records are issued together, before anything waits (a level at a time, the unwind paid one
        // memory latency per level of the longest job).
        const float4* vsn = nullptr;
        const float4* vsc = nullptr;
        float4 tc = {0.f, 0.f, 0.f, 0.f}, qc0 = tc, qc1 = tc, qc2 = tc, qc3 = tc, qn0 = tc, qn1 = tc, qn2 = tc, qn3 = tc;
        if (GSTACK) {
          vsn = a.vstack + ((size_t)(b * (P.max_depth + 1))) * a.vstride + gtid;  // level k at vsn[k * vstride]
          vsc = vsn + a.vplane;
          const int k = job_sp - 1;
          const size_t i0 = (size_t)(k > 0 ? k : 0) * a.vstride, i1 = (size_t)(k > 1 ? k - 1 : 0) * a.vstride;
          const size_t i2 = (size_t)(k > 2 ? k - 2 : 0) * a.vstride, i3 = (size_t)(k > 3 ? k - 3 : 0) * a.vstride;
          tc = vsc[(size_t)P.max_depth * a.vstride];
          qc0 = vsc[i0]; qc1 = vsc[i1]; qc2 = vsc[i2]; qc3 = vsc[i3];
          qn0 = vsn[i0]; qn1 = vsn[i1]; qn2 = vsn[i2]; qn3 = vsn[i3];
        }
        if (tprim != TERMINAL_IN_MEDIUM) {
          struct { float r, g, b; uint32_t material; } sh;
          if (GSTACK) {
            sh.r = tc.x; sh.g = tc.y; sh.b = tc.z; sh.material = __float_as_uint(tc.w);
          } else {
            const ShadeRec rec = (BVH ? MIPT_SC(shade) : shade_tab)[tprim];
            sh.r = rec.r; sh.g = rec.g; sh.b = rec.b; sh.material = rec.material;
          }
          cr = (light_col[0] * m) / den;
          if (light_gray) {
            cg = cr; cb = cr;
          } else {
            cg = (light_col[1] * m) / den;
            cb = (light_col[2] * m) / den;
          }
          if (fog) { cr = cr * tr; cg = cg * tr; cb = cb * tr; }
          if (sh.material != MIPT_MAT_GLASS) {  // pathtracer.cpp:88-93
            cr = (sh.r * cr) * 1.0f; cg = (sh.g * cg) * 1.0f; cb = (sh.b * cb) * 1.0f;
          }
        } else {
          const float phase = 1.0f / (4.0f * 3.14159f);
          cr = (((light_col[0] * phase) / den) * tr) * fog_albedo;
          cg = (((light_col[1] * phase) / den) * tr) * fog_albedo;
          cb = (((light_col[2] * phase) / den) * tr) * fog_albedo;
        }
        // unwind the recursion, deepest vertex first
        // (one straight body for the three kinds of vertex, selected by the tag: as three nested
        // per-lane branches a level cost ~30 scalar instructions for the execution masks, and the jobs of
        // a flush are of mixed kinds at every level anyway. x * 1.0f is x.)
        // glass: (c * 1) * weight (pathtracer.cpp:62); fog: c * albedo; diffuse, with the direction of
        // the last segment: (((c * colour) * dn) * 1) * weight (pathtracer.cpp:18)
        if (GSTACK) {
          // a level that does not exist multiplies by 1.0f, which leaves every float as it is
          auto level = [&](const float4& vc, const float4& vn, bool valid) {
            const uint32_t kind = __float_as_uint(vc.w);
            const bool is_dif = valid & (kind == VK_DIFFUSE);
            const float wk = kind == VK_GLASS ? glass_weight : (kind == VK_FOG ? fog_albedo : diffuse_weight);
            const float w = valid ? wk : 1.0f;
            const float dn = dot4(dfin, V4{vn.x, vn.y, vn.z, vn.w});
            float tr_ = is_dif ? cr * vc.x : cr, tg_ = is_dif ? cg * vc.y : cg, tb_ = is_dif ? cb * vc.z : cb;
            tr_ = is_dif ? tr_ * dn : tr_; tg_ = is_dif ? tg_ * dn : tg_; tb_ = is_dif ? tb_ * dn : tb_;
            cr = tr_ * w; cg = tg_ * w; cb = tb_ * w;
          };
          int k = job_sp - 1;
          level(qc0, qn0, k >= 0); level(qc1, qn1, k >= 1); level(qc2, qn2, k >= 2); level(qc3, qn3, k >= 3);
          for (k -= 4; k >= 0; k -= 4) {  // jobs of more than four vertices: four more per round
            const size_t i0 = (size_t)k * a.vstride, i1 = (size_t)(k > 1 ? k - 1 : 0) * a.vstride;
            const size_t i2 = (size_t)(k > 2 ? k - 2 : 0) * a.vstride, i3 = (size_t)(k > 3 ? k - 3 : 0) * a.vstride;
            qc0 = vsc[i0]; qc1 = vsc[i1]; qc2 = vsc[i2]; qc3 = vsc[i3];
            qn0 = vsn[i0]; qn1 = vsn[i1]; qn2 = vsn[i2]; qn3 = vsn[i3];
            level(qc0, qn0, true); level(qc1, qn1, k >= 1); level(qc2, qn2, k >= 2); level(qc3, qn3, k >= 3);
          }
        } else
        for (int k = job_sp - 1; k >= 0; --k) {
          const uint32_t tag = tags[((size_t)(b * P.max_depth + k)) * BLOCK + tid];
          const bool is_glass = tag == TAGV_GLASS, is_fog = tag == TAGV_FOG;
          const bool is_dif = !(is_glass | is_fog);
          const uint32_t prim = is_dif ? tag : 0u;  // the others read record 0 and do not use it
          const ShadeRec sv = (BVH ? MIPT_SC(shade) : shade_tab)[prim];
          V4 N = {sv.nx, sv.ny, sv.nz, sv.nw};
          if (MIPT_SC(n_spheres) > 0 && (int)prim >= MIPT_SC(n_tris)) {
            const float* q = sph_n + ((size_t)k * BLOCK + tid) * 4;
            N = {q[0], q[1], q[2], q[3]};
          }
          const float dn = dot4(dfin, N);
          const float w = is_glass ? glass_weight : (is_fog ? fog_albedo : diffuse_weight);
          float tr_ = is_dif ? cr * sv.r : cr, tg_ = is_dif ? cg * sv.g : cg, tb_ = is_dif ? cb * sv.b : cb;
          tr_ = is_dif ? tr_ * dn : tr_; tg_ = is_dif ? tg_ * dn : tg_; tb_ = is_dif ? tb_ * dn : tb_;
          cr = tr_ * w; cg = tg_ * w; cb = tb_ * w;
        }
        const float sr = park[c0 + 0][tid] + cr, sg = park[c0 + 1][tid] + cg, sb = park[c0 + 2][tid] + cb;  // main.cpp:137
        park[c0 + 0][tid] = sr; park[c0 + 1][tid] = sg; park[c0 + 2][tid] = sb;
        if (js & JS_LAST) {  // the job's item was waiting for this path alone
          float* dst = a.partial + (size_t)__float_as_uint(park[c0 + 3][tid]) * 3;
          dst[0] = sr; dst[1] = sg; dst[2] = sb;
        }
        js &= JS_CUR | JS_CTX;
      }
    }
    if (defer && want_job)  // the path's buffers become the job's; the next path writes the other pair
      js = ((js & (JS_CUR | JS_CTX)) ^ JS_CUR) | JS_PEND | ((js & JS_CTX) ? JS_JCTX : 0u) | ((ds & 0xFFu) << 8);
    if (path_done) {
      s++;
      fresh = true;
      if (s >= s_end) {
        // the item's samples were added in sample order by this one lane: plain store
        have = false;
        if (!LIST) {
          if ((js & JS_PEND) && (((js >> 3) ^ (js >> 1)) & 1u) == 0) {
            js |= JS_LAST;  // the sum is complete once the job has run
          } else {
            const int c0 = (js & JS_CTX) ? 11 : 0;
            float* dst = a.partial + (size_t)__float_as_uint(park[c0 + 3][tid]) * 3;
            dst[0] = park[c0 + 0][tid]; dst[1] = park[c0 + 1][tid]; dst[2] = park[c0 + 2][tid];
          }
        }
      }
    }
    prof.mark(PF_UNWIND);
  }
#ifdef MIPT_PROFILE_SECTIONS
  __syncthreads();
  if (tid < PF_COUNT && prof_lds[tid]) atomicAdd(a.stats + ST_COUNT + tid, (unsigned long long)prof_lds[tid]);
#endif

  if (LIST) {
    if (list_i < a.n_paths) {
      a.rgb[3 * list_i + 0] = park[0][tid]; a.rgb[3 * list_i + 1] = park[1][tid]; a.rgb[3 * list_i + 2] = park[2][tid];
      if (a.counts) { a.counts[2 * list_i] = st[ST_CLOSEST]; a.counts[2 * list_i + 1] = st[ST_SHADOW]; }
    }
  }

#undef MIPT_VTOP
#undef MIPT_SC
  if (STATS) {
    // wave-level sums, one atomic per counter per wave
#pragma unroll
    for (int k = 0; k < ST_COUNT; ++k) {
      unsigned long long v = st[k];
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if ((tid & 63) == 0 && v) atomicAdd(a.stats + k, v);
    }
  }
}

// ---------------------------------------------------------------------------------------
// Device-side self-check of the fast arithmetic against the literal one, at sizes no CPU oracle
// can reach. out[0]: floats b (all 2^32 bit patterns) whose rcp_exact differs from IEEE 1.0f/b;
// out[1]: largest error of the hardware estimate v_rcp_f32, in units of 2^-10 ulp, over the
// fast-path range; out[2] / out[3]: generated rays (camera rays, interior rays, origins on triangles
// and on edges, rays grazing a triangle's plane) whose FAST scan result (index, distance,
// barycentrics) differs from the literal scan, closest-hit / any-hit rule.

__global__ __launch_bounds__(BLOCK) void k_selfcheck_rcp(unsigned long long* out) {
  const uint32_t stride = gridDim.x * BLOCK;
  uint32_t bad = 0, worst = 0;
  uint32_t bits = blockIdx.x * BLOCK + threadIdx.x;
  for (uint32_t it = 0; it < (1u << 20); ++it, bits += stride) {  // gridDim.x * BLOCK == 2^12
    const float b = __uint_as_float(bits);
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float fast = rcp_exact(b, r0);
    const float ref = 1.0f / b;
    const bool same = (__float_as_uint(fast) == __float_as_uint(ref)) || (fast != fast && ref != ref);
    bad += same ? 0u : 1u;
    const float ab = fabsf(b);
    if (ab >= 7.888609052210118e-31f && ab <= 1.2676506002282294e30f) {
      const double exact = 1.0 / (double)b;
      const double ulp = (double)__uint_as_float((__float_as_uint(ref) & 0x7F800000u)) * 1.1920928955078125e-7;
      const double err = fabs((double)r0 - exact) / ulp;
      const uint32_t q = (uint32_t)(err * 1024.0);
      worst = q > worst ? q : worst;
    }
  }
  if (bad) atomicAdd(out + 0, (unsigned long long)bad);
  atomicMax(out + 1, (unsigned long long)worst);
}

// out[6]: floats in [0, 2 pi] for which the fused-arithmetic sincos differs from the plain one
__global__ __launch_bounds__(BLOCK) void k_selfcheck_sincos(unsigned long long* out) {
  const uint32_t stride = gridDim.x * BLOCK;
  uint32_t bad = 0;
  for (uint32_t bits = blockIdx.x * BLOCK + threadIdx.x; bits <= 0x40C90FDBu; bits += stride) {
    const float y = __uint_as_float(bits);
    float s0, c0, s1, c1;
    sincosf_glibc_impl<false>(y, s0, c0);
    sincosf_glibc_impl<true>(y, s1, c1);
    bad += (__float_as_uint(s0) != __float_as_uint(s1) || __float_as_uint(c0) != __float_as_uint(c1)) ? 1u : 0u;
  }
  if (bad) atomicAdd(out + 6, (unsigned long long)bad);
}

template <bool BVH>
__global__ __launch_bounds__(BLOCK) void k_selfcheck_scan(DeviceScene sc, unsigned long long n_rays, uint32_t seed,
                                                          unsigned long long* out) {
  const unsigned long long gid = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
  if (gid >= n_rays) return;
  // ray generator: five families chosen by the ray number
  uint32_t k = path_key(seed, (uint32_t)gid, (uint32_t)(gid >> 32));
  auto u01 = [&]() { k = mix32(k + 0x9E3779B9u); return (float)(k >> 8) * (1.0f / 16777216.0f); };
  V4 o, d;
  const int family = (int)(gid % 5ull);
  if (family == 0) {  // camera rays of a 1024^2 frame, direction not normalised
    o = {0.f, 0.f, -3.f, 1.f};
    d = {(u01() - 0.5f) * 1024.f, (u01() - 0.5f) * 1024.f, 1024.f, 0.f};
  } else if (family == 4 && sc.n_tris > 0) {
    // grazing rays: within ~0.01 degrees of a random triangle's plane, starting in or next to that
    // plane, aimed past the triangle at up to two edge lengths from it -- the rays for which the
    // reference's test is worst conditioned (its barycentrics are rounding noise), inside and
    // outside the triangle's box
    const TriRec T = sc.tris[(uint32_t)(u01() * (float)sc.n_tris) % (uint32_t)sc.n_tris];
    const V3 e1 = {T.e1x, T.e1y, T.e1z}, e2 = {T.e2x, T.e2y, T.e2z};
    V3 n = cross3(e1, e2);
    const float ln = sqrtf(dot3(n, n)), l1 = sqrtf(dot3(e1, e1));
    const float in = ln > 0.f ? 1.0f / ln : 0.f, i1 = l1 > 0.f ? 1.0f / l1 : 0.f;
    n = {n.x * in, n.y * in, n.z * in};
    const V3 a1 = {e1.x * i1, e1.y * i1, e1.z * i1};
    const V3 a2 = cross3(n, a1);
    const float phi = 6.2831853f * u01(), cp = cosf(phi), sp = sinf(phi);
    const V3 w = {cp * a1.x + sp * a2.x, cp * a1.y + sp * a2.y, cp * a1.z + sp * a2.z};
    const float a = u01() * 4.0f - 1.5f, b = u01() * 4.0f - 1.5f;
    const float back = u01() * 3.0f * l1;
    const uint32_t sel = (uint32_t)(u01() * 4.0f);
    const float h = (sel == 0u ? 0.f : (u01() - 0.5f) * (sel == 1u ? 2e-6f : 2e-3f)) * l1;
    const float tilt = sel == 3u ? 0.f : (u01() - 0.5f) * 4e-4f;
    o = {(T.v0x + a * e1.x) + b * e2.x - back * w.x + h * n.x, (T.v0y + a * e1.y) + b * e2.y - back * w.y + h * n.y,
         (T.v0z + a * e1.z) + b * e2.z - back * w.z + h * n.z, 1.f};
    d = normalize4({w.x + tilt * n.x, w.y + tilt * n.y, w.z + tilt * n.z, u01() < 0.5f ? 0.f : 1.0f});
  } else {
    if (family == 1 || family == 4 || sc.n_tris == 0) {  // incoherent rays from inside the box
      o = {u01() * 1.9f - 0.95f, u01() * 1.9f - 0.95f, u01() * 1.9f - 0.95f, 1.f};
    } else {  // origin exactly on a triangle (family 2) or exactly on one of its edges (family 3)
      const TriRec T = sc.tris[(uint32_t)(u01() * (float)sc.n_tris) % (uint32_t)sc.n_tris];
      float a = u01(), b = u01() * (1.0f - a);
      if (family == 3) b = 0.f;
      o = {(T.v0x + a * T.e1x) + b * T.e2x, (T.v0y + a * T.e1y) + b * T.e2y, (T.v0z + a * T.e1z) + b * T.e2z, 1.f};
    }
    const float z = 1.0f - 2.0f * u01(), phi = 6.2831853f * u01();
    const float rxy = sqrtf(fmaxf(0.f, 1.0f - z * z));
    d = normalize4({rxy * cosf(phi), rxy * sinf(phi), z, 1.0f});
  }
  const float max_t = u01() * 2.5f;
  for (int mode = 0; mode < 2; ++mode) {
    HitAcc f, e, g;
    f.set_lim(mode ? max_t : 3.402823466e+38f); f.idx = mode ? 0 : -1; f.by = f.bz = 0.f;
    e = f; g = f;
    uint32_t nt = 0, ng = 0;
    Prof prof{};
    scan_triangles<false, SCAN_LITERAL>(sc, o, d, mode != 0, e, nt, ng, prof, sc.tris);
    scan_triangles<false, SCAN_FAST>(sc, o, d, mode != 0, f, nt, ng, prof, sc.tris);
    uint32_t redo = 0, overflow = 0;
    scene_query<false, BVH>(sc, o, d, mode != 0, g, nt, ng, redo, overflow, prof, sc.tris);  // two-phase scan, or the LBVH
    if (redo) atomicAdd(out + 4, (unsigned long long)redo);
    if (overflow) atomicAdd(out + 5, (unsigned long long)overflow);
    auto same = [](const HitAcc& a, const HitAcc& b) {
      return a.idx == b.idx && __float_as_uint(a.lim) == __float_as_uint(b.lim) &&
             __float_as_uint(a.by) == __float_as_uint(b.by) && __float_as_uint(a.bz) == __float_as_uint(b.bz);
    };
    // any-hit through the tree reports only whether a triangle accepts, not which
    const bool g_ok = (mode && sc.bvh) ? (g.idx == e.idx) : same(g, e);
    // The tree on grazing rays is counted apart (out[7]): a triangle whose box such a ray never
    // enters can still be accepted by the reference's test out of pure rounding noise, and the walk
    // cannot know. The scans without a tree must be exact for every family.
    const bool apart = BVH && family == 4;
    if (!same(f, e) || (!g_ok && !apart)) atomicAdd(out + 2 + mode, 1ull);
    if (!g_ok && apart) atomicAdd(out + 7, 1ull);
  }
}

// accum[i] = (((accum[i] + partial[0][i]) + partial[1][i]) + ...): the item sums join the running sum one
// by one, in sample order. Adding samples [a, b) and then [b, c) therefore leaves the very bits that
// adding [a, c) at once leaves, whenever b - a is a whole number of work items.
__global__ __launch_bounds__(BLOCK) void k_reduce_chunks(const float* __restrict__ partial, uint32_t n_chunks,
                                                         size_t n, float* __restrict__ accum) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  float s = accum[i];
  for (uint32_t c = 0; c < n_chunks; ++c) s += partial[(size_t)c * n + i];
  accum[i] = s;
}

// main.cpp:141-145 + PutPixelSDL (SDLauxiliary.h:149-161)
__global__ __launch_bounds__(BLOCK) void k_resolve_argb(const float* __restrict__ accum, int W, int H,
                                                        uint32_t n_samples, uint32_t* __restrict__ screen) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= (size_t)W * H) return;
  const int x = (int)(i % W), y = (int)(i / W);
  const float div = (float)(int)n_samples * (1.0f / (2.0f * 3.1415f));
  const float* a = accum + ((size_t)x * H + y) * 3;
  uint32_t ch[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float c = 255.0f * (a[k] / div);
    const float lo = (c < 0.f) ? 0.f : c;        // glm::clamp = min(max(x, 0), 255)
    const float cl = (255.f < lo) ? 255.f : lo;
    ch[k] = (cl == cl) ? (uint32_t)cl : 0u;
  }
  screen[i] = (128u << 24) + (ch[0] << 16) + (ch[1] << 8) + ch[2];
}

}  // namespace mipt
