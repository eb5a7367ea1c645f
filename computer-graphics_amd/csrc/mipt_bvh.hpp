// mipt_bvh.hpp -- GPU-built LBVH and its traversal, for scenes too large for the linear scan.
//
// The reference has no acceleration structure (AABB.h is included by main.cpp:40 but never
// used); its closestIntersection is a linear scan whose result depends on the ORDER of the
// triangles only through near-ties. The traversal here therefore does not reimplement the
// reference's rule in tree order; it uses the tree to find the few triangles whose reference
// test can succeed near the front of the ray, evaluates those with the reference's own
// arithmetic, and replays the reference's sequential rule over them in index order:
//
//   build     triangle bounds -> 63-bit Morton keys of the centroids -> radix sort (rocPRIM) ->
//             Karras' parallel radix-tree construction -> bottom-up refit, one launch per level
//             wave (kernel boundaries order the writes; no inter-workgroup flags).
//   node      the binary tree is collapsed into nodes of up to four children, 64 bytes each with
//             the children's boxes on a per-node 8-bit grid (BvhNode below); references are
//             >= 0 inner node, < 0 leaf holding triangle -1 - ref.
//   leaf box  the triangle's bounds grown by 3 % of its extent: the candidate filter
//             (tri_candidate) admits hits up to 2 % of an edge outside the triangle.
//   closest   stack traversal, near child first; at a leaf the candidate filter, then the
//             reference's determinant ratio t, inverse-path distance z = |x.x| and barycentrics.
//             A triangle the reference could accept (t >= 0, barycentrics valid) is kept in a
//             small list sorted by index; boxes and hits beyond zmin (1 + 2^-9) + 4 delta are
//             dropped, zmin = smallest z so far and delta an absolute slack for the rounding noise
//             of hits at t ~ 0. At the end the reference's rule `t <= min -> min = z` runs over the
//             list in index order. While every hit met has |t - z| <= 2^-12 z + delta this equals
//             the rule run over all triangles: a dropped hit lies behind every kept one by more
//             than the noise, so it is either rejected by the reference too, or accepted before the
//             closest hit and then overridden by it. A hit whose t and z disagree more (ill-
//             conditioned for this ray), or a full list, sends the ray through the exhaustive
//             traversal: no pruning, one pass over the ray's boxes per triangle the reference
//             accepts.
//   any-hit   existsIntersection is an OR over the triangles (its `break` only shortens the
//             loop), so the traversal stops at the first leaf whose reference test accepts.
//
// Equality with the linear scan is exact except in one corner: a triangle whose padded box the
// ray never enters, or enters beyond the pruning distance, but whose reference test is so
// ill-conditioned for this ray (ray within ~0.01 degrees of the triangle's plane AND nearly in
// that plane) that its rounding noise alone produces an accepted hit in front of the true one.
#pragma once

#include "mipt_device.hpp"

namespace mipt {

// binary node of the radix tree: build-time only
struct __attribute__((aligned(64))) BvhBin {
  float lmin[3], lmax[3], rmin[3], rmax[3];
  int left, right;
  int pad0, pad1;
};

// What the walk reads: the binary tree collapsed two levels at a time into nodes of up to four
// children (half as many dependent fetches per ray). Child references: >= 0 node, < 0 leaf holding
// triangle -1 - ref, BVH_DONE = empty slot, never hit.
#ifndef MIPT_BVH_QUANT
#define MIPT_BVH_QUANT 1
#endif
#ifndef MIPT_BVH_GREEDY
#define MIPT_BVH_GREEDY 1
#endif
#if MIPT_BVH_QUANT
// 64 bytes: the children's boxes on a per-node grid of 256 steps per axis. A bound is
// org[axis] + q * 2^(e[axis] - 127), decoded with one fused multiply-add; the build rounds lower
// bounds down and upper bounds up against that very expression, so a decoded box always contains
// the exact one (the walk visits a few more nodes, never fewer). One node = four 16-byte loads
// per lane: the walk's rays diverge, so its cost is the number of lines the L1 has to look up.
struct __attribute__((aligned(64))) BvhNode {
  float org[3];
  uint32_t ex;      // byte k: biased exponent of axis k's grid step
  uint32_t qlo[3];  // word k, byte c: lower bound of child c on axis k, in grid steps
  uint32_t qhi[3];
  int ref[4];
  int pad[2];
};
#else
// 128 bytes: boxes as lo[axis][child] / hi[axis][child] in full precision
struct __attribute__((aligned(128))) BvhNode {
  float lo[3][4];
  float hi[3][4];
  int ref[4];
  int pad[4];
};
#endif

struct Aabb { float lo[3], hi[3]; };

// ---------------------------------------------------------------------------------------
// build kernels

__device__ __forceinline__ uint32_t float_flip(float f) {  // order-preserving float -> uint
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float float_unflip(uint32_t u) {
  const uint32_t v = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
  float f;
#if defined(__HIP_DEVICE_COMPILE__)
  f = __uint_as_float(v);
#else
  memcpy(&f, &v, 4);
#endif
  return f;
}

// per triangle: padded bounds; scene bounds of the centroids by atomics on flipped floats
// `subset` lists the triangles that go into the tree; box is indexed by triangle, centroid by position
__global__ __launch_bounds__(BLOCK) void k_bvh_bounds(const TriRec* __restrict__ tris, const int* __restrict__ subset,
                                                      int n, Aabb* __restrict__ box, float* __restrict__ centroid,
                                                      uint32_t* __restrict__ scene) {
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  const bool valid = i < n;  // lanes past the end stay for the wave-wide reduction below
  const int tri = valid ? subset[i] : 0;
  const TriRec T = valid ? tris[tri] : TriRec{};
  const float v0[3] = {T.v0x, T.v0y, T.v0z};
  const float e1[3] = {T.e1x, T.e1y, T.e1z}, e2[3] = {T.e2x, T.e2y, T.e2z};
  Aabb b;
  float ext = 0.f, mag = 0.f;
  for (int k = 0; k < 3; ++k) {
    const float a = v0[k], c = v0[k] + e1[k], d = v0[k] + e2[k];
    b.lo[k] = fminf(a, fminf(c, d));
    b.hi[k] = fmaxf(a, fmaxf(c, d));
    ext = fmaxf(ext, b.hi[k] - b.lo[k]);
    mag = fmaxf(mag, fmaxf(fabsf(b.lo[k]), fabsf(b.hi[k])));
  }
  const float pad = 0.03f * ext + 1e-5f * mag + 1e-30f;
  for (int k = 0; k < 3; ++k) {
    const float c = 0.5f * (b.lo[k] + b.hi[k]);
    // scene bounds of the centroids: reduced over the wave first, one atomic pair per wave and axis
    // (an atomic per triangle on six addresses took 1.1 ms of a 3.9 ms build of 10^6 triangles)
    uint32_t lo = valid ? float_flip(c) : 0xFFFFFFFFu, hi = valid ? float_flip(c) : 0u;
    for (int off = 32; off > 0; off >>= 1) {
      const uint32_t lo2 = (uint32_t)__shfl_xor((int)lo, off, 64), hi2 = (uint32_t)__shfl_xor((int)hi, off, 64);
      lo = lo2 < lo ? lo2 : lo;
      hi = hi2 > hi ? hi2 : hi;
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMin(scene + k, lo);
      atomicMax(scene + 3 + k, hi);
    }
    if (valid) centroid[3 * i + k] = c;
    b.lo[k] -= pad;
    b.hi[k] += pad;
  }
  if (valid) box[tri] = b;
}

__device__ __forceinline__ unsigned long long spread21(unsigned long long v) {  // 21 bits -> every third bit
  v &= 0x1FFFFFull;
  v = (v | v << 32) & 0x1F00000000FFFFull;
  v = (v | v << 16) & 0x1F0000FF0000FFull;
  v = (v | v << 8) & 0x100F00F00F00F00Full;
  v = (v | v << 4) & 0x10C30C30C30C30C3ull;
  v = (v | v << 2) & 0x1249249249249249ull;
  return v;
}

__global__ __launch_bounds__(BLOCK) void k_bvh_morton(const float* __restrict__ centroid, int n,
                                                      const uint32_t* __restrict__ scene,
                                                      const int* __restrict__ subset,
                                                      unsigned long long* __restrict__ keys, int* __restrict__ ids) {
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  unsigned long long code = 0;
  for (int k = 0; k < 3; ++k) {
    const float lo = float_unflip(scene[k]), hi = float_unflip(scene[3 + k]);
    const float w = hi - lo;
    float u = (w > 0.f) ? (centroid[3 * i + k] - lo) / w : 0.f;
    u = fminf(fmaxf(u, 0.f), 1.f);
    const unsigned long long q = (unsigned long long)fminf(u * 2097152.0f, 2097151.0f);
    code |= spread21(q) << k;
  }
  keys[i] = code;
  ids[i] = subset[i];
}

// common-prefix length of sorted keys i and j; equal keys are told apart by their position
__device__ __forceinline__ int bvh_delta(const unsigned long long* keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const unsigned long long a = keys[i], b = keys[j];
  if (a == b) return 64 + __clz((unsigned)(i ^ j));
  return __clzll((long long)(a ^ b));
}

// Karras 2012, "Maximizing parallelism in the construction of BVHs, octrees and k-d trees"
__global__ __launch_bounds__(BLOCK) void k_bvh_topology(const unsigned long long* __restrict__ keys,
                                                        const int* __restrict__ ids, int n,
                                                        BvhBin* __restrict__ nodes) {
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n - 1) return;
  const int d = (bvh_delta(keys, n, i, i + 1) - bvh_delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  const int dmin = bvh_delta(keys, n, i, i - d);
  int lmax = 2;
  while (bvh_delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1)
    if (bvh_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = bvh_delta(keys, n, i, j);
  int s = 0;
  for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
    if (bvh_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    if (t <= 1) break;
  }
  const int gamma = i + s * d + (d < 0 ? d : 0);
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  // a child that covers a single sorted key is a leaf holding that key's triangle
  nodes[i].left = (lo == gamma) ? -1 - ids[gamma] : gamma;
  nodes[i].right = (hi == gamma + 1) ? -1 - ids[gamma + 1] : gamma + 1;
}

// one refit wave: every inner node whose children both have boxes computes its own.
// node_box[0..n-2] inner, ready[] likewise; leaves take their box from tri_box.
__global__ __launch_bounds__(BLOCK) void k_bvh_refit(BvhBin* __restrict__ nodes, int n, const Aabb* __restrict__ tri_box,
                                                     Aabb* __restrict__ node_box, unsigned char* __restrict__ ready,
                                                     const unsigned char* __restrict__ ready_prev,
                                                     int* __restrict__ n_done) {
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n - 1) return;
  if (ready_prev[i]) { ready[i] = 1; return; }
  const int l = nodes[i].left, r = nodes[i].right;
  const bool lok = l < 0 || ready_prev[l], rok = r < 0 || ready_prev[r];
  if (!(lok && rok)) { ready[i] = 0; return; }
  const Aabb a = l < 0 ? tri_box[-1 - l] : node_box[l];
  const Aabb b = r < 0 ? tri_box[-1 - r] : node_box[r];
  Aabb u;
  for (int k = 0; k < 3; ++k) {
    nodes[i].lmin[k] = a.lo[k]; nodes[i].lmax[k] = a.hi[k];
    nodes[i].rmin[k] = b.lo[k]; nodes[i].rmax[k] = b.hi[k];
    u.lo[k] = fminf(a.lo[k], b.lo[k]);
    u.hi[k] = fmaxf(a.hi[k], b.hi[k]);
  }
  node_box[i] = u;
  ready[i] = 1;
  atomicAdd(n_done, 1);
}

// ---------------------------------------------------------------------------------------
// Binary tree by parallel locally-ordered clustering (Meister & Bittner 2018, "Parallel Locally-Ordered
// Clustering for Bounding Volume Hierarchy Construction"): bottom-up agglomeration along the Morton
// order. Every cluster looks at its PLOC_RADIUS neighbours on either side for the one whose union
// with it has the smallest surface; two clusters that choose each other merge into a new node; the
// survivors are compacted in order; repeat until one is left. The surfaces of the boxes, not the bits
// of the keys, decide what is grouped, which costs a ray a third fewer nodes on a mesh than the radix
// tree over the same keys. Depth: every cluster carries the depth of its subtree, and the builder falls
// back to the radix tree (whose depth is bounded by its key length) if the finished tree is deeper than
// the walk's stack allows, or if the clustering has not finished after PLOC_MAX_ROUNDS rounds.
#ifndef MIPT_BVH_PLOC
#define MIPT_BVH_PLOC 1
#endif
#ifndef MIPT_BVH_PLOC_RADIUS
#define MIPT_BVH_PLOC_RADIUS 16
#endif
constexpr int PLOC_RADIUS = MIPT_BVH_PLOC_RADIUS;
// The clustered tree is kept when its surface-area cost is below this share of the radix tree's. Measured,
// whole paths: cost 0.83 -> +4 %, 0.86 -> +5.5 %, but 0.91 (a closed mesh in a room, three quarters of
// whose rays never enter the tree) -> -1 ... -2 %: a few per cent of cost do not pay for the
// clustered tree's less regular shape.
constexpr double PLOC_KEEP_RATIO = 0.9;
constexpr int PLOC_MAX_ROUNDS = 512;
constexpr int PLOC_MAX_DEPTH = 47;  // binary levels: the walk's stack holds 3 entries for each of 48 four-wide levels

__device__ __forceinline__ float aabb_union_area(const Aabb& a, const Aabb& b) {
  const float dx = fmaxf(a.hi[0], b.hi[0]) - fminf(a.lo[0], b.lo[0]);
  const float dy = fmaxf(a.hi[1], b.hi[1]) - fminf(a.lo[1], b.lo[1]);
  const float dz = fmaxf(a.hi[2], b.hi[2]) - fminf(a.lo[2], b.lo[2]);
  return dx * dy + dy * dz + dz * dx;  // the same bits whichever box comes first
}

// the leaves as clusters, in Morton order: cluster i = triangle ids[i] with its padded box
__global__ __launch_bounds__(BLOCK) void k_ploc_init(const Aabb* __restrict__ tri_box, const int* __restrict__ ids, int n,
                                                     Aabb* __restrict__ cbox, int2* __restrict__ cid) {
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  cbox[i] = tri_box[ids[i]];
  cid[i] = make_int2(-1 - ids[i], 0);  // (reference, depth of the subtree)
}

// nearest neighbour within the radius; of equally near ones the lowest position. (With a symmetric
// distance and this tie rule the lowest-positioned cluster among those at the globally smallest
// distance and its choice choose each other: every round merges at least one pair.)
__global__ __launch_bounds__(BLOCK) void k_ploc_nearest(const Aabb* __restrict__ cbox, int m, int* __restrict__ nn) {
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= m) return;
  const Aabb a = cbox[i];
  const int lo = i - PLOC_RADIUS > 0 ? i - PLOC_RADIUS : 0, hi = i + PLOC_RADIUS < m - 1 ? i + PLOC_RADIUS : m - 1;
  float best = __builtin_inff();
  int bj = -1;
  for (int j = lo; j <= hi; ++j) {
    if (j == i) continue;
    const float d = aabb_union_area(a, cbox[j]);
    if (d < best || bj < 0) { best = d; bj = j; }
  }
  nn[i] = bj;
}

// mutual choices merge: the lower position keeps the merged cluster (a new binary node), the higher
// one leaves; keep[i] = 1 for every cluster that goes on to the next round
__global__ __launch_bounds__(BLOCK) void k_ploc_merge(Aabb* __restrict__ cbox, int2* __restrict__ cid,
                                                      const int* __restrict__ nn, int m, BvhBin* __restrict__ bin,
                                                      int* __restrict__ n_nodes, uint32_t* __restrict__ keep) {
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= m) return;
  const int j = nn[i];
  if (j < 0 || nn[j] != i) { keep[i] = 1u; return; }
  if (i > j) { keep[i] = 0u; return; }
  const Aabb a = cbox[i], b = cbox[j];
  const int k = atomicAdd(n_nodes, 1);
  BvhBin N;
  Aabb u;
  for (int c = 0; c < 3; ++c) {
    N.lmin[c] = a.lo[c]; N.lmax[c] = a.hi[c];
    N.rmin[c] = b.lo[c]; N.rmax[c] = b.hi[c];
    u.lo[c] = fminf(a.lo[c], b.lo[c]);
    u.hi[c] = fmaxf(a.hi[c], b.hi[c]);
  }
  const int2 ci = cid[i], cj = cid[j];
  N.left = ci.x; N.right = cj.x;
  N.pad0 = 0; N.pad1 = 0;
  bin[k] = N;
  cbox[i] = u;  // (only this thread reads or writes clusters i and j in this launch)
  cid[i] = make_int2(k, 1 + (ci.y > cj.y ? ci.y : cj.y));
  keep[i] = 1u;
}

// the survivors, in order: pos = exclusive prefix sum of keep
__global__ __launch_bounds__(BLOCK) void k_ploc_compact(const Aabb* __restrict__ cbox, const int2* __restrict__ cid,
                                                        const uint32_t* __restrict__ keep, const uint32_t* __restrict__ pos,
                                                        int m, Aabb* __restrict__ cbox_out, int2* __restrict__ cid_out) {
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= m || !keep[i]) return;
  cbox_out[pos[i]] = cbox[i];
  cid_out[pos[i]] = cid[i];
}

// Surface-area cost of a binary tree: the sum of the surfaces of all child boxes (up to the common
// factor of the root's surface, the expected number of boxes a random line through the scene meets).
// One partial sum per block, added in a fixed order: the same tree always gets the same cost.
__global__ __launch_bounds__(BLOCK) void k_bin_cost(const BvhBin* __restrict__ bin, int n_nodes, double* __restrict__ partial) {
  __shared__ double part[BLOCK];
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  double c = 0.0;
  if (i < n_nodes) {
    const BvhBin B = bin[i];
    const float lx = B.lmax[0] - B.lmin[0], ly = B.lmax[1] - B.lmin[1], lz = B.lmax[2] - B.lmin[2];
    const float rx = B.rmax[0] - B.rmin[0], ry = B.rmax[1] - B.rmin[1], rz = B.rmax[2] - B.rmin[2];
    c = ((double)lx * ly + (double)ly * lz + (double)lz * lx) + ((double)rx * ry + (double)ry * rz + (double)rz * rx);
  }
  part[threadIdx.x] = c;
  __syncthreads();
  for (int w = BLOCK / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = part[0];
}

// One wave of the collapse: every frontier entry (binary node b, wide slot w) fills wide node w with
// b's grandchildren (a child that is a leaf stays as it is), gives every inner grandchild a new
// wide slot and puts it on the next frontier. Launched level by level until the frontier is empty.
__global__ __launch_bounds__(BLOCK) void k_bvh_collapse(const BvhBin* __restrict__ bin,
                                                        const int2* __restrict__ frontier_in, int n_in,
                                                        int2* __restrict__ frontier_out, int* __restrict__ n_out,
                                                        BvhNode* __restrict__ wide, int* __restrict__ n_wide) {
  const int j = blockIdx.x * BLOCK + threadIdx.x;
  if (j >= n_in) return;
  const int b = frontier_in[j].x, w = frontier_in[j].y;
  const BvhBin B = bin[b];
  int ref[4];
  float lo[4][3], hi[4][3];
  ref[0] = B.left; ref[1] = B.right;
  for (int k = 0; k < 3; ++k) { lo[0][k] = B.lmin[k]; hi[0][k] = B.lmax[k]; lo[1][k] = B.rmin[k]; hi[1][k] = B.rmax[k]; }
  int m = 2;
#if MIPT_BVH_GREEDY
  // open the inner child with the largest surface until the node has four children (or only leaves)
  while (m < 4) {
    int pick = -1;
    float best = -1.f;
    for (int c = 0; c < m; ++c) {
      if (ref[c] < 0) continue;
      const float dx = hi[c][0] - lo[c][0], dy = hi[c][1] - lo[c][1], dz = hi[c][2] - lo[c][2];
      const float area = dx * dy + dy * dz + dz * dx;
      if (pick < 0 || area > best) { pick = c; best = area; }
    }
    if (pick < 0) break;
    const BvhBin C = bin[ref[pick]];
    ref[pick] = C.left;
    ref[m] = C.right;
    for (int k = 0; k < 3; ++k) { lo[pick][k] = C.lmin[k]; hi[pick][k] = C.lmax[k]; lo[m][k] = C.rmin[k]; hi[m][k] = C.rmax[k]; }
    ++m;
  }
#else
  // both children give way to their own children
  {
    const int l = B.left, r = B.right;
    if (r >= 0) {
      const BvhBin C = bin[r];
      ref[1] = C.left; ref[m] = C.right;
      for (int k = 0; k < 3; ++k) { lo[1][k] = C.lmin[k]; hi[1][k] = C.lmax[k]; lo[m][k] = C.rmin[k]; hi[m][k] = C.rmax[k]; }
      ++m;
    }
    if (l >= 0) {
      const BvhBin C = bin[l];
      ref[0] = C.left; ref[m] = C.right;
      for (int k = 0; k < 3; ++k) { lo[0][k] = C.lmin[k]; hi[0][k] = C.lmax[k]; lo[m][k] = C.rmin[k]; hi[m][k] = C.rmax[k]; }
      ++m;
    }
  }
#endif
  BvhNode N;
  // the inner children get consecutive slots: siblings a ray visits together share lines
  int n_inner = 0;
  for (int c = 0; c < m; ++c) n_inner += ref[c] >= 0;
  int slot = n_inner ? atomicAdd(n_wide, n_inner) : 0;
  int entry = n_inner ? atomicAdd(n_out, n_inner) : 0;
  for (int c = 0; c < 4; ++c) {
    if (c < m && ref[c] >= 0) {
      frontier_out[entry++] = make_int2(ref[c], slot);
      N.ref[c] = slot++;
    } else {
      N.ref[c] = c < m ? ref[c] : (int)0x80000000;
    }
  }
#if MIPT_BVH_QUANT
  N.ex = 0;
  N.pad[0] = N.pad[1] = 0;
  const float inf = __builtin_inff();
  for (int k = 0; k < 3; ++k) {
    float org = inf, top = -inf;
    for (int c = 0; c < m; ++c) { org = fminf(org, lo[c][k]); top = fmaxf(top, hi[c][k]); }
    if (!(fabsf(org) < inf)) org = 0.f;  // boxes of NaN triangles only: nothing to find in them
    // smallest power of two with 255 steps covering the extent, raised until every bound fits
    uint32_t e = ((__float_as_uint((top - org) / 255.f) >> 23) & 0xFFu) + 1u;
    e = e < 1u ? 1u : (e > 254u ? 254u : e);
    uint32_t wlo = 0, whi = 0;
    for (;;) {
      const float step = __uint_as_float(e << 23), inv_step = 1.f / step;
      bool ok = true;
      wlo = 0; whi = 0;
      for (int c = 0; c < 4; ++c) {
        float ql = 255.f, qh = 0.f;  // empty slot: inverted
        if (c < m) {
          ql = fminf(fmaxf(floorf((lo[c][k] - org) * inv_step), 0.f), 255.f);
          while (ql > 0.f && __fmaf_rn(ql, step, org) > lo[c][k]) ql -= 1.f;
          qh = fminf(fmaxf(ceilf((hi[c][k] - org) * inv_step), 0.f), 255.f);
          while (qh < 255.f && __fmaf_rn(qh, step, org) < hi[c][k]) qh += 1.f;
          if (__fmaf_rn(qh, step, org) < hi[c][k]) ok = false;
        }
        wlo |= (uint32_t)ql << (8 * c);
        whi |= (uint32_t)qh << (8 * c);
      }
      if (ok || e >= 254u) break;
      ++e;
    }
    N.org[k] = org;
    N.ex |= e << (8 * k);
    N.qlo[k] = wlo;
    N.qhi[k] = whi;
  }
#else
  for (int c = 0; c < 4; ++c) {
    for (int k = 0; k < 3; ++k) {
      N.lo[k][c] = c < m ? lo[c][k] : __builtin_inff();
      N.hi[k][c] = c < m ? hi[c][k] : -__builtin_inff();
    }
    N.pad[c] = 0;
  }
#endif
  wide[w] = N;
}

// ---------------------------------------------------------------------------------------
// traversal

// The reference's quantities for one triangle, without the sequential gate: t = detM / detA
// (intersections.cpp:23), x = inverse(A) * b (:26). Same operations as tri_test.
template <typename TP>
__device__ __forceinline__ void tri_eval(TP T, float ox, float oy, float oz, float px, float py, float pz,
                                         float& t, float& x0, float& x1, float& x2) {
  const float bx = ox - T->v0x, by = oy - T->v0y, bz = oz - T->v0z;
  const float e1x = T->e1x, e1y = T->e1y, e1z = T->e1z;
  const float e2x = T->e2x, e2y = T->e2y, e2z = T->e2z;
  const float c0 = T->c0;
  const float m1 = by * e2z - e2y * bz;
  const float m2 = by * e1z - e1y * bz;
  const float detM = ((bx * c0) - (e1x * m1)) + (e2x * m2);
  const float k1 = py * e2z - e2y * pz;
  const float k2 = py * e1z - e1y * pz;
  const float detA = ((px * c0) - (e1x * k1)) + (e2x * k2);
  t = detM / detA;
  const float ood = rcp_exact(detA, __builtin_amdgcn_rcpf(detA));
  const float i00 = c0 * ood;
  const float i10 = -(T->c1) * ood;
  const float i20 = T->c2 * ood;
  const float i01 = -k1 * ood;
  const float i11 = (px * e2z - e2x * pz) * ood;
  const float i21 = -(px * e2y - e2x * py) * ood;
  const float i02 = k2 * ood;
  const float i12 = -(px * e1z - e1x * pz) * ood;
  const float i22 = (px * e1y - e1x * py) * ood;
  x0 = (i00 * bx + i10 * by) + i20 * bz;
  x1 = (i01 * bx + i11 * by) + i21 * bz;
  x2 = (i02 * bx + i12 * by) + i22 * bz;
}

struct BvhHit { int idx; float t, z, by, bz; };

// A binary tree over sorted 63-bit keys, equal keys split by their 32-bit position, is at most 95
// levels deep, i.e. 48 levels of four-wide nodes, and the walk pushes at most three entries per
// level: a stack of 144 cannot overflow.
// weights of the burst's vote between a node step and a leaf test
#ifndef MIPT_BVH_VOTE_NODE
#define MIPT_BVH_VOTE_NODE 2
#endif
#ifndef MIPT_BVH_VOTE_LEAF
#define MIPT_BVH_VOTE_LEAF 3
#endif
#ifndef MIPT_BVH_STACK
#define MIPT_BVH_STACK 144
#endif
constexpr int BVH_STACK = MIPT_BVH_STACK;
// a burst of the walk ends when fewer than KEEP_NUM / KEEP_DEN of the lanes that entered it still walk
#ifndef MIPT_BVH_KEEP_NUM
#define MIPT_BVH_KEEP_NUM 3
#endif
#ifndef MIPT_BVH_KEEP_DEN
#define MIPT_BVH_KEEP_DEN 8
#endif
constexpr int BVH_KEEP_NUM = MIPT_BVH_KEEP_NUM, BVH_KEEP_DEN = MIPT_BVH_KEEP_DEN;
constexpr int BVH_DONE = (int)0x80000000;  // traversal finished (leaf references are -1 - index > this)
constexpr int BVH_LIST = 8;               // hits kept by the pruned traversal
constexpr float BVH_BAND = 1.001953125f;  // 1 + 2^-9

__device__ __forceinline__ bool slab(const float lo[3], const float hi[3], V3 o, V3 inv, float bound, float& tnear) {
  const float ax = (lo[0] - o.x) * inv.x, bx = (hi[0] - o.x) * inv.x;
  const float ay = (lo[1] - o.y) * inv.y, by = (hi[1] - o.y) * inv.y;
  const float az = (lo[2] - o.z) * inv.z, bz = (hi[2] - o.z) * inv.z;
  // fminf / fmaxf drop a NaN operand (0 * inf when the origin lies on a slab plane of a ray
  // parallel to it), which keeps the test conservative
  const float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
  const float t1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
  tnear = t0;
  return t0 <= t1 * 1.0000004f + 1e-30f && t1 >= 0.f && t0 <= bound;
}

// One child of a wide node against the ray: entry distance in tnear, true if the box is hit in
// front of the origin and not beyond `bound`.
#if MIPT_BVH_QUANT
// The ray in a node's grid: the distance to the plane `q` steps up axis k is q * ks[k] + kb[k].
// Rounding moves a plane by <= 2^-22 of (node extent + distance to the origin), far inside the
// padding of the leaf boxes. An infinite inverse direction gives NaN or a same-signed infinity on
// both planes of the axis: fminf / fmaxf drop the NaNs (no constraint from that axis, conservative)
// and the infinities reject exactly the rays outside the slab.
struct NodeRay { float ks[3], kb[3]; };
__device__ __forceinline__ NodeRay node_ray(const BvhNode& N, V3 o, V3 inv) {
  NodeRay R;
  const float oo[3] = {o.x, o.y, o.z}, ii[3] = {inv.x, inv.y, inv.z};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    R.ks[k] = __uint_as_float(((N.ex >> (8 * k)) & 0xFFu) << 23) * ii[k];
    R.kb[k] = (N.org[k] - oo[k]) * ii[k];
  }
  return R;
}
__device__ __forceinline__ bool slab4(const BvhNode& N, const NodeRay& R, int c, float bound, float& tnear) {
  float a[3], b[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    a[k] = __fmaf_rn((float)((N.qlo[k] >> (8 * c)) & 0xFFu), R.ks[k], R.kb[k]);
    b[k] = __fmaf_rn((float)((N.qhi[k] >> (8 * c)) & 0xFFu), R.ks[k], R.kb[k]);
  }
  const float t0 = fmaxf(fmaxf(fminf(a[0], b[0]), fminf(a[1], b[1])), fminf(a[2], b[2]));
  const float t1 = fminf(fminf(fmaxf(a[0], b[0]), fmaxf(a[1], b[1])), fmaxf(a[2], b[2]));
  tnear = t0;
  return t0 <= t1 * 1.0000004f + 1e-30f && t1 >= 0.f && t0 <= bound;
}
#else
struct NodeRay { V3 o, inv; };
__device__ __forceinline__ NodeRay node_ray(const BvhNode&, V3 o, V3 inv) { return NodeRay{o, inv}; }
__device__ __forceinline__ bool slab4(const BvhNode& N, const NodeRay& R, int c, float bound, float& tnear) {
  const float lo[3] = {N.lo[0][c], N.lo[1][c], N.lo[2][c]}, hi[3] = {N.hi[0][c], N.hi[1][c], N.hi[2][c]};
  return slab(lo, hi, R.o, R.inv, bound, tnear);
}
#endif

// Where a walk's stack lives. StackScratch: the lane's scratch array. StackLds: the first
// BVH_LDS_STACK entries in a column of LDS (entry e of lane tid at lds[e * BLOCK], conflict free),
// deeper ones in the scratch array -- a pop is on the walk's critical path (its result is the next
// node's address), and the LDS answers in a fraction of the time of a scratch load.
// Measured: 0 / 4 / 8 / 12 / 16 / 20 / 24 entries -> 0.87 / 1.11 / 1.38 / 1.42 / 1.42 / 1.42 / 1.37 Grays/s on
// incoherent rays against 10^6 triangles (deeper than 12 costs occupancy and gains nothing).
#ifndef MIPT_BVH_LDS_STACK
#define MIPT_BVH_LDS_STACK 12
#endif
constexpr int BVH_LDS_STACK = MIPT_BVH_LDS_STACK;
struct StackScratch {
  int* s;
  __device__ __forceinline__ void push(int& sp, int v) const { s[sp++] = v; }
  __device__ __forceinline__ int pop(int& sp) const { return s[--sp]; }
};
struct StackLds {
  int* lds;  // this lane's column
  int* s;
  __device__ __forceinline__ void push(int& sp, int v) const {
    if (sp < BVH_LDS_STACK) lds[sp * BLOCK] = v; else s[sp] = v;
    ++sp;
  }
  __device__ __forceinline__ int pop(int& sp) const {
    --sp;
    return sp < BVH_LDS_STACK ? lds[sp * BLOCK] : s[sp];
  }
};

// A step through an inner node: the nearest child hit becomes the next node, the other hit children
// go on the stack (farthest first, so that nearer ones are walked earlier and prune more). The four
// (entry distance, reference) pairs go through a five-exchange sorting network with misses at
// +inf; nothing is indexed by a run-time value, so the node stays in registers.
// Returns false if the stack is full (cannot happen: <= 3 pushes per level, <= 48 levels).
__device__ __forceinline__ void bvh_cswap(float& ta, int& ra, float& tb, int& rb) {
  const bool s = tb < ta;
  const float t = s ? tb : ta, u = s ? ta : tb;
  const int r = s ? rb : ra, q = s ? ra : rb;
  ta = t; tb = u; ra = r; rb = q;
}
template <typename STK>
__device__ __forceinline__ bool bvh_step(const BvhNode& N, V3 o, V3 inv, float bound, int& cur, const STK& stack, int& sp) {
  const float inf = __builtin_inff();
  float t0, t1, t2, t3;
  const NodeRay R = node_ray(N, o, inv);
  t0 = (slab4(N, R, 0, bound, t0) && N.ref[0] != BVH_DONE) ? t0 : inf;  // empty slots never hit
  t1 = (slab4(N, R, 1, bound, t1) && N.ref[1] != BVH_DONE) ? t1 : inf;
  t2 = (slab4(N, R, 2, bound, t2) && N.ref[2] != BVH_DONE) ? t2 : inf;
  t3 = (slab4(N, R, 3, bound, t3) && N.ref[3] != BVH_DONE) ? t3 : inf;
  int r0 = N.ref[0], r1 = N.ref[1], r2 = N.ref[2], r3 = N.ref[3];
  bvh_cswap(t0, r0, t1, r1);
  bvh_cswap(t2, r2, t3, r3);
  bvh_cswap(t0, r0, t2, r2);
  bvh_cswap(t1, r1, t3, r3);
  bvh_cswap(t1, r1, t2, r2);
  if (!(t0 < inf)) {
    cur = sp ? stack.pop(sp) : BVH_DONE;
    return true;
  }
  if (sp + 3 > BVH_STACK) return false;
  if (t3 < inf) stack.push(sp, r3);
  if (t2 < inf) stack.push(sp, r2);
  if (t1 < inf) stack.push(sp, r1);
  cur = r0;
  return true;
}

struct BvhRay {
  V4 o, d;
  V3 oo, inv;
  float px, py, pz, Dm, delta;
  uint32_t n_nodes, n_leaves;  // work counters (dead code unless the caller reads them)
};

// (the scene arguments of the walk's functions are POINTERS to a DeviceScene, in whatever address space
// the caller has it: a kernel that is short of scalar registers hands in its argument block and the
// fields are fetched where they are used, see late_args() in mipt_kernels.hpp)
__device__ __forceinline__ BvhRay bvh_ray(float scene_scale, V4 o, V4 d) {
  BvhRay r;
  r.o = o; r.d = d;
  r.px = -d.x; r.py = -d.y; r.pz = -d.z;
  const float Dabs = fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fabsf(d.z));
  const float Om = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
  const bool sane = Dabs >= 9.094947e-13f && Dabs <= 1.0995116e12f && Om <= 1.0995116e12f;
  r.Dm = sane ? Dabs : __builtin_inff();
  // absolute slack on ray parameters: the determinant ratio of a triangle the origin lies on is
  // rounding noise of about 2^-23 (|o| + |v|) / (|d| cos(angle to the normal))
  // (hardware reciprocals, within 1 ulp: delta is an order-of-magnitude slack, and an inverse direction
  // off by 2^-23 moves a box plane by less than the rounding of the slab distances themselves, far
  // inside the padding of the leaf boxes; a subnormal component counts as zero: +-inf like 1 / 0.
  // The path kernel calls this in every iteration of a walk: four IEEE divisions cost ~50 instructions)
  r.delta = sane ? (0.0001220703125f * (scene_scale + Om)) * __builtin_amdgcn_rcpf(Dabs) : 0.f;  // 2^-13
  r.n_nodes = 0; r.n_leaves = 0;
  r.oo = {o.x, o.y, o.z};
  r.inv = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z)};
  return r;
}

// could the reference accept this triangle at all? intersections.cpp:25 with min = FLT_MAX, :27
// FILTER = false: the caller has run the candidate filter on this triangle already (scan units)
template <bool FILTER = true>
__device__ __forceinline__ bool bvh_leaf_rec(const TriRec& T, const BvhRay& r, float& t, float& z, float& x1, float& x2);
template <bool FILTER = true, typename SP>
__device__ __forceinline__ bool bvh_leaf(SP sc, const BvhRay& r, int i, float& t, float& z,
                                         float& x1, float& x2) {
  const TriRec T = sc->tris[i];
  return bvh_leaf_rec<FILTER>(T, r, t, z, x1, x2);
}
template <bool FILTER>
__device__ __forceinline__ bool bvh_leaf_rec(const TriRec& T, const BvhRay& r, float& t, float& z, float& x1, float& x2) {
  if (FILTER && !tri_candidate(&T, r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, r.Dm)) return false;
  float x0;
  tri_eval(&T, r.o.x, r.o.y, r.o.z, r.px, r.py, r.pz, t, x0, x1, x2);
  z = fabsf(x0);
  return t >= 0.0f && t <= 3.402823466e+38f && x1 >= 0.0f && x2 >= 0.0f &&
         (x1 + x2) <= __uint_as_float(0x3F8020C4u);
}

// State of one ray's walk through the tree. The walk is resumable: bvh_burst advances it until it
// is finished or until too few lanes of the wave are still walking, so that a caller can hand
// finished lanes new work instead of letting them wait for the longest walk of the wave (walk
// lengths are heavy tailed: a wave that waits for its slowest ray runs at ~10 % lane occupancy).
struct BvhTrav {
  int cur;        // node (>= 0), leaf (< 0) or BVH_DONE
  int pend;       // a leaf met while other lanes still stepped through nodes, or BVH_DONE; cur != BVH_DONE while set
  int sp, n_list;
  float zmin, bound;
  // closest: the nearest hit so far (smallest z) lives in registers; only hits within the tie band
  // of it go to the list in scratch memory -- a few per cent of the rays have any
  BvhHit best;
  bool redo;      // closest: the pruned walk cannot vouch for this ray, run bvh_closest_exhaustive
  bool hit;       // any-hit: a triangle accepted
};
// The walk's arrays live beside the scalar state, not inside it: a struct holding dynamically
// indexed arrays is kept in scratch memory as a whole, scalars included.
struct BvhMem {
  int stack[BVH_STACK];
  BvhHit list[BVH_LIST];  // closest: surviving hits, sorted by triangle index
};

// A triangle the reference's test could accept (bvh_leaf returned true) enters the walk's state.
__device__ __forceinline__ void bvh_take(BvhTrav& T, BvhMem& M, const BvhRay& r, bool shadow, float max_t, int i,
                                         float t, float z, float x1, float x2) {
  if (shadow) {
    if (t <= max_t && (x1 + x2) <= 1.0f) { T.hit = true; T.cur = T.pend = BVH_DONE; }  // intersections.cpp:75-77
    return;
  }
  if (!(fabsf(t - z) <= 0.000244140625f * z + r.delta)) { T.redo = true; T.cur = T.pend = BVH_DONE; }  // 2^-12
  BvhHit extra = {i, t, z, x1, x2};  // what goes to the list, if anything (idx < 0: nothing)
  if (z < T.zmin) {
    // a new nearest hit; the old one stays only if it is within the band of the new one
    extra = T.best;
    T.best = {i, t, z, x1, x2};
    T.zmin = z;
    T.bound = T.zmin * BVH_BAND + 4.0f * r.delta;
    int w = 0;  // drop listed hits that fell out of the band
    for (int k = 0; k < T.n_list; ++k)
      if (M.list[k].t <= T.bound) M.list[w++] = M.list[k];
    T.n_list = w;
  }
  if (extra.idx >= 0 && extra.t <= T.bound) {
    if (T.n_list == BVH_LIST) {
      T.redo = true; T.cur = T.pend = BVH_DONE;
    } else {
      M.list[T.n_list++] = extra;  // in no particular order; bvh_finish replays by index
    }
  }
}

// big_recs: null, or the records of the big triangles in list order in fast memory (the path kernel's LDS copy)
template <typename SP>
__device__ __forceinline__ void bvh_begin(BvhTrav& T, BvhMem& M, SP sc, BvhRay& r, bool shadow,
                                          float max_t, const TriRec* big_recs = nullptr) {
  T.cur = sc->bvh_root;
  T.pend = BVH_DONE;
  T.sp = 0; T.n_list = 0;
  T.best = {-1, 0.f, 0.f, 0.f, 0.f};
  T.zmin = 3.402823466e+38f;
  // any-hit: t of an accepted triangle is <= max_t; its true distance can exceed that by the filter's slack
  T.bound = shadow ? fabsf(max_t) * 1.05f + 4.0f * r.delta + 1e-30f : 3.402823466e+38f;
  T.redo = false; T.hit = false;
  // The triangles too large for the tree first (wave-uniform loop): a wall spans the scene, its
  // box would swallow every ray; tested up front it also gives the walk its first pruning bound.
  // Filtered like the linear scan does it (scan units over the big list: walls come in coplanar
  // pairs), candidates then go through the reference's test in index order.
  for (int base = 0; base < sc->n_big; base += 32) {
    uint32_t mask = unit_mask(sc->units + base, r.o, r.d, r.Dm);
    r.n_leaves += (uint32_t)__popc(mask);
    while (mask != 0 && !(T.redo || T.hit)) {
      const int slot = base + __ffs((int)mask) - 1;
      const int i = sc->big_ids[slot];
      mask &= mask - 1u;
      float t, z, x1, x2;
      bool could;
      if (big_recs) { const TriRec Tb = big_recs[slot]; could = bvh_leaf_rec<false>(Tb, r, t, z, x1, x2); }
      else could = bvh_leaf<false>(sc, r, i, t, z, x1, x2);
      if (could) bvh_take(T, M, r, shadow, max_t, i, t, z, x1, x2);
    }
  }
  // a big triangle may have ended the query (any-hit found, or redo) before the tree had a say
  // (Measured and dropped: a test of the ray against the bounds of the whole tree here, which ends
  // three quarters of a room-with-a-mesh scene's queries before any node is fetched. The wave still
  // waits for the lanes that do walk, so nothing is gained: -1 ... -3 % paths/s on three scenes; and
  // letting the walkers wait until 16 ... 48 lanes have a walk to do lost 3 ... 20 %.)
  T.cur = (T.redo || T.hit) ? BVH_DONE : sc->bvh_root;
}

// Each round the wave does ONE kind of work, a node step or a leaf test, whichever more of its
// lanes are ready for. A lane that arrives at a leaf while the wave goes on stepping through nodes
// sets the leaf aside (T.pend) and carries on with the next entry of its stack, so it is ready for
// both kinds; the order in which leaves are tested does not matter to the result (head of this file).
// Closest hit (shadow = false): distance pruning + near-tie list. Any hit: existsIntersection is an
// OR over the triangles, the walk stops at the first acceptance.
// Returns when the lane's walk is done or when fewer than BVH_KEEP_NUM / BVH_KEEP_DEN of the lanes
// that entered are still walking.
// SET_ASIDE = false (the any-hit batch kernel): a lane that meets a leaf waits for the leaf round instead
// of walking on. An any-hit walk most likely ENDS at its next leaf test, so the nodes it steps through
// meanwhile are mostly wasted (10^6-triangle soup: +15 % any-hit rays/s, 102 k mesh +4 %); a closest-hit
// walk has to visit them anyway, and in the path kernel, where the two kinds share waves, waiting lost 1-4 %.
template <int KEEP_NUM = BVH_KEEP_NUM, int KEEP_DEN = BVH_KEEP_DEN, bool SET_ASIDE = true, typename STK, typename SP>
__device__ __forceinline__ void bvh_burst(BvhTrav& T, BvhMem& M, const STK& stack, SP sc,
                                          const BvhNode* __restrict__ nodes, BvhRay& r, bool shadow, float max_t,
                                          uint32_t& overflow, uint32_t* round_ctr = nullptr) {
  // round_ctr (developer builds): LDS counters {bursts, lanes entered, node rounds, lanes in them, leaf rounds, lanes in them}
  const int entered = __popcll(__ballot(T.cur != BVH_DONE));
  const bool tally = round_ctr && (int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1;
  if (tally && entered) { atomicAdd(round_ctr + 0, 1u); atomicAdd(round_ctr + 1, (uint32_t)entered); }
  while (T.cur != BVH_DONE) {
    if (SET_ASIDE && T.cur < 0 && T.pend == BVH_DONE && T.sp > 0) { T.pend = T.cur; T.cur = stack.pop(T.sp); }
    const int n_node = __popcll(__ballot(T.cur >= 0));
    const int n_leaf = __popcll(__ballot(T.pend != BVH_DONE || T.cur < 0));
    if (MIPT_BVH_VOTE_NODE * n_node >= MIPT_BVH_VOTE_LEAF * n_leaf) {
      if (round_ctr && (int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) { atomicAdd(round_ctr + 2, 1u); atomicAdd(round_ctr + 3, (uint32_t)n_node); }
      if (T.cur >= 0) {
        const BvhNode N = nodes[T.cur];
        r.n_nodes++;
        if (!bvh_step(N, r.oo, r.inv, T.bound, T.cur, stack, T.sp)) { overflow++; T.redo = true; T.cur = T.pend = BVH_DONE; }
      }
    } else {
      if (round_ctr && (int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) { atomicAdd(round_ctr + 4, 1u); atomicAdd(round_ctr + 5, (uint32_t)n_leaf); }
      int leaf = T.pend;
      if (leaf != BVH_DONE) {
        T.pend = BVH_DONE;
      } else if (T.cur < 0) {
        leaf = T.cur;
        T.cur = T.sp ? stack.pop(T.sp) : BVH_DONE;
      }
      if (leaf != BVH_DONE) {
        const int i = -1 - leaf;
        float t, z, x1, x2;
        r.n_leaves++;
        if (bvh_leaf(sc, r, i, t, z, x1, x2)) bvh_take(T, M, r, shadow, max_t, i, t, z, x1, x2);
      }
    }
    if (T.cur == BVH_DONE && T.pend != BVH_DONE) { T.cur = T.pend; T.pend = BVH_DONE; }
    // lanes still walking (this ballot is taken by the lanes inside the loop only)
    if (KEEP_DEN * (int)__popcll(__ballot(T.cur != BVH_DONE)) < KEEP_NUM * entered) break;
  }
}

// result of a finished walk into the scan's accumulator (the caller handles T.redo first)
__device__ __forceinline__ void bvh_finish(const BvhTrav& T, const BvhMem& M, bool shadow, HitAcc& h) {
  if (shadow) {
    if (T.hit) { h.set_lim(-1.0f); h.idx = 1; }
    return;
  }
  // the reference's sequential rule over the surviving hits, in index order (:25, :28)
  if (T.best.idx < 0) return;
  if (T.n_list == 0) {  // the usual case: one hit, no near-tie
    if (T.best.t <= h.lim) { h.set_lim(T.best.z); h.idx = T.best.idx; h.by = T.best.by; h.bz = T.best.bz; }
    return;
  }
  int last = -1;
  for (;;) {  // next larger index among the nearest hit and the listed ones (a triangle is met once)
    BvhHit e = T.best;
    bool found = e.idx > last;
    for (int k = 0; k < T.n_list; ++k) {
      const BvhHit c = M.list[k];
      if (c.idx > last && (!found || c.idx < e.idx)) { e = c; found = true; }
    }
    if (!found) break;
    last = e.idx;
    if (e.t <= h.lim) { h.set_lim(e.z); h.idx = e.idx; h.by = e.by; h.bz = e.bz; }
  }
}

// The same without any assumption about distances: the reference's scan accepts, after triangle
// `after`, the lowest-numbered triangle that passes its test against the current limit; each
// pass walks every box the ray touches to find that triangle, until a pass finds none.
template <typename SP>
__device__ __forceinline__ void bvh_closest_exhaustive(SP sc, const BvhNode* __restrict__ nodes,
                                                       BvhRay& r, HitAcc& h, uint32_t& overflow) {
  int after = -1;
  for (;;) {
    int best = 0x7FFFFFFF;
    float bz = 0.f, b1 = 0.f, b2 = 0.f;
    int stack[BVH_STACK];
    int sp = 0;
    for (int b = 0; b < sc->n_big; ++b) {
      const int i = sc->big_ids[b];
      float t, z, x1, x2;
      if (i > after && i < best && bvh_leaf(sc, r, i, t, z, x1, x2) && t <= h.lim) {
        best = i; bz = z; b1 = x1; b2 = x2;
      }
    }
    int cur = sc->bvh_root;
    for (;;) {
      if (cur == BVH_DONE) break;
      if (cur < 0) {
        const int i = -1 - cur;
        float t, z, x1, x2;
        if (i > after && i < best && bvh_leaf(sc, r, i, t, z, x1, x2) && t <= h.lim) {
          best = i; bz = z; b1 = x1; b2 = x2;
        }
        if (sp == 0) break;
        cur = stack[--sp];
        continue;
      }
      const BvhNode N = nodes[cur];
      r.n_nodes++;
      if (!bvh_step(N, r.oo, r.inv, 3.402823466e+38f, cur, StackScratch{stack}, sp)) { overflow++; return; }
    }
    if (best == 0x7FFFFFFF) return;
    h.set_lim(bz);
    h.idx = best; h.by = b1; h.bz = b2;
    after = best;
  }
}

}  // namespace mipt
