// mipt_device.hpp -- device-side data layout and arithmetic of the path-tracing hot path.
//
// Arithmetic contract: this translation unit is compiled with -ffp-contract=off, IEEE
// division / square root and subnormals kept, so every float product and sum below is
// rounded on its own, in the order written. The order is the reference's: GLM 0.9.7.2
// (CW/glm/glm/detail/func_geometric.inl, func_matrix.inl, type_mat3x3.inl) as used by
// CW/raytracer/Source/intersections.cpp, pathtracer.cpp, light.cpp, utils.cpp, sphere.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mipt.h"

namespace mipt {

constexpr int BLOCK = 256;  // threads per workgroup of every kernel

// ---------------------------------------------------------------------------------------
// HBM layout of a scene (built once per scene by mipt_scene_create)
//
// TriRec: what the intersection loop needs, one 64-byte record per triangle. Every lane of a
// wave reads the SAME record in the same iteration, so the records are read through the
// constant address space: scalar loads into SGPRs, shared by the 64 lanes, no VGPR cost.
//   e1 = v1 - v0, e2 = v2 - v0               (intersections.cpp:17-18)
//   c0 = e1.y*e2.z - e2.y*e1.z               ray-independent cofactors of A = (-dir, e1, e2):
//   c1 = e1.x*e2.z - e2.x*e1.z               first-column cofactors of glm::determinant
//   c2 = e1.x*e2.y - e2.x*e1.y               (func_matrix.inl:235-238) and first row of
//                                            glm::inverse (type_mat3x3.inl:45-47)
//   kA, kB, kT                               error-bound factors of the candidate filter
//                                            (tri_candidate): 2^-12 |e1|oo |e2|oo,
//                                            2^-17 max(|e1|oo, |e2|oo), 2^-17 |e1|oo |e2|oo;
//                                            kA = +inf marks a triangle the filter never rejects
// Order: the twelve values a lane's own test of ITS candidate fetches (v0, e1, e2, c0..c2) fill the
// first 48 bytes -- three 16-byte LDS reads.
struct __attribute__((aligned(64))) TriRec {
  float v0x, v0y, v0z, c0;
  float e1x, e1y, e1z, c1;
  float e2x, e2y, e2z, c2;
  float v0w, kA, kB, kT;
};

// ShadeRec: what shading needs about the ONE primitive a ray hit; gathered per lane. For
// triangles the normal is constant, so everything the path computes from the normal alone is
// computed once on the host with the same IEEE operations: nn = normalize(normal) (light.cpp:20)
// and (nt, nb) = orthonormalSystem(normal) (utils.cpp:16-22). Sphere records leave them zero.
struct __attribute__((aligned(32))) ShadeRec {
  float nx, ny, nz, nw;
  float r, g, b;
  uint32_t material;
  float nnx, nny, nnz, nnw;
  float ntx, nty, ntz, nbx;
  float nby, nbz, pad0, pad1;
};

struct __attribute__((aligned(16))) SphereRec {
  float cx, cy, cz, cw;
  float radius, r, g, b;
  uint32_t material, pad0, pad1, pad2;
};

// ScanUnit: what phase 1 of the linear scan (the candidate filter) reads, one 64-byte record per
// UNIT: a single triangle, or two triangles with consecutive indices that share an edge and lie in
// one plane (the two halves of a wall, of a box face). A pair is filtered with ONE evaluation of
// the ray / plane system in the frame (r; f1, f2): r = the first triangle's vertex off the shared
// edge, f1, f2 = the edges from r to the shared vertices. In that frame the first triangle is
// u, v >= 0, u + v <= 1 and the second has barycentrics lq = (u + v - 1) qc, u - qu lq, v - qv lq,
// (qu, qv) = its far vertex, qc = 1 / (qu + qv - 1). Units are stored per chunk of 32 triangles
// (records [32 c, 32 c + n_units), pairs first); meta of a chunk's first record carries the
// counts. The filter frame need not be the reference's labelling of the vertices: the filter is
// only ever asked "can the reference's test of this triangle possibly accept".
struct __attribute__((aligned(64))) ScanUnit {
  float v0x, v0y, v0z;
  float e1x, e1y, e1z;
  float e2x, e2y, e2z;
  float kA, kB;      // error-bound factors as in TriRec, widened for pairs (see build_scan_units)
  float emax;        // pairs: largest |edge|oo of the two triangles (0 for single triangles)
  float qc, qu, qv;  // pairs: the second triangle in the frame of the first
  uint32_t meta;     // bits 0-4 / 5-9: mask bit of the first / second triangle; first record of a
                     // chunk only: the chunk's unit counts (see unit_mask)
};
// A pair whose plane is normal to coordinate axis k (f1, f2 and the far vertex have an exactly zero
// k component: walls, floors, box tops) is stored first in its chunk and in another form, because
// its system needs no cross products: t = -s_k / d_k, in-plane hit point p = s + t d, and
// (u, v) = (p_i A + p_j B, p_i C + p_j D) with i, j = the axes after k. Fields reused:
// e1x, e1y, e1z, e2x = A, B, C, D (the inverse of the 2x2 system, rounded from double);
// v0 = (r_k, r_i, r_j), permuted like the ray; e2y = kA 2^-4 / |n|; e2z = k (integer bits); kA, kB = the pair's factors divided by
// |n| = |f1_i f2_j - f1_j f2_i|, so that with |det| = |n| |d_k| the bounds are those of the
// general form. The evaluation has fewer roundings than the general one and stays inside the same
// error budget (<= 10 eps S Dm / (|f1| |d_k|) against the 64 eps share of the filter).

// BoxUnit: several coplanar pairs that are faces of ONE parallelepiped (the walls of the room, the
// faces of a block) filtered in the parallelepiped's own frame. With corner r and edges A, B, C the
// frame coordinates of a point x are m (x - r), m = (A B C)^-1 rounded from double; face "w = 0" /
// "w = 1" of axis w is the unit square of the two other coordinates (p, q) = axes (w + 1, w + 2) mod 3
// in the plane where coordinate w is 0 / 1, and its two triangles are the halves p + q <= 1 ("lo")
// and p + q >= 1 ("hi") -- after p -> 1 - p when the face is split along its other diagonal. The ray
// is mapped into the frame once (18 multiply-adds) and every face then costs a reciprocal per axis,
// two multiply-adds for its plane hit and a handful of comparisons: no cross products at all.
// A box takes two consecutive ScanUnit slots at the head of its chunk; `meta` sits where
// ScanUnit::meta does. ax[w].faces: bits 0-11 the face w = 0, bits 12-23 the face w = 1, each:
// mask bit of the lo triangle (5 bits), of the hi triangle (5), flip p (1), face present (1).
// Error model (every bound below is folded into the per-axis factors on the host, make_box_unit):
//   * Frame coordinates are sums of three products with |m| rows of 1-norm R_i, so o' and d' are
//     within 6 eps R_i |s|oo and 6 eps R_i |d|oo of their exact values. With W = |d'_w|,
//     G = R_w |d|oo / W >= 1 (how grazing the ray is for axis w) and Z = 1 + rmax |s|oo (which bounds
//     1 + every |o'_i|), the computed (p, q) of a face of axis w are within
//     eps kappa Z G (30 + 6 G) of the exact plane hit, kappa = max R_i / min R_i.
//   * The reference's own barycentrics of a triangle of that face are, for a ray that is `well`
//     (W >= |d|oo kA_w: its determinant |d . (e1 x e2)| = W vol is at least
//     2^12 eps 8 |d|oo emax^2, as for pair units), within 0.01 + 2^-15 lambda Z G emax / (rmax R_w vol)
//     of the exact ones (tri_candidate's bound with |o - v0|oo <= |s|oo + 3 emax), lambda =
//     max(1, emax rmax); vertices may sit off the ideal corners by 10^-6 emax, which moves the exact
//     hit by at most 4 10^-6 lambda kappa G in frame units.
//   * The reference accepts only if x1, x2 >= 0 and x1 + x2 <= 1.001 as it computes them, hence
//     only if all three exact barycentrics are >= -(0.001 + 2 m), m = its error; whichever way it
//     labels the vertices, the three are (p, q, 1 - p - q) for lo and (1 - p, 1 - q, p + q - 1) for hi.
//     delta = 0.001 + 2 m + 2 f (f = the filter's error) therefore decides every comparison.
//   * t: the ray parameter is the same number in every frame; dt covers the reference's
//     numerator noise (2^-15 lambda Z emax^2 / (rmax vol W)) and the filter's own (10 eps Z / W).
struct __attribute__((aligned(64))) BoxUnit {
  float rx, ry, rz;
  float m[9];     // rows of (A B C)^-1
  float rmax;     // largest R_i, rounded up
  float c0;       // constant part of delta
  float k6;       // 2 * 6 eps kappa, with a safety factor
  uint32_t meta;  // chunk counts when this is the chunk's first record (see unit_mask)
  struct Axis {      // what one axis needs (the record's second 64 bytes: dwords 5 w .. 5 w + 4)
    float kA;        // `well` threshold
    float kh;        // first-order coefficient of delta
    float kT;        // coefficient of dt
    float rw;        // R_w
    uint32_t faces;  // the two faces of this axis
  } ax[3];
  uint32_t pad;
};
static_assert(sizeof(BoxUnit) == 128, "a box takes two scan-unit slots");

struct BvhNode;

struct DeviceScene {
  const BvhNode* bvh;  // LBVH over the small triangles, or null: linear scan
  const int* big_ids;  // tree scenes: triangles too large for the tree (walls), tested one by one
  int n_big;
  int bvh_root;        // 0 = inner root, -1 - i = the tree is the single triangle i, BVH_DONE = empty
  const ScanUnit* units;  // phase-1 records of the linear scan (null with an LBVH)
  const TriRec* tris;
  const ShadeRec* shade;  // n_tris triangle records followed by n_spheres sphere records
  const SphereRec* spheres;
  int n_tris;
  int n_spheres;
  float scale;  // largest |coordinate| of a triangle vertex (slack of the tree traversal)
};

enum StatSlot {
  ST_PATHS = 0, ST_CLOSEST, ST_SHADOW, ST_TESTS, ST_GATE, ST_SPHERE, ST_FOG, ST_DIFFUSE, ST_MIRROR,
  ST_GLASS, ST_BVH_REDO, ST_BVH_OVERFLOW, ST_BVH_NODES, ST_SPECULAR, ST_COUNT
};

#define MIPT_CONST __attribute__((address_space(4)))

template <typename T>
__device__ __forceinline__ const T MIPT_CONST* as_constant(const T* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  return (const T MIPT_CONST*)p;
#pragma clang diagnostic pop
}

// ---------------------------------------------------------------------------------------
// small vector helpers, GLM operation order

struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };

__host__ __device__ __forceinline__ float dot3(V3 a, V3 b) {  // func_geometric.inl:69-70
  float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
  return (tx + ty) + tz;
}
__host__ __device__ __forceinline__ float dot4(V4 a, V4 b) {  // func_geometric.inl:79-80
  float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z, tw = a.w * b.w;
  return (tx + ty) + (tz + tw);
}
__host__ __device__ __forceinline__ V4 scale4(V4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
__host__ __device__ __forceinline__ V4 add4(V4 a, V4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
__host__ __device__ __forceinline__ V4 sub4(V4 a, V4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
__host__ __device__ __forceinline__ V4 normalize4(V4 v) {  // func_geometric.inl:158, func_exponential.inl:66
  float s = 1.0f / sqrtf(dot4(v, v));
  return scale4(v, s);
}
__host__ __device__ __forceinline__ V3 cross3(V3 x, V3 y) {  // func_geometric.inl:138-141
  return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y};
}

// orthonormalSystem (utils.cpp:16-22): component-wise true division, then cross(N.xyz, Nt)
__host__ __device__ inline void orthonormal_system(V4 n, V3& Nt, V3& Nb) {
  if (fabsf(n.x) > fabsf(n.y)) {
    const float len = sqrtf(n.x * n.x + n.z * n.z);
    Nt = {n.z / len, 0.0f / len, -n.x / len};
  } else {
    const float len = sqrtf(n.y * n.y + n.z * n.z);
    Nt = {0.0f / len, -n.z / len, n.y / len};
  }
  Nb = {n.y * Nt.z - Nt.y * n.z, n.z * Nt.x - Nt.z * n.x, n.x * Nt.y - Nt.x * n.y};
}

// ---------------------------------------------------------------------------------------
// RNG

__device__ __forceinline__ uint32_t mix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}

// path_key(seed, pixel, sample) = sample_key(pixel_key(seed, pixel), sample); the pixel part is
// computed once per work item, the sample part once per path.
__host__ __device__ inline uint32_t pixel_key(uint32_t seed, uint32_t pixel) {
  uint32_t k = seed ^ 0x9E3779B9u;
  k ^= k >> 16; k *= 0x85EBCA6Bu; k ^= k >> 13; k *= 0xC2B2AE35u; k ^= k >> 16;
  k += pixel;
  k ^= k >> 16; k *= 0x85EBCA6Bu; k ^= k >> 13; k *= 0xC2B2AE35u; k ^= k >> 16;
  return k;
}
__host__ __device__ inline uint32_t sample_key(uint32_t pkey, uint32_t sample) {
  uint32_t k = pkey ^ (sample * 0x9E3779B1u + 0x7F4A7C15u);
  k ^= k >> 16; k *= 0x85EBCA6Bu; k ^= k >> 13; k *= 0xC2B2AE35u; k ^= k >> 16;
  return k;
}
__host__ __device__ inline uint32_t path_key(uint32_t seed, uint32_t pixel, uint32_t sample) {
  return sample_key(pixel_key(seed, pixel), sample);
}

// RNG_MODE MIPT_RNG_MINSTD_PATH: a = engine state.  MIPT_RNG_COUNTER: a = key, b = draw number * 0x9E3779B9.
struct Rng { uint32_t a, b; };

template <int RNG_MODE>
__device__ __forceinline__ void rng_begin(Rng& r, uint32_t pkey, uint32_t sample) {
  uint32_t key = sample_key(pkey, sample);
  if (RNG_MODE == MIPT_RNG_COUNTER) {
    r.a = key; r.b = 0;
  } else {
    // 1 + key mod (2^31 - 2): key < 2^32 = 2*(2^31 - 2) + 4, so two conditional subtractions
    uint32_t m = key;
    if (m >= 2147483646u) m -= 2147483646u;
    if (m >= 2147483646u) m -= 2147483646u;
    r.a = 1u + m; r.b = 0;
  }
}

template <int RNG_MODE>
__device__ __forceinline__ float rng_next(Rng& r) {
  if (RNG_MODE == MIPT_RNG_COUNTER) {
    r.b += 0x9E3779B9u;  // b = draw number * 0x9E3779B9 (mod 2^32), kept premultiplied
    uint32_t h = mix32(r.a + r.b);
    return (float)(h >> 8) * (1.0f / 16777216.0f);
  } else {
    // minstd_rand0: x <- 16807 x mod (2^31 - 1); 2^31 = 1 (mod 2^31 - 1) folds the 46-bit product
    uint64_t p = (uint64_t)r.a * 16807u;
    uint32_t x = (uint32_t)(p & 0x7FFFFFFFu) + (uint32_t)(p >> 31);
    if (x >= 0x7FFFFFFFu) x -= 0x7FFFFFFFu;
    r.a = x;
    // generate_canonical<float,24>: float(x - 1) / float(2147483646.0L) = float(x-1) * 2^-31,
    // clamped below 1 (libstdc++ random.tcc:3366-3378)
    float u = (float)(x - 1u) * 4.656612873077392578125e-10f;
    return (u >= 1.0f) ? 0.99999994f : u;
  }
}

// ---------------------------------------------------------------------------------------
// libm functions whose host results the reference depends on

// sinf/cosf of glibc >= 2.28 (the ARM optimized-routines algorithm: double-precision
// quadrant reduction and two short polynomials) for 0 <= y < 120. The double arithmetic is
// far finer than the final rounding to float, so this reproduces glibc's float results bit
// for bit on the range the path uses (phi = 2*3.14*u, utils.cpp:12): verified exhaustively
// for every float in [0, 6.28] on the build host.
// FUSED evaluates the same polynomials with fused multiply-adds (fewer double-precision
// instructions); mipt_debug_selfcheck compares the two variants for every float in [0, 2 pi].
template <bool FUSED>
__device__ __forceinline__ void sincosf_glibc_impl(float y, float& s_out, float& c_out) {
  const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5,
               C3 = -0x1.6c087e89a359dp-10, C4 = 0x1.99343027bf8c3p-16;
  const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
  auto mad = [](double a, double b, double c) { return FUSED ? __builtin_fma(a, b, c) : a * b + c; };
  double x = (double)y;
  double sgn = 1.0, csgn = 1.0;
  int n = 0;
  uint32_t top = (__float_as_uint(y) >> 20) & 0x7FFu;
  if (top < 0x398u) {  // |y| < 2^-12: sinf returns y, cosf returns 1
    s_out = y; c_out = 1.0f;
    return;
  }
  if (top >= 0x3F4u) {  // |y| >= pi/4 (compared on the top 12 bits, as glibc does)
    double r = x * 0x1.45F306DC9C883p+23;
    n = ((int32_t)r + 0x800000) >> 24;
    x = mad(-(double)n, 0x1.921FB54442D18p0, x);
    sgn = (n & 1) != ((n >> 1) & 1) ? -1.0 : 1.0;  // sign table {1,-1,-1,1}[n & 3]
    csgn = (n & 2) ? -1.0 : 1.0;                   // second coefficient table negates the cosine
  }
  double xs = x * sgn;
  double x2 = x * x;
  // sine polynomial
  double x3 = xs * x2;
  double s1 = mad(x2, S3, S2);
  double x7 = x3 * x2;
  double sp = mad(x3, S1, xs);
  float sin_poly = (float)mad(x7, s1, sp);
  // cosine polynomial
  double x4 = x2 * x2;
  double c2 = mad(x2, csgn * C4, csgn * C3);
  double c1 = mad(x2, csgn * C1, csgn * C0);
  double x6 = x4 * x2;
  double cp = mad(x4, csgn * C2, c1);
  float cos_poly = (float)mad(x6, c2, cp);
  // sinf uses polynomial (n & 1), cosf polynomial ((n ^ 1) & 1)
  s_out = (n & 1) ? cos_poly : sin_poly;
  c_out = (n & 1) ? sin_poly : cos_poly;
}
__device__ __forceinline__ void sincosf_glibc(float y, float& s_out, float& c_out) {
  sincosf_glibc_impl<true>(y, s_out, c_out);
}

// powf(x, 5.0f) (pathtracer.cpp:47). x^5 in double rounded once to float is the correctly
// rounded result; glibc's powf differs from it for < 0.07 % of x in [0,1], by one ulp, and
// the value only feeds the comparison `u > Rprob`.
__device__ __forceinline__ float pow5f(float x) {
  double d = (double)x;
  double d2 = d * d;
  return (float)((d2 * d2) * d);
}

// Transcendentals of the fog (which has no reference code): fixed sequences of binary32
// operations with explicit fused multiply-adds, so the host restatement and the device agree bit
// for bit whatever their libms do. Accuracy ~1 ulp (Cephes' single-precision argument reduction
// and polynomials); the polynomial and its evaluation order are part of the fog's definition here.
__host__ __device__ __forceinline__ float fog_bits_to_float(uint32_t u) {
  float f;
#if defined(__HIP_DEVICE_COMPILE__)
  f = __uint_as_float(u);
#else
  memcpy(&f, &u, 4);
#endif
  return f;
}
__host__ __device__ __forceinline__ float fog_expf(float x) {
  if (!(x > -87.0f)) return 0.0f;
  if (x > 88.0f) x = 88.0f;
  const float k = __builtin_rintf(x * 1.44269504088896341f);
  float r = __builtin_fmaf(k, -0.693359375f, x);
  r = __builtin_fmaf(k, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = __builtin_fmaf(p, r, 1.3981999507e-3f);
  p = __builtin_fmaf(p, r, 8.3334519073e-3f);
  p = __builtin_fmaf(p, r, 4.1665795894e-2f);
  p = __builtin_fmaf(p, r, 1.6666665459e-1f);
  p = __builtin_fmaf(p, r, 5.0000001201e-1f);
  const float y = __builtin_fmaf(p, r * r, r) + 1.0f;
  return y * fog_bits_to_float((uint32_t)((int32_t)k + 127) << 23);
}
// x > 0 and normal (the fog calls it on 1 - u, u in [0, 1))
__host__ __device__ __forceinline__ float fog_logf(float x) {
  if (!(x > 0.0f)) return -__builtin_inff();
  uint32_t ix;
#if defined(__HIP_DEVICE_COMPILE__)
  ix = __float_as_uint(x);
#else
  memcpy(&ix, &x, 4);
#endif
  int32_t e = (int32_t)(ix >> 23) - 126;  // x = m 2^e, m in [0.5, 1)
  float m = fog_bits_to_float((ix & 0x007FFFFFu) | 0x3F000000u);
  if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; } else { m = m - 1.0f; }
  const float z = m * m;
  float p = 7.0376836292e-2f;
  p = __builtin_fmaf(p, m, -1.1514610310e-1f);
  p = __builtin_fmaf(p, m, 1.1676998740e-1f);
  p = __builtin_fmaf(p, m, -1.2420140846e-1f);
  p = __builtin_fmaf(p, m, 1.4249322787e-1f);
  p = __builtin_fmaf(p, m, -1.6668057665e-1f);
  p = __builtin_fmaf(p, m, 2.0000714765e-1f);
  p = __builtin_fmaf(p, m, -2.4999993993e-1f);
  p = __builtin_fmaf(p, m, 3.3333331174e-1f);
  float y = (p * m) * z;
  const float fe = (float)e;
  y = __builtin_fmaf(fe, -2.12194440e-4f, y);
  y = __builtin_fmaf(z, -0.5f, y);
  return __builtin_fmaf(fe, 0.693359375f, m + y);
}

// ---------------------------------------------------------------------------------------
// ray / triangle test, intersections.cpp:11-37 and :61-82

struct HitAcc {
  float lim;  // closest: minimumDistance; shadow: inter.distance, -1 once occluded
  int idx;    // closest: winning primitive, -1 = none; shadow: 1 = occluded
  float by, bz;  // barycentrics x.y, x.z of the winning triangle
  float lim_lo, lim_hi;  // lim * (1 -/+ 2^-20): the band inside which the gate needs the exact quotient

  __device__ __forceinline__ void set_lim(float v) {
    lim = v;
    lim_lo = v * 0.99999904632568359375f;  // 1 - 2^-20
    lim_hi = v * 1.00000095367431640625f;  // 1 + 2^-20
  }
};

// Correctly rounded 1/b. Fast path: hardware reciprocal estimate (<= 1 ulp) + two Newton steps in
// fused arithmetic; it equals IEEE 1.0f/b for every b with 2^-100 <= |b| <= 2^100 (checked
// exhaustively on the GPU by mipt_debug_selfcheck). Anything else (zero, subnormal, huge, inf,
// NaN) takes the IEEE division. r0 is the caller's estimate v_rcp_f32(b).
__device__ __forceinline__ float rcp_exact(float b, float r0) {
  // 2^-100 <= |b| <= 2^100 as one unsigned comparison of the bit pattern (a NaN lands outside):
  // two float comparisons cost two scalar moves for their constants and a scalar OR on top
  if ((__float_as_uint(b) & 0x7FFFFFFFu) - 0x0D800000u <= 0x64000000u) {
    float e = __builtin_fmaf(-b, r0, 1.0f);
    float r = __builtin_fmaf(e, r0, r0);
    e = __builtin_fmaf(-b, r, 1.0f);
    return __builtin_fmaf(e, r, r);
  }
  return 1.0f / b;
}

// MIPT_FILTER_TIGHT = k: the filters demand a determinant k times further from zero before they
// trust their own (u, v, t) and in exchange shrink their constant margins by k.
#ifndef MIPT_FILTER_TIGHT
#define MIPT_FILTER_TIGHT 1
#endif
constexpr float FILTER_TIGHT = (float)MIPT_FILTER_TIGHT;

// Candidate filter of the two-phase scan: returns false only when the reference's test of this
// triangle provably cannot accept, whatever the running minimum is. It evaluates the same
// linear system by Moeller-Trumbore in fused arithmetic (u, v, t ~ x.y, x.z, t of the reference)
// and rejects when the triangle is well conditioned for this ray AND (u, v) is outside the
// accepted region or t is negative by more than an error bound:
//   * every 3x3 determinant here and in the reference is a sum of six triple products, so either
//     evaluation is within 32 eps M of the exact value, M = product of the operands' oo-norms
//     (eps = 2^-24). "Well conditioned": |det| >= 2^12 eps |d|oo |e1|oo |e2|oo, which makes the
//     two computed determinants agree in sign and to better than 1 %.
//   * then |u - x.y| <= 0.02 + 2^-17 |s|oo |d|oo max(|e1|oo,|e2|oo) / |det| (same for v), and a
//     quotient t below -(2^-17 |s|oo |e1|oo |e2|oo / |det| + 1e-30) means the reference's
//     determinant ratio is negative and not rounded to zero.
// NaNs, infinities and out-of-range magnitudes make every comparison false, i.e. "candidate".
// Dm is |dir|oo, or +inf when the ray's magnitudes are outside [2^-40, 2^40].
template <typename TP>
__device__ __forceinline__ bool tri_candidate(TP T, float ox, float oy, float oz, float dx, float dy,
                                              float dz, float Dm) {
#pragma clang fp contract(fast)
  const float sx = ox - T->v0x, sy = oy - T->v0y, sz = oz - T->v0z;
  const float e1x = T->e1x, e1y = T->e1y, e1z = T->e1z;
  const float e2x = T->e2x, e2y = T->e2y, e2z = T->e2z;
  const float pvx = dy * e2z - dz * e2y, pvy = dz * e2x - dx * e2z, pvz = dx * e2y - dy * e2x;
  const float det = e1x * pvx + e1y * pvy + e1z * pvz;
  const float un = sx * pvx + sy * pvy + sz * pvz;
  const float qvx = sy * e1z - sz * e1y, qvy = sz * e1x - sx * e1z, qvz = sx * e1y - sy * e1x;
  const float vn = dx * qvx + dy * qvy + dz * qvz;
  const float tn = e2x * qvx + e2y * qvy + e2z * qvz;
  const float S = fmaxf(fmaxf(fabsf(sx), fabsf(sy)), fabsf(sz));
  const bool well = fabsf(det) >= Dm * T->kA + 1e-35f;
  const float inv = __builtin_amdgcn_rcpf(det);
  const float ainv = fabsf(inv);
  const float u = un * inv, v = vn * inv, t = tn * inv;
  const float delta = ((S * Dm) * T->kB) * ainv + 0.02f / FILTER_TIGHT;
  const float dt = (S * T->kT) * ainv + 1e-30f;
  const bool outside = (u < -delta) | (v < -delta) | (u + v > 1.001f + 2.0f * delta) | (t < -dt);
  return !(well & outside);
}

// The candidate filter over a scan unit: tri_candidate's system solved once for a pair. Sets the
// units' mask bits in `mask` for every triangle of the unit the reference could accept.
// Error model of a pair: the frame's edges are not the reference's e1, e2, and the second
// triangle's own reference determinants are those of the frame times its area ratio; every
// |edge|oo involved is <= emax and every |o - vertex|oo <= |s|oo + emax <= 2 max(|s|oo, emax), so
// the bounds of tri_candidate hold with E1 = E2 = emax and S = 2 max(|s|oo, emax) (the factors
// are folded into kA, kB on the host). The second triangle's barycentrics amplify an error in
// (u, v) by at most 1 + 2 qc max(|qu|, |qv|) <= 9 (pairs beyond that are not merged), and the
// merge tolerance on coplanarity moves them by < 0.01 for a ray that passes `well`.
template <bool PAIR, typename UP>
__device__ __forceinline__ void unit_candidates(UP U, float ox, float oy, float oz, float dx, float dy,
                                                float dz, float Dm, uint32_t& mask) {
#pragma clang fp contract(fast)
  const float sx = ox - U->v0x, sy = oy - U->v0y, sz = oz - U->v0z;
  const float e1x = U->e1x, e1y = U->e1y, e1z = U->e1z;
  const float e2x = U->e2x, e2y = U->e2y, e2z = U->e2z;
  const float pvx = dy * e2z - dz * e2y, pvy = dz * e2x - dx * e2z, pvz = dx * e2y - dy * e2x;
  const float det = e1x * pvx + e1y * pvy + e1z * pvz;
  const float un = sx * pvx + sy * pvy + sz * pvz;
  const float qx = sy * e1z - sz * e1y, qy = sz * e1x - sx * e1z, qz = sx * e1y - sy * e1x;
  const float vn = dx * qx + dy * qy + dz * qz;
  const float tn = e2x * qx + e2y * qy + e2z * qz;
  float S = fmaxf(fmaxf(fabsf(sx), fabsf(sy)), fabsf(sz));
  if (PAIR) S = fmaxf(S, U->emax);
  const float kA = U->kA;
  const bool well = fabsf(det) >= Dm * kA + 1e-35f;
  const float inv = __builtin_amdgcn_rcpf(det);
  const float u = un * inv, v = vn * inv, t = tn * inv;
  const float w1 = S * fabsf(inv);
  const float drel = (w1 * Dm) * U->kB;
  const float delta = drel + (PAIR ? 0.02f / FILTER_TIGHT + 0.002f : 0.02f / FILTER_TIGHT);
  // kT of tri_candidate is kA 2^-5; a pair's S carries a factor 2 that is folded in here
  const float dt = (w1 * kA) * (PAIR ? 0.0625f : 0.03125f) + 1e-30f;
  const bool behind = t < -dt;
  const float uv = u + v;
  const bool outA = (u < -delta) | (v < -delta) | (uv > 1.001f + 2.0f * delta) | behind;
  const uint32_t meta = U->meta;
  if (!(well & outA)) mask |= 1u << (meta & 31u);
  if (PAIR) {
    const float lq = (uv - 1.0f) * U->qc;
    const float l1 = u - U->qu * lq, l2 = v - U->qv * lq;
    const float dB = drel * 9.0f + (0.05f / FILTER_TIGHT + 0.01f);
    const bool outB = (lq < -dB) | (l1 < -dB) | (l2 < -dB) | behind;
    if (!(well & outB)) mask |= 1u << ((meta >> 5) & 31u);
  }
}

template <typename UP>
__device__ __forceinline__ void axis_pair_candidates(UP U, float sk, float si, float sj, float dk, float di,
                                                     float dj, float Dm, uint32_t& mask) {
#pragma clang fp contract(fast)
  const float S = fmaxf(fmaxf(fmaxf(fabsf(sk), fabsf(si)), fabsf(sj)), U->emax);
  const bool well = fabsf(dk) >= Dm * U->kA + 1e-35f;
  const float inv = __builtin_amdgcn_rcpf(dk);
  const float t = -(sk * inv);
  const float pi = si + t * di, pj = sj + t * dj;
  const float u = pi * U->e1x + pj * U->e1y, v = pi * U->e1z + pj * U->e2x;
  const float w1 = S * fabsf(inv);
  const float drel = (w1 * Dm) * U->kB;
  const float delta = drel + (0.02f / FILTER_TIGHT + 0.002f);
  const float dt = w1 * U->e2y + 1e-30f;
  const bool behind = t < -dt;
  const float uv = u + v;
  const bool outA = (u < -delta) | (v < -delta) | (uv > 1.001f + 2.0f * delta) | behind;
  const uint32_t meta = U->meta;
  if (!(well & outA)) mask |= 1u << (meta & 31u);
  const float lq = (uv - 1.0f) * U->qc;
  const float l1 = u - U->qu * lq, l2 = v - U->qv * lq;
  const float dB = drel * 9.0f + (0.05f / FILTER_TIGHT + 0.01f);
  const bool outB = (lq < -dB) | (l1 < -dB) | (l2 < -dB) | behind;
  if (!(well & outB)) mask |= 1u << ((meta >> 5) & 31u);
}

// The candidate filter over a box unit (see BoxUnit).
template <typename BP>
__device__ __forceinline__ void box_candidates(BP B, V4 o, V4 d, float Dm, uint32_t& mask) {
#pragma clang fp contract(fast)
  const float sx = o.x - B->rx, sy = o.y - B->ry, sz = o.z - B->rz;
  const float S = fmaxf(fmaxf(fabsf(sx), fabsf(sy)), fabsf(sz));
  const float Z = S * B->rmax + 1.0f;
  float oc[3], dc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    oc[i] = B->m[3 * i] * sx + B->m[3 * i + 1] * sy + B->m[3 * i + 2] * sz;
    dc[i] = B->m[3 * i] * d.x + B->m[3 * i + 1] * d.y + B->m[3 * i + 2] * d.z;
  }
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    const uint32_t fw = B->ax[w].faces;  // wave-uniform
    if (!(fw & 0x800800u)) continue;
    const int ip = (w + 1) % 3, iq = (w + 2) % 3;
    const float dw = dc[w];
    const bool well = fabsf(dw) >= Dm * B->ax[w].kA + 1e-35f;
    const float inv = __builtin_amdgcn_rcpf(dw);
    const float ainv = fabsf(inv);
    const float G = (Dm * B->ax[w].rw) * ainv;
    const float h = G * B->k6 + B->ax[w].kh;
    const float delta = (Z * G) * h + B->c0;
    const float dt = (Z * ainv) * B->ax[w].kT + 1e-30f;
    const float one_p = 1.0f + delta, one_m = 1.0f - delta;
    const float t0 = -(oc[w] * inv);
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      const uint32_t f = (fw >> (12 * side)) & 0xFFFu;
      if (!(f & 0x800u)) continue;
      const float t = side ? t0 + inv : t0;
      const float p0 = t * dc[ip] + oc[ip], q = t * dc[iq] + oc[iq];
      const float p = (f & 0x400u) ? 1.0f - p0 : p0;
      const float sig = p + q;
      const bool behind = t < -dt;
      const bool lo_out = (p < -delta) | (q < -delta) | (sig > one_p) | behind;
      const bool hi_out = (p > one_p) | (q > one_p) | (sig < one_m) | behind;
      if (!(well & lo_out)) mask |= 1u << (f & 31u);
      if (!(well & hi_out)) mask |= 1u << ((f >> 5) & 31u);
    }
  }
}

// Phase 1 over one chunk of scan units (records [0, n_units) at `chunk`, pairs first): the mask of
// the chunk's triangles the reference could accept. Unrolled so that the scalar loads of the next
// record overlap the arithmetic of this one (rolled, every unit waits out its own load: -9 %).
__device__ __forceinline__ uint32_t unit_mask(const ScanUnit* chunk, V4 o, V4 d, float Dm) {
  const ScanUnit MIPT_CONST* units = as_constant(chunk);
  // counts in the chunk's first record (a box keeps its meta where a scan unit does): bits 10-14
  // axis-plane pairs, 15-19 all pairs, 20-25 pairs + singles, 26-28 boxes (two slots each, first)
  const uint32_t m0 = units[0].meta;
  const int n_box = (int)((m0 >> 26) & 7u);
  const int n_axis = (int)((m0 >> 10) & 31u), n_pairs = (int)((m0 >> 15) & 31u), n_units = (int)((m0 >> 20) & 63u);
  uint32_t mask = 0;
  {
    const BoxUnit MIPT_CONST* boxes = reinterpret_cast<const BoxUnit MIPT_CONST*>(units);
    // (rolled; the whole record in two 64-byte scalar loads issued together: read field by field,
    // the compiler sinks the loads into the per-axis code and waits out each of them there)
#pragma unroll 1
    for (int k = 0; k < n_box; ++k) {
      typedef uint32_t Dwords16 __attribute__((ext_vector_type(16)));
      const Dwords16 MIPT_CONST* P = reinterpret_cast<const Dwords16 MIPT_CONST*>(boxes + k);
      const Dwords16 lo16 = P[0], hi16 = P[1];
      BoxUnit R;
      R.rx = __uint_as_float(lo16[0]); R.ry = __uint_as_float(lo16[1]); R.rz = __uint_as_float(lo16[2]);
#pragma unroll
      for (int i = 0; i < 9; ++i) R.m[i] = __uint_as_float(lo16[3 + i]);
      R.rmax = __uint_as_float(lo16[12]); R.c0 = __uint_as_float(lo16[13]); R.k6 = __uint_as_float(lo16[14]);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        R.ax[i].kA = __uint_as_float(hi16[5 * i]); R.ax[i].kh = __uint_as_float(hi16[5 * i + 1]);
        R.ax[i].kT = __uint_as_float(hi16[5 * i + 2]); R.ax[i].rw = __uint_as_float(hi16[5 * i + 3]);
        R.ax[i].faces = hi16[5 * i + 4];
      }
      box_candidates(&R, o, d, Dm, mask);
    }
    units += 2 * n_box;
  }
  // axis-plane pairs come sorted by axis; the ray is permuted once per run of equal axes.
  // (Rolled: unrolled by 2 or 4 the kernel spills registers; requesting the next record before
  // this one is evaluated costs more in register copies than the wait it hides.)
  uint32_t cur = 3u;
  float ok = 0.f, oi = 0.f, oj = 0.f, dk = 0.f, di = 0.f, dj = 0.f;
#pragma unroll 1
  for (int k = 0; k < n_axis; ++k) {
    // every field read up front: one scalar load of the whole record, axis included
    const ScanUnit MIPT_CONST* U = units + k;
    ScanUnit R;
    R.v0x = U->v0x; R.v0y = U->v0y; R.v0z = U->v0z; R.e1x = U->e1x; R.e1y = U->e1y; R.e1z = U->e1z;
    R.e2x = U->e2x; R.e2y = U->e2y; R.e2z = U->e2z; R.kA = U->kA; R.kB = U->kB; R.emax = U->emax;
    R.qc = U->qc; R.qu = U->qu; R.qv = U->qv; R.meta = U->meta;
    const uint32_t axis = __float_as_uint(R.e2z);  // wave-uniform
    if (axis != cur) {
      cur = axis;
      ok = axis == 0u ? o.x : axis == 1u ? o.y : o.z;
      oi = axis == 0u ? o.y : axis == 1u ? o.z : o.x;
      oj = axis == 0u ? o.z : axis == 1u ? o.x : o.y;
      dk = axis == 0u ? d.x : axis == 1u ? d.y : d.z;
      di = axis == 0u ? d.y : axis == 1u ? d.z : d.x;
      dj = axis == 0u ? d.z : axis == 1u ? d.x : d.y;
    }
    axis_pair_candidates(&R, ok - R.v0x, oi - R.v0y, oj - R.v0z, dk, di, dj, Dm, mask);
  }
#pragma unroll 4
  for (int k = n_axis; k < n_pairs; ++k)
    unit_candidates<true>(units + k, o.x, o.y, o.z, d.x, d.y, d.z, Dm, mask);
#pragma unroll 4
  for (int k = n_pairs; k < n_units; ++k)
    unit_candidates<false>(units + k, o.x, o.y, o.z, d.x, d.y, d.z, Dm, mask);
  return mask;
}

// One triangle of the scan. FAST = false is the literal reference arithmetic. FAST = true decides
// the gate `t >= 0 && t <= lim` (t = detM / detA, used for nothing else) from the estimate
// detM * rcp(detA) whenever that estimate is outside a relative band of 2^-20 around 0 and lim --
// the estimate is within 3e-7 of the rounded quotient -- and computes the IEEE quotient only
// inside the band; the reciprocal of the inverse path comes from rcp_exact. Same decisions, same
// stored bits, ~20 fewer instructions per test.
// T is a pointer to the record: through the constant address space when every lane tests the same
// triangle (scalar loads), or to a lane-private copy when each lane tests its own candidate.
// occluded: optional, a word cleared when a shadow ray is accepted (the two-phase scan's list of
// candidates still to test: existsIntersection is an OR, nothing later can change the answer).
template <bool STATS, bool FAST, typename TP>
__device__ __forceinline__ void tri_test(TP T, int i, float ox, float oy, float oz, float px, float py,
                                         float pz, bool shadow, HitAcc& h, uint32_t& n_tests,
                                         uint32_t& n_gate, uint32_t* occluded = nullptr) {
  // existsIntersection leaves its loop at the first accepted triangle (:79); the lane keeps
  // scanning here (its gate can no longer pass), so those iterations are not counted as tests
  if (STATS) n_tests += (shadow && h.idx) ? 0u : 1u;
  // b = origin - v0 (:19); A = (-dir, e1, e2), m = A with column 0 replaced by b (:20-22)
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
  bool pass;
  float r0 = 0.f;
  if (FAST) {
    r0 = __builtin_amdgcn_rcpf(detA);
    const float qa = detM * r0;
    // 1e-30 <= qa <= 1e30 (finite and clearly positive) and -1e30 <= qa <= -1e-30, each as ONE
    // unsigned comparison of the bit pattern (floats of one sign order like their bits; a NaN, a
    // zero, a subnormal or a value of the other sign lands outside): half the vector and scalar
    // instructions of four float comparisons with their constants and the ANDs between them
    const uint32_t qi = __float_as_uint(qa);
    const bool pos = qi - 0x0DA24260u <= 0x63A7B06Au;
    const bool neg = qi - 0x8DA24260u <= 0x63A7B06Au;
    const bool sure_pass = pos && qa <= h.lim_lo;
    const bool sure_fail = neg || (pos && qa >= h.lim_hi);
    pass = sure_pass;
    if (!sure_pass && !sure_fail) {
      const float t = detM / detA;  // :23
      pass = t >= 0.0f && t <= h.lim;
    }
  } else {
    const float t = detM / detA;  // :23
    pass = t >= 0.0f && t <= h.lim;  // :25 / :75
  }
  if (pass) {
    if (STATS) n_gate++;
    // glm::inverse(A) * b (:26): type_mat3x3.inl:39-53 then :510-512
    const float ood = FAST ? rcp_exact(detA, r0) : 1.0f / detA;
    const float i00 = c0 * ood;
    const float i10 = -(T->c1) * ood;
    const float i20 = T->c2 * ood;
    const float i01 = -k1 * ood;
    const float i11 = (px * e2z - e2x * pz) * ood;
    const float i21 = -(px * e2y - e2x * py) * ood;
    const float i02 = k2 * ood;
    const float i12 = -(px * e1z - e1x * pz) * ood;
    const float i22 = (px * e1y - e1x * py) * ood;
    const float x0 = (i00 * bx + i10 * by) + i20 * bz;
    const float x1 = (i01 * bx + i11 * by) + i21 * bz;
    const float x2 = (i02 * bx + i12 * by) + i22 * bz;
    // :27 compares the float sum with the double 1.001, i.e. with the largest float below it;
    // :77 (shadow rays) compares with 1
    const float bound = shadow ? 1.0f : __uint_as_float(0x3F8020C4u);
    if (x1 >= 0.0f && x2 >= 0.0f && (x1 + x2) <= bound) {
      if (shadow) {
        h.set_lim(-1.0f);  // `break` (:79): no later triangle can pass t >= 0 && t <= -1
        h.idx = 1;
        if (occluded) *occluded = 0u;
      } else {
        h.set_lim(fabsf(x0));  // :28
        h.idx = i;
        h.by = x1;
        h.bz = x2;
      }
    }
  }
}

// Sphere::intersect + solveQuadratic (sphere.h:23-45, utils.cpp:34-49) and the acceptance test
// of intersections.cpp:42. Returns true when the sphere becomes the closest hit.
__device__ __forceinline__ bool sphere_test(const SphereRec& S, V4 o, V4 d, float& lim, V4& pos, V4& nrm) {
  V4 c = {S.cx, S.cy, S.cz, S.cw};
  V4 L = sub4(o, c);
  float a = dot4(d, d);
  float b = 2.0f * dot4(d, L);
  float cc = dot4(L, L) - S.radius * S.radius;
  float discr = b * b - (4.0f * a) * cc;
  float x0, x1;
  if (discr < 0.0f) return false;
  if (discr == 0.0f) {
    x0 = x1 = (float)((-0.5 * (double)b) / (double)a);
  } else {
    float sq = sqrtf(discr);
    float q = (b > 0.0f) ? (float)(-0.5 * (double)(b + sq)) : (float)(-0.5 * (double)(b - sq));
    x0 = q / a;
    x1 = cc / q;
  }
  if (x0 > x1) { float tmp = x0; x0 = x1; x1 = tmp; }
  float t0 = x0;
  if (t0 < 0.0f) {
    t0 = x1;
    if (t0 < 0.0f) return false;
  }
  if (!((double)t0 >= 1e-6 && t0 <= lim)) return false;
  lim = t0;
  pos = add4(o, scale4(d, t0));
  nrm = normalize4(sub4(pos, c));
  return true;
}

}  // namespace mipt
