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

#include "mipt_device.hpp"

namespace mipt {

constexpr int BLOCK = 256;
// a block renders an 8 (x) by 32 (y) pixel tile; each wave an 8 x 8 tile. y runs fastest in
// the accumulation buffer (x-major, like the reference's pixels[x][y]).
constexpr int TILE_X = 8;
constexpr int TILE_Y = 32;

constexpr uint32_t TAG_GLASS = 0xFFFFFFFEu;
constexpr uint32_t TAG_FOG = 0xFFFFFFFDu;

struct PathArgs {
  DeviceScene sc;
  float M[16];  // cameraRotationX * cameraRotationY (main.cpp:133), column-major
  float cam_pos[4];
  float focal;
  float light_pos[4];
  float light_col[3];
  MiptParams p;
  uint32_t sample_begin;
  uint32_t sample_count;
  uint32_t chunk;  // samples per chunk; gridDim.y chunks
  uint32_t tiles_x;
  float* partial;  // [gridDim.y][W*H*3]
  // list mode (mipt_trace_paths)
  const int32_t* px;
  const int32_t* py;
  const uint32_t* samp;
  uint32_t n_paths;
  float* rgb;
  uint32_t* counts;
  unsigned long long* stats;  // ST_COUNT device counters, STATS builds only
};

// ---------------------------------------------------------------------------------------

template <bool STATS>
__device__ __forceinline__ void scan_triangles(const DeviceScene& sc, V4 o, V4 d, bool shadow, HitAcc& h,
                                               uint32_t& n_tests, uint32_t& n_gate) {
  const TriRec MIPT_CONST* tris = as_constant(sc.tris);
  const float px = -d.x, py = -d.y, pz = -d.z;  // vec3(-ray.direction), intersections.cpp:20
  const int n = sc.n_tris;
  for (int i = 0; i < n; ++i) tri_test<STATS>(tris + i, i, o.x, o.y, o.z, px, py, pz, shadow, h, n_tests, n_gate);
}

__global__ __launch_bounds__(BLOCK) void k_closest_hit(DeviceScene sc, const float4* __restrict__ origin,
                                                       const float4* __restrict__ dir, size_t n_rays,
                                                       int32_t* __restrict__ tri_idx, float* __restrict__ t_out,
                                                       float4* __restrict__ pos_out) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n_rays) return;
  const float4 of = origin[i], df = dir[i];
  const V4 o = {of.x, of.y, of.z, of.w}, d = {df.x, df.y, df.z, df.w};
  HitAcc h = {3.402823466e+38f, -1, 0.f, 0.f};
  uint32_t n_tests = 0, n_gate = 0;
  scan_triangles<false>(sc, o, d, false, h, n_tests, n_gate);
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

__global__ __launch_bounds__(BLOCK) void k_any_hit(DeviceScene sc, const float4* __restrict__ origin,
                                                   const float4* __restrict__ dir,
                                                   const float* __restrict__ max_t, size_t n_rays,
                                                   uint8_t* __restrict__ occluded) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n_rays) return;
  const float4 of = origin[i], df = dir[i];
  const V4 o = {of.x, of.y, of.z, of.w}, d = {df.x, df.y, df.z, df.w};
  HitAcc h = {max_t[i], 0, 0.f, 0.f};
  uint32_t n_tests = 0, n_gate = 0;
  scan_triangles<false>(sc, o, d, true, h, n_tests, n_gate);
  occluded[i] = (uint8_t)h.idx;
}

// ---------------------------------------------------------------------------------------
// The path kernel.
//
// One lane owns one pixel and walks its samples in order, so a pixel's radiance sum is formed
// in sample order by one lane and written with a plain store: no atomics, deterministic image.
// The reference's recursion (trace -> bounce* -> trace ...) is single-child, so it is run as a
// loop with one ray/scene query per iteration. A lane whose path ended starts its next sample
// in the same iteration ("regeneration"), and a terminal vertex turns the lane's next query
// into the shadow ray of directLight, handled by the same triangle scan with the any-hit
// rule -- the scan, ~90 % of the work, therefore always runs with every unfinished lane of
// the wave active. Samples are split over gridDim.y chunks so that the grid has many more
// blocks than the chip has slots; chunk images are summed in chunk order by k_reduce_chunks.
//
// pathtracer.cpp passes `Ray&` down the recursion, and bounceDiffuse reads ray.direction AFTER
// the recursive call returns (pathtracer.cpp:17-18): the cosine factor of every diffuse vertex
// is taken with the direction of the path's LAST segment. The per-vertex factors are therefore
// applied when the path ends, deepest vertex first, from a small per-lane stack of primitive
// tags kept in LDS.

#ifdef MIPT_WAVES_PER_EU
#define MIPT_PATHS_OCC __attribute__((amdgpu_waves_per_eu(MIPT_WAVES_PER_EU, MIPT_WAVES_PER_EU)))
#else
#define MIPT_PATHS_OCC
#endif

template <int RNG_MODE, bool STATS, bool LIST>
__global__ __launch_bounds__(BLOCK) MIPT_PATHS_OCC void k_paths(const PathArgs a) {
  extern __shared__ uint32_t lds_stack[];  // [max_depth][BLOCK] tags, then sphere normals
  const int tid = threadIdx.x;
  const MiptParams& P = a.p;
  const int W = P.width, H = P.height;

  int x = 0, y = 0;
  uint32_t s = 0, s_end = 0;
  size_t list_i = 0;
  if (LIST) {
    list_i = (size_t)blockIdx.x * BLOCK + tid;
    if (list_i < a.n_paths) {
      x = a.px[list_i]; y = a.py[list_i];
      s = a.samp[list_i]; s_end = s + 1;
    }
  } else {
    const int lane = tid & 63, wave = tid >> 6;
    const int tx = blockIdx.x % a.tiles_x, ty = blockIdx.x / a.tiles_x;
    x = tx * TILE_X + (lane >> 3);
    y = ty * TILE_Y + wave * 8 + (lane & 7);
    if (x < W && y < H) {
      s = a.sample_begin + blockIdx.y * a.chunk;
      const uint32_t last = a.sample_begin + a.sample_count;
      s_end = (s + a.chunk < last) ? s + a.chunk : last;
      if (s > s_end) s = s_end;
    }
  }
  const uint32_t pixel = (uint32_t)(x * H + y);
  bool active = s < s_end;

  float sum_r = 0.f, sum_g = 0.f, sum_b = 0.f;
  uint32_t st[ST_COUNT];
  if (STATS || LIST) {
#pragma unroll
    for (int k = 0; k < ST_COUNT; ++k) st[k] = 0;
  }

  // path state
  V4 o = {0, 0, 0, 0}, d = {0, 0, 0, 0}, dfin = {0, 0, 0, 0};
  Rng rng = {1u, 0u};
  int depth = 0, sp = 0;
  bool shadow = false, fresh = true;
  float shadow_max = 0.f;
  float Lr = 0.f, Lg = 0.f, Lb = 0.f;  // radiance of the terminal vertex, pending the shadow ray
  uint32_t* tags = lds_stack;
  float* sph_n = reinterpret_cast<float*>(lds_stack + (size_t)P.max_depth * BLOCK);
  const bool fog = P.fog_sigma_t > 0.f;

  while (active) {
    if (fresh) {
      // main.cpp:124-133: two draws jitter the pixel, direction = Rx*Ry*d, not normalised
      rng_begin<RNG_MODE>(rng, P.seed, pixel, s);
      const float u1 = rng_next<RNG_MODE>(rng);
      const float u2 = rng_next<RNG_MODE>(rng);
      const float dvx = (float)((double)(x - W / 2) + (double)(u1 * 2.0f - 1.0f) * 0.1);
      const float dvy = (float)((double)(y - H / 2) + (double)(u2 * 2.0f - 1.0f) * 0.1);
      const float dvz = a.focal, dvw = 0.0f;
      // mat4 * vec4 (type_mat4x4.inl:641-652): (m0*v0 + m1*v1) + (m2*v2 + m3*v3)
      d.x = (a.M[0] * dvx + a.M[4] * dvy) + (a.M[8] * dvz + a.M[12] * dvw);
      d.y = (a.M[1] * dvx + a.M[5] * dvy) + (a.M[9] * dvz + a.M[13] * dvw);
      d.z = (a.M[2] * dvx + a.M[6] * dvy) + (a.M[10] * dvz + a.M[14] * dvw);
      d.w = (a.M[3] * dvx + a.M[7] * dvy) + (a.M[11] * dvz + a.M[15] * dvw);
      o = {a.cam_pos[0], a.cam_pos[1], a.cam_pos[2], a.cam_pos[3]};
      depth = 0; sp = 0; shadow = false; fresh = false;
      if (STATS || LIST) st[ST_PATHS]++;
    }

    // ---- one ray/scene query: closestIntersection, or existsIntersection for a shadow ray
    HitAcc h;
    h.lim = shadow ? shadow_max : 3.402823466e+38f;
    h.idx = shadow ? 0 : -1;
    h.by = 0.f; h.bz = 0.f;
    uint32_t n_tests = 0, n_gate = 0;
    scan_triangles<STATS>(a.sc, o, d, shadow, h, n_tests, n_gate);
    if (STATS || LIST) {
      st[shadow ? ST_SHADOW : ST_CLOSEST]++;
      if (STATS) { st[ST_TESTS] += n_tests; st[ST_GATE] += n_gate; }
    }

    bool path_done = false;
    float cr = 0.f, cg = 0.f, cb = 0.f;  // radiance at the deepest vertex when the path ends lit

    if (shadow) {
      // light.cpp:26-29: occluded -> zero
      if (h.idx == 0) { cr = Lr; cg = Lg; cb = Lb; }
      path_done = true;
    } else {
      V4 pos = {0, 0, 0, 0}, nrm = {0, 0, 0, 0};
      bool sphere_hit = false;
      for (int j = 0; j < a.sc.n_spheres; ++j) {  // intersections.cpp:39-52
        if (STATS) st[ST_SPHERE]++;
        if (sphere_test(a.sc.spheres[j], o, d, h.lim, pos, nrm)) {
          h.idx = a.sc.n_tris + j;
          sphere_hit = true;
        }
      }
      if (h.idx < 0) {
        path_done = true;  // pathtracer.cpp:74-75: miss, colour stays 0
      } else {
        const ShadeRec sh = a.sc.shade[h.idx];
        if (!sphere_hit) {
          const TriRec T = a.sc.tris[h.idx];
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

        const bool terminal = depth >= P.max_depth || rng_next<RNG_MODE>(rng) <= P.rr_prob;  // :80
        if (terminal) {
          // pathtracer.cpp:81-93 + directLight (light.cpp:13-31)
          V4 lp;
          lp.x = a.light_pos[0] + (rng_next<RNG_MODE>(rng) * 2.0f - 1.0f) * 0.1f;
          lp.y = a.light_pos[1];
          lp.z = a.light_pos[2] + (rng_next<RNG_MODE>(rng) * 2.0f - 1.0f) * 0.1f;
          lp.w = a.light_pos[3];
          const V4 at = medium ? o : pos;
          const V4 v = sub4(lp, at);
          const V4 vecr = normalize4(v);
          const float r = sqrtf(dot4(v, v));
          const float den = (float)(((4.0 * 3.14159) * (double)r) * (double)r);  // light.cpp:22
          if (!medium) {
            const V4 nn = normalize4(nrm);
            const float c = dot4(vecr, nn);
            const float m = (c < 0.f) ? 0.f : c;  // std::max(c, 0.f)
            Lr = (a.light_col[0] * m) / den;
            Lg = (a.light_col[1] * m) / den;
            Lb = (a.light_col[2] * m) / den;
            if (fog) {
              const float tr = fog_expf(-(P.fog_sigma_t * r));
              Lr = Lr * tr; Lg = Lg * tr; Lb = Lb * tr;
            }
            if (sh.material != MIPT_MAT_GLASS) {  // pathtracer.cpp:88-93
              Lr = (sh.r * Lr) * 1.0f; Lg = (sh.g * Lg) * 1.0f; Lb = (sh.b * Lb) * 1.0f;
            }
            o = add4(pos, scale4(vecr, 0.001f));  // light.cpp:25
          } else {
            const float tr = fog_expf(-(P.fog_sigma_t * r));
            const float phase = 1.0f / (4.0f * 3.14159f);
            Lr = (((a.light_col[0] * phase) / den) * tr) * P.fog_albedo;
            Lg = (((a.light_col[1] * phase) / den) * tr) * P.fog_albedo;
            Lb = (((a.light_col[2] * phase) / den) * tr) * P.fog_albedo;
          }
          dfin = d;
          d = normalize4(vecr);  // Ray ctor normalises again (ray.h:11-12)
          shadow_max = r;
          shadow = true;
        } else if (medium) {
          // isotropic scattering, weight = single-scattering albedo
          const float u1 = rng_next<RNG_MODE>(rng);
          const float u2 = rng_next<RNG_MODE>(rng);
          const float z = 1.0f - 2.0f * u1;
          const float s2 = 1.0f - z * z;
          const float rxy = sqrtf(s2 < 0.f ? 0.f : s2);
          const float phi = 6.28318530718f * u2;
          float sn, cs;
          sincosf_glibc(phi, sn, cs);
          d = {rxy * cs, rxy * sn, z, 0.0f};
          tags[sp * BLOCK + tid] = TAG_FOG;
          sp++; depth++;
        } else {
          o = pos;  // pathtracer.cpp:100
          const uint32_t mat = sh.material;
          if (mat == MIPT_MAT_DIFFUSE) {
            // bounceDiffuse (pathtracer.cpp:5-19), orthonormalSystem / diffuseHemisphere (utils.cpp:10-22)
            if (STATS) st[ST_DIFFUSE]++;
            V3 Nt;
            if (fabsf(nrm.x) > fabsf(nrm.y)) {
              const float len = sqrtf(nrm.x * nrm.x + nrm.z * nrm.z);
              Nt = {nrm.z / len, 0.0f / len, -nrm.x / len};
            } else {
              const float len = sqrtf(nrm.y * nrm.y + nrm.z * nrm.z);
              Nt = {0.0f / len, -nrm.z / len, nrm.y / len};
            }
            const V3 Nb = cross3({nrm.x, nrm.y, nrm.z}, Nt);
            // g++ evaluates the two argument draws right to left: first draw -> b, second -> a
            const float ub = rng_next<RNG_MODE>(rng);
            const float ua = rng_next<RNG_MODE>(rng);
            const float rr = sqrtf(1.0f - ua * ua);
            const float phi = (float)((2 * 3.14) * (double)ub);
            float sn, cs;
            sincosf_glibc(phi, sn, cs);
            const float sx = cs * rr, sy = ua, sz = sn * rr;
            V4 rd;
            rd.x = (sx * Nb.x + sy * nrm.x) + sz * Nt.x;
            rd.y = (sx * Nb.y + sy * nrm.y) + sz * Nt.y;
            rd.z = (sx * Nb.z + sy * nrm.z) + sz * Nt.z;
            rd.w = 1.0f;
            d = normalize4(rd);
            o = add4(o, scale4(d, 0.001f));
            tags[sp * BLOCK + tid] = (uint32_t)h.idx;
            if (sphere_hit) {
              float* q = sph_n + ((size_t)sp * BLOCK + tid) * 4;
              q[0] = nrm.x; q[1] = nrm.y; q[2] = nrm.z; q[3] = nrm.w;
            }
            sp++; depth++;
          } else if (mat == MIPT_MAT_MIRROR) {
            // bounceMirror (pathtracer.cpp:21-27); weight 1, nothing to unwind
            if (STATS) st[ST_MIRROR]++;
            const float ct = dot4(nrm, d) * -1.0f;
            d = normalize4(add4(d, scale4(nrm, ct * 2.0f)));
            depth++;
          } else if (mat == MIPT_MAT_GLASS) {
            // bounceGlass (pathtracer.cpp:29-63)
            if (STATS) st[ST_GLASS]++;
            float R0 = (P.ior_air - P.ior_glass) / (P.ior_air + P.ior_glass);
            R0 = R0 * R0;
            float eta = P.ior_glass;
            V4 n = nrm;
            if (dot4(n, d) > 0.f) {
              n = scale4(n, -1.0f);
              eta = 1.0f / eta;
            }
            eta = 1.0f / eta;
            const float c = dot4(n, d) * -1.0f;
            const float cost1 = (c < 0.f) ? 0.f : c;
            const float cost2 = 1.0f - (eta * eta) * (1.0f - cost1 * cost1);
            const float Rprob = R0 + (1.0f - R0) * pow5f(1.0f - cost1);
            if (cost2 > 0.f && rng_next<RNG_MODE>(rng) > Rprob) {
              const float kk = eta * cost1 - sqrtf(cost2);
              d = normalize4(add4(scale4(d, eta), scale4(n, kk)));
            } else {
              d = normalize4(add4(d, scale4(n, cost1 * 2.0f)));
            }
            tags[sp * BLOCK + tid] = TAG_GLASS;
            sp++; depth++;
          } else if (mat == MIPT_MAT_SPECULAR) {
            // bounceSpecular (pathtracer.cpp:65-70); weight 1
            const float k2 = dot4(d, nrm) * 2.f;
            d = normalize4(sub4(d, scale4(nrm, k2)));
            depth++;
          } else {
            path_done = true;  // no branch of pathtracer.cpp:103-110 taken: colour stays 0
          }
        }
      }
    }

    if (path_done) {
      if (cr != 0.f || cg != 0.f || cb != 0.f) {
        // unwind the recursion, deepest vertex first
        for (int k = sp - 1; k >= 0; --k) {
          const uint32_t tag = tags[k * BLOCK + tid];
          if (tag == TAG_GLASS) {  // pathtracer.cpp:62
            cr = (cr * 1.0f) * P.glass_weight; cg = (cg * 1.0f) * P.glass_weight; cb = (cb * 1.0f) * P.glass_weight;
          } else if (tag == TAG_FOG) {
            cr = cr * P.fog_albedo; cg = cg * P.fog_albedo; cb = cb * P.fog_albedo;
          } else {  // pathtracer.cpp:18, with the direction of the last segment
            const ShadeRec sv = a.sc.shade[tag];
            V4 N = {sv.nx, sv.ny, sv.nz, sv.nw};
            if ((int)tag >= a.sc.n_tris) {
              const float* q = sph_n + ((size_t)k * BLOCK + tid) * 4;
              N = {q[0], q[1], q[2], q[3]};
            }
            const float dn = dot4(dfin, N);
            cr = (((cr * sv.r) * dn) * 1.0f) * P.diffuse_weight;
            cg = (((cg * sv.g) * dn) * 1.0f) * P.diffuse_weight;
            cb = (((cb * sv.b) * dn) * 1.0f) * P.diffuse_weight;
          }
        }
        sum_r += cr; sum_g += cg; sum_b += cb;  // main.cpp:137
      }
      s++;
      fresh = true;
      active = s < s_end;
    }
  }

  if (LIST) {
    if (list_i < a.n_paths) {
      a.rgb[3 * list_i + 0] = sum_r; a.rgb[3 * list_i + 1] = sum_g; a.rgb[3 * list_i + 2] = sum_b;
      if (a.counts) { a.counts[2 * list_i] = st[ST_CLOSEST]; a.counts[2 * list_i + 1] = st[ST_SHADOW]; }
    }
  } else if (x < W && y < H) {
    float* dst = a.partial + (size_t)blockIdx.y * ((size_t)W * H * 3) + (size_t)pixel * 3;
    dst[0] = sum_r; dst[1] = sum_g; dst[2] = sum_b;
  }

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

// accum[i] += sum over chunks (in chunk order) of partial[c][i]
__global__ __launch_bounds__(BLOCK) void k_reduce_chunks(const float* __restrict__ partial, uint32_t n_chunks,
                                                         size_t n, float* __restrict__ accum) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  float s = partial[i];
  for (uint32_t c = 1; c < n_chunks; ++c) s += partial[(size_t)c * n + i];
  accum[i] += s;
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
