/*
 * pt_oracle.c -- CPU restatement of the reference path-tracing hot path.
 * TEST INFRASTRUCTURE (see pt_oracle.h for who may use it and for the parity status).
 *
 * Every function cites the reference lines it follows; paths are relative to
 * CW/raytracer/Source/ of the reference tree, GLM paths to CW/glm/glm/detail/.
 * Compile with -O2 -ffp-contract=off: all float arithmetic below relies on each
 * operation being rounded to binary32 on its own, in the order written.
 */
#include "pt_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ GLM primitives */

/* compute_dot<tvec3> (func_geometric.inl:65-72): tmp.x + tmp.y + tmp.z */
float or_dot3(const float a[3], const float b[3]) {
  float tx = a[0] * b[0], ty = a[1] * b[1], tz = a[2] * b[2];
  return (tx + ty) + tz;
}

/* compute_dot<tvec4> (func_geometric.inl:75-81): (tmp.x + tmp.y) + (tmp.z + tmp.w) */
float or_dot4(const float a[4], const float b[4]) {
  float tx = a[0] * b[0], ty = a[1] * b[1], tz = a[2] * b[2], tw = a[3] * b[3];
  return (tx + ty) + (tz + tw);
}

/* cross (func_geometric.inl:134-142) */
void or_cross3(const float x[3], const float y[3], float out[3]) {
  float c0 = x[1] * y[2] - y[1] * x[2];
  float c1 = x[2] * y[0] - y[2] * x[0];
  float c2 = x[0] * y[1] - y[0] * x[1];
  out[0] = c0; out[1] = c1; out[2] = c2;
}

/* normalize = x * inversesqrt(dot(x,x)) (func_geometric.inl:154-159),
 * inversesqrt = 1 / sqrt(x) (func_exponential.inl:62-68) */
void or_normalize3(const float v[3], float out[3]) {
  float s = 1.0f / sqrtf(or_dot3(v, v));
  out[0] = v[0] * s; out[1] = v[1] * s; out[2] = v[2] * s;
}

void or_normalize4(const float v[4], float out[4]) {
  float s = 1.0f / sqrtf(or_dot4(v, v));
  out[0] = v[0] * s; out[1] = v[1] * s; out[2] = v[2] * s; out[3] = v[3] * s;
}

/* length = sqrt(dot(v,v)) (func_geometric.inl:94-100) */
float or_length4(const float v[4]) { return sqrtf(or_dot4(v, v)); }

/* compute_determinant<tmat3x3> (func_matrix.inl:230-240), columns p, q, r */
float or_det3(const float p[3], const float q[3], const float r[3]) {
  float t0 = p[0] * (q[1] * r[2] - r[1] * q[2]);
  float t1 = q[0] * (p[1] * r[2] - r[1] * p[2]);
  float t2 = r[0] * (p[1] * q[2] - q[1] * p[2]);
  return (t0 - t1) + t2;
}

/* x = inverse(mat3(p,q,r)) * b: compute_inverse (type_mat3x3.inl:37-56) followed by
 * mat3 * vec3 (type_mat3x3.inl:507-513). */
void or_inverse3_mul(const float p[3], const float q[3], const float r[3], const float b[3], float x[3]) {
  float ood = 1.0f / or_det3(p, q, r);
  float i00 = +(q[1] * r[2] - r[1] * q[2]) * ood;
  float i10 = -(q[0] * r[2] - r[0] * q[2]) * ood;
  float i20 = +(q[0] * r[1] - r[0] * q[1]) * ood;
  float i01 = -(p[1] * r[2] - r[1] * p[2]) * ood;
  float i11 = +(p[0] * r[2] - r[0] * p[2]) * ood;
  float i21 = -(p[0] * r[1] - r[0] * p[1]) * ood;
  float i02 = +(p[1] * q[2] - q[1] * p[2]) * ood;
  float i12 = -(p[0] * q[2] - q[0] * p[2]) * ood;
  float i22 = +(p[0] * q[1] - q[0] * p[1]) * ood;
  x[0] = (i00 * b[0] + i10 * b[1]) + i20 * b[2];
  x[1] = (i01 * b[0] + i11 * b[1]) + i21 * b[2];
  x[2] = (i02 * b[0] + i12 * b[1]) + i22 * b[2];
}

/* mat4 * mat4 (type_mat4x4.inl:709-728): Result[j] = ((A0*B[j][0] + A1*B[j][1]) + A2*B[j][2]) + A3*B[j][3] */
void or_mat4_mul(const float a[16], const float b[16], float out[16]) {
  float r[16];
  for (int j = 0; j < 4; j++)
    for (int k = 0; k < 4; k++) {
      float s0 = a[0 * 4 + k] * b[j * 4 + 0];
      float s1 = a[1 * 4 + k] * b[j * 4 + 1];
      float s2 = a[2 * 4 + k] * b[j * 4 + 2];
      float s3 = a[3 * 4 + k] * b[j * 4 + 3];
      r[j * 4 + k] = ((s0 + s1) + s2) + s3;
    }
  memcpy(out, r, sizeof r);
}

/* mat4 * vec4 (type_mat4x4.inl:641-652): (m[0]*v0 + m[1]*v1) + (m[2]*v2 + m[3]*v3) */
void or_mat4_mul_vec4(const float m[16], const float v[4], float out[4]) {
  float r[4];
  for (int k = 0; k < 4; k++) {
    float m0 = m[0 * 4 + k] * v[0];
    float m1 = m[1 * 4 + k] * v[1];
    float m2 = m[2 * 4 + k] * v[2];
    float m3 = m[3 * 4 + k] * v[3];
    r[k] = (m0 + m1) + (m2 + m3);
  }
  memcpy(out, r, sizeof r);
}

/* ------------------------------------------------------------------ RNG */

static uint32_t mix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}

uint32_t or_path_key(uint32_t seed, uint32_t pixel, uint32_t sample) {
  uint32_t k = mix32(seed ^ 0x9E3779B9u);
  k = mix32(k + pixel);
  k = mix32(k ^ (sample * 0x9E3779B1u + 0x7F4A7C15u));
  return k;
}

float or_counter_uniform(uint32_t key, uint32_t k) {
  uint32_t h = mix32(key + (k + 1u) * 0x9E3779B9u);
  return (float)(h >> 8) * (1.0f / 16777216.0f);
}

/* minstd_rand0: x <- 16807 x mod (2^31 - 1). generate_canonical<float,24> with this
 * engine takes ONE draw: sum = float(x - 1), tmp = float(2147483646.0L) = 2^31,
 * ret = sum / tmp, clamped to nextafter(1, 0) (random.tcc:3353-3379). */
float or_rng_canonical(uint32_t* state) {
  uint32_t x = (uint32_t)(((uint64_t)(*state) * 16807u) % 2147483647u);
  *state = x;
  float sum = (float)(x - 1u);
  float ret = sum / 2147483648.0f;
  if (ret >= 1.0f) ret = nextafterf(1.0f, 0.0f);
  return ret;
}

typedef struct {
  uint32_t mode;
  uint32_t state; /* minstd engine state */
  uint32_t key;   /* counter mode */
  uint32_t ctr;
} OrRng;

static void rng_begin_path(OrRng* r, const MiptParams* p, uint32_t pixel, uint32_t sample) {
  if (r->mode == MIPT_RNG_MINSTD_GLOBAL) return; /* one stream for the whole frame */
  uint32_t key = or_path_key(p->seed, pixel, sample);
  r->key = key;
  r->ctr = 0;
  r->state = 1u + key % 2147483646u; /* in [1, 2^31 - 2], valid minstd state */
}

static float rng_next(OrRng* r) {
  if (r->mode == MIPT_RNG_COUNTER) return or_counter_uniform(r->key, r->ctr++);
  return or_rng_canonical(&r->state);
}

/* ------------------------------------------------------------------ scene builders */

static void compute_normal(MiptTriangle* t) {
  /* Triangle::ComputeNormal (TestModelH.h:39-48): normalize(cross(e2, e1)), w = 1 */
  float e1[3], e2[3], c[3], n[3];
  for (int k = 0; k < 3; k++) { e1[k] = t->v1[k] - t->v0[k]; e2[k] = t->v2[k] - t->v0[k]; }
  or_cross3(e2, e1, c);
  or_normalize3(c, n);
  t->normal[0] = n[0]; t->normal[1] = n[1]; t->normal[2] = n[2]; t->normal[3] = 1.0f;
}

static void push_tri(MiptTriangle* out, int* n, const float a[3], const float b[3], const float c[3],
                     const float col[3], uint32_t mat) {
  MiptTriangle* t = &out[(*n)++];
  memset(t, 0, sizeof *t);
  for (int k = 0; k < 3; k++) { t->v0[k] = a[k]; t->v1[k] = b[k]; t->v2[k] = c[k]; t->color[k] = col[k]; }
  t->v0[3] = t->v1[3] = t->v2[3] = 1.0f;
  t->material = mat;
}

/* the ten faces of a block given its eight corners (TestModelH.h:120-138) */
static void push_block(MiptTriangle* out, int* n, float P[8][3], const float col[3], uint32_t mat) {
  enum { A, B, C, D, E, F, G, H };
  push_tri(out, n, P[E], P[B], P[A], col, mat); push_tri(out, n, P[E], P[F], P[B], col, mat);
  push_tri(out, n, P[F], P[D], P[B], col, mat); push_tri(out, n, P[F], P[H], P[D], col, mat);
  push_tri(out, n, P[H], P[C], P[D], col, mat); push_tri(out, n, P[H], P[G], P[C], col, mat);
  push_tri(out, n, P[G], P[E], P[C], col, mat); push_tri(out, n, P[E], P[A], P[C], col, mat);
  push_tri(out, n, P[G], P[F], P[E], col, mat); push_tri(out, n, P[G], P[H], P[F], col, mat);
}

static void block_corners(float P[8][3], const float base[4][2], float h) {
  for (int i = 0; i < 4; i++) {
    P[i][0] = base[i][0]; P[i][1] = 0.0f; P[i][2] = base[i][1];
    P[i + 4][0] = base[i][0]; P[i + 4][1] = h; P[i + 4][2] = base[i][1];
  }
}

int or_load_test_model(int which, MiptTriangle* out) {
  const float red[3] = {0.75f, 0.15f, 0.15f}, yellow[3] = {0.75f, 0.75f, 0.15f};
  const float purple[3] = {0.75f, 0.15f, 0.75f}, white[3] = {0.75f, 0.75f, 0.75f};
  const float black[3] = {0.f, 0.f, 0.f};
  const float L = 555.0f;
  int n = 0;
  enum { A, B, C, D, E, F, G, H };
  float R[8][3] = {{L, 0, 0}, {0, 0, 0}, {L, 0, L}, {0, 0, L}, {L, L, 0}, {0, L, 0}, {L, L, L}, {0, L, L}};
  int cornell = (which == OR_SCENE_CORNELL || which == OR_SCENE_COLOR_BLEEDING);
  if (!cornell && which != OR_SCENE_GLASS_CUBE && which != OR_SCENE_NONE) return -1;
  /* room: floor, left, right, ceiling, back (TestModelH.h:87-105 / 384-402 / 516-534) */
  const float* floorc = cornell ? white : red;
  const float* leftc = cornell ? purple : red;
  const float* rightc = cornell ? yellow : red;
  push_tri(out, &n, R[C], R[B], R[A], floorc, MIPT_MAT_DIFFUSE);
  push_tri(out, &n, R[C], R[D], R[B], floorc, MIPT_MAT_DIFFUSE);
  push_tri(out, &n, R[A], R[E], R[C], leftc, MIPT_MAT_DIFFUSE);
  push_tri(out, &n, R[C], R[E], R[G], leftc, MIPT_MAT_DIFFUSE);
  push_tri(out, &n, R[F], R[B], R[D], rightc, MIPT_MAT_DIFFUSE);
  push_tri(out, &n, R[H], R[F], R[D], rightc, MIPT_MAT_DIFFUSE);
  push_tri(out, &n, R[E], R[F], R[G], floorc, MIPT_MAT_DIFFUSE);
  push_tri(out, &n, R[F], R[H], R[G], floorc, MIPT_MAT_DIFFUSE);
  push_tri(out, &n, R[G], R[D], R[C], floorc, MIPT_MAT_DIFFUSE);
  push_tri(out, &n, R[G], R[H], R[D], floorc, MIPT_MAT_DIFFUSE);
  if (which != OR_SCENE_NONE) {
    /* short block (TestModelH.h:110-138 / 407-435) */
    const float sb[4][2] = {{290, 114}, {130, 65}, {240, 272}, {82, 225}};
    float P[8][3];
    block_corners(P, sb, 165.0f);
    push_block(out, &n, P, white, cornell ? MIPT_MAT_DIFFUSE : MIPT_MAT_GLASS);
    if (cornell) {
      /* tall block (TestModelH.h:143-171) */
      const float tb[4][2] = {{423, 247}, {265, 296}, {472, 406}, {314, 456}};
      block_corners(P, tb, 330.0f);
      push_block(out, &n, P, white, MIPT_MAT_DIFFUSE);
    } else {
      /* one mirror quad: front face of a moved tall block (TestModelH.h:440-452) */
      const float tb[4][2] = {{323, 347}, {165, 396}, {472, 406}, {314, 456}};
      block_corners(P, tb, 330.0f);
      push_tri(out, &n, P[E], P[B], P[A], black, MIPT_MAT_MIRROR);
      push_tri(out, &n, P[E], P[F], P[B], black, MIPT_MAT_MIRROR);
    }
  }
  /* scale to [-1,1]^3, flip x and y, w = 1, recompute normals (TestModelH.h:177-200) */
  const float s = 2.0f / L;
  for (int i = 0; i < n; i++) {
    float* vs[3] = {out[i].v0, out[i].v1, out[i].v2};
    for (int v = 0; v < 3; v++) {
      for (int k = 0; k < 4; k++) vs[v][k] = vs[v][k] * s;
      for (int k = 0; k < 4; k++) vs[v][k] = vs[v][k] - 1.0f;
      vs[v][0] = vs[v][0] * -1.0f;
      vs[v][1] = vs[v][1] * -1.0f;
      vs[v][3] = 1.0f;
    }
    compute_normal(&out[i]);
  }
  return n;
}

void or_default_camera(MiptCamera* cam, int width) {
  /* reset_camera (camera.cpp:4-13): focalLength = SCREEN_WIDTH */
  memset(cam, 0, sizeof *cam);
  cam->focal = (float)width;
  cam->pos[0] = 0.f; cam->pos[1] = 0.f; cam->pos[2] = -3.f; cam->pos[3] = 1.f;
  for (int i = 0; i < 4; i++) { cam->rot_x[i * 4 + i] = 1.f; cam->rot_y[i * 4 + i] = 1.f; }
}

void or_default_light(MiptLight* l) {
  /* reset_light (light.cpp:6-9) */
  l->pos[0] = 0.f; l->pos[1] = -0.5f; l->pos[2] = -0.7f; l->pos[3] = 1.0f;
  l->color[0] = l->color[1] = l->color[2] = 14.0f * 1.0f;
}

void or_params_default(MiptParams* p, int width, int height) {
  memset(p, 0, sizeof *p);
  p->width = width; p->height = height;
  p->max_depth = 10;          /* definitions.h:47 */
  p->rr_prob = 0.15f;         /* :48 */
  p->diffuse_weight = 0.5f;   /* :49 */
  p->glass_weight = 1.15f;    /* :50 */
  p->ior_glass = 1.512f;      /* :54 */
  p->ior_air = 1.000f;        /* :55 */
  p->rng_mode = MIPT_RNG_MINSTD_PATH;
  p->seed = 0;
  p->fog_sigma_t = 0.0f;
  p->fog_albedo = 0.8f;
  p->flags = 0;
}

/* ------------------------------------------------------------------ intersections */

typedef struct {
  const MiptTriangle* tris; size_t n_tris;
  const MiptSphere* sph; size_t n_sph;
} OrScene;

typedef struct {
  float position[4];
  float distance;
  float color[3];
  float normal[4];
  uint32_t material;
  int index;
} OrHit;

typedef struct {
  uint64_t closest_rays, shadow_rays, tri_tests, gate_passes, sphere_tests, fog_events;
  uint64_t diffuse_bounces, mirror_bounces, glass_bounces, specular_bounces;
} OrCounters;

/* solveQuadratic (utils.cpp:34-49). The 0.5 literals are double, sqrt is the float overload. */
static int solve_quadratic(float a, float b, float c, float* x0, float* x1) {
  float discr = b * b - (4.0f * a) * c;
  if (discr < 0) return 0;
  else if (discr == 0) { *x0 = *x1 = (float)((-0.5 * (double)b) / (double)a); }
  else {
    float q = (b > 0) ? (float)(-0.5 * (double)(b + sqrtf(discr)))
                      : (float)(-0.5 * (double)(b - sqrtf(discr)));
    *x0 = q / a;
    *x1 = c / q;
  }
  if (*x0 > *x1) { float t = *x0; *x0 = *x1; *x1 = t; }
  return 1;
}

/* Sphere::intersect (sphere.h:23-45) */
static int sphere_intersect(const MiptSphere* s, const float o[4], const float d[4], OrHit* res) {
  float Lv[4];
  for (int k = 0; k < 4; k++) Lv[k] = o[k] - s->center[k];
  float a = or_dot4(d, d);
  float b = 2.0f * or_dot4(d, Lv);
  float c = or_dot4(Lv, Lv) - s->radius * s->radius;
  float t0, t1;
  if (!solve_quadratic(a, b, c, &t0, &t1)) return 0;
  if (t0 > t1) { float t = t0; t0 = t1; t1 = t; }
  if (t0 < 0) {
    t0 = t1;
    if (t0 < 0) return 0;
  }
  res->distance = t0;
  float pc[4];
  for (int k = 0; k < 4; k++) { res->position[k] = o[k] + t0 * d[k]; pc[k] = res->position[k] - s->center[k]; }
  for (int k = 0; k < 3; k++) res->color[k] = s->color[k];
  res->material = s->material;
  or_normalize4(pc, res->normal);
  return 1;
}

/* closestIntersection (intersections.cpp:6-55) */
static int closest_intersection(const OrScene* sc, const float o[4], const float d[4], OrHit* out, OrCounters* cn) {
  float minimum = FLT_MAX;
  int okay = 0;
  cn->closest_rays++;
  for (size_t i = 0; i < sc->n_tris; i++) {
    const MiptTriangle* tr = &sc->tris[i];
    float e1[3], e2[3], b[3], p[3];
    for (int k = 0; k < 3; k++) {
      e1[k] = tr->v1[k] - tr->v0[k];
      e2[k] = tr->v2[k] - tr->v0[k];
      b[k] = o[k] - tr->v0[k];
      p[k] = -d[k];
    }
    /* :20-23  A = (-dir, e1, e2);  m = A with column 0 = b;  t = det(m)/det(A) */
    float t = or_det3(b, e1, e2) / or_det3(p, e1, e2);
    cn->tri_tests++;
    if (t >= 0 && t <= minimum) {
      cn->gate_passes++;
      float x[3];
      or_inverse3_mul(p, e1, e2, b, x); /* :26 */
      /* :27  the sum is float, the comparison with 1.001 is double */
      if (x[1] >= 0 && x[2] >= 0 && ((double)(x[1] + x[2]) <= 1.001)) {
        minimum = fabsf(x[0]);
        for (int k = 0; k < 3; k++) out->position[k] = (tr->v0[k] + x[1] * e1[k]) + x[2] * e2[k];
        out->position[3] = (tr->v0[3] + x[1] * 0.0f) + x[2] * 0.0f;
        out->distance = fabsf(x[0]);
        for (int k = 0; k < 3; k++) out->color[k] = tr->color[k];
        for (int k = 0; k < 4; k++) out->normal[k] = tr->normal[k];
        out->material = tr->material;
        out->index = (int)i;
        okay = 1;
      }
    }
  }
  for (size_t i = 0; i < sc->n_sph; i++) { /* :39-52 */
    OrHit res;
    cn->sphere_tests++;
    int valid = sphere_intersect(&sc->sph[i], o, d, &res);
    if (valid && (double)res.distance >= 1e-6 && res.distance <= minimum) {
      minimum = res.distance;
      res.index = (int)(sc->n_tris + i);
      *out = res;
      okay = 1;
    }
  }
  return okay;
}

/* existsIntersection (intersections.cpp:57-85) */
static int exists_intersection(const OrScene* sc, const float o[4], const float d[4], float max_t, OrCounters* cn) {
  cn->shadow_rays++;
  for (size_t i = 0; i < sc->n_tris; i++) {
    const MiptTriangle* tr = &sc->tris[i];
    float e1[3], e2[3], b[3], p[3];
    for (int k = 0; k < 3; k++) {
      e1[k] = tr->v1[k] - tr->v0[k];
      e2[k] = tr->v2[k] - tr->v0[k];
      b[k] = o[k] - tr->v0[k];
      p[k] = -d[k];
    }
    float t = or_det3(b, e1, e2) / or_det3(p, e1, e2);
    cn->tri_tests++;
    if (t >= 0 && t <= max_t) {
      cn->gate_passes++;
      float x[3];
      or_inverse3_mul(p, e1, e2, b, x);
      if (x[1] >= 0 && x[2] >= 0 && (x[1] + x[2] <= 1)) return 1;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------ shading */

typedef struct {
  const OrScene* sc;
  const MiptParams* p;
  const MiptLight* light;
  OrRng* rng;
  OrCounters* cn;
} OrCtx;

/* The fog's transcendental functions: fixed sequences of binary32 operations with explicit
 * fused multiply-adds (fmaf is exact by definition), so that any host and the device agree bit
 * for bit. Cephes-style argument reduction and polynomials, ~1 ulp. The fog has no reference
 * code; these sequences are part of its definition here. */
static float fog_bits_to_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
float or_fog_expf(float x) {
  if (!(x > -87.0f)) return 0.0f;
  if (x > 88.0f) x = 88.0f;
  const float k = rintf(x * 1.44269504088896341f);
  float r = fmaf(k, -0.693359375f, x);
  r = fmaf(k, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  const float y = fmaf(p, r * r, r) + 1.0f;
  return y * fog_bits_to_float((uint32_t)((int32_t)k + 127) << 23);
}
float or_fog_logf(float x) { /* x > 0 and normal */
  if (!(x > 0.0f)) return -INFINITY;
  uint32_t ix;
  memcpy(&ix, &x, 4);
  int32_t e = (int32_t)(ix >> 23) - 126;
  float m = fog_bits_to_float((ix & 0x007FFFFFu) | 0x3F000000u);
  if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; } else { m = m - 1.0f; }
  const float z = m * m;
  float p = 7.0376836292e-2f;
  p = fmaf(p, m, -1.1514610310e-1f);
  p = fmaf(p, m, 1.1676998740e-1f);
  p = fmaf(p, m, -1.2420140846e-1f);
  p = fmaf(p, m, 1.4249322787e-1f);
  p = fmaf(p, m, -1.6668057665e-1f);
  p = fmaf(p, m, 2.0000714765e-1f);
  p = fmaf(p, m, -2.4999993993e-1f);
  p = fmaf(p, m, 3.3333331174e-1f);
  float y = (p * m) * z;
  const float fe = (float)e;
  y = fmaf(fe, -2.12194440e-4f, y);
  y = fmaf(z, -0.5f, y);
  return fmaf(fe, 0.693359375f, m + y);
}
#define fog_expf or_fog_expf
#define fog_logf or_fog_logf

/* directLight (light.cpp:13-31). fog: transmittance along the light segment. */
static void direct_light(const OrCtx* cx, const float pos[4], const float normal[4],
                         const float lpos[4], const float lcol[3], float out[3]) {
  float v[4], vecr[4], nn[4];
  for (int k = 0; k < 4; k++) v[k] = lpos[k] - pos[k];
  or_normalize4(v, vecr);
  float r = or_length4(v);
  or_normalize4(normal, nn);
  float c = or_dot4(vecr, nn);
  float m = (c < 0.f) ? 0.f : c;                                /* std::max(c, 0.f) */
  float den = (float)(((4.0 * 3.14159) * (double)r) * (double)r); /* :22, double island */
  for (int k = 0; k < 3; k++) out[k] = (lcol[k] * m) / den;
  float so[4], sd[4];
  for (int k = 0; k < 4; k++) so[k] = pos[k] + vecr[k] * 0.001f;
  or_normalize4(vecr, sd); /* Ray ctor normalises again (ray.h:11-12) */
  if (exists_intersection(cx->sc, so, sd, r, cx->cn)) { out[0] = out[1] = out[2] = 0.f; return; }
  if (cx->p->fog_sigma_t > 0.f) {
    float tr = fog_expf(-(cx->p->fog_sigma_t * r));
    for (int k = 0; k < 3; k++) out[k] = out[k] * tr;
  }
}

static void jitter_light(const OrCtx* cx, float lpos[4]) {
  /* pathtracer.cpp:81-85: x first, then z */
  lpos[0] = cx->light->pos[0] + (rng_next(cx->rng) * 2.0f - 1.0f) * 0.1f;
  lpos[1] = cx->light->pos[1];
  lpos[2] = cx->light->pos[2] + (rng_next(cx->rng) * 2.0f - 1.0f) * 0.1f;
  lpos[3] = cx->light->pos[3];
}

static void trace(const OrCtx* cx, float o[4], float d[4], int depth, float color[3]);

/* Medium vertex of the homogeneous fog (no reference counterpart). Same terminal
 * rule as a surface vertex; isotropic phase function 1/(4 pi). */
static void fog_vertex(const OrCtx* cx, float o[4], float d[4], int depth, float color[3]) {
  const MiptParams* p = cx->p;
  cx->cn->fog_events++;
  if (depth >= p->max_depth || rng_next(cx->rng) <= p->rr_prob) {
    float lpos[4];
    jitter_light(cx, lpos);
    float v[4], vecr[4];
    for (int k = 0; k < 4; k++) v[k] = lpos[k] - o[k];
    or_normalize4(v, vecr);
    float r = or_length4(v);
    float den = (float)(((4.0 * 3.14159) * (double)r) * (double)r);
    float sd[4];
    or_normalize4(vecr, sd);
    if (exists_intersection(cx->sc, o, sd, r, cx->cn)) return;
    float tr = fog_expf(-(p->fog_sigma_t * r));
    float phase = 1.0f / (4.0f * 3.14159f);
    for (int k = 0; k < 3; k++) color[k] = (((cx->light->color[k] * phase) / den) * tr) * p->fog_albedo;
    return;
  }
  float u1 = rng_next(cx->rng);
  float u2 = rng_next(cx->rng);
  float z = 1.0f - 2.0f * u1;
  float s2 = 1.0f - z * z;
  float rxy = sqrtf(s2 < 0.f ? 0.f : s2);
  float phi = 6.28318530718f * u2;
  d[0] = rxy * cosf(phi); d[1] = rxy * sinf(phi); d[2] = z; d[3] = 0.0f;
  float tmp[3] = {0.f, 0.f, 0.f};
  trace(cx, o, d, depth + 1, tmp);
  for (int k = 0; k < 3; k++) color[k] = 0.0f + tmp[k] * p->fog_albedo;
}

/* bounceDiffuse (pathtracer.cpp:5-19) with orthonormalSystem / diffuseHemisphere (utils.cpp:10-22) */
static void bounce_diffuse(const OrCtx* cx, float o[4], float d[4], int depth, float color[3], const OrHit* hit) {
  const float* N = hit->normal;
  float Nt[3], Nb[3];
  if (fabsf(N[0]) > fabsf(N[1])) {
    float len = sqrtf(N[0] * N[0] + N[2] * N[2]);
    Nt[0] = N[2] / len; Nt[1] = 0.0f / len; Nt[2] = -N[0] / len;
  } else {
    float len = sqrtf(N[1] * N[1] + N[2] * N[2]);
    Nt[0] = 0.0f / len; Nt[1] = -N[2] / len; Nt[2] = N[1] / len;
  }
  or_cross3(N, Nt, Nb);
  const float* rotX = Nt; const float* rotY = Nb;
  /* g++ evaluates the two argument draws right to left: first draw -> b, second -> a */
  float b = rng_next(cx->rng);
  float a = rng_next(cx->rng);
  float r = sqrtf(1.0f - a * a);
  float phi = (float)((2 * 3.14) * (double)b);
  float s[3] = {cosf(phi) * r, a, sinf(phi) * r};
  float rd[4];
  for (int k = 0; k < 3; k++) rd[k] = (s[0] * rotY[k] + s[1] * N[k]) + s[2] * rotX[k];
  rd[3] = 1.0f;
  or_normalize4(rd, d);
  for (int k = 0; k < 4; k++) o[k] = o[k] + d[k] * 0.001f;
  float tmp[3] = {0.f, 0.f, 0.f};
  trace(cx, o, d, depth + 1, tmp);
  float dn = or_dot4(d, N);
  for (int k = 0; k < 3; k++)
    color[k] = color[k] + (((tmp[k] * hit->color[k]) * dn) * 1.0f) * cx->p->diffuse_weight;
}

/* bounceMirror (pathtracer.cpp:21-27) */
static void bounce_mirror(const OrCtx* cx, float o[4], float d[4], int depth, float color[3], const float n[4]) {
  float ct = or_dot4(n, d) * -1.0f;
  float k2 = ct * 2.0f;
  float nd[4];
  for (int k = 0; k < 4; k++) nd[k] = d[k] + n[k] * k2;
  or_normalize4(nd, d);
  float tmp[3] = {0.f, 0.f, 0.f};
  trace(cx, o, d, depth + 1, tmp);
  for (int k = 0; k < 3; k++) color[k] = color[k] + tmp[k] * 1.0f;
}

/* bounceGlass (pathtracer.cpp:29-63) */
static void bounce_glass(const OrCtx* cx, float o[4], float d[4], int depth, float color[3], const float n_in[4]) {
  const MiptParams* p = cx->p;
  float n[4] = {n_in[0], n_in[1], n_in[2], n_in[3]};
  float R0 = (p->ior_air - p->ior_glass) / (p->ior_air + p->ior_glass);
  R0 = R0 * R0;
  float eta = p->ior_glass;
  if (or_dot4(n, d) > 0) {
    for (int k = 0; k < 4; k++) n[k] = n[k] * -1.0f;
    eta = 1.0f / eta;
  }
  eta = 1.0f / eta;
  float c = or_dot4(n, d) * -1.0f;
  float cost1 = (c < 0.f) ? 0.f : c; /* std::max(c, 0.f) */
  float cost2 = 1.0f - (eta * eta) * (1.0f - cost1 * cost1);
  float Rprob = R0 + (1.0f - R0) * powf(1.0f - cost1, 5.0f);
  float nd[4];
  if (cost2 > 0 && rng_next(cx->rng) > Rprob) {
    float kk = eta * cost1 - sqrtf(cost2);
    for (int k = 0; k < 4; k++) nd[k] = (d[k] * eta) + (n[k] * kk);
  } else {
    float kk = cost1 * 2.0f;
    for (int k = 0; k < 4; k++) nd[k] = d[k] + n[k] * kk;
  }
  or_normalize4(nd, d);
  float tmp[3] = {0.f, 0.f, 0.f};
  trace(cx, o, d, depth + 1, tmp);
  for (int k = 0; k < 3; k++) color[k] = color[k] + (tmp[k] * 1.0f) * p->glass_weight;
}

/* bounceSpecular (pathtracer.cpp:65-70) */
static void bounce_specular(const OrCtx* cx, float o[4], float d[4], int depth, float color[3], const float n[4]) {
  float k2 = or_dot4(d, n) * 2.f;
  float nd[4];
  for (int k = 0; k < 4; k++) nd[k] = d[k] - n[k] * k2;
  or_normalize4(nd, d);
  float tmp[3] = {0.f, 0.f, 0.f};
  trace(cx, o, d, depth + 1, tmp);
  for (int k = 0; k < 3; k++) color[k] = color[k] + tmp[k];
}

/* trace (pathtracer.cpp:72-111) */
static void trace(const OrCtx* cx, float o[4], float d[4], int depth, float color[3]) {
  const MiptParams* p = cx->p;
  OrHit hit;
  if (!closest_intersection(cx->sc, o, d, &hit, cx->cn)) return;

  if (p->fog_sigma_t > 0.f) {
    /* free-flight distance in the medium, compared with the world-space hit distance */
    float len = sqrtf(or_dot3(d, d));
    float dist = hit.distance * len;
    float uf = rng_next(cx->rng);
    float s = -fog_logf(1.0f - uf) / p->fog_sigma_t;
    if (s < dist) {
      float k = s / len;
      for (int q = 0; q < 3; q++) o[q] = o[q] + d[q] * k;
      o[3] = 1.0f;
      fog_vertex(cx, o, d, depth, color);
      return;
    }
  }

  if (depth >= p->max_depth || rng_next(cx->rng) <= p->rr_prob) { /* :80 */
    float lpos[4], dl[3];
    jitter_light(cx, lpos);
    direct_light(cx, hit.position, hit.normal, lpos, cx->light->color, dl);
    if (hit.material == MIPT_MAT_GLASS) {
      for (int k = 0; k < 3; k++) color[k] = dl[k];
    } else {
      for (int k = 0; k < 3; k++) color[k] = (hit.color[k] * dl[k]) * 1.0f;
    }
    return;
  }
  for (int k = 0; k < 4; k++) o[k] = hit.position[k]; /* :100 */
  switch (hit.material) {
    case MIPT_MAT_DIFFUSE: cx->cn->diffuse_bounces++; bounce_diffuse(cx, o, d, depth, color, &hit); break;
    case MIPT_MAT_MIRROR: cx->cn->mirror_bounces++; bounce_mirror(cx, o, d, depth, color, hit.normal); break;
    case MIPT_MAT_GLASS: cx->cn->glass_bounces++; bounce_glass(cx, o, d, depth, color, hit.normal); break;
    case MIPT_MAT_SPECULAR: cx->cn->specular_bounces++; bounce_specular(cx, o, d, depth, color, hit.normal); break;
    default: break;
  }
}

/* the per-pixel body of Draw (main.cpp:124-137) */
static void pixel_sample(const OrCtx* cx, const MiptCamera* cam, const float M[16], int x, int y,
                         uint32_t sample, float color[3]) {
  const MiptParams* p = cx->p;
  rng_begin_path(cx->rng, p, (uint32_t)(x * p->height + y), sample);
  float u1 = rng_next(cx->rng);
  float u2 = rng_next(cx->rng);
  float dv[4];
  dv[0] = (float)((double)(x - p->width / 2) + (double)(u1 * 2.0f - 1.0f) * 0.1);
  dv[1] = (float)((double)(y - p->height / 2) + (double)(u2 * 2.0f - 1.0f) * 0.1);
  dv[2] = cam->focal;
  dv[3] = 0.0f;
  float o[4], d[4];
  for (int k = 0; k < 4; k++) o[k] = cam->pos[k];
  or_mat4_mul_vec4(M, dv, d);
  color[0] = color[1] = color[2] = 0.f;
  trace(cx, o, d, 0, color);
}

/* ------------------------------------------------------------------ exported batch entry points */

int or_closest_hit(const MiptTriangle* tris, size_t n_tris, const MiptSphere* sph, size_t n_sph,
                   const float* origin4, const float* dir4, size_t n_rays,
                   int32_t* tri_idx, float* t, float* position4) {
  OrScene sc = {tris, n_tris, sph, n_sph};
  OrCounters cn;
  memset(&cn, 0, sizeof cn);
  for (size_t i = 0; i < n_rays; i++) {
    OrHit h;
    if (closest_intersection(&sc, origin4 + 4 * i, dir4 + 4 * i, &h, &cn)) {
      tri_idx[i] = h.index;
      t[i] = h.distance;
      if (position4) memcpy(position4 + 4 * i, h.position, 16);
    } else {
      tri_idx[i] = -1;
    }
  }
  return 0;
}

int or_any_hit(const MiptTriangle* tris, size_t n_tris, const float* origin4, const float* dir4,
               const float* max_t, size_t n_rays, uint8_t* occluded) {
  OrScene sc = {tris, n_tris, NULL, 0};
  OrCounters cn;
  memset(&cn, 0, sizeof cn);
  for (size_t i = 0; i < n_rays; i++)
    occluded[i] = (uint8_t)exists_intersection(&sc, origin4 + 4 * i, dir4 + 4 * i, max_t[i], &cn);
  return 0;
}

int or_trace_paths(const MiptTriangle* tris, size_t n_tris, const MiptSphere* sph, size_t n_sph,
                   const MiptCamera* cam, const MiptLight* light, const MiptParams* params,
                   const int32_t* px, const int32_t* py, const uint32_t* sample, size_t n_paths,
                   float* rgb, uint32_t* ray_counts) {
  if (params->rng_mode == MIPT_RNG_MINSTD_GLOBAL) return -1;
  OrScene sc = {tris, n_tris, sph, n_sph};
  float M[16];
  or_mat4_mul(cam->rot_x, cam->rot_y, M); /* main.cpp:133 */
  /* listed paths are independent (own engine, own counters): spread over the host threads */
#pragma omp parallel for schedule(dynamic, 64)
  for (long long i = 0; i < (long long)n_paths; i++) {
    OrCounters cn;
    memset(&cn, 0, sizeof cn);
    OrRng rng = {params->rng_mode, 1u, 0u, 0u};
    OrCtx cx = {&sc, params, light, &rng, &cn};
    pixel_sample(&cx, cam, M, px[i], py[i], sample[i], rgb + 3 * i);
    if (ray_counts) { ray_counts[2 * i] = (uint32_t)cn.closest_rays; ray_counts[2 * i + 1] = (uint32_t)cn.shadow_rays; }
  }
  return 0;
}

int or_render_accumulate(const MiptTriangle* tris, size_t n_tris, const MiptSphere* sph, size_t n_sph,
                         const MiptCamera* cam, const MiptLight* light, const MiptParams* params,
                         uint32_t sample_begin, uint32_t sample_count, float* accum_rgb,
                         int threads, MiptStats* stats) {
  OrScene sc = {tris, n_tris, sph, n_sph};
  const int W = params->width, H = params->height;
  float M[16];
  or_mat4_mul(cam->rot_x, cam->rot_y, M);
  OrCounters total;
  memset(&total, 0, sizeof total);
  if (params->rng_mode == MIPT_RNG_MINSTD_GLOBAL || threads <= 1) {
    /* default-constructed engine: state 1; it keeps running across pixels and frames */
    OrRng rng = {params->rng_mode, 1u, 0u, 0u};
    OrCtx cx = {&sc, params, light, &rng, &total};
    for (uint32_t s = sample_begin; s < sample_begin + sample_count; s++)
      for (int x = 0; x < W; x++)
        for (int y = 0; y < H; y++) {
          float c[3];
          pixel_sample(&cx, cam, M, x, y, s, c);
          float* a = accum_rgb + ((size_t)x * H + y) * 3;
          a[0] += c[0]; a[1] += c[1]; a[2] += c[2];
        }
  } else {
#ifdef _OPENMP
    omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
      OrCounters local;
      memset(&local, 0, sizeof local);
      OrRng rng = {params->rng_mode, 1u, 0u, 0u};
      OrCtx cx = {&sc, params, light, &rng, &local};
#pragma omp for schedule(dynamic)
      for (int x = 0; x < W; x++)
        for (int y = 0; y < H; y++) {
          float* a = accum_rgb + ((size_t)x * H + y) * 3;
          for (uint32_t s = sample_begin; s < sample_begin + sample_count; s++) {
            float c[3];
            pixel_sample(&cx, cam, M, x, y, s, c);
            a[0] += c[0]; a[1] += c[1]; a[2] += c[2];
          }
        }
#pragma omp critical
      {
        total.closest_rays += local.closest_rays; total.shadow_rays += local.shadow_rays;
        total.tri_tests += local.tri_tests; total.gate_passes += local.gate_passes;
        total.sphere_tests += local.sphere_tests; total.fog_events += local.fog_events;
        total.diffuse_bounces += local.diffuse_bounces; total.mirror_bounces += local.mirror_bounces;
        total.glass_bounces += local.glass_bounces; total.specular_bounces += local.specular_bounces;
      }
    }
  }
  if (stats) {
    memset(stats, 0, sizeof *stats);
    stats->paths = (uint64_t)W * H * sample_count;
    stats->closest_rays = total.closest_rays; stats->shadow_rays = total.shadow_rays;
    stats->tri_tests = total.tri_tests; stats->gate_passes = total.gate_passes;
    stats->sphere_tests = total.sphere_tests; stats->fog_events = total.fog_events;
    stats->diffuse_bounces = total.diffuse_bounces; stats->mirror_bounces = total.mirror_bounces;
    stats->glass_bounces = total.glass_bounces; stats->specular_bounces = total.specular_bounces;
  }
  return 0;
}

void or_resolve_argb(const float* accum_rgb, int width, int height, uint32_t n_samples, uint32_t* screen_buffer) {
  /* main.cpp:143: pixels[x][y] / (samplesSeenSoFar * (1 / (2 * 3.1415f))) */
  float div = (float)(int)n_samples * (1.0f / (2.0f * 3.1415f));
  for (int x = 0; x < width; x++)
    for (int y = 0; y < height; y++) {
      const float* a = accum_rgb + ((size_t)x * height + y) * 3;
      uint32_t ch[3];
      for (int k = 0; k < 3; k++) {
        float c = 255.0f * (a[k] / div);
        /* glm::clamp(x, 0, 255) = min(max(x, 0), 255) (SDLauxiliary.h:156-158) */
        float lo = (c < 0.f) ? 0.f : c;
        float cl = (255.f < lo) ? 255.f : lo;
        ch[k] = (cl == cl) ? (uint32_t)cl : 0u;
      }
      screen_buffer[(size_t)y * width + x] = (128u << 24) + (ch[0] << 16) + (ch[1] << 8) + ch[2];
    }
}
