/*
 * pt_oracle.h -- CPU restatement of the reference path-tracing hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT. Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library, and only as the checker /
 * the reported CPU baseline. libmipt.so never links or calls it.
 *
 * PARITY STATUS. The reference's integrated path (Draw -> trace ->
 * closestIntersection) cannot be built in this image: every translation unit of
 * CW/raytracer/Source includes definitions.h, which includes <SDL.h>, and SDL2 is
 * not installed. What IS pinned by running reference code here:
 *   - the four scene builders (TestModelH.h compiles on its own with the vendored
 *     GLM): tests/golden/scene_*.bin are its outputs, matched bit-for-bit;
 *   - every GLM primitive the path uses (determinant, inverse, mat*vec, dot,
 *     cross, normalize, length): tests/golden/glm_kat.bin, matched bit-for-bit;
 *   - the libstdc++ RNG stream (system library the reference links against).
 * The composition of those pieces (gate order, tie rule, draw order, Russian
 * roulette, Fresnel branch) follows the cited source lines and is pinned by
 * known answers the reference produced (tests/golden/survey_kat.json): the
 * centre-pixel hit distance bit pattern and the exact whole-frame counts of
 * closestIntersection / existsIntersection / bounce* / glm::determinant /
 * glm::inverse calls for four (scene, size) frames, all reproduced exactly by
 * MIPT_RNG_MINSTD_GLOBAL renders of this file. Radiance VALUES are pinned at
 * image level by the reference's own rendering of its glassWolf scene
 * (CW/raytracer/screenshot.bmp, kept as tests/golden/glasswolf_screen_500x500.npz):
 * this file's paths through the diffuse room reproduce that image's pixel values
 * to Monte-Carlo noise, and built with the reference's own flags (-Ofast) it also
 * reproduces the image's glass region to noise (0.98 .. 1.00), the strict build
 * being 1.15x brighter there (tests/test_reference_screenshot.py). Per-path bit patterns
 * have no reference vector in this image. Fog has no reference code at all:
 * "parity unpinned".
 *
 * Arithmetic: build with -O2 -ffp-contract=off (see Makefile); every float
 * product and sum is rounded separately, as in the reference compiled without
 * -Ofast. Plain C, no dependencies beyond libm.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include "../include/mipt.h" /* POD scene / camera / params types only */

#ifdef __cplusplus
extern "C" {
#endif

enum { OR_SCENE_CORNELL = 0, OR_SCENE_COLOR_BLEEDING = 1, OR_SCENE_GLASS_CUBE = 2, OR_SCENE_NONE = 3 };

/* LoadTestModel / ...ColorBleeding / ...GlassCube / ...None
 * (CW/raytracer/Source/TestModelH.h:55-201, 203-349, 351-482, 484-580).
 * `out` needs room for 30 triangles; returns the triangle count (30/30/22/10). */
int or_load_test_model(int which, MiptTriangle* out);

/* reset_camera (camera.cpp:4-13) with focal = width; reset_light (light.cpp:6-11). */
void or_default_camera(MiptCamera* cam, int width);
void or_default_light(MiptLight* light);
void or_params_default(MiptParams* p, int width, int height);

/* minstd_rand0 step followed by generate_canonical<float,24> (definitions.h:40-41;
 * /usr/include/c++/11/bits/random.tcc:3348-3380). *state is the engine state. */
float or_rng_canonical(uint32_t* state);
uint32_t or_path_key(uint32_t seed, uint32_t pixel, uint32_t sample);
/* draw k (0-based) of the counter-based stream of `key` */
float or_counter_uniform(uint32_t key, uint32_t k);

/* closestIntersection (intersections.cpp:6-55), one call per ray. */
int or_closest_hit(const MiptTriangle* tris, size_t n_tris, const MiptSphere* sph, size_t n_sph,
                   const float* origin4, const float* dir4, size_t n_rays,
                   int32_t* tri_idx, float* t, float* position4);
/* existsIntersection (intersections.cpp:57-85). */
int or_any_hit(const MiptTriangle* tris, size_t n_tris, const float* origin4, const float* dir4,
               const float* max_t, size_t n_rays, uint8_t* occluded);

/* One `trace` (pathtracer.cpp:72-111) per listed (pixel, sample), primary ray as in
 * main.cpp:128-133. rgb: 3 floats per path; ray_counts (nullable): closest, shadow. */
int or_trace_paths(const MiptTriangle* tris, size_t n_tris, const MiptSphere* sph, size_t n_sph,
                   const MiptCamera* cam, const MiptLight* light, const MiptParams* params,
                   const int32_t* px, const int32_t* py, const uint32_t* sample, size_t n_paths,
                   float* rgb, uint32_t* ray_counts);

/* The pixel loop of Draw (main.cpp:121-139) for samples [begin, begin+count),
 * adding into accum_rgb[((x*H)+y)*3]. threads <= 1: single thread (required for
 * MIPT_RNG_MINSTD_GLOBAL); otherwise OpenMP over x like main.cpp:121.
 * stats (nullable) receives call / test counters. */
int or_render_accumulate(const MiptTriangle* tris, size_t n_tris, const MiptSphere* sph, size_t n_sph,
                         const MiptCamera* cam, const MiptLight* light, const MiptParams* params,
                         uint32_t sample_begin, uint32_t sample_count, float* accum_rgb,
                         int threads, MiptStats* stats);

/* Draw's tonemap loop (main.cpp:141-145) + PutPixelSDL (SDLauxiliary.h:149-161). */
void or_resolve_argb(const float* accum_rgb, int width, int height, uint32_t n_samples,
                     uint32_t* screen_buffer);

/* GLM primitives exactly as the path uses them, exported so that the golden
 * vectors produced by the vendored GLM can be checked one primitive at a time. */
float or_det3(const float p[3], const float q[3], const float r[3]);
void or_inverse3_mul(const float p[3], const float q[3], const float r[3], const float b[3], float x[3]);
float or_dot3(const float a[3], const float b[3]);
float or_dot4(const float a[4], const float b[4]);
void or_cross3(const float x[3], const float y[3], float out[3]);
void or_normalize3(const float v[3], float out[3]);
void or_normalize4(const float v[4], float out[4]);
float or_length4(const float v[4]);
void or_mat4_mul(const float a[16], const float b[16], float out[16]);
void or_mat4_mul_vec4(const float m[16], const float v[4], float out[4]);

/* the fog's exp / log (fixed binary32 operation sequences, see pt_oracle.c) */
float or_fog_expf(float x);
float or_fog_logf(float x);

#ifdef __cplusplus
}
#endif
#endif
