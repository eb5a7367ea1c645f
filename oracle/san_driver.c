/* san_driver.c -- runs the CPU oracle's entry points once under the host sanitizers
 * (make -C oracle sanitize: -fsanitize=address,undefined): closest-hit and any-hit
 * batches incl. degenerate rays, listed paths, a multi-threaded render with fog and a sphere, the
 * tonemap. Exits 0 when nothing was reported (the sanitizers abort otherwise). Test infrastructure. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pt_oracle.h"

int main(void) {
  MiptTriangle tris[30];
  const int n = or_load_test_model(OR_SCENE_GLASS_CUBE, tris);
  MiptSphere sph;
  memset(&sph, 0, sizeof sph);
  sph.radius = 0.3f;
  sph.center[0] = -0.45f; sph.center[1] = 0.7f; sph.center[2] = -0.55f; sph.center[3] = 1.f;
  sph.color[0] = 1.f;
  sph.material = MIPT_MAT_GLASS;
  enum { W = 24, H = 20, NR = 512 };
  MiptCamera cam;
  MiptLight light;
  MiptParams p;
  or_default_camera(&cam, W);
  or_default_light(&light);
  light.pos[0] = 0.7f; light.pos[1] = 0.95f; light.pos[2] = -0.7f;
  or_params_default(&p, W, H);
  p.fog_sigma_t = 0.2f;
  p.fog_albedo = 0.8f;

  float* o = malloc(sizeof(float) * 4 * NR);
  float* d = malloc(sizeof(float) * 4 * NR);
  float* mx = malloc(sizeof(float) * NR);
  int32_t* idx = malloc(sizeof(int32_t) * NR);
  float* t = malloc(sizeof(float) * NR);
  float* pos = malloc(sizeof(float) * 4 * NR);
  uint8_t* occ = malloc(NR);
  uint32_t st = 12345u;
  for (int i = 0; i < NR; i++) {
    for (int k = 0; k < 3; k++) {
      o[4 * i + k] = or_rng_canonical(&st) * 1.8f - 0.9f;
      d[4 * i + k] = or_rng_canonical(&st) * 2.f - 1.f;
    }
    o[4 * i + 3] = 1.f;
    d[4 * i + 3] = (i & 1) ? 1.f : 0.f;
    if (i % 17 == 0) d[4 * i] = d[4 * i + 1] = d[4 * i + 2] = 0.f; /* zero direction: 0/0 */
    if (i % 19 == 0) { d[4 * i] *= 1e30f; d[4 * i + 1] *= 1e-30f; }  /* overflow / subnormals */
    mx[i] = or_rng_canonical(&st) * 2.f;
  }
  if (or_closest_hit(tris, (size_t)n, &sph, 1, o, d, NR, idx, t, pos) != 0) return 2;
  if (or_any_hit(tris, (size_t)n, o, d, mx, NR, occ) != 0) return 3;

  int32_t px[64], py[64];
  uint32_t sm[64];
  float rgb[64 * 3];
  uint32_t counts[64 * 2];
  for (int i = 0; i < 64; i++) { px[i] = i % W; py[i] = (i * 7) % H; sm[i] = (uint32_t)i; }
  for (uint32_t mode = 0; mode < 2; mode++) {
    p.rng_mode = mode;
    if (or_trace_paths(tris, (size_t)n, &sph, 1, &cam, &light, &p, px, py, sm, 64, rgb, counts) != 0) return 4;
  }
  float* acc = calloc((size_t)W * H * 3, sizeof(float));
  MiptStats stats;
  p.rng_mode = MIPT_RNG_COUNTER;
  if (or_render_accumulate(tris, (size_t)n, &sph, 1, &cam, &light, &p, 0, 6, acc, 4, &stats) != 0) return 5;
  p.rng_mode = MIPT_RNG_MINSTD_GLOBAL;
  p.fog_sigma_t = 0.f;
  if (or_render_accumulate(tris, (size_t)n, NULL, 0, &cam, &light, &p, 0, 2, acc, 1, &stats) != 0) return 6;
  uint32_t* screen = malloc(sizeof(uint32_t) * W * H);
  or_resolve_argb(acc, W, H, 8, screen);
  double sum = 0;
  for (int i = 0; i < W * H * 3; i++) sum += acc[i];
  printf("sanitizer run ok: %llu closest rays, image sum %.6g\n", (unsigned long long)stats.closest_rays, sum);
  free(o); free(d); free(mx); free(idx); free(t); free(pos); free(occ); free(acc); free(screen);
  return isfinite(sum) ? 0 : 7;
}
