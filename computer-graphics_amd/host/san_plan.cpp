// The scan-unit planner of the linear scan's candidate filter (mipt_plan_scan_units: coplanar pairs,
// axis pairs, parallelepiped faces) and the other entry points that run without a GPU, under
// AddressSanitizer + UndefinedBehaviorSanitizer (`make -C computer-graphics_amd/host sanitize`).
// The library's HIP translation unit is compiled with the sanitizers on its host side only; no
// kernel is launched.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <random>
#include <vector>

#include "mipt.h"

static int failures = 0;
#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) { std::printf("FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #cond, mipt_last_error()); ++failures; } \
  } while (0)

static MiptTriangle tri(const float a[3], const float b[3], const float c[3]) {
  MiptTriangle t;
  std::memset(&t, 0, sizeof t);
  for (int k = 0; k < 3; ++k) { t.v0[k] = a[k]; t.v1[k] = b[k]; t.v2[k] = c[k]; }
  t.v0[3] = t.v1[3] = t.v2[3] = 1.f;
  const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
  float n[3] = {e2[1] * e1[2] - e2[2] * e1[1], e2[2] * e1[0] - e2[0] * e1[2], e2[0] * e1[1] - e2[1] * e1[0]};
  const float len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
  for (int k = 0; k < 3; ++k) t.normal[k] = len > 0.f ? n[k] / len : 0.f;
  t.normal[3] = 1.f;
  t.color[0] = t.color[1] = t.color[2] = 0.5f;
  t.material = MIPT_MAT_DIFFUSE;
  return t;
}

// a quad as two triangles sharing the diagonal, the way TestModelH.h writes its walls
static void quad(std::vector<MiptTriangle>& out, const float p[3], const float u[3], const float v[3]) {
  float a[3], b[3], c[3], d[3];
  for (int k = 0; k < 3; ++k) { a[k] = p[k]; b[k] = p[k] + u[k]; c[k] = p[k] + u[k] + v[k]; d[k] = p[k] + v[k]; }
  out.push_back(tri(a, b, c));
  out.push_back(tri(a, c, d));
}

static void plan(const std::vector<MiptTriangle>& t, const char* what) {
  int pairs = -1, axis = -1, boxes = -1;
  const int rc = mipt_plan_scan_units(t.data(), t.size(), &pairs, &axis, &boxes);
  if (rc != MIPT_OK) { std::printf("FAILED plan(%s): %s\n", what, mipt_last_error()); ++failures; return; }
  CHECK(pairs >= 0 && axis >= 0 && axis <= pairs && boxes >= 0 && (size_t)pairs * 2 <= t.size());
}

int main() {
  // the reference's scenes
  for (int which = 0; which < 4; ++which) {
    std::vector<MiptTriangle> t(30);
    const int n = mipt_load_test_model(which, t.data(), t.size());
    CHECK(n > 0);
    t.resize(n > 0 ? (size_t)n : 0);
    plan(t, "test model");
    for (size_t cut = 0; cut < t.size(); ++cut) plan(std::vector<MiptTriangle>(t.begin(), t.begin() + (long)cut), "prefix");
  }
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  // random boxes (turned, sheared, faces missing) and loose quads, up to the linear scan's sizes
  for (int round = 0; round < 40; ++round) {
    std::vector<MiptTriangle> t;
    const int n_boxes = 1 + round % 12;
    for (int b = 0; b < n_boxes; ++b) {
      float p[3], e[3][3];
      for (int k = 0; k < 3; ++k) p[k] = U(rng);
      for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) e[j][k] = (j == k ? 0.3f : 0.f) + (round % 3 ? 0.2f * U(rng) : 0.f);
      for (int f = 0; f < 6; ++f) {
        if ((rng() & 7u) == 0) continue;  // a face left out
        const int ax = f >> 1, u = (ax + 1) % 3, v = (ax + 2) % 3;
        float q[3];
        for (int k = 0; k < 3; ++k) q[k] = p[k] + ((f & 1) ? e[ax][k] : 0.f);
        quad(t, q, e[u], e[v]);
      }
    }
    plan(t, "random boxes");
  }
  // eight tilted two-face boxes in one chunk of 32 triangles (the count field of a chunk holds seven)
  {
    std::vector<MiptTriangle> t;
    for (int b = 0; b < 8; ++b) {
      const float p[3] = {0.2f * b - 0.8f, 0.1f * b, 0.05f * b};
      const float u[3] = {0.1f, 0.02f, 0.f}, v[3] = {-0.02f, 0.1f, 0.01f}, w[3] = {0.f, -0.01f, 0.1f};
      quad(t, p, u, v);
      quad(t, p, u, w);
    }
    plan(t, "eight two-face boxes");
  }
  // degenerate and hostile geometry
  {
    const float inf = std::numeric_limits<float>::infinity(), nan = std::numeric_limits<float>::quiet_NaN();
    const float big = 3.0e38f, tiny = 1.0e-45f;
    const float vals[] = {0.f, -0.f, 1.f, inf, -inf, nan, big, -big, tiny, 1.0e12f, 1.0e-30f};
    std::vector<MiptTriangle> t;
    for (int i = 0; i < 300; ++i) {
      float a[3], b[3], c[3];
      for (int k = 0; k < 3; ++k) {
        a[k] = vals[rng() % 11]; b[k] = vals[rng() % 11]; c[k] = (rng() & 1u) ? a[k] : vals[rng() % 11];
      }
      t.push_back(tri(a, b, c));
      if (i % 3 == 0) t.push_back(t.back());  // exact duplicates, consecutive
    }
    plan(t, "hostile");
    const float z[3] = {0.f, 0.f, 0.f};
    plan(std::vector<MiptTriangle>(64, tri(z, z, z)), "64 points");
  }
  plan(std::vector<MiptTriangle>(), "empty");
  int p = 0;
  CHECK(mipt_plan_scan_units(nullptr, 5, &p, &p, &p) < 0);
  {
    std::vector<MiptTriangle> one(1);
    std::memset(one.data(), 0, sizeof(MiptTriangle));
    CHECK(mipt_plan_scan_units(one.data(), 1, nullptr, nullptr, nullptr) == MIPT_OK);
  }

  // what else runs without a GPU
  MiptParams par;
  mipt_params_default(&par, 64, 48);
  CHECK(par.width == 64 && par.height == 48);
  CHECK(mipt_version() != nullptr && std::strlen(mipt_version()) > 0);
  (void)mipt_path_key(1u, 2u, 3u);
  for (int n = 1; n <= 9; ++n) {
    uint32_t covered = 0, covered_items = 0;
    for (int r = 0; r < n; ++r) {
      uint32_t b = 0, c = 0;
      mipt_multi_sample_range(5u, 1000u, r, n, &b, &c);
      CHECK(b == 5u + covered);
      covered += c;
      mipt_multi_item_range(5u, 1000u, r, n, &b, &c);
      CHECK(b == 5u + covered_items);
      covered_items += c;
    }
    CHECK(covered == 1000u && covered_items == 1000u);
  }
  if (failures) { std::printf("%d check(s) failed\n", failures); return 1; }
  std::printf("planner sanitizer run ok\n");
  return 0;
}
