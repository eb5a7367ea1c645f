// ref_dump.cpp -- runs the parts of the reference that build in this image and
// writes their outputs as golden vectors. TEST INFRASTRUCTURE.
//
// Compiled against the reference tree where it lies (see ../Makefile, target
// _ref/ref_dump); nothing from the reference is copied into this repository.
// Included from the reference, unmodified:
//   CW/raytracer/Source/TestModelH.h  (Triangle, LoadTestModel*; pulls material.h)
//   CW/glm/glm/glm.hpp                (vendored GLM 0.9.7.2)
// The other sources of the raytracer include <SDL.h> through definitions.h and
// cannot be built here (SDL2 is not installed), so the integrated path is not
// part of this harness.
//
// usage: ref_dump <output directory>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include <glm/glm.hpp>
#include "TestModelH.h"

static uint32_t material_code(const char* type) {
  if (!strcmp(type, "diffuse")) return 0;
  if (!strcmp(type, "mirror")) return 1;
  if (!strcmp(type, "glass")) return 2;
  if (!strcmp(type, "specular")) return 3;
  return 4;
}

static void put(FILE* f, const void* p, size_t n) {
  if (fwrite(p, 1, n, f) != n) { perror("fwrite"); exit(1); }
}

static void dump_scene(const std::string& path, void (*load)(std::vector<Triangle>&)) {
  std::vector<Triangle> tris;
  load(tris);
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) { perror(path.c_str()); exit(1); }
  for (const Triangle& t : tris) {
    // 80-byte record: v0, v1, v2 (vec4), color (vec3), normal (vec4), material code
    put(f, &t.v0[0], 16); put(f, &t.v1[0], 16); put(f, &t.v2[0], 16);
    put(f, &t.color[0], 12); put(f, &t.normal[0], 16);
    uint32_t m = material_code(t.material.type);
    put(f, &m, 4);
  }
  fclose(f);
  printf("%s: %zu triangles, sizeof(Triangle)=%zu\n", path.c_str(), tris.size(), sizeof(Triangle));
}

// deterministic inputs spanning several magnitudes (values like those on the path:
// box coordinates in [-1,1], un-normalised primary directions ~1e3, tiny offsets)
static uint32_t lcg_state = 0x1234567u;
static float frand() {
  lcg_state = lcg_state * 1664525u + 1013904223u;
  float u = (float)(lcg_state >> 8) * (1.0f / 16777216.0f);
  return u * 2.0f - 1.0f;
}
static float fscaled(int rec) {
  static const float scales[6] = {1.0f, 1.0f, 1.0f, 1024.0f, 1e-3f, 37.5f};
  return frand() * scales[rec % 6];
}

static void dump_glm(const std::string& path, int n_records) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) { perror(path.c_str()); exit(1); }
  for (int rec = 0; rec < n_records; rec++) {
    glm::vec3 p(fscaled(rec), fscaled(rec), fscaled(rec));
    glm::vec3 q(frand(), frand(), frand());
    glm::vec3 r(frand(), frand(), frand());
    glm::vec3 b(fscaled(rec + 1), fscaled(rec + 1), fscaled(rec + 1));
    glm::vec4 a4(frand(), frand(), frand(), (rec & 1) ? 1.0f : frand());
    glm::vec4 b4(fscaled(rec), fscaled(rec), fscaled(rec), (rec & 2) ? 0.0f : 1.0f);
    glm::mat4 A, B;
    for (int c = 0; c < 4; c++)
      for (int k = 0; k < 4; k++) { A[c][k] = frand(); B[c][k] = frand(); }
    float s = fscaled(rec + 2);

    glm::mat3 m3(p, q, r);
    float det = glm::determinant(m3);
    glm::mat3 mb = m3; mb[0] = b;
    float detb = glm::determinant(mb);
    glm::vec3 x = glm::inverse(m3) * b;
    float d3 = glm::dot(p, q);
    glm::vec3 cr = glm::cross(p, q);
    glm::vec3 n3 = glm::normalize(p);
    float d4 = glm::dot(a4, b4);
    glm::vec4 n4 = glm::normalize(b4);
    float l4 = glm::length(b4);
    glm::vec3 dv = p / s;
    glm::mat4 AB = A * B;
    glm::vec4 Av = A * b4;

    // inputs: p q r b (12) a4 b4 (8) A B (32) s (1) = 53 floats
    put(f, &p[0], 12); put(f, &q[0], 12); put(f, &r[0], 12); put(f, &b[0], 12);
    put(f, &a4[0], 16); put(f, &b4[0], 16);
    put(f, &A[0][0], 64); put(f, &B[0][0], 64);
    put(f, &s, 4);
    // outputs: det detb x(3) d3 cr(3) n3(3) d4 n4(4) l4 dv(3) AB(16) Av(4) = 41 floats
    put(f, &det, 4); put(f, &detb, 4); put(f, &x[0], 12); put(f, &d3, 4);
    put(f, &cr[0], 12); put(f, &n3[0], 12); put(f, &d4, 4); put(f, &n4[0], 16);
    put(f, &l4, 4); put(f, &dv[0], 12); put(f, &AB[0][0], 64); put(f, &Av[0], 16);
  }
  fclose(f);
  printf("%s: %d records of 94 floats\n", path.c_str(), n_records);
}

// std::default_random_engine + uniform_real_distribution<float>(0,1): the reference's
// `generator` / `distribution` (definitions.h:40-41), here straight from libstdc++.
static void dump_rng(const std::string& path) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) { perror(path.c_str()); exit(1); }
  const uint32_t seeds[4] = {0u /* default-constructed */, 12345u, 2147483646u, 2147483647u};
  for (int s = 0; s < 4; s++) {
    std::default_random_engine gen;
    if (s > 0) gen.seed(seeds[s]);
    std::uniform_real_distribution<float> dist(0, 1);
    for (int i = 0; i < 64; i++) { float u = dist(gen); put(f, &u, 4); }
  }
  fclose(f);
  printf("%s: 4 x 64 floats (default seed, 12345, 2147483646, 2147483647)\n", path.c_str());
}

// Which argument of f(g(), g()) does this compiler evaluate first? The reference calls
// diffuseHemisphere(distribution(generator), distribution(generator)) (pathtracer.cpp:8).
static int counter = 0;
static int next_id() { return counter++; }
static int first_arg_id(int a, int b) { (void)b; return a; }

int main(int argc, char** argv) {
  if (argc != 2) { fprintf(stderr, "usage: ref_dump <outdir>\n"); return 2; }
  std::string out = argv[1];
  dump_scene(out + "/scene_cornell.bin", LoadTestModel);
  dump_scene(out + "/scene_colorbleeding.bin", LoadTestModelColorBleeding);
  dump_scene(out + "/scene_glasscube.bin", LoadTestModelGlassCube);
  dump_scene(out + "/scene_none.bin", LoadTestModelNone);
  dump_glm(out + "/glm_kat.bin", 3000);
  dump_rng(out + "/rng_kat.bin");
  int id = first_arg_id(next_id(), next_id());
  FILE* f = fopen((out + "/argorder.txt").c_str(), "w");
  // id == 1: the LAST argument was evaluated first (right to left)
  fprintf(f, "%s\n", id == 1 ? "right-to-left" : "left-to-right");
  fclose(f);
  printf("argument evaluation: %s\n", id == 1 ? "right-to-left" : "left-to-right");
  return 0;
}
