// host_demo.cpp -- a stand-alone C++ host that drives libmipt.so through mipt_adapter.hpp the
// way the reference's main.cpp would: scene vectors of `Triangle` / `Sphere`, `Camera`, `Light`,
// `closestIntersection` for single rays and `Draw` into a `screen` buffer. The vector types below
// carry the member names the reference's glm types have (glm itself is not needed here).
//
//   host_demo <scene 0..3> <size> <frames> <samples_per_frame> [multi|multi-reduce|multi-frame]
// (multi: Draw goes through mipt_host::MultiGpuScene on device 0 -- the several-GPU path with one rank,
//  image handed from device to device; multi-reduce: the RCCL sum-reduce of the per-device images;
//  multi-frame: mipt_host::MultiFrame over two scene replicas sharing device 0 -- the image stays on
//  the device, only the screen comes back, `pixels` is read once at the end)
// prints one line of key=value results that tests/test_gpu_host_cpp.py compares with the
// Python host's results for the same inputs.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mipt_adapter.hpp"

struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };
struct mat4 {
  vec4 c[4];
  struct Col { const vec4* v; float operator[](int r) const { return (&v->x)[r]; } };
  Col operator[](int col) const { return Col{&c[col]}; }
};
struct Material { const char* type; };
struct Triangle { vec4 v0, v1, v2; vec3 color; vec4 normal; Material material; };
struct Sphere {
  float radius; vec4 center; vec3 color; Material material;
  vec4 normal(const vec4& p) const { (void)p; return vec4{0, 0, 0, 0}; }
};
struct Camera { float focalLength; vec4 cameraPos; mat4 cameraRotationX, cameraRotationY; };
struct Light { vec4 lightPos; vec3 lightColor; };
struct Ray { vec4 origin, direction; };
struct Intersection { vec4 position; float distance; vec3 color; vec4 normal; Material material; };
struct screen { int height, width; uint32_t* buffer; };

int main(int argc, char** argv) {
  const int which = argc > 1 ? atoi(argv[1]) : 2;
  const int size = argc > 2 ? atoi(argv[2]) : 64;
  const int frames = argc > 3 ? atoi(argv[3]) : 2;
  const int spf = argc > 4 ? atoi(argv[4]) : 4;
  const bool reduce = argc > 5 && strcmp(argv[5], "multi-reduce") == 0;
  const bool multi = reduce || (argc > 5 && strcmp(argv[5], "multi") == 0);
  const bool multi_frame = argc > 5 && strcmp(argv[5], "multi-frame") == 0;
  try {
    MiptTriangle raw[30];
    const int n = mipt_load_test_model(which, raw, 30);
    if (n <= 0) { fprintf(stderr, "bad scene\n"); return 2; }
    std::vector<Triangle> triangles;
    for (int i = 0; i < n; ++i) {
      const MiptTriangle& r = raw[i];
      Triangle t;
      t.v0 = {r.v0[0], r.v0[1], r.v0[2], r.v0[3]};
      t.v1 = {r.v1[0], r.v1[1], r.v1[2], r.v1[3]};
      t.v2 = {r.v2[0], r.v2[1], r.v2[2], r.v2[3]};
      t.color = {r.color[0], r.color[1], r.color[2]};
      t.normal = {r.normal[0], r.normal[1], r.normal[2], r.normal[3]};
      t.material = Material{mipt_host::material_name(r.material)};
      triangles.push_back(t);
    }
    std::vector<Sphere> spheres;
    MiptCamera mc; mipt_default_camera(&mc, size);
    MiptLight ml; mipt_default_light(&ml, which >= 2);
    Camera cam;
    cam.focalLength = mc.focal;
    cam.cameraPos = {mc.pos[0], mc.pos[1], mc.pos[2], mc.pos[3]};
    for (int c = 0; c < 4; ++c) {
      cam.cameraRotationX.c[c] = {mc.rot_x[4 * c], mc.rot_x[4 * c + 1], mc.rot_x[4 * c + 2], mc.rot_x[4 * c + 3]};
      cam.cameraRotationY.c[c] = {mc.rot_y[4 * c], mc.rot_y[4 * c + 1], mc.rot_y[4 * c + 2], mc.rot_y[4 * c + 3]};
    }
    Light light{{ml.pos[0], ml.pos[1], ml.pos[2], ml.pos[3]}, {ml.color[0], ml.color[1], ml.color[2]}};

    mipt_host::GpuScene gpu(triangles, spheres);

    // single-ray queries, as trace() would issue them
    Ray ray{{0, 0, -3, 1}, {0.25f * size, -0.125f * size, (float)size, 0}};
    Intersection inter{};
    const bool hit = mipt_host::closestIntersection(gpu, ray, triangles, spheres, inter);
    uint32_t tbits = 0;
    memcpy(&tbits, &inter.distance, 4);
    Ray up{{0.f, 0.9f, -0.5f, 1.f}, {0.f, -1.f, 0.f, 0.f}};
    Intersection lim{};
    lim.distance = 1.5f;
    const bool occluded = mipt_host::existsIntersection(gpu, up, triangles, lim);

    // progressive rendering into the reference's column-wise pixel store
    std::vector<uint32_t> buf((size_t)size * size);
    screen scr{size, size, buf.data()};
    vec3** pixels = new vec3*[size];
    for (int x = 0; x < size; ++x) pixels[x] = new vec3[size]();
    int samplesSeenSoFar = 0;
    if (multi_frame) {
      mipt_host::MultiGpuScene all(triangles, spheres, std::vector<int>{0, 0}, true);
      mipt_host::MultiFrame frame(all, size, size);
      for (int f = 0; f < frames; ++f) {
        frame.add_samples(cam, light, spf);
        frame.present(&scr);
      }
      while (!frame.done()) {}
      frame.read_pixels(pixels);
      samplesSeenSoFar = frame.samplesSeenSoFar();
    } else if (multi) {
      mipt_host::MultiGpuScene all(triangles, spheres, std::vector<int>{0});
      all.sum_reduce = reduce;
      for (int f = 0; f < frames; ++f) mipt_host::Draw(all, &scr, cam, light, pixels, samplesSeenSoFar, spf);
    } else {
      for (int f = 0; f < frames; ++f) mipt_host::Draw(gpu, &scr, cam, light, pixels, samplesSeenSoFar, spf);
    }
    double psum = 0;
    uint64_t ssum = 0;
    for (int x = 0; x < size; ++x)
      for (int y = 0; y < size; ++y) psum += (double)pixels[x][y].x + pixels[x][y].y + pixels[x][y].z;
    for (uint32_t v : buf) ssum += v;
    printf("n_tris=%d hit=%d material=%s tbits=%08x occluded=%d samples=%d pixel_sum=%.9e screen_sum=%llu\n", n,
           (int)hit, hit ? inter.material.type : "-", tbits, (int)occluded, samplesSeenSoFar, psum,
           (unsigned long long)ssum);
    for (int x = 0; x < size; ++x) delete[] pixels[x];
    delete[] pixels;
  } catch (const std::exception& e) {
    fprintf(stderr, "host_demo: %s\n", e.what());
    return 1;
  }
  return 0;
}
