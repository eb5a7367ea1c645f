// mipt_adapter.hpp -- C++ host adapter between the reference raytracer's own types and the
// C ABI of libmipt.so.
//
// The reference (CW/raytracer/Source) is one translation unit built on glm::vec3/vec4/mat4.
// This header is duck-typed on member names (.x .y .z .w, operator[] for matrices), so it
// takes the reference's `Triangle` (TestModelH.h:11-49), `Sphere` (sphere.h:9-16), `Camera`,
// `Light`, `Intersection` (definitions.h:16-37) and `Ray` (ray.h:6-13) as they are, and any
// other struct with the same member names. It offers the reference's call shapes:
//
//   mipt_host::GpuScene gpu(triangles, spheres);                        after scene load, main.cpp:63-94
//   mipt_host::closestIntersection(gpu, ray, triangles, spheres, inter) intersections.h:7
//   mipt_host::existsIntersection(gpu, ray, triangles, inter)           intersections.h:12
//   mipt_host::Draw(gpu, screen, cam, light, pixels, samplesSeenSoFar)  definitions.h:61
//
// Errors: the reference has no error channel (its functions return bool / void), so the
// adapter throws std::runtime_error carrying mipt_last_error().
#pragma once

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mipt.h"

namespace mipt_host {

inline void check(int rc) {
  if (rc != MIPT_OK) throw std::runtime_error(std::string("mipt: ") + mipt_last_error());
}

// material.h:5-11 stores `const char* type`; pathtracer.cpp:88,103-110 compares it with strcmp
inline uint32_t material_code(const char* type) {
  if (!type) return MIPT_MAT_NONE;
  if (!std::strcmp(type, "diffuse")) return MIPT_MAT_DIFFUSE;
  if (!std::strcmp(type, "mirror")) return MIPT_MAT_MIRROR;
  if (!std::strcmp(type, "glass")) return MIPT_MAT_GLASS;
  if (!std::strcmp(type, "specular")) return MIPT_MAT_SPECULAR;
  return MIPT_MAT_NONE;
}

inline const char* material_name(uint32_t code) {
  static const char* names[] = {"diffuse", "mirror", "glass", "specular", "none"};
  return names[code <= (uint32_t)MIPT_MAT_NONE ? code : (uint32_t)MIPT_MAT_NONE];
}

template <class V4> inline void put4(float out[4], const V4& v) { out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w; }
template <class V3> inline void put3(float out[3], const V3& v) { out[0] = v.x; out[1] = v.y; out[2] = v.z; }

template <class TriangleT> inline MiptTriangle to_mipt(const TriangleT& t) {
  MiptTriangle m;
  put4(m.v0, t.v0); put4(m.v1, t.v1); put4(m.v2, t.v2);
  put3(m.color, t.color); put4(m.normal, t.normal);
  m.material = material_code(t.material.type);
  return m;
}

template <class SphereT> inline MiptSphere to_mipt_sphere(const SphereT& s) {
  MiptSphere m;
  m.radius = s.radius;
  put4(m.center, s.center); put3(m.color, s.color);
  m.material = material_code(s.material.type);
  return m;
}

template <class CameraT> inline MiptCamera to_mipt_camera(const CameraT& c) {
  MiptCamera m;
  m.focal = c.focalLength;
  put4(m.pos, c.cameraPos);
  for (int col = 0; col < 4; ++col)
    for (int row = 0; row < 4; ++row) {
      m.rot_x[4 * col + row] = c.cameraRotationX[col][row];
      m.rot_y[4 * col + row] = c.cameraRotationY[col][row];
    }
  return m;
}

template <class LightT> inline MiptLight to_mipt_light(const LightT& l) {
  MiptLight m;
  put4(m.pos, l.lightPos); put3(m.color, l.lightColor);
  return m;
}

// Owns the device copy of an (immutable after load) scene.
class GpuScene {
 public:
  template <class TriangleT, class SphereT>
  GpuScene(const std::vector<TriangleT>& triangles, const std::vector<SphereT>& spheres, int device = 0) {
    std::vector<MiptTriangle> t;
    t.reserve(triangles.size());
    for (const auto& x : triangles) t.push_back(to_mipt(x));
    std::vector<MiptSphere> s;
    s.reserve(spheres.size());
    for (const auto& x : spheres) s.push_back(to_mipt_sphere(x));
    n_tris_ = t.size();
    check(mipt_scene_create(t.data(), t.size(), s.data(), s.size(), device, &h_));
  }
  ~GpuScene() { mipt_scene_destroy(h_); }
  GpuScene(const GpuScene&) = delete;
  GpuScene& operator=(const GpuScene&) = delete;
  MiptScene* handle() const { return h_; }
  size_t n_tris() const { return n_tris_; }
  MiptParams params;  // filled by Draw() from the screen size on first use; edit to taste
  bool params_set = false;

 private:
  MiptScene* h_ = nullptr;
  size_t n_tris_ = 0;
};

// closestIntersection (intersections.cpp:6-55) for one ray. `triangles` / `spheres` are the
// host vectors the scene was built from; colour, normal and material of the hit primitive are
// copied from them exactly as the reference does (:31-33, :47-49).
template <class RayT, class TriangleT, class SphereT, class IntersectionT>
inline bool closestIntersection(GpuScene& gpu, const RayT& ray, const std::vector<TriangleT>& triangles,
                                const std::vector<SphereT>& spheres, IntersectionT& out) {
  float o[4], d[4], pos[4] = {0, 0, 0, 0}, t = 0.f;
  int32_t idx = -1;
  put4(o, ray.origin); put4(d, ray.direction);
  check(mipt_closest_hit(gpu.handle(), o, d, 1, &idx, &t, pos));
  if (idx < 0) return false;
  out.position.x = pos[0]; out.position.y = pos[1]; out.position.z = pos[2]; out.position.w = pos[3];
  out.distance = t;
  if ((size_t)idx < triangles.size()) {
    out.color = triangles[idx].color;
    out.normal = triangles[idx].normal;
    out.material = triangles[idx].material;
  } else {
    const SphereT& s = spheres[idx - triangles.size()];
    out.color = s.color;
    out.material = s.material;
    out.normal = s.normal(out.position);
  }
  return true;
}

// existsIntersection (intersections.cpp:57-85): inter.distance is the maximum t.
template <class RayT, class TriangleT, class IntersectionT>
inline bool existsIntersection(GpuScene& gpu, const RayT& ray, const std::vector<TriangleT>&, const IntersectionT& inter) {
  float o[4], d[4], max_t = inter.distance;
  uint8_t occ = 0;
  put4(o, ray.origin); put4(d, ray.direction);
  check(mipt_any_hit(gpu.handle(), o, d, &max_t, 1, &occ));
  return occ != 0;
}

// Draw (main.cpp:110-146): `samples_per_call` samples of every pixel are added to `pixels`
// (W separately allocated columns of H vec3, main.cpp:55-57), samplesSeenSoFar advances by the
// same amount, and screen->buffer receives the tonemapped ARGB image.
template <class ScreenT, class CameraT, class LightT, class Vec3T>
inline void Draw(GpuScene& gpu, ScreenT* screen, const CameraT& cam, const LightT& light, Vec3T** pixels,
                 int& samplesSeenSoFar, int samples_per_call = 1) {
  const int W = screen->width, H = screen->height;
  if (!gpu.params_set) { mipt_params_default(&gpu.params, W, H); gpu.params_set = true; }
  gpu.params.width = W; gpu.params.height = H;
  std::vector<float> accum((size_t)W * H * 3);
  for (int x = 0; x < W; ++x)
    for (int y = 0; y < H; ++y) {
      float* a = &accum[((size_t)x * H + y) * 3];
      a[0] = pixels[x][y].x; a[1] = pixels[x][y].y; a[2] = pixels[x][y].z;
    }
  const MiptCamera c = to_mipt_camera(cam);
  const MiptLight l = to_mipt_light(light);
  check(mipt_render_accumulate(gpu.handle(), &c, &l, &gpu.params, (uint32_t)samplesSeenSoFar,
                               (uint32_t)samples_per_call, accum.data()));
  samplesSeenSoFar += samples_per_call;
  for (int x = 0; x < W; ++x)
    for (int y = 0; y < H; ++y) {
      const float* a = &accum[((size_t)x * H + y) * 3];
      pixels[x][y].x = a[0]; pixels[x][y].y = a[1]; pixels[x][y].z = a[2];
    }
  check(mipt_resolve_argb(gpu.handle(), accum.data(), W, H, (uint32_t)samplesSeenSoFar, screen->buffer));
}

// The scene on several GPUs of the node, driven from this one process: what the OpenMP loop over
// image columns (main.cpp:121) becomes. Draw() below splits every call's samples over the devices.
// By default the image is handed from device to device, each adding its samples in the order one GPU
// adds them (mipt_multi_render_accumulate_chained): the pixels are those of a one-GPU run bit for bit.
// sum_reduce = true takes one RCCL reduce of the per-device images instead (equal up to rounding).
class MultiGpuScene {
 public:
  template <class TriangleT, class SphereT>
  // allow_shared_device: a device may be listed more than once (rehearsal on a one-GPU machine; relay only)
  MultiGpuScene(const std::vector<TriangleT>& triangles, const std::vector<SphereT>& spheres,
                const std::vector<int>& devices, bool allow_shared_device = false) {
    std::vector<MiptTriangle> t;
    std::vector<MiptSphere> s;
    t.reserve(triangles.size());
    for (const auto& x : triangles) t.push_back(to_mipt(x));
    for (const auto& x : spheres) s.push_back(to_mipt_sphere(x));
    check(mipt_multi_create_opts(t.data(), t.size(), s.data(), s.size(), devices.data(), (int)devices.size(),
                                 allow_shared_device ? (uint32_t)MIPT_MULTI_ALLOW_SHARED_DEVICE : 0u, &h_));
  }
  ~MultiGpuScene() { mipt_multi_destroy(h_); }
  MultiGpuScene(const MultiGpuScene&) = delete;
  MultiGpuScene& operator=(const MultiGpuScene&) = delete;
  MiptMulti* handle() const { return h_; }
  MiptParams params;
  bool params_set = false;
  bool sum_reduce = false;

 private:
  MiptMulti* h_ = nullptr;
};

// Draw (main.cpp:110-146) on all the scene's GPUs
template <class ScreenT, class CameraT, class LightT, class Vec3T>
inline void Draw(MultiGpuScene& gpu, ScreenT* screen, const CameraT& cam, const LightT& light, Vec3T** pixels,
                 int& samplesSeenSoFar, int samples_per_call = 64) {
  const int W = screen->width, H = screen->height;
  if (!gpu.params_set) { mipt_params_default(&gpu.params, W, H); gpu.params_set = true; }
  gpu.params.width = W; gpu.params.height = H;
  std::vector<float> accum((size_t)W * H * 3);
  for (int x = 0; x < W; ++x)
    for (int y = 0; y < H; ++y) {
      float* a = &accum[((size_t)x * H + y) * 3];
      a[0] = pixels[x][y].x; a[1] = pixels[x][y].y; a[2] = pixels[x][y].z;
    }
  const MiptCamera c = to_mipt_camera(cam);
  const MiptLight l = to_mipt_light(light);
  check((gpu.sum_reduce ? mipt_multi_render_accumulate : mipt_multi_render_accumulate_chained)(
      gpu.handle(), &c, &l, &gpu.params, (uint32_t)samplesSeenSoFar, (uint32_t)samples_per_call, accum.data()));
  samplesSeenSoFar += samples_per_call;
  for (int x = 0; x < W; ++x)
    for (int y = 0; y < H; ++y) {
      const float* a = &accum[((size_t)x * H + y) * 3];
      pixels[x][y].x = a[0]; pixels[x][y].y = a[1]; pixels[x][y].z = a[2];
    }
  check(mipt_resolve_argb(mipt_multi_scene(gpu.handle(), 0), accum.data(), W, H, (uint32_t)samplesSeenSoFar,
                          screen->buffer));
}

// The interactive loop of main.cpp:98-103 with the image kept on the device: `pixels` and
// `samplesSeenSoFar` become one Frame; only the ARGB screen crosses PCIe per frame.
//     mipt_host::Frame frame(*gpu, SCREEN_WIDTH, SCREEN_HEIGHT);
//     while (NoQuitMessageSDL()) {
//       if (handleKeyboard(...changed camera or light...)) frame.reset();   // reset_evolutionModel
//       frame.add_samples(cam, light, 16);                                   // Draw's pixel loop
//       frame.present(screen);                                               // Draw's tonemap loop
//       SDL_Renderframe(screen);
//     }
class Frame {
 public:
  Frame(GpuScene& gpu, int width, int height) : gpu_(gpu) {
    check(mipt_frame_create(gpu.handle(), width, height, &f_));
    if (!gpu.params_set) { mipt_params_default(&gpu.params, width, height); gpu.params_set = true; }
  }
  ~Frame() { mipt_frame_destroy(f_); }
  Frame(const Frame&) = delete;
  Frame& operator=(const Frame&) = delete;
  void reset() { check(mipt_frame_reset(f_)); }
  template <class CameraT, class LightT>
  void add_samples(const CameraT& cam, const LightT& light, int n) {
    const MiptCamera c = to_mipt_camera(cam);
    const MiptLight l = to_mipt_light(light);
    check(mipt_frame_add_samples(f_, &c, &l, &gpu_.params, (uint32_t)n));
  }
  template <class ScreenT> void present(ScreenT* screen) { check(mipt_frame_present(f_, screen->buffer)); }
  int samplesSeenSoFar() const { return mipt_frame_samples(f_); }
  bool done() {  // non-blocking: have all samples added so far been rendered?
    const int rc = mipt_frame_poll(f_);
    if (rc < 0) check(rc);
    return rc == 1;
  }

 private:
  GpuScene& gpu_;
  MiptFrame* f_ = nullptr;
};

// The same loop on all the GPUs of a MultiGpuScene: the image never leaves the devices, adding samples
// returns as soon as the work is queued, present() brings back the ARGB screen alone.
//     mipt_host::MultiFrame frame(*all_gpus, SCREEN_WIDTH, SCREEN_HEIGHT);      // exact = true: the one-GPU bits
//     while (NoQuitMessageSDL()) { ...; frame.add_samples(cam, light, 64); frame.present(screen); ... }
class MultiFrame {
 public:
  MultiFrame(MultiGpuScene& gpu, int width, int height, bool exact = true) : gpu_(gpu) {
    check(mipt_multi_frame_create(gpu.handle(), width, height, exact ? MIPT_MULTI_FRAME_EXACT : MIPT_MULTI_FRAME_REDUCE, &f_));
    if (!gpu.params_set) { mipt_params_default(&gpu.params, width, height); gpu.params_set = true; }
    width_ = width; height_ = height;
  }
  ~MultiFrame() { mipt_multi_frame_destroy(f_); }
  MultiFrame(const MultiFrame&) = delete;
  MultiFrame& operator=(const MultiFrame&) = delete;
  void reset() { check(mipt_multi_frame_reset(f_)); }
  template <class CameraT, class LightT>
  void add_samples(const CameraT& cam, const LightT& light, int n) {
    const MiptCamera c = to_mipt_camera(cam);
    const MiptLight l = to_mipt_light(light);
    check(mipt_multi_frame_add_samples(f_, &c, &l, &gpu_.params, (uint32_t)n));
  }
  template <class ScreenT> void present(ScreenT* screen) { check(mipt_multi_frame_present(f_, screen->buffer)); }
  // the accumulated sums into the reference's column-wise `pixels` (main.cpp:55-57), e.g. before a save
  template <class Vec3T> void read_pixels(Vec3T** pixels) {
    std::vector<float> accum((size_t)width_ * height_ * 3);
    check(mipt_multi_frame_read_pixels(f_, accum.data()));
    for (int x = 0; x < width_; ++x)
      for (int y = 0; y < height_; ++y) {
        const float* a = &accum[((size_t)x * height_ + y) * 3];
        pixels[x][y].x = a[0]; pixels[x][y].y = a[1]; pixels[x][y].z = a[2];
      }
  }
  int samplesSeenSoFar() const { return mipt_multi_frame_samples(f_); }
  bool done() {
    const int rc = mipt_multi_frame_poll(f_);
    if (rc < 0) check(rc);
    return rc == 1;
  }

 private:
  MultiGpuScene& gpu_;
  MiptMultiFrame* f_ = nullptr;
  int width_ = 0, height_ = 0;
};

// SDL_SaveImage (SDLauxiliary.h:25-54) for the reference's `screen` struct (buffer, width, height)
template <class ScreenT>
inline void SDL_SaveImage(ScreenT* s, const char* filename) {
  check(mipt_save_bmp(filename, s->buffer, s->width, s->height));
}

// LoadObject (parser.cpp:8-46): appends the OBJ file's triangles, built with the reference's own
// five-argument Triangle constructor (TestModelH.h:20-24).
template <class TriangleT>
inline void LoadObject(const std::string& path, std::vector<TriangleT>& triangles) {
  const int n = mipt_load_obj(path.c_str(), nullptr, 0);
  if (n < 0) check(n);
  std::vector<MiptTriangle> m((size_t)n);
  if (n) check(mipt_load_obj(path.c_str(), m.data(), m.size()) == n ? MIPT_OK : MIPT_E_INVALID);
  typedef decltype(TriangleT::v0) V4;
  typedef decltype(TriangleT::color) V3;
  for (const MiptTriangle& t : m)
    triangles.push_back(TriangleT(V4(t.v0[0], t.v0[1], t.v0[2], t.v0[3]), V4(t.v1[0], t.v1[1], t.v1[2], t.v1[3]),
                                  V4(t.v2[0], t.v2[1], t.v2[2], t.v2[3]), V3(t.color[0], t.color[1], t.color[2]),
                                  V4(t.normal[0], t.normal[1], t.normal[2], t.normal[3])));
}

// normaliseTriangles (utils.cpp:52-140), same argument list; `type` must be a string that outlives
// the triangles (the reference stores the pointer, material.h:5-11).
template <class TriangleT>
inline void normaliseTriangles(std::vector<TriangleT>& raw, float scale = 1, float displacementX = 0,
                               float displacementY = 0, float displacementZ = 0, float rotateX = 0,
                               float rotateY = 0, float rotateZ = 0, float adjustNormalX = 1,
                               float adjustNormalY = 1, float adjustNormalZ = 1, const char* type = "diffuse") {
  std::vector<MiptTriangle> m;
  m.reserve(raw.size());
  for (const auto& t : raw) m.push_back(to_mipt(t));
  const float d[3] = {displacementX, displacementY, displacementZ}, r[3] = {rotateX, rotateY, rotateZ};
  const float a[3] = {adjustNormalX, adjustNormalY, adjustNormalZ};
  // the library's material codes cover the four strings the path dispatches on; any other string
  // behaves like "none" on the path, and the triangle keeps the caller's pointer either way
  check(mipt_normalise_triangles(m.data(), m.size(), scale, d, r, a, material_code(type)));
  for (size_t i = 0; i < raw.size(); ++i) {
    auto& t = raw[i];
    const MiptTriangle& s = m[i];
    t.v0.x = s.v0[0]; t.v0.y = s.v0[1]; t.v0.z = s.v0[2]; t.v0.w = s.v0[3];
    t.v1.x = s.v1[0]; t.v1.y = s.v1[1]; t.v1.z = s.v1[2]; t.v1.w = s.v1[3];
    t.v2.x = s.v2[0]; t.v2.y = s.v2[1]; t.v2.z = s.v2[2]; t.v2.w = s.v2[3];
    t.normal.x = s.normal[0]; t.normal.y = s.normal[1]; t.normal.z = s.normal[2]; t.normal.w = s.normal[3];
    t.material.type = type;
  }
}

}  // namespace mipt_host
