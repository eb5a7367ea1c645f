// scene_builders.cpp -- host-side scene model helpers of libmipt.so.
//
// The four built-in Cornell boxes of the reference (CW/raytracer/Source/TestModelH.h:55-580),
// its default camera (camera.cpp:4-13), lights (light.cpp:6-9, main.cpp:68-69) and constants
// (definitions.h:43-55), produced with the same float operations in the same order so that
// the vertex, normal and colour bits equal what LoadTestModel* hands to the renderer.
// Built with -ffp-contract=off; intermediate products go through `volatile` so that no
// compiler setting can fuse or widen them.
#include <cmath>
#include <cstring>

#include "../../include/mipt.h"

namespace {

struct P3 { float x, y, z; };

// corner tables in the reference's units (box side 555), named A..H as in TestModelH.h
constexpr float kL = 555.0f;
const P3 kRoom[8] = {{kL, 0, 0}, {0, 0, 0}, {kL, 0, kL}, {0, 0, kL}, {kL, kL, 0}, {0, kL, 0}, {kL, kL, kL}, {0, kL, kL}};
const float kShort[4][2] = {{290, 114}, {130, 65}, {240, 272}, {82, 225}};    // height 165
const float kTall[4][2] = {{423, 247}, {265, 296}, {472, 406}, {314, 456}};   // height 330
const float kMirror[4][2] = {{323, 347}, {165, 396}, {472, 406}, {314, 456}}; // height 330

enum Corner { A, B, C, D, E, F, G, H };
// room faces: floor, left wall, right wall, ceiling, back wall
const Corner kRoomFaces[10][3] = {{C, B, A}, {C, D, B}, {A, E, C}, {C, E, G}, {F, B, D},
                                  {H, F, D}, {E, F, G}, {F, H, G}, {G, D, C}, {G, H, D}};
// block faces: front, front, back, left, top
const Corner kBlockFaces[10][3] = {{E, B, A}, {E, F, B}, {F, D, B}, {F, H, D}, {H, C, D},
                                   {H, G, C}, {G, E, C}, {E, A, C}, {G, F, E}, {G, H, F}};

struct Rgb { float r, g, b; };
const Rgb kRed = {0.75f, 0.15f, 0.15f}, kYellow = {0.75f, 0.75f, 0.15f}, kPurple = {0.75f, 0.15f, 0.75f},
          kWhite = {0.75f, 0.75f, 0.75f}, kBlack = {0.f, 0.f, 0.f};

void corners_of_block(const float base[4][2], float height, P3 out[8]) {
  for (int i = 0; i < 4; ++i) {
    out[i] = {base[i][0], 0.f, base[i][1]};
    out[i + 4] = {base[i][0], height, base[i][1]};
  }
}

void emit(MiptTriangle* out, size_t& n, const P3* pts, const Corner f[3], Rgb col, uint32_t mat) {
  MiptTriangle& t = out[n++];
  std::memset(&t, 0, sizeof t);
  const P3 &a = pts[f[0]], &b = pts[f[1]], &c = pts[f[2]];
  t.v0[0] = a.x; t.v0[1] = a.y; t.v0[2] = a.z; t.v0[3] = 1.f;
  t.v1[0] = b.x; t.v1[1] = b.y; t.v1[2] = b.z; t.v1[3] = 1.f;
  t.v2[0] = c.x; t.v2[1] = c.y; t.v2[2] = c.z; t.v2[3] = 1.f;
  t.color[0] = col.r; t.color[1] = col.g; t.color[2] = col.b;
  t.material = mat;
}

// TestModelH.h:177-200 for one vertex: v *= 2/L; v -= 1; x,y *= -1; w = 1
void to_unit_box(float v[4]) {
  const float s = 2.0f / kL;
  for (int k = 0; k < 4; ++k) { volatile float m = v[k] * s; volatile float d = m - 1.0f; v[k] = d; }
  volatile float nx = v[0] * -1.0f, ny = v[1] * -1.0f;
  v[0] = nx; v[1] = ny; v[3] = 1.0f;
}

// Triangle::ComputeNormal (TestModelH.h:39-48): normalize(cross(e2, e1)), w = 1
void flat_normal(MiptTriangle& t) {
  volatile float e1[3], e2[3];
  for (int k = 0; k < 3; ++k) { e1[k] = t.v1[k] - t.v0[k]; e2[k] = t.v2[k] - t.v0[k]; }
  volatile float p0 = e2[1] * e1[2], q0 = e1[1] * e2[2];
  volatile float p1 = e2[2] * e1[0], q1 = e1[2] * e2[0];
  volatile float p2 = e2[0] * e1[1], q2 = e1[0] * e2[1];
  volatile float c0 = p0 - q0, c1 = p1 - q1, c2 = p2 - q2;
  volatile float s0 = c0 * c0, s1 = c1 * c1, s2 = c2 * c2;
  volatile float s01 = s0 + s1;
  volatile float len2 = s01 + s2;
  volatile float inv = 1.0f / std::sqrt((float)len2);
  t.normal[0] = c0 * inv; t.normal[1] = c1 * inv; t.normal[2] = c2 * inv; t.normal[3] = 1.0f;
}

}  // namespace

extern "C" int mipt_load_test_model(int which, MiptTriangle* out, size_t capacity) {
  const bool cornell = (which == 0 || which == 1);  // LoadTestModel / LoadTestModelColorBleeding
  const bool glass_cube = (which == 2), bare = (which == 3);
  if (!out || !(cornell || glass_cube || bare)) return MIPT_E_INVALID;
  const size_t need = cornell ? 30 : (glass_cube ? 22 : 10);
  if (capacity < need) return MIPT_E_INVALID;
  size_t n = 0;
  for (int f = 0; f < 10; ++f) {
    Rgb col = kRed;  // the glass scenes paint the whole room red
    if (cornell) col = (f == 2 || f == 3) ? kPurple : (f == 4 || f == 5) ? kYellow : kWhite;
    emit(out, n, kRoom, kRoomFaces[f], col, MIPT_MAT_DIFFUSE);
  }
  P3 blk[8];
  if (!bare) {
    corners_of_block(kShort, 165.f, blk);
    for (int f = 0; f < 10; ++f) emit(out, n, blk, kBlockFaces[f], kWhite, cornell ? MIPT_MAT_DIFFUSE : MIPT_MAT_GLASS);
  }
  if (cornell) {
    corners_of_block(kTall, 330.f, blk);
    for (int f = 0; f < 10; ++f) emit(out, n, blk, kBlockFaces[f], kWhite, MIPT_MAT_DIFFUSE);
  } else if (glass_cube) {
    corners_of_block(kMirror, 330.f, blk);  // only the front face of the moved tall block, as a mirror
    for (int f = 0; f < 2; ++f) emit(out, n, blk, kBlockFaces[f], kBlack, MIPT_MAT_MIRROR);
  }
  for (size_t i = 0; i < n; ++i) {
    to_unit_box(out[i].v0); to_unit_box(out[i].v1); to_unit_box(out[i].v2);
    flat_normal(out[i]);
  }
  return (int)n;
}

extern "C" void mipt_default_camera(MiptCamera* cam, int width) {
  std::memset(cam, 0, sizeof *cam);
  cam->focal = (float)width;  // camera.cpp:5, focalLength = SCREEN_WIDTH
  cam->pos[2] = -3.f; cam->pos[3] = 1.f;
  for (int i = 0; i < 4; ++i) cam->rot_x[5 * i] = cam->rot_y[5 * i] = 1.f;
}

extern "C" void mipt_default_light(MiptLight* light, int glass_scene) {
  if (glass_scene) {  // main.cpp:68-69 (double literals narrowed to float by the vec4 constructor)
    light->pos[0] = (float)0.7; light->pos[1] = (float)0.95; light->pos[2] = (float)-0.7;
  } else {            // light.cpp:7
    light->pos[0] = 0.f; light->pos[1] = (float)-0.5; light->pos[2] = (float)-0.7;
  }
  light->pos[3] = 1.0f;
  light->color[0] = light->color[1] = light->color[2] = 14.0f * 1.f;
}

extern "C" void mipt_params_default(MiptParams* p, int width, int height) {
  std::memset(p, 0, sizeof *p);
  p->width = width; p->height = height;
  p->max_depth = 10;
  p->rr_prob = 0.15f;
  p->diffuse_weight = 0.5f;
  p->glass_weight = 1.15f;
  p->ior_glass = 1.512f;
  p->ior_air = 1.000f;
  p->rng_mode = MIPT_RNG_MINSTD_PATH;
  p->seed = 0;
  p->fog_sigma_t = 0.f;
  p->fog_albedo = 0.8f;
  p->flags = 0;
}
