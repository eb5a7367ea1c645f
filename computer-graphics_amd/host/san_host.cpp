// The library's host-only entry points under AddressSanitizer + UndefinedBehaviorSanitizer
// (`make -C computer-graphics_amd/host sanitize`): the scene builders, the OBJ loader on the reference's wolf and on malformed
// files, normaliseTriangles, the .bmp writer. No GPU code is linked; the library's error hook is
// defined here.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mipt.h"

static std::string g_last;
extern "C" int mipt_internal_fail(int code, const char* msg) {
  g_last = msg ? msg : "";
  return code;
}

static int failures = 0;
#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } \
  } while (0)

static std::string write_file(const std::string& dir, const char* name, const std::string& text) {
  const std::string p = dir + "/" + name;
  FILE* f = std::fopen(p.c_str(), "wb");
  if (f) { std::fwrite(text.data(), 1, text.size(), f); std::fclose(f); }
  return p;
}

int main(int argc, char** argv) {
  const std::string wolf = argc > 1 ? argv[1] : "../../tests/golden/obj/wolf.obj";
  char tmpl[] = "/tmp/mipt_san_XXXXXX";
  const char* made = mkdtemp(tmpl);
  CHECK(made != nullptr);
  const std::string dir = made ? made : "/tmp";

  // scene builders: every model, exact and short capacity
  const int want[4] = {30, 30, 22, 10};
  for (int which = 0; which < 4; ++which) {
    std::vector<MiptTriangle> t(30);
    CHECK(mipt_load_test_model(which, t.data(), t.size()) == want[which]);
    std::vector<MiptTriangle> tight((size_t)want[which]);
    CHECK(mipt_load_test_model(which, tight.data(), tight.size()) == want[which]);
    std::vector<MiptTriangle> small(3);
    CHECK(mipt_load_test_model(which, small.data(), small.size()) < 0);
  }
  CHECK(mipt_load_test_model(17, nullptr, 0) < 0);
  MiptCamera cam; MiptLight light;
  mipt_default_camera(&cam, 1024);
  mipt_default_light(&light, 1);
  mipt_default_light(&light, 0);

  // the reference's mesh: size, load, load into a short buffer, place it as main.cpp does
  const int n = mipt_load_obj(wolf.c_str(), nullptr, 0);
  CHECK(n == 1784);
  if (n > 0) {
    std::vector<MiptTriangle> t((size_t)n);
    CHECK(mipt_load_obj(wolf.c_str(), t.data(), t.size()) == n);
    std::vector<MiptTriangle> part(100);
    CHECK(mipt_load_obj(wolf.c_str(), part.data(), part.size()) == n);
    const float disp[3] = {-0.2f, -1.f, -0.5f}, rot[3] = {3.15f, -2.5f, 0.f}, adj[3] = {-1.f, 1.f, 1.f};
    CHECK(mipt_normalise_triangles(t.data(), t.size(), 1.f, disp, rot, adj, MIPT_MAT_GLASS) == MIPT_OK);
    CHECK(mipt_normalise_triangles(t.data(), 0, 1.f, disp, rot, adj, MIPT_MAT_GLASS) == MIPT_OK);
    CHECK(mipt_normalise_triangles(nullptr, 5, 1.f, disp, rot, adj, MIPT_MAT_GLASS) < 0);
  }

  // files the loader must survive: whatever it returns, no out-of-bounds access and no overflow
  const struct { const char* name; std::string text; } odd[] = {
      {"empty.obj", ""},
      {"no_faces.obj", "v 0 0 0\nv 1 0 0\nv 0 1 0\n"},
      {"plain.obj", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n"},
      {"negative.obj", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf -3 -2 -1\n"},
      {"zero_index.obj", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 0 1 2\n"},
      {"beyond.obj", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 9\nf 4000000000 2 3\nf -9 2 3\n"},
      {"normals.obj", "v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nvt 0 0\nf 1/1/1 2/1/1 3/1/1\nf 1//1 2//1 3//1\nf 1/1 2/1 3/1\nf 1//7 2//-7 3//0\n"},
      {"polygon.obj", "v 0 0 0\nv 2 0 0\nv 2 2 0\nv 1 0.5 0\nv 0 2 0\nv 1 1 1\nf 1 2 3 4 5\nf 1 2 3 4 5 6\nf 1 1 1 1\nf 1 2\nf 1\nf\n"},
      {"collinear.obj", "v 0 0 0\nv 1 0 0\nv 2 0 0\nv 3 0 0\nv 4 0 0\nf 1 2 3 4 5\n"},
      {"garbage.obj", "v a b c\nv 1e999 -1e999 nan\nv\nf x y z\nf 1/ /2 //\nf 1/2/3/4 5/6/7/8 9\n#\n\n\r\n  \t \nvn\nvt\nusemtl\no\ng\ns\n"},
      {"long_line.obj", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf" + std::string(100000, ' ') + "1 2 3" + std::string(5000, '/') + "\n"},
      {"no_newline.obj", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3"},
      {"binary.obj", std::string("v 0 0 0\n") + std::string(3, '\0') + "\xff\xfe f 1 2 3\nv" + std::string(1, '\0') + " 1 2 3\n"},
  };
  for (const auto& o : odd) {
    const std::string p = write_file(dir, o.name, o.text);
    const int cnt = mipt_load_obj(p.c_str(), nullptr, 0);
    if (cnt > 0) {
      std::vector<MiptTriangle> t((size_t)cnt);
      CHECK(mipt_load_obj(p.c_str(), t.data(), t.size()) == cnt);
      std::vector<MiptTriangle> one(1);
      CHECK(mipt_load_obj(p.c_str(), one.data(), one.size()) == cnt);
    }
    std::remove(p.c_str());
  }
  CHECK(mipt_load_obj(nullptr, nullptr, 0) < 0);
  CHECK(mipt_load_obj((dir + "/missing.obj").c_str(), nullptr, 0) < 0);
  CHECK(mipt_load_obj((dir + "/name.txt").c_str(), nullptr, 0) < 0);
  CHECK(mipt_load_obj("", nullptr, 0) < 0);

  // the image writer
  {
    std::vector<uint32_t> px(7 * 5);
    for (size_t i = 0; i < px.size(); ++i) px[i] = 0xFF000000u | (uint32_t)(i * 2654435761u);
    const std::string p = dir + "/a.bmp";
    CHECK(mipt_save_bmp(p.c_str(), px.data(), 7, 5) == MIPT_OK);
    CHECK(mipt_save_bmp(p.c_str(), px.data(), 1, 1) == MIPT_OK);
    CHECK(mipt_save_bmp(p.c_str(), px.data(), 0, 5) < 0);
    CHECK(mipt_save_bmp(p.c_str(), px.data(), -7, 5) < 0);
    CHECK(mipt_save_bmp(p.c_str(), nullptr, 7, 5) < 0);
    CHECK(mipt_save_bmp(nullptr, px.data(), 7, 5) < 0);
    CHECK(mipt_save_bmp((dir + "/no/such/dir/a.bmp").c_str(), px.data(), 7, 5) < 0);
    std::remove(p.c_str());
  }
  if (made) std::remove(dir.c_str());
  if (failures) { std::printf("%d check(s) failed (last error: %s)\n", failures, g_last.c_str()); return 1; }
  std::printf("host sanitizer run ok\n");
  return 0;
}
