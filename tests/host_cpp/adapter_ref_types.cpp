// adapter_ref_types.cpp -- compiles host/mipt_adapter.hpp against the reference's OWN types:
// `Triangle` / `Material` from CW/raytracer/Source/TestModelH.h and glm vectors from the vendored
// GLM (both read in place from the reference tree; only built where that tree exists). Camera,
// Light and Sphere here are local structs with the member names of definitions.h:16-37 and
// sphere.h:9-16 over real glm types (definitions.h itself needs SDL and cannot be included).
// Writes the converted scenes as 80-byte records to stdout; the test compares them with
// tests/golden/scene_*.bin.
#include <cstdio>
#include <vector>

#include <glm/glm.hpp>
#include "TestModelH.h"

#include "mipt_adapter.hpp"

struct Camera { float focalLength; glm::vec4 cameraPos; glm::mat4 cameraRotationX, cameraRotationY; float yaw, pitch; };
struct Light { glm::vec4 lightPos; glm::vec3 lightColor; };
struct Sphere { float radius; glm::vec4 center; glm::vec3 color; Material material; };

int main(int argc, char** argv) {
  if (argc == 2) {  // glassWolf through the adapter's LoadObject / normaliseTriangles (main.cpp:78-89)
    std::vector<Triangle> scene, wolf;
    LoadTestModelNone(scene);
    mipt_host::LoadObject(argv[1], wolf);
    mipt_host::normaliseTriangles(wolf, 1.0, -0.2, -1, -0.5, 3.15, -2.5, 0, -1, 1, 1, "glass");
    scene.insert(scene.end(), wolf.begin(), wolf.end());
    for (const Triangle& t : scene) {
      MiptTriangle m = mipt_host::to_mipt(t);
      fwrite(&m, sizeof m, 1, stdout);
    }
    return 0;
  }
  void (*loaders[4])(std::vector<Triangle>&) = {LoadTestModel, LoadTestModelColorBleeding, LoadTestModelGlassCube,
                                               LoadTestModelNone};
  for (auto load : loaders) {
    std::vector<Triangle> tris;
    load(tris);
    for (const Triangle& t : tris) {
      MiptTriangle m = mipt_host::to_mipt(t);
      fwrite(&m, sizeof m, 1, stdout);
    }
  }
  // camera / light / sphere conversions, appended as raw structs
  Camera cam;
  cam.focalLength = 500;
  cam.cameraPos = glm::vec4(0, 0, -3, 1);
  cam.cameraRotationX = glm::mat4(1, 0, 0, 0, 0, 0.5f, 0.25f, 0, 0, -0.25f, 0.5f, 0, 0, 0, 0, 1);
  cam.cameraRotationY = glm::mat4(1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16);
  MiptCamera mc = mipt_host::to_mipt_camera(cam);
  fwrite(&mc, sizeof mc, 1, stdout);
  Light light{glm::vec4(0.7, 0.95, -0.7, 1.0), 14.0f * glm::vec3(1, 1, 1)};
  MiptLight ml = mipt_host::to_mipt_light(light);
  fwrite(&ml, sizeof ml, 1, stdout);
  Sphere s{0.3f, glm::vec4(-0.45, 0.7, -0.55, 1), glm::vec3(1, 0, 0), Material("glass")};
  MiptSphere ms = mipt_host::to_mipt_sphere(s);
  fwrite(&ms, sizeof ms, 1, stdout);
  return 0;
}
