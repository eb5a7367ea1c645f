// ref_obj_dump.cpp -- golden vectors for OBJ ingestion. TEST INFRASTRUCTURE.
//
// Compiled against the reference tree where it lies (../Makefile, target _ref/ref_obj_dump).
// Included from the reference, unmodified: CW/raytracer/Source/OBJ_Loader.h (the objl loader the
// reference's LoadObject drives), TestModelH.h (Triangle, LoadTestModelNone), the vendored GLM.
// parser.cpp and utils.cpp themselves include <SDL.h> through definitions.h and cannot be built
// in this image, so the two functions that sit on top of objl and GLM are restated here against
// the reference's own types: LoadObject (parser.cpp:8-46) and normaliseTriangles
// (utils.cpp:52-140). Everything they call -- objl::Loader, glm::mat4 products, Triangle's
// ComputeNormal -- is the reference's code.
//
// usage: ref_obj_dump <file.obj> <raw.bin> [<placed.bin> scale dx dy dz rx ry rz ax ay az material [room]]
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <limits>
#include <string>
#include <vector>

#include <glm/glm.hpp>
#include "TestModelH.h"
#include "OBJ_Loader.h"

using namespace std;
using glm::vec3;
using glm::vec4;
using glm::mat4;

static void load_object(const string& path, vector<Triangle>& triangles) {  // parser.cpp:8-46
  objl::Loader loader;
  if (!loader.LoadFile(path)) return;
  for (size_t i = 0; i < loader.LoadedMeshes.size(); i++) {
    objl::Mesh mesh = loader.LoadedMeshes[i];
    for (size_t j = 0; j < mesh.Indices.size(); j += 3) {
      auto a = mesh.Vertices[mesh.Indices[j]];
      auto b = mesh.Vertices[mesh.Indices[j + 1]];
      auto c = mesh.Vertices[mesh.Indices[j + 2]];
      vec4 p1(a.Position.X, a.Position.Y, a.Position.Z, 1);
      vec4 p2(b.Position.X, b.Position.Y, b.Position.Z, 1);
      vec4 p3(c.Position.X, c.Position.Y, c.Position.Z, 1);
      vec4 normal(a.Normal.X, a.Normal.Y, b.Normal.Z, 1);  // parser.cpp:38: x, y of V1, z of V2
      triangles.push_back(Triangle(p1, p2, p3, vec3(1, 1, 1), normal));
    }
  }
}

static void normalise_triangles(vector<Triangle>& raw, float scale, float dx, float dy, float dz,
                                float rx, float ry, float rz, float ax, float ay, float az,
                                const char* type) {  // utils.cpp:52-140
  float L = -numeric_limits<int>::max();
  for (auto t : raw) {
    const vec4* vs[3] = {&t.v0, &t.v1, &t.v2};
    for (int k = 0; k < 3; k++) {
      if (glm::abs(vs[k]->x) > L) L = glm::abs(vs[k]->x);
      if (glm::abs(vs[k]->y) > L) L = glm::abs(vs[k]->y);
      if (glm::abs(vs[k]->z) > L) L = glm::abs(vs[k]->z);
    }
  }
  vec4 v1(1, 0, 0, 0), v2(0, cos(rx), sin(rx), 0), v3(0, -sin(rx), cos(rx), 0), v4(0, 0, 0, 1);
  mat4 Rx(v1, v2, v3, v4);
  vec4 v5(cos(ry), 0, -sin(ry), 0), v6(0, 1, 0, 0), v7(sin(ry), 0, cos(ry), 0), v8(0, 0, 0, 1);
  mat4 Ry(v5, v6, v7, v8);
  vec4 v9(cos(rz), sin(rz), 0, 0), v10(-sin(rz), cos(rz), 0, 0), v11(0, 0, 1, 0), v12(0, 0, 0, 1);
  mat4 Rz(v9, v10, v11, v12);
  mat4 R = Rx * Ry * Rz;
  for (auto& t : raw) {
    t.v0 /= scale * L; t.v1 /= scale * L; t.v2 /= scale * L;
    t.v0 += vec4(dx, dy, dz, 0); t.v1 += vec4(dx, dy, dz, 0); t.v2 += vec4(dx, dy, dz, 0);
    t.v0 = R * t.v0; t.v1 = R * t.v1; t.v2 = R * t.v2;
    if (!t.normal.x) t.ComputeNormal();
    t.normal.x = ax * t.normal.x;
    t.normal.y = ay * t.normal.y;
    t.normal.z = az * t.normal.z;
    t.normal = R * t.normal;
    t.material.type = type;
  }
}

static uint32_t material_code(const char* type) {
  if (!strcmp(type, "diffuse")) return 0;
  if (!strcmp(type, "mirror")) return 1;
  if (!strcmp(type, "glass")) return 2;
  if (!strcmp(type, "specular")) return 3;
  return 4;
}

static void dump(const char* path, const vector<Triangle>& tris) {
  FILE* f = fopen(path, "wb");
  if (!f) { perror(path); exit(1); }
  for (const Triangle& t : tris) {
    fwrite(&t.v0[0], 1, 16, f); fwrite(&t.v1[0], 1, 16, f); fwrite(&t.v2[0], 1, 16, f);
    fwrite(&t.color[0], 1, 12, f); fwrite(&t.normal[0], 1, 16, f);
    uint32_t m = material_code(t.material.type);
    fwrite(&m, 1, 4, f);
  }
  fclose(f);
  printf("%s: %zu triangles\n", path, tris.size());
}

int main(int argc, char** argv) {
  if (argc != 3 && argc != 15 && argc != 16) { fprintf(stderr, "usage: see the file header\n"); return 2; }
  vector<Triangle> tris;
  load_object(argv[1], tris);
  dump(argv[2], tris);
  if (argc >= 15) {
    float p[10];
    for (int k = 0; k < 10; k++) p[k] = strtof(argv[4 + k], nullptr);
    static string material = argv[14];
    normalise_triangles(tris, p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], material.c_str());
    vector<Triangle> scene;
    if (argc == 16) LoadTestModelNone(scene);  // main.cpp:78-89: room first, then the mesh
    scene.insert(scene.end(), tris.begin(), tris.end());
    dump(argv[3], scene);
  }
  return 0;
}
