// obj_loader.cpp -- OBJ ingestion and placement of libmipt.so (host code).
//
// mipt_load_obj            = LoadObject (CW/raytracer/Source/parser.cpp:8-46) over the objl loader
//                            it drives (OBJ_Loader.h:430-702 LoadFile, :716-823 vertices of a face,
//                            :827-992 triangulation);
// mipt_normalise_triangles = normaliseTriangles (utils.cpp:52-140).
// Same float operations in the same order, so the triangles equal the reference's bit for bit:
//   * numbers go through strtof like std::stof; indices are 1-based, negative ones count from
//     the end (OBJ_Loader.h:397-405);
//   * a face without normals gets ONE unnormalised cross product (P0-P1) x (P2-P1) on all its
//     vertices (:811-822); LoadObject then builds each triangle's normal from the x, y of its
//     first vertex and the z of its SECOND (parser.cpp:38), w = 1, colour (1,1,1), "diffuse";
//   * triangles leave in file order (objl's meshes partition the faces in order);
//   * polygons are cut the way objl cuts them, including what it does with repeated positions;
//   * placement divides all FOUR components by scale * L (glm's vec4 /= scalar), so w = 1 / (scale L)
//     afterwards; a normal is recomputed only when its x is exactly 0 (utils.cpp:129-131).
// Built with -ffp-contract=off; products that feed sums go through `volatile`.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <string>
#include <new>
#include <vector>

#include "../../include/mipt.h"

extern "C" int mipt_internal_fail(int code, const char* msg);  // sets mipt_last_error()

namespace {

struct V3 { float x, y, z; };
inline bool same(const V3& a, const V3& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline V3 sub(const V3& a, const V3& b) {
  volatile float x = a.x - b.x, y = a.y - b.y, z = a.z - b.z;
  return {x, y, z};
}
inline V3 cross(const V3& a, const V3& b) {  // math::CrossV3, OBJ_Loader.h:224-229
  volatile float p0 = a.y * b.z, q0 = a.z * b.y, p1 = a.z * b.x, q1 = a.x * b.z, p2 = a.x * b.y, q2 = a.y * b.x;
  volatile float x = p0 - q0, y = p1 - q1, z = p2 - q2;
  return {x, y, z};
}
inline float dot(const V3& a, const V3& b) {  // math::DotV3, :238-241
  volatile float p0 = a.x * b.x, p1 = a.y * b.y, p2 = a.z * b.z;
  volatile float s = p0 + p1;
  return s + p2;
}
inline float magnitude(const V3& v) {  // math::MagnitudeV3, :232-235
  volatile float a = powf(v.x, 2), b = powf(v.y, 2), c = powf(v.z, 2);
  volatile float s = a + b;
  volatile float t = s + c;
  return sqrtf(t);
}
bool same_side(const V3& p1, const V3& p2, const V3& a, const V3& b) {  // algorithm::SameSide, :272-281
  const V3 ab = sub(b, a);
  return dot(cross(ab, sub(p1, a)), cross(ab, sub(p2, a))) >= 0;
}
bool in_triangle(const V3& point, const V3& t1, const V3& t2, const V3& t3) {  // algorithm::inTriangle, :295-317
  if (!(same_side(point, t1, t2, t3) && same_side(point, t2, t1, t3) && same_side(point, t3, t1, t2))) return false;
  const V3 n = cross(sub(t2, t1), sub(t3, t1));
  const float len = magnitude(n);
  volatile float bx = n.x / len, by = n.y / len, bz = n.z / len;  // math::ProjV3, :252-256
  const V3 bn = {bx, by, bz};
  const float d = dot(point, bn);
  volatile float px = bn.x * d, py = bn.y * d, pz = bn.z * d;
  return magnitude({px, py, pz}) == 0;
}

struct FaceVertex { V3 pos, normal; };

// algorithm::tail / firstToken (:359-393) on the separators they use
std::string first_token(const std::string& in) {
  const size_t b = in.find_first_not_of(" \t");
  if (b == std::string::npos) return "";
  const size_t e = in.find_first_of(" \t", b);
  return e == std::string::npos ? in.substr(b) : in.substr(b, e - b);
}
std::string tail(const std::string& in) {
  const size_t token_start = in.find_first_not_of(" \t");
  const size_t space_start = in.find_first_of(" \t", token_start);
  const size_t tail_start = in.find_first_not_of(" \t", space_start);
  const size_t tail_end = in.find_last_not_of(" \t");
  if (tail_start != std::string::npos && tail_end != std::string::npos) return in.substr(tail_start, tail_end - tail_start + 1);
  if (tail_start != std::string::npos) return in.substr(tail_start);
  return "";
}
// algorithm::split (:320-356) for a one-character separator: empty fields between adjacent
// separators are kept, a trailing separator adds nothing
void split(const std::string& in, char sep, std::vector<std::string>& out) {
  out.clear();
  std::string cur;
  for (size_t i = 0; i < in.size(); ++i) {
    if (in[i] == sep) {
      out.push_back(cur);  // pushes "" when cur is empty, like objl
      cur.clear();
    } else if (i + 1 >= in.size()) {
      cur += in[i];
      out.push_back(cur);
      break;
    } else {
      cur += in[i];
    }
  }
}

bool parse_float(const std::string& s, float& v) {
  const char* b = s.c_str();
  char* e = nullptr;
  v = strtof(b, &e);
  return e != b;
}
bool parse_index(const std::string& s, size_t count, size_t& idx) {  // algorithm::getElement, :397-405
  const char* b = s.c_str();
  char* e = nullptr;
  const long v = strtol(b, &e, 10);
  if (e == b) return false;
  const long i = v < 0 ? (long)count + v : v - 1;
  if (i < 0 || (size_t)i >= count) return false;
  idx = (size_t)i;
  return true;
}

// Loader::VertexTriangluation (:827-992): indices into `iv`, three per triangle
void triangulate(const std::vector<FaceVertex>& iv, std::vector<unsigned>& out) {
  if (iv.size() < 3) return;
  if (iv.size() == 3) { out.insert(out.end(), {0u, 1u, 2u}); return; }
  std::vector<FaceVertex> tv = iv;
  auto push_matches = [&](const V3& a, const V3& b, const V3& c, size_t limit) {
    for (size_t j = 0; j < limit; ++j) {
      if (same(iv[j].pos, a)) out.push_back((unsigned)j);
      if (same(iv[j].pos, b)) out.push_back((unsigned)j);
      if (same(iv[j].pos, c)) out.push_back((unsigned)j);
    }
  };
  for (;;) {
    const size_t tv_before = tv.size(), out_before = out.size();
    for (int i = 0; i < (int)tv.size(); ++i) {
      const FaceVertex prev = tv[i == 0 ? tv.size() - 1 : (size_t)i - 1];
      const FaceVertex cur = tv[(size_t)i];
      const FaceVertex next = tv[(size_t)i == tv.size() - 1 ? 0 : (size_t)i + 1];
      if (tv.size() == 3) {
        push_matches(cur.pos, prev.pos, next.pos, tv.size());  // :884 loops to tVerts.size()
        tv.clear();
        break;
      }
      if (tv.size() == 4) {
        push_matches(cur.pos, prev.pos, next.pos, iv.size());
        V3 other = {0.f, 0.f, 0.f};  // objl leaves it default-constructed when no vertex qualifies
        for (size_t j = 0; j < tv.size(); ++j)
          if (!same(tv[j].pos, cur.pos) && !same(tv[j].pos, prev.pos) && !same(tv[j].pos, next.pos)) { other = tv[j].pos; break; }
        push_matches(prev.pos, next.pos, other, iv.size());
        tv.clear();
        break;
      }
      // :938-940 compares an angle with `<= 0 && >= 180`, which never holds: no effect
      bool in_tri = false;
      for (size_t j = 0; j < iv.size(); ++j)
        if (in_triangle(iv[j].pos, prev.pos, cur.pos, next.pos) && !same(iv[j].pos, prev.pos) &&
            !same(iv[j].pos, cur.pos) && !same(iv[j].pos, next.pos)) { in_tri = true; break; }
      if (in_tri) continue;
      push_matches(cur.pos, prev.pos, next.pos, iv.size());
      for (size_t j = 0; j < tv.size(); ++j)
        if (same(tv[j].pos, cur.pos)) { tv.erase(tv.begin() + (long)j); break; }
      i = -1;
    }
    if (out.empty()) break;
    if (tv.empty()) break;
    // a pass that clipped nothing would repeat forever (objl itself never returns from such a
    // polygon): stop with the triangles found so far
    if (tv.size() == tv_before && out.size() == out_before) break;
  }
}

void mat4_mul(const float a[16], const float b[16], float out[16]) {  // glm type_mat4x4.inl:722-727
  for (int j = 0; j < 4; ++j)
    for (int k = 0; k < 4; ++k) {
      volatile float s0 = a[0 * 4 + k] * b[j * 4 + 0], s1 = a[1 * 4 + k] * b[j * 4 + 1];
      volatile float s2 = a[2 * 4 + k] * b[j * 4 + 2], s3 = a[3 * 4 + k] * b[j * 4 + 3];
      volatile float t0 = s0 + s1;
      volatile float t1 = t0 + s2;
      out[j * 4 + k] = t1 + s3;
    }
}
void mat4_mul_vec4(const float m[16], float v[4]) {  // type_mat4x4.inl:641-652: (m0 v0 + m1 v1) + (m2 v2 + m3 v3)
  float r[4];
  for (int k = 0; k < 4; ++k) {
    volatile float a = m[k] * v[0], b = m[4 + k] * v[1], c = m[8 + k] * v[2], d = m[12 + k] * v[3];
    volatile float ab = a + b, cd = c + d;
    r[k] = ab + cd;
  }
  memcpy(v, r, sizeof r);
}

void flat_normal(MiptTriangle& t) {  // Triangle::ComputeNormal, TestModelH.h:39-48
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

static int load_obj_impl(const char* path, MiptTriangle* out, size_t capacity) {
  if (!path) return mipt_internal_fail(MIPT_E_INVALID, "mipt_load_obj: path is null");
  const std::string p = path;
  if (p.size() < 4 || p.substr(p.size() - 4) != ".obj")  // OBJ_Loader.h:433
    return mipt_internal_fail(MIPT_E_INVALID, "mipt_load_obj: not an .obj file");
  std::ifstream file(p);
  if (!file.is_open()) return mipt_internal_fail(MIPT_E_INVALID, "mipt_load_obj: cannot open the file");
  std::vector<V3> positions, normals;
  size_t n_tex = 0, n_out = 0, line_no = 0;
  std::string line;
  std::vector<std::string> fields, parts;
  std::vector<FaceVertex> fv;
  std::vector<unsigned> idx;
  char msg[160];
  while (std::getline(file, line)) {
    ++line_no;
    const std::string tok = first_token(line);
    if (tok == "v" || tok == "vn") {
      split(tail(line), ' ', fields);
      V3 v;
      if (fields.size() < 3 || !parse_float(fields[0], v.x) || !parse_float(fields[1], v.y) || !parse_float(fields[2], v.z)) {
        snprintf(msg, sizeof msg, "mipt_load_obj: line %zu: malformed '%s' record", line_no, tok.c_str());
        return mipt_internal_fail(MIPT_E_INVALID, msg);
      }
      (tok == "v" ? positions : normals).push_back(v);
    } else if (tok == "vt") {
      ++n_tex;  // texture coordinates never reach a Triangle; only their count validates indices
    } else if (tok == "f") {
      split(tail(line), ' ', fields);
      fv.clear();
      bool no_normal = false, bad = false;
      FaceVertex cur = {{0, 0, 0}, {0, 0, 0}};  // objl reuses one Vertex across the face (:723)
      for (const std::string& f : fields) {
        split(f, '/', parts);
        size_t ip = 0, in = 0, it = 0;
        if (parts.size() == 1 || parts.size() == 2) {
          if (!parse_index(parts[0], positions.size(), ip) || (parts.size() == 2 && !parse_index(parts[1], n_tex, it))) { bad = true; break; }
          cur.pos = positions[ip];
          no_normal = true;
          fv.push_back(cur);
        } else if (parts.size() == 3) {
          if (!parse_index(parts[0], positions.size(), ip) || !parse_index(parts[2], normals.size(), in) ||
              (parts[1] != "" && !parse_index(parts[1], n_tex, it))) { bad = true; break; }
          cur.pos = positions[ip];
          cur.normal = normals[in];
          fv.push_back(cur);
        }  // any other field shape adds no vertex (:801-804)
      }
      if (bad || (no_normal && fv.size() < 3)) {
        snprintf(msg, sizeof msg, "mipt_load_obj: line %zu: malformed face", line_no);
        return mipt_internal_fail(MIPT_E_INVALID, msg);
      }
      if (no_normal) {  // :811-822
        const V3 n = cross(sub(fv[0].pos, fv[1].pos), sub(fv[2].pos, fv[1].pos));
        for (FaceVertex& v : fv) v.normal = n;
      }
      idx.clear();
      triangulate(fv, idx);
      for (size_t j = 0; j + 2 < idx.size(); j += 3) {  // parser.cpp:28-43
        const FaceVertex &a = fv[idx[j]], &b = fv[idx[j + 1]], &c = fv[idx[j + 2]];
        if (out && n_out < capacity) {
          MiptTriangle& t = out[n_out];
          memset(&t, 0, sizeof t);
          t.v0[0] = a.pos.x; t.v0[1] = a.pos.y; t.v0[2] = a.pos.z; t.v0[3] = 1.f;
          t.v1[0] = b.pos.x; t.v1[1] = b.pos.y; t.v1[2] = b.pos.z; t.v1[3] = 1.f;
          t.v2[0] = c.pos.x; t.v2[1] = c.pos.y; t.v2[2] = c.pos.z; t.v2[3] = 1.f;
          t.color[0] = t.color[1] = t.color[2] = 1.f;
          t.normal[0] = a.normal.x; t.normal[1] = a.normal.y; t.normal[2] = b.normal.z; t.normal[3] = 1.f;
          t.material = MIPT_MAT_DIFFUSE;
        }
        ++n_out;
      }
    }
  }
  if (n_out > (size_t)std::numeric_limits<int>::max()) return mipt_internal_fail(MIPT_E_INVALID, "mipt_load_obj: too many triangles");
  return (int)n_out;
}

// nothing may throw across the C ABI: allocation failures become MIPT_E_NOMEM
extern "C" int mipt_load_obj(const char* path, MiptTriangle* out, size_t capacity) {
  try {
    return load_obj_impl(path, out, capacity);
  } catch (const std::bad_alloc&) {
    return mipt_internal_fail(MIPT_E_NOMEM, "mipt_load_obj: host allocation failed");
  } catch (...) {
    return mipt_internal_fail(MIPT_E_INVALID, "mipt_load_obj: unexpected exception");
  }
}

extern "C" int mipt_normalise_triangles(MiptTriangle* tris, size_t n, float scale, const float displacement[3],
                                        const float rotate[3], const float adjust_normal[3], uint32_t material) {
  if ((n && !tris) || !displacement || !rotate || !adjust_normal)
    return mipt_internal_fail(MIPT_E_INVALID, "mipt_normalise_triangles: null argument");
  if (material > MIPT_MAT_NONE) return mipt_internal_fail(MIPT_E_INVALID, "mipt_normalise_triangles: bad material code");
  float L = -(float)std::numeric_limits<int>::max();  // utils.cpp:57
  for (size_t i = 0; i < n; ++i) {
    const float* vs[3] = {tris[i].v0, tris[i].v1, tris[i].v2};
    for (int k = 0; k < 3; ++k)
      for (int c = 0; c < 3; ++c)
        if (std::fabs(vs[k][c]) > L) L = std::fabs(vs[k][c]);
  }
  const float cx = std::cos(rotate[0]), sx = std::sin(rotate[0]);  // float overloads: cosf / sinf
  const float cy = std::cos(rotate[1]), sy = std::sin(rotate[1]);
  const float cz = std::cos(rotate[2]), sz = std::sin(rotate[2]);
  const float Rx[16] = {1, 0, 0, 0, 0, cx, sx, 0, 0, -sx, cx, 0, 0, 0, 0, 1};  // column-major like glm::mat4
  const float Ry[16] = {cy, 0, -sy, 0, 0, 1, 0, 0, sy, 0, cy, 0, 0, 0, 0, 1};
  const float Rz[16] = {cz, sz, 0, 0, -sz, cz, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  float RxRy[16], R[16];
  mat4_mul(Rx, Ry, RxRy);
  mat4_mul(RxRy, Rz, R);
  volatile float div = scale * L;
  const float disp[4] = {displacement[0], displacement[1], displacement[2], 0.f};
  for (size_t i = 0; i < n; ++i) {
    MiptTriangle& t = tris[i];
    float* vs[3] = {t.v0, t.v1, t.v2};
    for (int k = 0; k < 3; ++k) {
      for (int c = 0; c < 4; ++c) { volatile float q = vs[k][c] / div; volatile float s = q + disp[c]; vs[k][c] = s; }
      mat4_mul_vec4(R, vs[k]);
    }
    if (!t.normal[0]) flat_normal(t);
    volatile float nx = adjust_normal[0] * t.normal[0], ny = adjust_normal[1] * t.normal[1], nz = adjust_normal[2] * t.normal[2];
    t.normal[0] = nx; t.normal[1] = ny; t.normal[2] = nz;
    mat4_mul_vec4(R, t.normal);
    t.material = material;
  }
  return MIPT_OK;
}
