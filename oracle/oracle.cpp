// oracle.cpp — fp64 CPU restatement of the reference path tracer's hot path.
//
// TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED at the reference
// boundary: the reference holds no golden vectors for this path and cannot be
// built here, so every function below follows the reference source text, cited as
// file:line under ray/src/.
//
// Arithmetic follows the reference expression by expression in IEEE double
// (compile with -ffp-contract=off).  glm semantics restated: column-major
// matrices, dot = (x*x' + y*y') + z*z', normalize(v) = v * (1/sqrt(dot(v,v))),
// reflect(I,N) = I - N*dot(N,I)*2, mix(x,y,a) = x*(1-a) + y*a, inverse by cofactors.
//
// Deliberate, documented departures (the reference is undefined there):
//  * a hit with t >= t_max, or a mesh hit with local t >= t_max, is a miss; the
//    reference would dereference a null material (bvh.h:172, ray.cpp:6-9);
//  * a NaN sample is stored as 0 (glm::clamp passes NaN through, then
//    (int)(255*NaN) is undefined, RayTracer.cpp:54,97);
//  * random numbers come from a counter-based generator (the reference's rand()
//    is shared by all threads and its area-light engine is seeded from
//    std::random_device, RayTracer.cpp:228, light.h:121);
//  * recursion stops at p->max_depth (the reference is unbounded);
//  * the shadow walk gives up after 4096 steps.
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_err;

// Mutation switch, 0 in every use but one: tests/test_reference_screenshots.py turns single
// quirks of the reference OFF (or common mistakes ON) to measure which of them the
// screenshot pin can tell apart from the faithful restatement.  Each value names one
// departure from the reference; the cited line is the one the mutant gets wrong.
enum Mutation {
  MUT_NONE = 0,
  MUT_NO_RR_COMPENSATION = 1,   // drops the division by 1 - q            (RayTracer.cpp:277)
  MUT_NO_MIRROR_HALVING = 2,    // (C + L2) instead of (C + L2) / 2       (RayTracer.cpp:273-274)
  MUT_SHADOWED_SPECULAR = 3,    // direct specular scaled by the light's visibility (material.cpp:132-143 has no shadow term)
  MUT_FIXED_SAMPLE_POINT = 4,   // corner + u*uI + v*vI, the intended formula (light.cpp:109 adds vVec whole and vI to all components)
  MUT_COLOURED_SPECULAR = 5,    // direct specular times the light colour (material.cpp:142 omits it)
  MUT_SPHERE_NORMAL_FACES_RAY = 6,  // sphere normal flipped toward the ray from inside (Sphere.cpp:33-38 never flips)
  MUT_COSINE_PDF = 7,           // divides the bounce radiance by 1/pi instead of 1/(2 pi) (RayTracer.cpp:261-263)
  MUT_CLAMP_AFTER_MEAN = 8,     // no per-sample clamp                    (RayTracer.cpp:54)
  MUT_ABS_N_DOT_V = 9,          // |n.V| in the specular denominators     (material.cpp:141,160 use the signed value)
  MUT_SQUARE_NORMAL_FIXED = 10, // square normal always +z                (Square.cpp:33-37 flips it toward the ray)
  MUT_NO_RR_AT_DEPTH_0 = 11,    // camera-ray hits always survive         (RayTracer.cpp:228-231 applies q at every depth)
  MUT_ROUGHNESS_AS_ALPHA = 12,  // alpha = roughness instead of its square (material.cpp:103)
  MUT_NO_INDIRECT_SPECULAR = 13,// drops the indirect specular term       (material.cpp:153-163)
  MUT_DIFFUSE_IGNORES_METALLIC = 14, // no (1 - metallic) on the diffuse terms (material.cpp:129,149)
  MUT_FLOAT_FREE_HEMISPHERE = 15,    // hemisphere x, z kept in double    (RayTracer.cpp:248-249 round them to float)
  MUT_COUNT
};
int g_mut = MUT_NONE;

const double RAY_EPSILON = 0.00000001;  // scene/ray.h:147

// ---------------------------------------------------------------- vectors ----
struct V3 {
  double x, y, z;
  V3() : x(0), y(0), z(0) {}
  V3(double a, double b, double c) : x(a), y(b), z(c) {}
  explicit V3(double a) : x(a), y(a), z(a) {}
  double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
  double &at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, V3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator*(V3 a, double s) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(double s, V3 a) { return V3(s * a.x, s * a.y, s * a.z); }
inline V3 operator/(V3 a, double s) { return V3(a.x / s, a.y / s, a.z / s); }
inline V3 operator/(V3 a, V3 b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline V3 &operator+=(V3 &a, V3 b) { a = a + b; return a; }
inline V3 &operator*=(V3 &a, V3 b) { a = a * b; return a; }
inline V3 &operator*=(V3 &a, double s) { a = a * s; return a; }
inline V3 &operator/=(V3 &a, double s) { a = a / s; return a; }
inline double dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
  return V3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
inline double length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(V3 a) { return a * (1.0 / std::sqrt(dot(a, a))); }
inline double distance(V3 a, V3 b) { return length(b - a); }
inline V3 vmin(V3 a, V3 b) { return V3(std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)); }
inline V3 vmax(V3 a, V3 b) { return V3(std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)); }
inline V3 vpow(V3 a, double e) { return V3(std::pow(a.x, e), std::pow(a.y, e), std::pow(a.z, e)); }

// --------------------------------------------------------------- matrices ----
// Column-major 4x4, m[4*c + r] (glm::dmat4 memory order).
struct M4 {
  double m[16];
  double operator()(int r, int c) const { return m[4 * c + r]; }
};
struct M3 {
  double m[9];  // m[3*c + r]
  double operator()(int r, int c) const { return m[3 * c + r]; }
};

// operator*(dmat4, dvec3) of scene/scene.h:59-63: w = 1, fourth row dropped.
inline V3 xform_point(const M4 &M, V3 v) {
  return V3(M(0, 0) * v.x + M(0, 1) * v.y + M(0, 2) * v.z + M(0, 3),
            M(1, 0) * v.x + M(1, 1) * v.y + M(1, 2) * v.z + M(1, 3),
            M(2, 0) * v.x + M(2, 1) * v.y + M(2, 2) * v.z + M(2, 3));
}
inline V3 mul(const M3 &M, V3 v) {
  return V3(M(0, 0) * v.x + M(0, 1) * v.y + M(0, 2) * v.z, M(1, 0) * v.x + M(1, 1) * v.y + M(1, 2) * v.z,
            M(2, 0) * v.x + M(2, 1) * v.y + M(2, 2) * v.z);
}

// General 4x4 inverse by cofactors (what glm::inverse computes).
M4 inverse(const M4 &A) {
  const double *a = A.m;
  double inv[16];
  inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
  inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
  inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
  inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
  inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
  inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
  inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
  inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
  inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
  inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
  inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
  inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
  inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
  inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
  inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
  inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
  double det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
  double ood = 1.0 / det;
  M4 R;
  for (int i = 0; i < 16; i++) R.m[i] = inv[i] * ood;
  return R;
}

// transpose(inverse(mat3(M))) — scene/scene.h:76.
M3 normal_matrix(const M4 &A) {
  double a = A(0, 0), b = A(0, 1), c = A(0, 2);
  double d = A(1, 0), e = A(1, 1), f = A(1, 2);
  double g = A(2, 0), h = A(2, 1), i = A(2, 2);
  double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
  double ood = 1.0 / det;
  // inverse(r,c), then transposed: N(r,c) = inv(c,r)
  double inv[3][3];
  inv[0][0] = (e * i - f * h) * ood; inv[0][1] = (c * h - b * i) * ood; inv[0][2] = (b * f - c * e) * ood;
  inv[1][0] = (f * g - d * i) * ood; inv[1][1] = (a * i - c * g) * ood; inv[1][2] = (c * d - a * f) * ood;
  inv[2][0] = (d * h - e * g) * ood; inv[2][1] = (b * g - a * h) * ood; inv[2][2] = (a * e - b * d) * ood;
  M3 N;
  for (int r = 0; r < 3; r++)
    for (int col = 0; col < 3; col++) N.m[3 * col + r] = inv[col][r];
  return N;
}

// ------------------------------------------------------------------- rng ----
inline uint64_t mix64(uint64_t key, uint64_t j) {
  uint64_t z = key + (j + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
inline double u01(uint64_t z) { return (double)(z >> 11) * (1.0 / 9007199254740992.0); }
inline uint64_t sample_key(uint64_t seed, uint64_t pixel, uint64_t sample) {
  return mix64(mix64(mix64(seed, 0x5EEDull), pixel), sample);
}
// Stream slots of one path node.
enum { SLOT_RR = 0, SLOT_COS = 1, SLOT_PHI = 2, SLOT_MIRROR = 3, SLOT_CHILD_BOUNCE = 4, SLOT_CHILD_MIRROR = 5, SLOT_AREA = 16 };

inline double radical_inverse2(uint32_t v) {
  v = (v << 16) | (v >> 16);
  v = ((v & 0x00ff00ffu) << 8) | ((v & 0xff00ff00u) >> 8);
  v = ((v & 0x0f0f0f0fu) << 4) | ((v & 0xf0f0f0f0u) >> 4);
  v = ((v & 0x33333333u) << 2) | ((v & 0xccccccccu) >> 2);
  v = ((v & 0x55555555u) << 1) | ((v & 0xaaaaaaaau) >> 1);
  return (double)v * (1.0 / 4294967296.0);
}

// Random source of one path node: counter mode reads slot j of the node key,
// sequential mode advances one per-thread state in call order.
struct Rng {
  int mode;
  uint64_t *seq;  // sequential state (per thread)
  double draw(uint64_t key, uint64_t slot) const {
    if (mode == ORC_RNG_COUNTER) return u01(mix64(key, slot));
    *seq += 0x9E3779B97F4A7C15ull;
    uint64_t z = *seq;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return u01(z ^ (z >> 31));
  }
};

// ---------------------------------------------------------------- geometry ----
struct Ray {
  V3 p, d;
  V3 at(double t) const { return p + (t * d); }  // ray.h:47
};

// scene/bbox.{h,cc}
struct BBox {
  bool empty = true;
  V3 mn, mx;
  void merge(const BBox &b) {  // bbox.cc:87-98
    if (b.empty) return;
    for (int a = 0; a < 3; a++) {
      if (empty || b.mn[a] < mn[a]) mn.at(a) = b.mn[a];
      if (empty || b.mx[a] > mx.at(a)) mx.at(a) = b.mx[a];
    }
    empty = false;
  }
  // bbox.cc:27-63 (Kay/Kajiya; the early-outs sit inside the axis loop).
  bool intersect(const Ray &r) const {
    double tMin = -1.0e308, tMax = 1.0e308;
    for (int a = 0; a < 3; a++) {
      double vd = r.d[a];
      if (vd == 0.0) continue;
      double v1 = mn[a] - r.p[a];
      double v2 = mx[a] - r.p[a];
      double t1 = v1 / vd;
      double t2 = v2 / vd;
      if (t1 > t2) std::swap(t1, t2);
      if (t1 > tMin) tMin = t1;
      if (t2 < tMax) tMax = t2;
      if (tMin > tMax) return false;
      if (tMax < RAY_EPSILON) return false;
    }
    return true;
  }
};

struct Material {
  V3 kd, ka, kt;
  double rough_raw, metal_raw, index_raw;
  int kdTex = 0, kaTex = 0;  // 0 = constant, k > 0 = textures[k - 1] (MaterialParameter, material.h:76-120)
  // MaterialParameter::intensityValue, material.cpp:232-241: scalars are stored as
  // (v,v,v) and read back through luminance weights.
  static double intensity(double v) { return (0.299 * v) + (0.587 * v) + (0.114 * v); }
  double roughness() const { return intensity(rough_raw); }
  double metallic() const { return intensity(metal_raw); }
  double index() const { return intensity(index_raw); }
};

struct Isect {  // ray.h:67-145 (fields used on this path)
  double t = 0.0;
  V3 N;
  int obj = -1;   // object index (file order) or -1 == NULL
  int prim = -1;  // reference-order primitive index
  int mat = -1;
  double u = 0.0, v = 0.0;  // uvCoordinates, default-constructed to (0, 0) (ray.h:71)
  bool hasColor = false;    // the hit carries its own material copy with kd replaced (trimesh.cpp:179-189)
  V3 color;
};

struct Counters {
  uint64_t node_visits = 0, prim_tests = 0, path_rays = 0, shadow_rays = 0, shaded = 0, nan_samples = 0, samples = 0;
};

// scene/bvh.h — one template for objects and faces.  `Item` needs bbox().
struct BVHNode {
  BBox bounds;
  int left = -1, right = -1;
  bool leaf = false;
  int first = 0, amt = 0;
};

struct BVH {
  std::vector<BVHNode> nodes;
  std::vector<int> order;    // allNodes[k]->geoIdx
  std::vector<BBox> boxes;   // allNodes[k]->nodeBounds (permuted with order)

  void build(const std::vector<BBox> &itemBoxes) {  // bvh.h:112-131
    int n = (int)itemBoxes.size();
    order.resize(n);
    boxes = itemBoxes;
    for (int i = 0; i < n; i++) order[i] = i;
    nodes.clear();
    nodes.reserve(2 * (size_t)n + 1);
    nodes.emplace_back();
    make(0, 0, n);
  }

  void make(int cur, int begin, int amt) {  // bvh.h:40-110
    int end = begin + amt - 1;
    for (int k = begin; k <= end; k++) nodes[cur].bounds.merge(boxes[k]);
    nodes[cur].first = begin;
    nodes[cur].amt = amt;
    if (amt <= 2) {
      nodes[cur].leaf = true;
      return;
    }
    V3 lengths = nodes[cur].bounds.mx - nodes[cur].bounds.mn;
    int axis = 0;
    if (lengths[1] > lengths[axis]) axis = 1;
    if (lengths[2] > lengths[axis]) axis = 2;
    double half = lengths[axis] / 2 + nodes[cur].bounds.mn[axis];
    int i = begin, j = end;
    while (i <= j) {
      double mid = (boxes[i].mn[axis] + boxes[i].mx[axis]) / 2;
      if (mid >= half) {
        std::swap(boxes[i], boxes[j]);
        std::swap(order[i], order[j]);
        j--;
      } else {
        i++;
      }
    }
    int leftCount = i - begin;
    if (leftCount == 0 || leftCount == amt) {
      nodes[cur].leaf = true;
      return;
    }
    int rightCount = amt - leftCount;
    int l = (int)nodes.size();
    nodes.emplace_back();
    make(l, begin, leftCount);
    int r = (int)nodes.size();
    nodes.emplace_back();
    make(r, i, rightCount);
    nodes[cur].left = l;
    nodes[cur].right = r;
  }
};

struct Mesh {
  std::vector<V3> verts, normals;
  std::vector<double> uvs;  // 2 per vertex, or empty
  std::vector<V3> colors;   // vertColors, or empty
  struct Face {
    int ids[3];
    V3 normal;
    double A00, A01, A10, A11;  // AMat, trimesh.h:152-157 (glm::dmat2 column-major: Acr)
    BBox box;
  };
  std::vector<Face> faces;  // non-degenerate, in addFace order
  BVH tree;                 // BVH<TrimeshFace>
};

struct Object {
  int type, material, mesh;
  M4 xform, inv;
  M3 normi;
  BBox bounds;   // world
  int primBase;  // reference-order index of this object's first primitive
};

struct Light {
  int type;
  V3 color, position, orientation, corner, uVec, vVec, center;
  double uL, vL;
  float a, b, c;
};

struct Tex {
  int w = 0, h = 0;
  std::vector<uint8_t> data;
  V3 pixel(int x, int y) const {  // material.cpp:212-222
    if (x < 0 || x >= w || y < 0 || y >= h || data.size() < (size_t)3 * w * h) return V3();
    int s = (x + y * w) * 3;
    return V3(double(data[s]), double(data[s + 1]), double(data[s + 2])) / 255.0;
  }
  V3 mapped(double cu, double cv) const {  // material.cpp:186-210
    double x = cu * double(w - 1), y = cv * double(h - 1);
    double x1 = double(int(x)), y1 = double(int(y));
    double x2 = x1 + 1, y2 = y1 + 1;
    V3 c11 = pixel(int(x1), int(y1));
    V3 c12 = pixel(int(x1), int(y2));
    V3 c21 = pixel(int(x2), int(y1));
    V3 c22 = pixel(int(x2), int(y2));
    return c11 * (x2 - x) * (y2 - y) + c21 * (x - x1) * (y2 - y) + c12 * (x2 - x) * (y - y1) + c22 * (x - x1) * (y - y1);
  }
};

}  // namespace

struct orc_scene {
  mipt_camera cam;
  V3 ambient;
  std::vector<Material> materials;
  std::vector<Object> objects;
  std::vector<Mesh> meshes;
  std::vector<Light> lights;
  BVH tree;  // BVH<Geometry>
  bool hasCube = false;
  Tex cube[6];
  std::vector<Tex> textures;
  int64_t nPrims = 0;

  // MaterialParameter::value (material.cpp:224-230) for the two colours shadeBRDF reads
  // through a texture: the material as the hit sees it.
  Material materialAt(const Isect &i) const {
    Material m = materials[i.mat];
    if (i.hasColor) m.kd = i.color;  // setDiffuse(interpolatedColor): a constant, whatever kd was
    else if (m.kdTex > 0) m.kd = textures[(size_t)m.kdTex - 1].mapped(i.u, i.v);
    if (m.kaTex > 0) m.ka = textures[(size_t)m.kaTex - 1].mapped(i.u, i.v);
    return m;
  }

  // ---- primitives ----------------------------------------------------------
  // Sphere.cpp:9-40
  static bool sphereLocal(Ray &r, Isect &i) {
    r.d = normalize(r.d);
    V3 v = -r.p;
    double b = dot(v, r.d);
    double disc = b * b - dot(v, v) + 1;
    if (disc < 0.0) return false;
    disc = std::sqrt(disc);
    double t2 = b + disc;
    if (t2 <= RAY_EPSILON) return false;
    double t1 = b - disc;
    if (t1 > RAY_EPSILON) {
      i.t = t1;
      i.N = normalize(r.at(t1));
    } else {
      i.t = t2;
      i.N = normalize(r.at(t2));
    }
    if (g_mut == MUT_SPHERE_NORMAL_FACES_RAY && dot(i.N, r.d) > 0) i.N = -i.N;
    return true;
  }
  // Box.cpp:11-74
  static bool boxLocal(const Ray &r, Isect &i) {
    V3 p = r.p, d = r.d;
    double bestT = 1e100;
    int bestIndex = -1;
    for (int it = 0; it < 6; it++) {
      int mod0 = it % 3;
      if (d[mod0] == 0) continue;
      double t = ((it / 3) - 0.5 - p[mod0]) / d[mod0];
      if (t < RAY_EPSILON || t > bestT) continue;
      int mod1 = (it + 1) % 3, mod2 = (it + 2) % 3;
      double x = p[mod1] + t * d[mod1];
      double y = p[mod2] + t * d[mod2];
      if (x <= 0.5 && x >= -0.5 && y <= 0.5 && y >= -0.5) {
        if (bestT > t) {
          bestT = t;
          bestIndex = it;
        }
      }
    }
    if (bestIndex < 0) return false;
    i.t = bestT;
    V3 ip = r.at(bestT);  // Box.cpp:56-71
    int i1 = (bestIndex + 1) % 3, i2 = (bestIndex + 2) % 3;
    if (bestIndex < 3) {
      i.N = V3(-double(bestIndex == 0), -double(bestIndex == 1), -double(bestIndex == 2));
      i.u = 0.5 - ip[std::min(i1, i2)];
      i.v = 0.5 + ip[std::max(i1, i2)];
    } else {
      i.N = V3(double(bestIndex == 3), double(bestIndex == 4), double(bestIndex == 5));
      i.u = 0.5 + ip[std::min(i1, i2)];
      i.v = 0.5 + ip[std::max(i1, i2)];
    }
    return true;
  }
  // Square.cpp:8-43
  static bool squareLocal(const Ray &r, Isect &i) {
    V3 p = r.p, d = r.d;
    if (d.z == 0.0) return false;
    double t = -p.z / d.z;
    if (t <= RAY_EPSILON) return false;
    V3 P = r.at(t);
    if (P.x < -0.5 || P.x > 0.5) return false;
    if (P.y < -0.5 || P.y > 0.5) return false;
    i.t = t;
    i.N = d.z > 0.0 ? V3(0.0, 0.0, -1.0) : V3(0.0, 0.0, 1.0);
    if (g_mut == MUT_SQUARE_NORMAL_FIXED) i.N = V3(0.0, 0.0, 1.0);
    i.u = P.x + 0.5;  // Square.cpp:41
    i.v = P.y + 0.5;
    return true;
  }
  // trimesh.cpp:98-200
  static bool faceLocal(const Mesh &m, const Mesh::Face &f, const Ray &r, Isect &i) {
    V3 a = m.verts[f.ids[0]], b = m.verts[f.ids[1]], c = m.verts[f.ids[2]];
    V3 normal = f.normal;
    double denom = dot(normal, r.d);
    if (std::fabs(denom) < RAY_EPSILON) return false;
    double num = dot(a - r.p, normal);
    double t = num / denom;
    if (t < 0) return false;
    V3 P = r.at(t);
    double b1 = dot(P - a, b - a);
    double b2 = dot(P - a, c - a);
    // glm: (M * v).x = M[0][0]*v.x + M[1][0]*v.y ; .y = M[0][1]*v.x + M[1][1]*v.y
    double pb0 = f.A00 * b1 + f.A10 * b2;
    double pb1 = f.A01 * b1 + f.A11 * b2;
    double w0 = 1 - pb0 - pb1, w1 = pb0, w2 = pb1;
    if (!(w0 >= 0 && w0 <= 1 && w1 >= 0 && w1 <= 1 && w2 >= 0 && w2 <= 1)) return false;
    if (!m.normals.empty()) {
      V3 nA = m.normals[f.ids[0]] * w0;
      V3 nB = m.normals[f.ids[1]] * w1;
      V3 nC = m.normals[f.ids[2]] * w2;
      i.N = normalize(nA + nB + nC);
    } else {
      i.N = normal;
    }
    if (!m.uvs.empty()) {  // trimesh.cpp:165-172
      double uA = m.uvs[2 * (size_t)f.ids[0]] * w0, vA = m.uvs[2 * (size_t)f.ids[0] + 1] * w0;
      double uB = m.uvs[2 * (size_t)f.ids[1]] * w1, vB = m.uvs[2 * (size_t)f.ids[1] + 1] * w1;
      double uC = m.uvs[2 * (size_t)f.ids[2]] * w2, vC = m.uvs[2 * (size_t)f.ids[2] + 1] * w2;
      i.u = (uA + uB) + uC;
      i.v = (vA + vB) + vC;
    } else if (!m.colors.empty()) {  // trimesh.cpp:179-189
      V3 cA = m.colors[f.ids[0]] * w0;
      V3 cB = m.colors[f.ids[1]] * w1;
      V3 cC = m.colors[f.ids[2]] * w2;
      i.color = (cA + cB) + cC;
      i.hasColor = true;
    }
    i.t = t;
    return true;
  }

  // BVH<TrimeshFace>::checkIntersect, bvh.h:160-185
  bool meshWalk(const Mesh &m, int cur, const Ray &r, Isect &i, int objPrimBase, Counters &c) const {
    const BVHNode &nd = m.tree.nodes[cur];
    c.node_visits++;
    if (!nd.bounds.intersect(r)) return false;
    if (nd.leaf) {
      bool hitAny = false;
      for (int j = 0; j < nd.amt; j++) {
        Isect test;
        c.prim_tests++;
        bool check = faceLocal(m, m.faces[m.tree.order[nd.first + j]], r, test);
        if (check && test.t < i.t) {
          i.t = test.t;
          i.N = test.N;
          i.u = test.u;
          i.v = test.v;
          i.hasColor = test.hasColor;
          i.color = test.color;
          i.prim = objPrimBase + nd.first + j;
        }
        hitAny |= check;
      }
      return hitAny;
    }
    bool res = meshWalk(m, nd.left, r, i, objPrimBase, c);
    res |= meshWalk(m, nd.right, r, i, objPrimBase, c);
    return res;
  }

  // Geometry::intersect, scene.cpp:13-40
  bool objectIntersect(int oi, const Ray &rw, Isect &i, double tmax, Counters &c) const {
    const Object &o = objects[oi];
    if (!o.bounds.intersect(rw)) return false;
    Ray r;
    r.p = xform_point(o.inv, rw.p);
    V3 dir = xform_point(o.inv, rw.p + rw.d) - r.p;
    double len = length(dir);
    r.d = normalize(dir);
    bool hit = false;
    switch (o.type) {
      case MIPT_OBJ_SPHERE: hit = sphereLocal(r, i); break;
      case MIPT_OBJ_BOX: hit = boxLocal(r, i); break;
      case MIPT_OBJ_SQUARE: hit = squareLocal(r, i); break;
      case MIPT_OBJ_TRIMESH: {
        // Trimesh::intersectLocal, trimesh.cpp:70-88
        const Mesh &m = meshes[o.mesh];
        i.t = tmax;
        i.prim = -1;
        if (!m.faces.empty()) meshWalk(m, 0, r, i, o.primBase, c);
        hit = i.prim >= 0;  // a face hit at local t >= t_max is a miss (see file header)
        break;
      }
      default: break;
    }
    if (!hit) return false;
    i.N = normalize(mul(o.normi, i.N));  // scene.h:92-94
    i.t = i.t / len;
    i.obj = oi;
    i.mat = o.material;
    if (o.type != MIPT_OBJ_TRIMESH) i.prim = o.primBase;
    return true;
  }

  // BVH<Geometry>::checkIntersect, bvh.h:160-185
  void sceneWalk(int cur, const Ray &r, Isect &i, double tmax, Counters &c) const {
    const BVHNode &nd = tree.nodes[cur];
    c.node_visits++;
    if (!nd.bounds.intersect(r)) return;
    if (nd.leaf) {
      for (int j = 0; j < nd.amt; j++) {
        Isect test;
        c.prim_tests++;
        bool check = objectIntersect(tree.order[nd.first + j], r, test, tmax, c);
        if (check && test.t < i.t) i = test;
      }
      return;
    }
    sceneWalk(nd.left, r, i, tmax, c);
    sceneWalk(nd.right, r, i, tmax, c);
  }

  // Scene::intersect, scene.cpp:114-138.  Leaves every field of `i` but t untouched
  // on a miss, as the reference does (the shadow walk depends on that).
  bool intersect(const Ray &r, Isect &i, double tmax, Counters &c) const {
    i.t = tmax;
    Isect best = i;
    best.obj = -1;
    if (!objects.empty()) sceneWalk(0, r, best, tmax, c);
    if (best.obj >= 0 && best.t < tmax) {
      i = best;
      return true;
    }
    return false;
  }

  bool intersectBrute(const Ray &r, Isect &i, double tmax) const {
    Counters c;
    i.t = tmax;
    Isect best = i;
    best.obj = -1;
    for (size_t k = 0; k < tree.order.size(); k++) {
      int oi = tree.order[k];
      const Object &o = objects[oi];
      Isect test;
      bool check = false;
      if (o.type == MIPT_OBJ_TRIMESH) {
        if (o.bounds.intersect(r)) {
          const Mesh &m = meshes[o.mesh];
          Ray rl;
          rl.p = xform_point(o.inv, r.p);
          V3 dir = xform_point(o.inv, r.p + r.d) - rl.p;
          double len = length(dir);
          rl.d = normalize(dir);
          test.t = tmax;
          test.prim = -1;
          for (size_t fi = 0; fi < m.faces.size(); fi++) {
            Isect ft;
            if (faceLocal(m, m.faces[m.tree.order[fi]], rl, ft) && ft.t < test.t) {
              test.t = ft.t;
              test.N = ft.N;
              test.u = ft.u;
              test.v = ft.v;
              test.hasColor = ft.hasColor;
              test.color = ft.color;
              test.prim = o.primBase + (int)fi;
            }
          }
          if (test.prim >= 0) {
            test.N = normalize(mul(o.normi, test.N));
            test.t = test.t / len;
            test.obj = oi;
            test.mat = o.material;
            check = true;
          }
        }
      } else {
        check = objectIntersect(oi, r, test, tmax, c);
      }
      if (check && test.t < best.t) best = test;
    }
    if (best.obj >= 0 && best.t < tmax) {
      i = best;
      return true;
    }
    return false;
  }

  // ---- lights ---------------------------------------------------------------
  V3 lightDirection(const Light &L, V3 P) const {
    switch (L.type) {
      case MIPT_LIGHT_DIRECTIONAL: return -L.orientation;          // light.cpp:39-41
      case MIPT_LIGHT_POINT: return normalize(L.position - P);     // light.cpp:55-57
      default: return normalize(L.center - P);                     // light.cpp:96-98
    }
  }
  double distanceAttenuation(const Light &L, V3 P) const {
    if (L.type == MIPT_LIGHT_POINT) {  // light.cpp:47-51
      double d = distance(L.position, P);
      double denom = L.a + L.b * d + L.c * std::pow(d, 2);
      return std::min(1.0, 1 / denom);
    }
    return 1.0;  // light.cpp:10-13, 88-90
  }

  // The occluder walk shared by the three lights (light.cpp:21-33, 70-83, 125-138).
  // `target` == nullptr: directional (limit stays t_max).
  V3 shadowWalk(V3 light, Ray sr, const V3 *target, double lightT, double tmax, Counters &c) const {
    Isect point;
    c.shadow_rays++;
    intersect(sr, point, tmax, c);
    int guard = 0;
    while (point.t < lightT && guard++ < 4096) {
      V3 entry = sr.at(point.t);
      sr.p = sr.at(point.t + RAY_EPSILON);
      intersect(sr, point, tmax, c);
      V3 exitp = sr.at(point.t);
      double dist = distance(entry, exitp);
      V3 kt = point.mat >= 0 ? materials[point.mat].kt : V3();
      light *= vpow(kt, dist);
      sr.p = sr.at(point.t + RAY_EPSILON);
      intersect(sr, point, tmax, c);
      if (target) lightT = std::sqrt(dot(*target - sr.p, *target - sr.p));
    }
    return light;
  }

  // Light::shadowAttenuation.  `r` is the shadow ray built by shadeBRDF
  // (material.cpp:117-120), p = firePos.
  V3 shadowAttenuation(int li, const Ray &r, V3 p, const mipt_params &prm, const Rng &rng, uint64_t key, Counters &c) const {
    const Light &L = lights[li];
    double tmax = prm.t_max;
    if (L.type == MIPT_LIGHT_DIRECTIONAL) {  // light.cpp:15-35
      return shadowWalk(L.color, r, nullptr, tmax, tmax, c);
    }
    if (L.type == MIPT_LIGHT_POINT) {  // light.cpp:63-85
      double lightT = std::sqrt(dot(L.position - p, L.position - p));
      return shadowWalk(L.color, r, &L.position, lightT, tmax, c);
    }
    // RectangleAreaLight, light.cpp:114-147
    V3 finalLight(0, 0, 0);
    int S = prm.area_samples;
    for (int s = 0; s < S; s++) {
      V3 light = areaSample(li, s, r, p, prm, rng, key, c);
      finalLight += light;
    }
    finalLight /= (double)S;  // the reference divides by the literal 10 == its sample count
    return finalLight;
  }

  // One pass of the loop of RectangleAreaLight::shadowAttenuation (light.cpp:118-144): sample s of area light li.
  V3 areaSample(int li, int s, const Ray &r, V3 p, const mipt_params &prm, const Rng &rng, uint64_t key, Counters &c) const {
    const Light &L = lights[li];
    const double tmax = prm.t_max;
    const int S = prm.area_samples;
    {
      double uI, vI;
      if (prm.flags & MIPT_FLAG_FIXED_AREA_PATTERN) {
        uI = L.uL * ((s + 0.5) / S);
        vI = L.vL * radical_inverse2((uint32_t)s);
      } else {
        uint64_t slot = SLOT_AREA + 2 * ((uint64_t)li * (uint64_t)S + (uint64_t)s);
        uI = L.uL * rng.draw(key, slot);
        vI = L.vL * rng.draw(key, slot + 1);
      }
      // samplePoint, light.cpp:105-111: corner + uVec*uI + vVec + vI (vI added to every component)
      V3 position = (L.corner + L.uVec * uI + L.vVec) + V3(vI);
      if (g_mut == MUT_FIXED_SAMPLE_POINT) position = (L.corner + L.uVec * uI) + L.vVec * vI;
      double lightT = std::sqrt(dot(position - p, position - p));
      Ray sr;
      sr.p = r.p;
      sr.d = normalize(position - r.p);
      V3 light = shadowWalk(L.color, sr, &position, lightT, tmax, c);
      double d = distance(position, r.p);
      double denom = L.a + L.b * d + L.c * std::pow(d, 2);
      light *= std::min(1.0, 1 / denom);
      return light;
    }
  }

  // ---- BRDF -----------------------------------------------------------------
  static double ggxGeometry(V3 n, V3 x, double alpha) {  // material.cpp:68-73
    double nDotX = std::fabs(dot(n, x));
    double denom = nDotX + std::sqrt((alpha * alpha) + (1 - (alpha * alpha)) * (nDotX * nDotX));
    return (2 * nDotX) / denom;
  }
  static V3 fresnel(V3 F0, V3 V, V3 H) {  // material.cpp:75-78
    return F0 + (V3(1) - F0) * std::pow((1 - std::fabs(dot(V, H))), 5);
  }
  static double ndf(double alpha, V3 N, V3 H) {  // material.cpp:80-86
    double alphaSquared = alpha * alpha;
    double nDotH = std::fabs(dot(N, H));
    double denom = std::pow((std::pow(nDotH, 2) * (alphaSquared - 1) + 1), 2) * M_PI;
    return alphaSquared / denom;
  }

  // Material::shadeBRDF, material.cpp:88-170.  wInDir = wIn.getDirection(),
  // P/V = wOut position/direction, n = i.getN().
  V3 shadeBRDF(const Material &m, V3 wInDir, V3 P, V3 V, V3 indirect, V3 n, const mipt_params &prm, const Rng &rng, uint64_t key, Counters &c) const {
    V3 diffuseBRDF(0);
    V3 ambientTerm = m.ka * ambient;
    V3 specularTerm(0, 0, 0);
    double roughness = m.roughness();
    if (roughness == 0) roughness = 0.001;
    double alpha = roughness * roughness;
    if (g_mut == MUT_ROUGHNESS_AS_ALPHA) alpha = roughness;
    double idx = m.index();
    V3 F0 = V3(std::pow((1.0 - idx) / (1.0 + idx), 2));
    double metallic = m.metallic();
    if (metallic > 0) F0 = F0 * (1.0 - metallic) + m.kd * metallic;  // glm::mix
    for (size_t li = 0; li < lights.size(); li++) {
      const Light &L = lights[li];
      V3 lightDir = lightDirection(L, P);
      V3 H = normalize(lightDir + V);
      V3 diffuseTerm(0, 0, 0);
      V3 firePos = P + n * RAY_EPSILON * 3.0;
      Ray shadowRay;
      shadowRay.p = firePos;
      shadowRay.d = lightDirection(L, firePos);
      V3 contributionD = shadowAttenuation((int)li, shadowRay, firePos, prm, rng, key, c);
      const V3 shadowAtt = contributionD;
      contributionD *= distanceAttenuation(L, P);
      contributionD *= m.kd;
      contributionD *= std::fabs(dot(n, lightDirection(L, P)));
      diffuseTerm += contributionD / M_PI;
      if (g_mut != MUT_DIFFUSE_IGNORES_METALLIC) diffuseTerm *= (1 - metallic);
      diffuseBRDF += diffuseTerm;
      V3 F = fresnel(F0, V, H);
      double D = ndf(alpha, n, H);
      double G = ggxGeometry(n, lightDirection(L, P), alpha) * ggxGeometry(n, V, alpha);
      double nDotl = std::fabs(dot(n, lightDirection(L, P)));
      V3 spec = ((F * D * G) / (4 * nDotl * dot(n, V)));
      if (g_mut == MUT_ABS_N_DOT_V) spec = ((F * D * G) / (4 * nDotl * std::fabs(dot(n, V))));
      if (g_mut == MUT_COLOURED_SPECULAR) spec = spec * L.color;
      if (g_mut == MUT_SHADOWED_SPECULAR) {  // visibility = attenuation over light colour, channel by channel
        for (int k = 0; k < 3; k++) spec.at(k) *= L.color[k] != 0.0 ? shadowAtt[k] / L.color[k] : 0.0;
      }
      specularTerm += spec * nDotl * distanceAttenuation(L, P);
    }
    V3 w = -wInDir;
    V3 H = normalize(w + V);
    diffuseBRDF += (m.kd * indirect) * std::fabs(dot(n, w)) / M_PI * (g_mut == MUT_DIFFUSE_IGNORES_METALLIC ? 1.0 : (1 - metallic));
    V3 F = fresnel(F0, V, H);
    double D = ndf(alpha, n, H);
    double G = ggxGeometry(n, w, alpha) * ggxGeometry(n, V, alpha);
    double nDotl = std::fabs(dot(n, w));
    V3 indirectSpecular = ((F * G * D) / (4 * nDotl * dot(n, V))) * nDotl * indirect;
    if (g_mut == MUT_ABS_N_DOT_V) indirectSpecular = ((F * G * D) / (4 * nDotl * std::fabs(dot(n, V)))) * nDotl * indirect;
    if (g_mut != MUT_NO_INDIRECT_SPECULAR) specularTerm += indirectSpecular;
    V3 ret = diffuseBRDF;
    ret += ambientTerm;
    ret += specularTerm;
    return ret;
  }

  // ---- environment ------------------------------------------------------------
  V3 cubeColor(V3 dirIn) const {  // cubeMap.cpp:7-64
    V3 dir = normalize(dirIn);
    double absX = std::fabs(dir.x), absY = std::fabs(dir.y), absZ = std::fabs(dir.z);
    double maxAxis = 0, u = 0, v = 0;
    int idx = 0;
    dir.z *= -1;
    bool px = dir.x > 0, py = dir.y > 0, pz = dir.z > 0;
    if (px && absX >= absY && absX >= absZ) { maxAxis = absX; u = -dir.z; v = dir.y; idx = 0; }
    if (!px && absX >= absY && absX >= absZ) { maxAxis = absX; u = dir.z; v = dir.y; idx = 1; }
    if (py && absY >= absX && absY >= absZ) { maxAxis = absY; u = dir.x; v = -dir.z; idx = 2; }
    if (!py && absY >= absX && absY >= absZ) { maxAxis = absY; u = dir.x; v = dir.z; idx = 3; }
    if (pz && absZ >= absX && absZ >= absY) { maxAxis = absZ; u = dir.x; v = dir.y; idx = 4; }
    if (!pz && absZ >= absX && absZ >= absY) { maxAxis = absZ; u = -dir.x; v = dir.y; idx = 5; }
    double cu = 0.5 * (u / maxAxis + 1.0), cv = 0.5 * (v / maxAxis + 1.0);
    return cube[idx].mapped(cu, cv);
  }

  // ---- integrator -------------------------------------------------------------
  // RayTracer::tracePath, RayTracer.cpp:221-306.
  V3 tracePath(const Ray &r, int depth, uint64_t key, const mipt_params &prm, const Rng &rng, Counters &c) const {
    if (prm.max_depth > 0 && depth >= prm.max_depth) return V3(0);
    Isect i;
    c.path_rays++;
    if (intersect(r, i, prm.t_max, c)) {
      const Material m = materialAt(i);
      double russianRoulette = rng.draw(key, SLOT_RR);
      if (russianRoulette < prm.rr_q && !(g_mut == MUT_NO_RR_AT_DEPTH_0 && depth == 0)) return V3(0);
      c.shaded++;
      V3 normal = i.N;
      V3 startPos = r.at(i.t);
      V3 Nt;
      if (std::fabs(normal.x) > std::fabs(normal.y))
        Nt = V3(normal.z, 0, -normal.x) / std::sqrt(normal.x * normal.x + normal.z * normal.z);
      else
        Nt = V3(0, -normal.z, normal.y) / std::sqrt(normal.y * normal.y + normal.z * normal.z);
      V3 Nb = cross(normal, Nt);
      double r1 = rng.draw(key, SLOT_COS);
      double sinTheta = std::sqrt(1 - r1 * r1);
      double phi = rng.draw(key, SLOT_PHI) * 2 * M_PI;
      float x = (float)(sinTheta * std::cos(phi));
      float z = (float)(sinTheta * std::sin(phi));
      V3 rd((double)x, r1, (double)z);
      if (g_mut == MUT_FLOAT_FREE_HEMISPHERE) rd = V3(sinTheta * std::cos(phi), r1, sinTheta * std::sin(phi));
      V3 conv(rd.x * Nb.x + rd.y * normal.x + rd.z * Nt.x, rd.x * Nb.y + rd.y * normal.y + rd.z * Nt.y,
              rd.x * Nb.z + rd.y * normal.z + rd.z * Nt.z);
      conv = normalize(conv);
      Ray randomRay;
      randomRay.p = startPos + conv * RAY_EPSILON;
      randomRay.d = conv;
      double pdf = 1 / (2 * M_PI);
      if (g_mut == MUT_COSINE_PDF) pdf = 1 / M_PI;
      V3 indirect(0);
      indirect += tracePath(randomRay, depth + 1, mix64(key, SLOT_CHILD_BOUNCE), prm, rng, c);
      indirect /= pdf;
      V3 wInDir = -conv;
      V3 V = normalize(-r.d);
      V3 colorC = shadeBRDF(m, wInDir, startPos, V, indirect, normal, prm, rng, key, c);
      double fireReflection = rng.draw(key, SLOT_MIRROR);
      if (m.roughness() < fireReflection) {
        V3 reflDir = r.d - normal * dot(normal, r.d) * 2.0;  // glm::reflect
        Ray reflRay;
        reflRay.p = startPos + RAY_EPSILON * normal;
        reflRay.d = reflDir;
        V3 reflResult = tracePath(reflRay, depth + 1, mix64(key, SLOT_CHILD_MIRROR), prm, rng, c);
        colorC += reflResult;
        if (g_mut != MUT_NO_MIRROR_HALVING) colorC /= 2;
      }
      if (g_mut != MUT_NO_RR_COMPENSATION && !(g_mut == MUT_NO_RR_AT_DEPTH_0 && depth == 0)) colorC /= (1.0 - prm.rr_q);
      return colorC;
    }
    if (hasCube) return cubeColor(r.d);
    return V3(0.0, 0.0, 0.0);
  }

  // RayTracer::trace, RayTracer.cpp:38-56.
  V3 trace(double x, double y, uint64_t key, const mipt_params &prm, const Rng &rng, Counters &c) const {
    V3 look(cam.look[0], cam.look[1], cam.look[2]), u(cam.u[0], cam.u[1], cam.u[2]), v(cam.v[0], cam.v[1], cam.v[2]);
    x -= 0.5;  // camera.cpp:20-29
    y -= 0.5;
    Ray r;
    r.d = normalize(look + x * u + y * v);
    r.p = V3(cam.eye[0], cam.eye[1], cam.eye[2]);
    V3 ret = tracePath(r, 0, key, prm, rng, c);
    c.samples++;
    double *q = &ret.x;
    (void)q;
    V3 out;
    bool nan = false;
    for (int k = 0; k < 3; k++) {
      double vv = ret[k];
      if (vv != vv) { nan = true; vv = 0.0; }
      if (g_mut != MUT_CLAMP_AFTER_MEAN) vv = std::min(std::max(vv, 0.0), 1.0);  // glm::clamp
      out.at(k) = vv;
    }
    if (nan) c.nan_samples++;
    return out;
  }

  static bool ownsRow(const mipt_params &p, int j) {
    int cnt = p.shard_count > 0 ? p.shard_count : 1;
    int band = p.band_rows > 0 ? p.band_rows : 1;
    return ((j / band) % cnt) == p.shard_index;
  }

  // RayTracer::tracePixel, RayTracer.cpp:59-101.  Returns the SUM over this call's
  // sample range; the caller divides by p.spp.
  V3 pixelSum(int i, int j, const mipt_params &p, const Rng &rng, Counters &c) const {
    double x = double(i) / double(p.width);
    double y = double(j) / double(p.height);
    V3 color(0, 0, 0);
    for (int s = p.sample_begin; s < p.sample_end; s++) {
      uint64_t key = sample_key(p.seed, (uint64_t)i + (uint64_t)j * (uint64_t)p.width, (uint64_t)s);
      color += trace(x, y, key, p, rng, c);
    }
    return color;
  }
};

// ------------------------------------------------------------------ C API ----
extern "C" {

const char *orc_last_error(void) { return g_err.c_str(); }

// 0 = the faithful restatement; anything else is one of the mutants above (process-wide).
int orc_set_mutation(int m) {
  if (m < 0 || m >= MUT_COUNT) return -1;
  g_mut = m;
  return 0;
}
int orc_mutation_count(void) { return MUT_COUNT; }

double orc_uniform(uint64_t key, uint64_t j) { return u01(mix64(key, j)); }
uint64_t orc_mix(uint64_t key, uint64_t j) { return mix64(key, j); }
uint64_t orc_sample_key(uint64_t seed, uint64_t pixel, uint64_t sample) { return sample_key(seed, pixel, sample); }

orc_scene *orc_create(const mipt_scene_desc *d) {
  if (!d) { g_err = "null scene description"; return nullptr; }
  std::unique_ptr<orc_scene> s(new orc_scene());
  s->cam = d->camera;
  s->ambient = V3(d->ambient[0], d->ambient[1], d->ambient[2]);
  for (int k = 0; k < d->n_materials; k++) {
    const mipt_material &m = d->materials[k];
    Material mm;
    mm.kd = V3(m.kd[0], m.kd[1], m.kd[2]);
    mm.ka = V3(m.ka[0], m.ka[1], m.ka[2]);
    mm.kt = V3(m.kt[0], m.kt[1], m.kt[2]);
    mm.rough_raw = m.roughness;
    mm.metal_raw = m.metallic;
    mm.index_raw = m.index;
    mm.kdTex = m.kd_texture;
    mm.kaTex = m.ka_texture;
    if (mm.kdTex < 0 || mm.kdTex > d->n_textures || mm.kaTex < 0 || mm.kaTex > d->n_textures) { g_err = "material texture index out of range"; return nullptr; }
    s->materials.push_back(mm);
  }
  for (int k = 0; k < d->n_textures; k++) {
    Tex t;
    t.w = d->textures[k].width;
    t.h = d->textures[k].height;
    t.data.assign(d->textures[k].rgb, d->textures[k].rgb + (size_t)3 * t.w * t.h);
    s->textures.push_back(std::move(t));
  }
  for (int k = 0; k < d->n_meshes; k++) {
    const mipt_mesh &src = d->meshes[k];
    Mesh m;
    m.verts.resize((size_t)src.n_vertices);
    for (int64_t v = 0; v < src.n_vertices; v++) m.verts[v] = V3(src.vertices[3 * v], src.vertices[3 * v + 1], src.vertices[3 * v + 2]);
    if (src.normals) {
      m.normals.resize((size_t)src.n_vertices);
      for (int64_t v = 0; v < src.n_vertices; v++) m.normals[v] = V3(src.normals[3 * v], src.normals[3 * v + 1], src.normals[3 * v + 2]);
    }
    if (src.uvs) m.uvs.assign(src.uvs, src.uvs + 2 * (size_t)src.n_vertices);
    if (src.colors)
      for (int64_t v = 0; v < src.n_vertices; v++) m.colors.push_back(V3(src.colors[3 * v], src.colors[3 * v + 1], src.colors[3 * v + 2]));
    std::vector<BBox> fboxes;
    for (int64_t f = 0; f < src.n_faces; f++) {
      int a = src.faces[3 * f], b = src.faces[3 * f + 1], c = src.faces[3 * f + 2];
      if (a < 0 || b < 0 || c < 0 || a >= src.n_vertices || b >= src.n_vertices || c >= src.n_vertices) {
        g_err = "mesh face index out of range";  // Trimesh::addFace returns false, trimesh.cpp:34-36
        return nullptr;
      }
      // TrimeshFace ctor, trimesh.h:123-158
      V3 A = m.verts[a], B = m.verts[b], C = m.verts[c];
      V3 vab = B - A, vac = C - A, vcb = B - C;
      if (length(vab) == 0.0 || length(vac) == 0.0 || length(vcb) == 0.0) continue;  // degen: dropped
      Mesh::Face face;
      face.ids[0] = a; face.ids[1] = b; face.ids[2] = c;
      face.normal = normalize(cross(B - A, C - A));
      face.box.empty = false;
      face.box.mx = vmax(C, vmax(A, B));
      face.box.mn = vmin(C, vmin(A, B));
      double t1 = dot(B - A, B - A), t2 = dot(B - A, C - A), t3 = dot(C - A, B - A), t4 = dot(C - A, C - A);
      // glm::dmat2(t1,t2,t3,t4): columns (t1,t2),(t3,t4); glm::inverse:
      double ood = 1.0 / (t1 * t4 - t3 * t2);
      face.A00 = t4 * ood;   // [0][0]
      face.A01 = -t2 * ood;  // [0][1]
      face.A10 = -t3 * ood;  // [1][0]
      face.A11 = t1 * ood;   // [1][1]
      m.faces.push_back(face);
      fboxes.push_back(face.box);
    }
    if (!m.faces.empty()) m.tree.build(fboxes);
    s->meshes.push_back(std::move(m));
  }
  std::vector<BBox> oboxes;
  for (int k = 0; k < d->n_objects; k++) {
    const mipt_object &src = d->objects[k];
    Object o;
    o.type = src.type;
    o.material = src.material;
    o.mesh = src.mesh;
    if (o.material < 0 || o.material >= d->n_materials) { g_err = "object material out of range"; return nullptr; }
    if (o.type == MIPT_OBJ_TRIMESH && (o.mesh < 0 || o.mesh >= d->n_meshes)) { g_err = "object mesh out of range"; return nullptr; }
    if (o.type < 0 || o.type > MIPT_OBJ_TRIMESH) { g_err = "unknown object type"; return nullptr; }
    std::memcpy(o.xform.m, src.xform, sizeof(o.xform.m));
    o.inv = inverse(o.xform);
    o.normi = normal_matrix(o.xform);
    // local boxes: Sphere.h:13-18, Box.h:13-18, Square.h:13-18, trimesh.h:75-90
    V3 mn, mx;
    bool haveBox = true;
    if (o.type == MIPT_OBJ_SPHERE) { mn = V3(-1.0); mx = V3(1.0); }
    else if (o.type == MIPT_OBJ_BOX) { mn = V3(-0.5); mx = V3(0.5); }
    else if (o.type == MIPT_OBJ_SQUARE) { mn = V3(-0.5, -0.5, -RAY_EPSILON); mx = V3(0.5, 0.5, RAY_EPSILON); }
    else {
      const Mesh &m = s->meshes[o.mesh];
      if (m.verts.empty()) haveBox = false;
      else {
        mn = mx = m.verts[0];
        for (const V3 &v : m.verts) { mx = vmax(mx, v); mn = vmin(mn, v); }
      }
    }
    // Geometry::ComputeBoundingBox, scene.cpp:54-92
    if (haveBox) {
      V3 nmn, nmx;
      for (int corner = 0; corner < 8; corner++) {
        V3 p((corner & 1) ? mx.x : mn.x, (corner & 2) ? mx.y : mn.y, (corner & 4) ? mx.z : mn.z);
        V3 w = xform_point(o.xform, p);
        if (corner == 0) { nmn = nmx = w; }
        else { nmx = vmax(nmx, w); nmn = vmin(nmn, w); }
      }
      o.bounds.empty = false;
      o.bounds.mn = nmn;
      o.bounds.mx = nmx;
    }
    o.primBase = 0;
    s->objects.push_back(o);
    oboxes.push_back(o.bounds);
  }
  if (!s->objects.empty()) s->tree.build(oboxes);
  int64_t base = 0;
  for (size_t k = 0; k < s->tree.order.size(); k++) {
    Object &o = s->objects[s->tree.order[k]];
    o.primBase = (int)base;
    base += (o.type == MIPT_OBJ_TRIMESH) ? (int64_t)s->meshes[o.mesh].faces.size() : 1;
  }
  s->nPrims = base;
  for (int k = 0; k < d->n_lights; k++) {
    const mipt_light &src = d->lights[k];
    Light L;
    L.type = src.type;
    L.color = V3(src.color[0], src.color[1], src.color[2]);
    L.position = V3(src.position[0], src.position[1], src.position[2]);
    L.corner = L.position;
    V3 dir(src.direction[0], src.direction[1], src.direction[2]);
    L.orientation = (src.type == MIPT_LIGHT_DIRECTIONAL) ? normalize(dir) : dir;  // light.h:46
    L.uVec = V3(src.u[0], src.u[1], src.u[2]);
    L.vVec = V3(src.v[0], src.v[1], src.v[2]);
    L.uL = src.uL;
    L.vL = src.vL;
    L.center = src.uL / 2 * L.uVec + src.vL / 2 * L.vVec + L.corner;  // light.h:124
    L.a = src.atten[0];
    L.b = src.atten[1];
    L.c = src.atten[2];
    s->lights.push_back(L);
  }
  if (d->cubemap) {
    s->hasCube = true;
    for (int f = 0; f < 6; f++) {
      s->cube[f].w = d->cubemap[f].width;
      s->cube[f].h = d->cubemap[f].height;
      s->cube[f].data.assign(d->cubemap[f].rgb, d->cubemap[f].rgb + (size_t)3 * d->cubemap[f].width * d->cubemap[f].height);
    }
  }
  return s.release();
}

void orc_destroy(orc_scene *s) { delete s; }

static bool check_params(const mipt_params *p) {
  if (!p || p->width <= 0 || p->height <= 0 || p->spp <= 0 || p->area_samples <= 0 || p->sample_begin < 0 ||
      p->sample_end > p->spp || p->sample_begin > p->sample_end || p->shard_index < 0 ||
      (p->shard_count > 0 && p->shard_index >= p->shard_count) || !(p->rr_q >= 0.0 && p->rr_q < 1.0)) {
    g_err = "invalid render parameters";
    return false;
  }
  return true;
}

int orc_render(orc_scene *s, const mipt_params *p, int threads, int rng_mode, double *mean_rgb, uint8_t *rgb8, mipt_counters *counters) {
  if (!s || !check_params(p)) return MIPT_ERR_INVALID;
  if (threads < 1) threads = 1;
  int w = p->width, h = p->height;
  if (mean_rgb) std::fill(mean_rgb, mean_rgb + (size_t)3 * w * h, 0.0);
  if (rgb8) std::fill(rgb8, rgb8 + (size_t)3 * w * h, (uint8_t)0);  // traceSetup, RayTracer.cpp:427
  std::vector<Counters> cs((size_t)threads);
  std::vector<std::thread> pool;
  // traceImage, RayTracer.cpp:482-487: contiguous column chunks, last takes the remainder.
  int chunk = w / threads;
  for (int t = 0; t < threads; t++) {
    int start = t * chunk;
    int end = (t == threads - 1) ? w : start + chunk;
    pool.emplace_back([=, &cs]() {
      uint64_t seq = mix64(p->seed, 0xABCDull + (uint64_t)t);
      Rng rng{rng_mode, &seq};
      Counters &c = cs[(size_t)t];
      for (int i = start; i < end; i++) {       // processChunk, RayTracer.cpp:444-450
        for (int j = 0; j < h; j++) {
          if (!orc_scene::ownsRow(*p, j)) continue;
          V3 sum = s->pixelSum(i, j, *p, rng, c);
          V3 color = sum / V3((double)p->spp);   // RayTracer.cpp:75
          size_t o = ((size_t)i + (size_t)j * w) * 3;
          if (mean_rgb) { mean_rgb[o] = color.x; mean_rgb[o + 1] = color.y; mean_rgb[o + 2] = color.z; }
          if (rgb8) {                             // RayTracer.cpp:97-99
            rgb8[o] = (uint8_t)(int)(255.0 * color.x);
            rgb8[o + 1] = (uint8_t)(int)(255.0 * color.y);
            rgb8[o + 2] = (uint8_t)(int)(255.0 * color.z);
          }
        }
      }
    });
  }
  for (auto &th : pool) th.join();  // waitRender, RayTracer.cpp:514-526
  if (counters) {
    std::memset(counters, 0, sizeof(*counters));
    for (auto &c : cs) {
      counters->samples += c.samples;
      counters->path_rays += c.path_rays;
      counters->shadow_rays += c.shadow_rays;
      counters->shaded_hits += c.shaded;
      counters->node_visits += c.node_visits;
      counters->prim_tests += c.prim_tests;
      counters->nan_samples += c.nan_samples;
    }
  }
  return MIPT_OK;
}

int orc_trace_pixel(orc_scene *s, const mipt_params *p, int i, int j, double out[3]) {
  if (!s || !check_params(p) || i < 0 || j < 0 || i >= p->width || j >= p->height) return MIPT_ERR_INVALID;
  uint64_t seq = 0;
  Rng rng{ORC_RNG_COUNTER, &seq};
  Counters c;
  V3 sum = s->pixelSum(i, j, *p, rng, c);
  V3 color = sum / V3((double)p->spp);
  out[0] = color.x; out[1] = color.y; out[2] = color.z;
  return MIPT_OK;
}

static int intersect_many(orc_scene *s, bool brute, int64_t n, const double *org, const double *dir, double tmax, double *t, int32_t *prim, double *normal) {
  if (!s || n < 0) return MIPT_ERR_INVALID;
  Counters c;
  for (int64_t k = 0; k < n; k++) {
    Ray r;
    r.p = V3(org[3 * k], org[3 * k + 1], org[3 * k + 2]);
    r.d = V3(dir[3 * k], dir[3 * k + 1], dir[3 * k + 2]);
    Isect i;
    bool hit = brute ? s->intersectBrute(r, i, tmax) : s->intersect(r, i, tmax, c);
    if (t) t[k] = hit ? i.t : tmax;
    if (prim) prim[k] = hit ? i.prim : -1;
    if (normal) {
      normal[3 * k] = hit ? i.N.x : 0.0;
      normal[3 * k + 1] = hit ? i.N.y : 0.0;
      normal[3 * k + 2] = hit ? i.N.z : 0.0;
    }
  }
  return MIPT_OK;
}

int orc_intersect(orc_scene *s, int64_t n, const double *org, const double *dir, double tmax, double *t, int32_t *prim, double *normal) {
  return intersect_many(s, false, n, org, dir, tmax, t, prim, normal);
}
int orc_intersect_brute(orc_scene *s, int64_t n, const double *org, const double *dir, double tmax, double *t, int32_t *prim, double *normal) {
  return intersect_many(s, true, n, org, dir, tmax, t, prim, normal);
}

int orc_hit_material(orc_scene *s, int64_t n, const double *org, const double *dir, double tmax, double *uv, double *kd, double *ka) {
  if (!s || !org || !dir || !uv || !kd || !ka) { g_err = "null argument"; return -1; }
  for (int64_t k = 0; k < n; k++) {
    Ray r;
    r.p = V3(org[3 * k], org[3 * k + 1], org[3 * k + 2]);
    r.d = V3(dir[3 * k], dir[3 * k + 1], dir[3 * k + 2]);
    Isect i;
    Counters c;
    V3 d(0), a(0);
    double u = 0, v = 0;
    if (s->intersect(r, i, tmax, c)) {
      Material m = s->materialAt(i);
      d = m.kd;
      a = m.ka;
      u = i.u;
      v = i.v;
    }
    uv[2 * k] = u; uv[2 * k + 1] = v;
    for (int q = 0; q < 3; q++) { kd[3 * k + q] = d[q]; ka[3 * k + q] = a[q]; }
  }
  return 0;
}

int orc_shadow_attenuation(orc_scene *s, const mipt_params *p, int light, int64_t n, const double *fire, const uint64_t *key, double *out) {
  if (!s || !check_params(p) || light < 0 || light >= (int)s->lights.size()) return MIPT_ERR_INVALID;
  uint64_t seq = 0;
  Rng rng{ORC_RNG_COUNTER, &seq};
  Counters c;
  for (int64_t k = 0; k < n; k++) {
    V3 firePos(fire[3 * k], fire[3 * k + 1], fire[3 * k + 2]);
    Ray r;
    r.p = firePos;
    r.d = s->lightDirection(s->lights[light], firePos);
    V3 res = s->shadowAttenuation(light, r, firePos, *p, rng, key ? key[k] : 0, c);
    out[3 * k] = res.x; out[3 * k + 1] = res.y; out[3 * k + 2] = res.z;
  }
  return MIPT_OK;
}

int orc_shadow_samples(orc_scene *s, const mipt_params *p, int light, int64_t n, const double *fire, const uint64_t *key, double *out) {
  if (!s || !check_params(p) || light < 0 || light >= (int)s->lights.size()) return MIPT_ERR_INVALID;
  uint64_t seq = 0;
  Rng rng{ORC_RNG_COUNTER, &seq};
  Counters c;
  const bool area = s->lights[light].type == MIPT_LIGHT_RECT_AREA;
  const int S = area ? p->area_samples : 1;
  for (int64_t k = 0; k < n; k++) {
    V3 firePos(fire[3 * k], fire[3 * k + 1], fire[3 * k + 2]);
    Ray r;
    r.p = firePos;
    r.d = s->lightDirection(s->lights[light], firePos);
    for (int sm = 0; sm < S; sm++) {
      V3 res = area ? s->areaSample(light, sm, r, firePos, *p, rng, key ? key[k] : 0, c) : s->shadowAttenuation(light, r, firePos, *p, rng, key ? key[k] : 0, c);
      double *o = out + 3 * ((size_t)k * S + sm);
      o[0] = res.x; o[1] = res.y; o[2] = res.z;
    }
  }
  return MIPT_OK;
}

int orc_shade_brdf(orc_scene *s, const mipt_params *p, int material, const double P[3], const double N[3], const double V[3], const double w[3], const double Lin[3], uint64_t key, double out[3]) {
  if (!s || !check_params(p) || material < 0 || material >= (int)s->materials.size()) return MIPT_ERR_INVALID;
  uint64_t seq = 0;
  Rng rng{ORC_RNG_COUNTER, &seq};
  Counters c;
  V3 res = s->shadeBRDF(s->materials[material], -V3(w[0], w[1], w[2]), V3(P[0], P[1], P[2]), V3(V[0], V[1], V[2]),
                        V3(Lin[0], Lin[1], Lin[2]), V3(N[0], N[1], N[2]), *p, rng, key, c);
  out[0] = res.x; out[1] = res.y; out[2] = res.z;
  return MIPT_OK;
}

int orc_object_order(orc_scene *s, int32_t *order) {
  if (!s || !order) return MIPT_ERR_INVALID;
  for (size_t k = 0; k < s->tree.order.size(); k++) order[k] = s->tree.order[k];
  return MIPT_OK;
}

int orc_cubemap_color(orc_scene *s, int64_t n, const double *dir, double *out) {
  if (!s || !s->hasCube) return MIPT_ERR_INVALID;
  for (int64_t k = 0; k < n; k++) {
    V3 c = s->cubeColor(V3(dir[3 * k], dir[3 * k + 1], dir[3 * k + 2]));
    out[3 * k] = c.x; out[3 * k + 1] = c.y; out[3 * k + 2] = c.z;
  }
  return MIPT_OK;
}

}  // extern "C"
