// Host-side check of the scene builder (path_tracer_amd/csrc/scene_build.cpp): builds random
// one-mesh scenes (plus a few spheres) and verifies the four-wide quantised BVH:
//   * every primitive sits in exactly one leaf, analytic primitives before triangles;
//   * the dequantised box of every child contains the fp64 boxes of all primitives below it
//     (the boxes may only be too large: they cull, they never decide a hit);
//   * stack_need bounds the entries a depth-first walk can hold.
// Prints "ok <scenes>" and exits 0, or describes the first violation and exits 1.
#include <cmath>
#include <cstdio>
#include <limits>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "../../path_tracer_amd/csrc/scene_build.h"

using namespace mipt;

struct B6 { double mn[3], mx[3]; };

static int fail(const char *what, long a = 0, long b = 0) {
  std::printf("FAIL: %s (%ld, %ld)\n", what, a, b);
  return 1;
}

struct Checker {
  const HostScene &S;
  const std::vector<B6> &primBox;  // by storage index
  std::vector<int> seen;
  int worstStack = 0;
  std::string err;

  // returns the fp64 box of everything below `ref`; held = stack entries held on the way here
  bool walk(int32_t ref, int held, B6 *out) {
    for (int a = 0; a < 3; a++) { out->mn[a] = 1e300; out->mx[a] = -1e300; }
    if (ref < 0) {
      uint32_t code = (uint32_t)~ref;
      uint32_t first = code >> 3, count = (code & 7u) + 1u;
      for (uint32_t k = 0; k < count; k++) {
        uint32_t s = first + k;
        if (s >= primBox.size()) { err = "leaf points past the primitive arrays"; return false; }
        seen[s]++;
        for (int a = 0; a < 3; a++) {
          out->mn[a] = std::min(out->mn[a], primBox[s].mn[a]);
          out->mx[a] = std::max(out->mx[a], primBox[s].mx[a]);
        }
      }
      worstStack = std::max(worstStack, held);
      return true;
    }
    if ((size_t)ref >= S.nodes.size()) { err = "child index out of range"; return false; }
    const Node4 &nd = S.nodes[(size_t)ref];
    int n = 0;
    for (int k = 0; k < 4; k++) {
      if (nd.child[k] == kNoChild) continue;
      if (k != n) { err = "children are not packed to the front"; return false; }
      n++;
    }
    if (n != nd.n_children) { err = "n_children does not match the child array"; return false; }
    for (int k = 0; k < n; k++) {
      B6 sub;
      if (!walk(nd.child[k], held + (n - 1), &sub)) return false;
      for (int a = 0; a < 3; a++) {
        double step = std::ldexp(1.0, (int)nd.exp[a] - 127);
        double lo = (double)nd.origin[a] + (double)((nd.lo[a] >> (8 * k)) & 0xff) * step;
        double hi = (double)nd.origin[a] + (double)((nd.hi[a] >> (8 * k)) & 0xff) * step;
        if (sub.mn[a] <= sub.mx[a] && (lo > sub.mn[a] || hi < sub.mx[a])) {
          char buf[200];
          std::snprintf(buf, sizeof buf, "node %d child %d axis %d: quantised [%.17g, %.17g] does not contain [%.17g, %.17g]", ref, k, a, lo, hi,
                        sub.mn[a], sub.mx[a]);
          err = buf;
          return false;
        }
        out->mn[a] = std::min(out->mn[a], sub.mn[a]);
        out->mx[a] = std::max(out->mx[a], sub.mx[a]);
      }
    }
    return true;
  }
};

static int check_scene(unsigned seed, int nTris, int nSpheres, double extent) {
  std::mt19937_64 rng(seed);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  std::vector<double> verts;
  std::vector<int32_t> faces;
  for (int t = 0; t < nTris; t++) {
    double c[3] = {U(rng) * extent, U(rng) * extent * 0.3, U(rng) * extent - 2 * extent};
    double sz = (seed % 3 == 0) ? 1e-4 * extent : 0.02 * extent * (0.2 + std::fabs(U(rng)));
    for (int v = 0; v < 3; v++)
      for (int a = 0; a < 3; a++) verts.push_back(c[a] + U(rng) * sz);
    faces.push_back(3 * t); faces.push_back(3 * t + 1); faces.push_back(3 * t + 2);
  }
  mipt_material mat{};
  mat.kd[0] = mat.kd[1] = mat.kd[2] = 0.5;
  mat.roughness = 1; mat.index = 1.5;
  std::vector<mipt_object> objs;
  const double I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  for (int k = 0; k < nSpheres; k++) {
    mipt_object o{};
    o.type = MIPT_OBJ_SPHERE; o.material = 0; o.mesh = -1;
    for (int i = 0; i < 16; i++) o.xform[i] = I[i];
    o.xform[12] = U(rng) * extent; o.xform[13] = U(rng) * extent; o.xform[14] = U(rng) * extent - 2 * extent;
    objs.push_back(o);
  }
  mipt_mesh mesh{verts.data(), (int64_t)(verts.size() / 3), faces.data(), (int64_t)nTris, nullptr};
  if (nTris) {
    mipt_object o{};
    o.type = MIPT_OBJ_TRIMESH; o.material = 0; o.mesh = 0;
    for (int i = 0; i < 16; i++) o.xform[i] = I[i];
    objs.push_back(o);
  }
  mipt_scene_desc d{};
  d.camera.look[2] = -1; d.camera.u[0] = 1; d.camera.v[1] = 1;
  d.materials = &mat; d.n_materials = 1;
  d.objects = objs.data(); d.n_objects = (int32_t)objs.size();
  d.meshes = &mesh; d.n_meshes = nTris ? 1 : 0;
  HostScene S;
  std::string err;
  if (build_host_scene(&d, &S, &err) != MIPT_OK) return fail(err.c_str());
  size_t n = S.prim_info.size();
  if (n != (size_t)(nTris + nSpheres)) return fail("primitive count", (long)n, nTris + nSpheres);
  // fp64 box of every stored primitive: triangles from their vertices, spheres from centre +- 1
  std::vector<double> fboxes;
  for (int t = 0; t < nTris; t++) {
    for (int a = 0; a < 3; a++) fboxes.push_back(std::min(verts[9 * t + a], std::min(verts[9 * t + 3 + a], verts[9 * t + 6 + a])));
    for (int a = 0; a < 3; a++) fboxes.push_back(std::max(verts[9 * t + a], std::max(verts[9 * t + 3 + a], verts[9 * t + 6 + a])));
  }
  std::vector<B6> primBox(n);
  for (size_t s = 0; s < n; s++) {
    uint32_t info = S.prim_info[s];
    uint32_t kind = info >> 30, obj = info & 0x3fffffffu;
    bool analytic = kind != KIND_TRI;
    if (analytic != (s < (size_t)S.n_analytic)) return fail("analytic primitives must come first", (long)s, S.n_analytic);
    if (analytic) {
      for (int a = 0; a < 3; a++) { primBox[s].mn[a] = objs[obj].xform[12 + a] - 1; primBox[s].mx[a] = objs[obj].xform[12 + a] + 1; }
    } else {
      // Tri::a is the first vertex of the face: find the face by it (vertices are distinct random numbers)
      int face = -1;
      for (int t = 0; t < nTris && face < 0; t++)
        if (verts[9 * t] == S.tris[s].a[0] && verts[9 * t + 1] == S.tris[s].a[1] && verts[9 * t + 2] == S.tris[s].a[2]) face = t;
      if (face < 0) return fail("stored triangle matches no input face", (long)s);
      for (int a = 0; a < 3; a++) { primBox[s].mn[a] = fboxes[6 * face + a]; primBox[s].mx[a] = fboxes[6 * face + 3 + a]; }
    }
  }
  Checker ck{S, primBox, std::vector<int>(n, 0)};
  B6 all;
  if (!ck.walk(0, 0, &all)) return fail(ck.err.c_str());
  for (size_t s = 0; s < n; s++)
    if (ck.seen[s] != 1) return fail("primitive is not in exactly one leaf", (long)s, ck.seen[s]);
  if (ck.worstStack > S.stack_need) return fail("stack_need is too small", ck.worstStack, S.stack_need);
  return 0;
}

// Coordinates the builder cannot do arithmetic on are refused at the boundary, not built into a tree.
static int check_refusals() {
  const double nan = std::numeric_limits<double>::quiet_NaN(), inf = std::numeric_limits<double>::infinity();
  const double bads[] = {nan, inf, -inf, 1e308};
  for (int which = 0; which < 3; which++)
    for (double bad : bads) {
      std::vector<double> verts = {0, 0, 0, 1, 0, 0, 0, 1, 0, 2, 2, 2, 3, 2, 2, 2, 3, 2};
      std::vector<int32_t> faces = {0, 1, 2, 3, 4, 5};
      mipt_material mat{};
      mat.roughness = 1; mat.index = 1;
      mipt_object o{};
      o.type = MIPT_OBJ_TRIMESH; o.material = 0; o.mesh = 0;
      o.xform[0] = o.xform[5] = o.xform[10] = o.xform[15] = 1;
      mipt_mesh mesh{verts.data(), 6, faces.data(), 2, nullptr};
      mipt_scene_desc d{};
      d.camera.look[2] = -1; d.camera.u[0] = 1; d.camera.v[1] = 1;
      d.materials = &mat; d.n_materials = 1;
      d.objects = &o; d.n_objects = 1;
      d.meshes = &mesh; d.n_meshes = 1;
      if (which == 0) verts[10] = bad;
      if (which == 1) o.xform[13] = bad;
      if (which == 2) d.camera.eye[1] = bad;
      if (which == 2 && bad == 1e308) continue;  // a finite camera position is the caller's business
      HostScene S;
      std::string err;
      if (build_host_scene(&d, &S, &err) != MIPT_ERR_INVALID || err.empty()) return fail("non-finite input was not refused", which);
    }
  return 0;
}

int main() {
  if (check_refusals()) return 1;
  int scenes = 0;
  struct Case { int tris, spheres; double extent; };
  const Case cases[] = {{0, 0, 1}, {0, 1, 1}, {1, 0, 1}, {2, 0, 5}, {3, 2, 5}, {17, 0, 1}, {64, 5, 10}, {500, 0, 1e-3}, {500, 3, 1e4}, {3000, 7, 40}};
  for (const Case &c : cases)
    for (unsigned seed = 1; seed <= 3; seed++) {
      if (check_scene(seed, c.tris, c.spheres, c.extent)) {
        std::printf("  in case tris=%d spheres=%d extent=%g seed=%u\n", c.tris, c.spheres, c.extent, seed);
        return 1;
      }
      scenes++;
    }
  std::printf("ok %d\n", scenes);
  return 0;
}
