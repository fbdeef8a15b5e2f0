// intersect.h — closest-hit / any-hit traversal of the flat BVH and the fp64
// primitive tests.  Replaces Scene::intersect (scene/scene.cpp:114-138) and what it
// calls: BVH::checkIntersect (scene/bvh.h:160-185), BoundingBox::intersect
// (scene/bbox.cc:27-63), Geometry::intersect (scene/scene.cpp:13-40) and the
// intersectLocal of Sphere / Box / Square / TrimeshFace.
//
// Result contract: the closest primitive hit with t < t_max, ties in t broken by the
// reference's visiting order (ref_order) — what the reference's unordered, un-pruned
// walk with `test.t < i.t` returns.  Box tests are fp32 on 8-bit quantised boxes and
// conservative (they may only accept too much), so the result does not depend on the
// tree shape or on the quantisation.
#ifndef MIPT_INTERSECT_H_
#define MIPT_INTERSECT_H_

#include "device_math.h"
#include "layout.h"

namespace mipt {

#define MIPT_RAY_EPSILON 0.00000001  // scene/ray.h:147

struct SceneDev {
  const Node4 *nodes;
  const Tri *tris;
  const TriNormals *tri_normals;  // may be null
  const TriUV *tri_uvs;           // may be null
  const TriColor *tri_colors;     // may be null
  const TexRec *textures;         // material textures, n_textures of them
  int n_textures;
  const uint32_t *prim_info;
  const uint32_t *ref_order;
  const ObjRec *objects;
  const MatRec *materials;
  const LightRec *lights;
  const uint16_t *ray_light;  // light of every shadow ray of a hit (rays_per_hit entries)
  int n_lights;
  int n_materials;
  int n_prims;
  int n_analytic;  // primitives [0, n_analytic) are analytic, [n_analytic, n_prims) triangles
  const double *object_boxes;  // per object: world box (min xyz, max xyz), slightly grown
  int lean_any;    // few analytic primitives beside meshes: any-hit walks test them ahead of the walk (k_shadow_any)
  int rays_per_hit;
  int stack_entries;  // LDS traversal stack entries per lane; deeper entries go to stack_ovf
  uint32_t *stack_ovf;     // overflow of the traversal stacks: entry k of thread g at [k * ovf_stride + g]
  uint32_t ovf_stride;     // threads the overflow area is laid out for
  int translucent;
  int vis_bytes;  // opaque scene whose lights are all flat: a shadow ray's result is one byte (reached its light or not)
  int local_meshes;  // some mesh object has ObjRec::local_units set: triangle tests look their object up
  double t_cap;      // the render's t_max: the cap Trimesh::intersectLocal starts every mesh walk with, in LOCAL units (trimesh.cpp:85-86)
  double radius;
  double ambient[3];
  CameraRec camera;
  int has_cubemap;
  const uint8_t *cube[6];
  int cube_w[6], cube_h[6];
};

// ---- analytic primitives, local space --------------------------------------------
// Each returns the local hit parameter (or a negative value for a miss) and the
// local normal.

// Sphere::intersectLocal, Sphere.cpp:9-40.
__device__ __forceinline__ bool sphere_local(d3 p, d3 d, double &t, d3 &N) {
  d = normalize(d);
  d3 v = -p;
  double b = dot(v, d);
  double disc = b * b - dot(v, v) + 1;
  if (disc < 0.0) return false;
  disc = sqrt(disc);
  double t2 = b + disc;
  if (t2 <= MIPT_RAY_EPSILON) return false;
  double t1 = b - disc;
  t = (t1 > MIPT_RAY_EPSILON) ? t1 : t2;
  N = normalize(p + t * d);
  return true;
}

// Box::intersectLocal, Box.cpp:11-74.
__device__ __forceinline__ bool box_local(d3 p, d3 d, double &t, d3 &N, int *face = nullptr) {
  double bestT = 1e100;
  int bestIndex = -1;
#pragma unroll
  for (int it = 0; it < 6; it++) {
    const int mod0 = it % 3;
    double dm = comp(d, mod0);
    if (dm == 0) continue;
    double tt = ((double)(it / 3) - 0.5 - comp(p, mod0)) / dm;
    if (tt < MIPT_RAY_EPSILON || tt > bestT) continue;
    const int mod1 = (it + 1) % 3, mod2 = (it + 2) % 3;
    double x = comp(p, mod1) + tt * comp(d, mod1);
    double y = comp(p, mod2) + tt * comp(d, mod2);
    if (x <= 0.5 && x >= -0.5 && y <= 0.5 && y >= -0.5) {
      if (bestT > tt) {
        bestT = tt;
        bestIndex = it;
      }
    }
  }
  if (bestIndex < 0) return false;
  t = bestT;
  if (face) *face = bestIndex;
  double s = bestIndex < 3 ? -1.0 : 1.0;
  int ax = bestIndex % 3;
  // the reference writes -double(bestIndex == k): the untouched components are -0.0
  // for the negative faces; the sign of zero never reaches a result
  N = d3{ax == 0 ? s : 0.0, ax == 1 ? s : 0.0, ax == 2 ? s : 0.0};
  return true;
}

// Square::intersectLocal, Square.cpp:8-43.
__device__ __forceinline__ bool square_local(d3 p, d3 d, double &t, d3 &N) {
  if (d.z == 0.0) return false;
  double tt = -p.z / d.z;
  if (tt <= MIPT_RAY_EPSILON) return false;
  double px = p.x + tt * d.x, py = p.y + tt * d.y;
  if (px < -0.5 || px > 0.5) return false;
  if (py < -0.5 || py > 0.5) return false;
  t = tt;
  N = d3{0.0, 0.0, d.z > 0.0 ? -1.0 : 1.0};
  return true;
}

// Geometry::intersect, scene.cpp:13-40, for the analytic kinds: world ray -> local,
// intersectLocal, t back to world.  Nlocal is returned untransformed.
__device__ __forceinline__ bool analytic_hit(const ObjRec *o, uint32_t kind, d3 org, d3 dir, double &t, d3 &Nlocal) {
  d3 p = xform_point(o->inv, org);
  d3 dl = xform_point(o->inv, org + dir) - p;
  double len = length(dl);
  dl = normalize(dl);
  double tl;
  bool hit;
  if (kind == KIND_SPHERE) hit = sphere_local(p, dl, tl, Nlocal);
  else if (kind == KIND_BOX) hit = box_local(p, dl, tl, Nlocal);
  else hit = square_local(p, dl, tl, Nlocal);
  if (!hit) return false;
  t = tl / len;
  return true;
}

// TrimeshFace::intersectLocal, trimesh.cpp:98-200, on the world-space triangle.
// w1/w2 are the barycentric weights of the second and third vertex.
// `tfar`: hits beyond it do not matter to the caller (the walk's current best); they are
// rejected before the barycentric half of the record is fetched.
// `lo`: null, or the object of a mesh whose local units matter (ObjRec::local_units).  The
// reference runs this test on the local-space ray (p', d' = inv3 d / len, len = |inv3 d|,
// scene.cpp:18-22): its |n'.d'| equals |n.d| / (|fwdt n| * len) and its t' equals t * len, so
// the parallel-ray threshold (trimesh.cpp:107-109) becomes |n.d| < 1e-8 * |fwdt n| * len and
// the cap the mesh walk starts with (trimesh.cpp:85-86) becomes t * len < t_cap.
// `recheck`: the caller already accepted this hit and only wants its barycentrics back:
// no threshold applies.
__device__ __forceinline__ bool tri_hit(const Tri *tr, d3 org, d3 dir, double &t, double &w1, double &w2, double tfar = 1.0e300,
                                        const ObjRec *lo = nullptr, double t_cap = 0.0, bool recheck = true) {
  const double2 *w = reinterpret_cast<const double2 *>(tr);
  const double2 w0 = w[0], w1_ = w[1], w2_ = w[2];  // one round trip for the plane
  d3 n{w0.x, w0.y, w1_.x};
  double denom = dot(n, dir);
  double eps = recheck ? 0.0 : MIPT_RAY_EPSILON;
  double len = 1.0;
  if (lo) {
    len = length(d3{lo->inv[0] * dir.x + lo->inv[1] * dir.y + lo->inv[2] * dir.z, lo->inv[4] * dir.x + lo->inv[5] * dir.y + lo->inv[6] * dir.z,
                    lo->inv[8] * dir.x + lo->inv[9] * dir.y + lo->inv[10] * dir.z});
    eps = MIPT_RAY_EPSILON * (length(mul3(lo->fwdt, n)) * len);
  }
  if (fabs(denom) < eps) return false;
  d3 a{w1_.y, w2_.x, w2_.y};
  double num = dot(a - org, n);
  double tt = num / denom;
  if (!(tt >= 0) || tt > tfar) return false;
  if (lo && !(tt * len < t_cap)) return false;
  const double2 w3 = w[3], w4 = w[4], w5 = w[5];  // a second one for the edges
  d3 pa = (org + tt * dir) - a;
  double b1 = dot(pa, d3{w3.x, w3.y, w4.x});
  double b2 = dot(pa, d3{w4.y, w5.x, w5.y});
  double b0 = 1 - b1 - b2;
  if (!(b0 >= 0 && b0 <= 1 && b1 >= 0 && b1 <= 1 && b2 >= 0 && b2 <= 1)) return false;
  t = tt;
  w1 = b1;
  w2 = b2;
  return true;
}

// ---- conservative fp32 slab culling --------------------------------------------
// A plane at coordinate b is crossed at t = b * id + off, with off lowered (near planes)
// or raised (far planes) by pad = 1.25 * 2^-21 * (R + |o|) * |1/d|: more than the error of
// rounding the origin and the reciprocal to fp32 and of evaluating b * id in two fmas
// (node_test), 2^-21 of it, plus the rounding of off itself to fp32 (to nearest: |off| <=
// (R + |o|) |1/d| (1 + 2^-20), so it moves by less than the 2^-23 share; a conversion with
// directed rounding is twelve instructions on this chip, to nearest it is one).
// Axis-parallel rays use |d| = 1e-30.
struct Cull {
  float idx, idy, idz;
  float nx, ny, nz;  // near-plane offsets
  float fx, fy, fz;  // far-plane offsets
};

__device__ __forceinline__ void cull_axis(double o, double d, double R, float &id, float &offN, float &offF) {
  double dd = fabs(d) < 1e-30 ? copysign(1e-30, d) : d;
  // 1/dd to ~1e-15 (the hardware estimate is good to 2^-26, one Newton step squares that): it only feeds the fp32 culling
  double inv = __builtin_amdgcn_rcp(dd);
  inv = fma(fma(-dd, inv, 1.0), inv, inv);
  double oi = o * inv;
  double pad = 5.9604644775390625e-07 * (R + fabs(o)) * fabs(inv);
  id = (float)inv;
  offN = (float)(-(oi + pad));
  offF = (float)(-(oi - pad));
}

__device__ __forceinline__ Cull make_cull(d3 o, d3 d, double R) {
  Cull c;
  cull_axis(o.x, d.x, R, c.idx, c.nx, c.fx);
  cull_axis(o.y, d.y, R, c.idy, c.ny, c.fy);
  cull_axis(o.z, d.z, R, c.idz, c.nz, c.fz);
  return c;
}

// Can the ray meet the box `b` (min xyz, max xyz) at 0 <= t <= tmax?  May say yes too often, never no.
__device__ __forceinline__ bool ray_meets_box(d3 o, d3 d, double tmax, const double *b) {
  double t0 = 0.0, t1 = tmax;
  const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
#pragma unroll
  for (int a = 0; a < 3; a++) {
    if (dd[a] == 0.0) {
      if (oo[a] < b[a] || oo[a] > b[3 + a]) return false;
    } else {
      const double inv = 1.0 / dd[a];
      const double ta = (b[a] - oo[a]) * inv, tb = (b[3 + a] - oo[a]) * inv;
      t0 = fmax(t0, fmin(ta, tb));
      t1 = fmin(t1, fmax(ta, tb));
    }
  }
  return t0 <= t1 * (1.0 + 1e-12) + 1e-12;
}

struct Hit {
  double t;
  int prim;  // storage index, -1 = miss
};

struct TravCounters {
  uint32_t nodes;
  uint32_t prims;
};

// ---- traversal -------------------------------------------------------------------
// State of one ray's walk through the BVH.  `cur` is the node to visit next: >= 0 an
// inner node, < 0 a leaf (~cur = first << 3 | count - 1), MIPT_TRAV_DONE when the walk
// is over.
#define MIPT_TRAV_DONE ((int)0x80000000)  // ~code of no valid leaf (needs < 2^28 - 1 primitives)
struct Trav {
  d3 org, dir;
  Cull c;
  float tlimit;
  int cur, sp;
  Hit best;
};

// Traversal stack of one lane: the first S.stack_entries entries live in LDS, one 4-byte
// column per lane (`lds[k * stride]`); a walk that needs more spills the rest to a
// column in global memory.  `gtid` numbers the threads of the launch (< S.ovf_stride).
struct Stack {
  uint32_t *lds;
  int stride;
  uint32_t gtid;
};
__device__ __forceinline__ void stack_push(const SceneDev &S, const Stack &st, int &sp, uint32_t v) {
  if (sp < S.stack_entries) st.lds[sp * st.stride] = v;
  else S.stack_ovf[(size_t)(sp - S.stack_entries) * S.ovf_stride + st.gtid] = v;
  sp++;
}
__device__ __forceinline__ uint32_t stack_pop(const SceneDev &S, const Stack &st, int &sp) {
  sp--;
  // always an LDS read (of the last LDS entry when the top of the stack is in the overflow
  // area), so the common path stays a plain ds_read
  int k = sp < S.stack_entries ? sp : S.stack_entries - 1;
  uint32_t v = st.lds[k * st.stride];
  asm volatile("" : "+v"(v));  // keeps the two reads apart: merged, they become one flat load through the texture path
  if (sp >= S.stack_entries) v = S.stack_ovf[(size_t)(sp - S.stack_entries) * S.ovf_stride + st.gtid];
  return v;
}

__device__ __forceinline__ void trav_init(Trav &T, const SceneDev &S, d3 org, d3 dir, double tmax) {
  T.org = org;
  T.dir = dir;
  T.c = make_cull(org, dir, S.radius);
  T.tlimit = (float)tmax * 1.000001f;  // >= tmax: to nearest loses 2^-24, the factor adds 2^-20
  T.cur = 0;  // the root is always an inner node
  T.sp = 0;
  T.best.t = tmax;
  T.best.prim = -1;
}

__device__ __forceinline__ float ubyte_f(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xffu); }  // v_cvt_f32_ubyteK

// The descend loop of a round is left when fewer lanes than this are still in it.
#ifndef MIPT_MIN_DESC_CLOSEST
#define MIPT_MIN_DESC_CLOSEST 16
#endif
#ifndef MIPT_MIN_DESC_ANY
#define MIPT_MIN_DESC_ANY 16
#endif

#define MIPT_CSWAP(a, b)                                        \
  if (t[b] < t[a]) {                                            \
    float tt_ = t[a]; t[a] = t[b]; t[b] = tt_;                  \
    int cc_ = ch[a]; ch[a] = ch[b]; ch[b] = cc_;                \
  }

// The children of inner node `node` the ray can enter: ch[k] = child reference, hit[k] = the
// ray meets the child's box, and (NEED_T) t[k] = entry distance, 3e38 for a missed or absent child.
// The ray is moved into the node's quantised frame once (per axis
// t = q * (step / d) + ((origin - o) / d -+ pad)), then each child costs six byte->float
// conversions and six fmas.  The errors stay inside the pad of make_cull: in units of
// 2^-24 (R + |o|) |1/d|, of which the pad holds 10 (2 of them for its own rounding), the rounded reciprocal costs one and each
// of the two chained fmas one (|t| of a plane never exceeds (R + |o|) |1/d| by more than the
// 1 % the quantised root box may stick out), so near and far values can be compared as they are.
// node_test_words: the arithmetic on a node's four 16-byte words (also what tools/micro/valu_issue.hip times
// on register-resident words); node_test: the fetch in front of it.
template <bool NEED_T>
__device__ __forceinline__ void node_test_words(const Trav &T, uint4 q0, uint4 q1, uint4 q2, int4 q3, int ch[4], float t[4], bool hit[4]) {
  const float ax = __uint_as_float((q0.w & 0xffu) << 23) * T.c.idx;
  const float ay = __uint_as_float(((q0.w >> 8) & 0xffu) << 23) * T.c.idy;
  const float az = __uint_as_float(((q0.w >> 16) & 0xffu) << 23) * T.c.idz;
  const float ox = __uint_as_float(q0.x), oy = __uint_as_float(q0.y), oz = __uint_as_float(q0.z);
  const float bnx = __fmaf_rn(ox, T.c.idx, T.c.nx), bfx = __fmaf_rn(ox, T.c.idx, T.c.fx);
  const float bny = __fmaf_rn(oy, T.c.idy, T.c.ny), bfy = __fmaf_rn(oy, T.c.idy, T.c.fy);
  const float bnz = __fmaf_rn(oz, T.c.idz, T.c.nz), bfz = __fmaf_rn(oz, T.c.idz, T.c.fz);
  // the plane a ray enters through is the upper one on axes it travels down
  const uint32_t nX = T.c.idx < 0.0f ? q1.w : q1.x, fX = T.c.idx < 0.0f ? q1.x : q1.w;
  const uint32_t nY = T.c.idy < 0.0f ? q2.x : q1.y, fY = T.c.idy < 0.0f ? q1.y : q2.x;
  const uint32_t nZ = T.c.idz < 0.0f ? q2.y : q1.z, fZ = T.c.idz < 0.0f ? q1.z : q2.y;
  ch[0] = q3.x; ch[1] = q3.y; ch[2] = q3.z; ch[3] = q3.w;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    float tn = fmaxf(fmaxf(__fmaf_rn(ubyte_f(nX, k), ax, bnx), __fmaf_rn(ubyte_f(nY, k), ay, bny)),
                     fmaxf(__fmaf_rn(ubyte_f(nZ, k), az, bnz), 0.0f));
    float tf = fminf(fminf(__fmaf_rn(ubyte_f(fX, k), ax, bfx), __fmaf_rn(ubyte_f(fY, k), ay, bfy)),
                     fminf(__fmaf_rn(ubyte_f(fZ, k), az, bfz), T.tlimit));
    hit[k] = tn <= tf && ch[k] != kNoChild;  // tf <= tlimit is finite, so an overflowed tn misses
    if (NEED_T) t[k] = hit[k] ? tn : 3.0e38f;
  }
}
template <bool NEED_T>
__device__ __forceinline__ void node_test(const Trav &T, const SceneDev &S, int node, int ch[4], float t[4], bool hit[4]) {
  const uint4 *np = reinterpret_cast<const uint4 *>(S.nodes + node);
  uint4 q0 = np[0], q1 = np[1], q2 = np[2];
  int4 q3 = *reinterpret_cast<const int4 *>(np + 3);
  node_test_words<NEED_T>(T, q0, q1, q2, q3, ch, t, hit);
}

// Tests primitive `s` (storage index) against the ray.  Closest hit (ANY = false): keeps the
// smallest (t, ref_order).  Any hit: every t < T.best.t counts.  Returns whether T.best changed.
// LOCAL: the scene has a mesh with ObjRec::local_units set (SceneDev::local_meshes), so a triangle
// test looks its object up; scenes without one run kernels compiled without that path.
// TRI_ONLY: `s` is known to be a triangle (the analytic tests are not compiled in).
template <bool ANY, bool LOCAL, bool TRI_ONLY = false, bool NO_TRI = false>  // NO_TRI: the scene has no triangles (their test is not compiled in)
__device__ __forceinline__ bool prim_test(Trav &T, const SceneDev &S, uint32_t s) {
  double t;
  bool hit;
  if (TRI_ONLY || (!NO_TRI && s >= (uint32_t)S.n_analytic)) {
    double w1, w2;
    if (LOCAL) {
      const ObjRec *o = S.objects + (S.prim_info[s] & 0x3fffffffu);
      hit = tri_hit(S.tris + s, T.org, T.dir, t, w1, w2, T.best.t, o->local_units ? o : nullptr, S.t_cap, false);
    } else {
      hit = tri_hit(S.tris + s, T.org, T.dir, t, w1, w2, T.best.t, nullptr, 0.0, false);
    }
  } else {
    uint32_t info = S.prim_info[s];
    d3 Nl;
    hit = analytic_hit(S.objects + (info & 0x3fffffffu), info >> 30, T.org, T.dir, t, Nl);
  }
  if (!hit) return false;
  bool closer = t < T.best.t;
  if (!ANY && t == T.best.t && T.best.prim >= 0) closer = S.ref_order[s] < S.ref_order[T.best.prim];
  if (!closer) return false;
  T.best.t = t;
  T.best.prim = (int)s;
  T.tlimit = (float)t * 1.000001f;
  return true;
}

// One round of the "while-while" walk: descend through inner nodes until the lane holds a
// leaf (or is done), then test that leaf's primitives and pop the next node.  Keeping the
// two phases apart lets the lanes of a wave run the fp64 primitive tests together instead
// of paying for both bodies on every iteration.
// Closest hit (ANY = false): children are entered near to far.  Any hit (ANY = true):
// children are entered in slot order (the builder puts large boxes first) and the first
// primitive with t < tmax ends the walk.
// TRI_ONLY: the leaves' analytic primitives are passed over (the caller has tested them).
template <bool ANY, bool COUNT, bool LOCAL, bool TRI_ONLY = false, bool NO_TRI = false>
__device__ __forceinline__ void trav_round(Trav &T, const SceneDev &S, const Stack &stk, TravCounters &tc) {
  int cur = T.cur, sp = T.sp;
  while (cur >= 0) {
    // any-hit walks do not sort: nearest-first saves 2 % of their node visits, the sort costs more
    constexpr bool SORTED = !ANY;
    int ch[4];
    float t[4];
    bool hit[4];
    node_test<SORTED>(T, S, cur, ch, t, hit);
#ifdef MIPT_LANE_STATS  // lane slots instead of visits: 64 per wave-level iteration
    if (COUNT) { unsigned long long m_ = __ballot(1); if (__lane_id() == (unsigned)__ffsll((long long)m_) - 1u) tc.nodes += 64; }
#else
    if (COUNT) tc.nodes++;
#endif
    if (SORTED) {
      MIPT_CSWAP(0, 1) MIPT_CSWAP(2, 3) MIPT_CSWAP(0, 2) MIPT_CSWAP(1, 3) MIPT_CSWAP(1, 2)
    }
    // enter the first child that was hit, stack the others so that they pop in order
    const bool h0 = SORTED ? t[0] < 2.0e38f : hit[0], h1 = SORTED ? t[1] < 2.0e38f : hit[1];
    const bool h2 = SORTED ? t[2] < 2.0e38f : hit[2], h3 = SORTED ? t[3] < 2.0e38f : hit[3];
    // sorted (closest hit): the hits are the first entries, so child k >= 1 is stacked iff hit
    const bool p1 = !SORTED ? (h1 && h0) : h1;
    const bool p2 = !SORTED ? (h2 && (h0 || h1)) : h2;
    const bool p3 = !SORTED ? (h3 && (h0 || h1 || h2)) : h3;
    const int enter = h0 ? ch[0] : h1 ? ch[1] : h2 ? ch[2] : ch[3];
    if (sp + 3 <= S.stack_entries) {
      // all in LDS: three unconditional writes; one that is not to be kept lands above the
      // top of the stack and is overwritten by the next (no exec-mask changes, no branches)
      stk.lds[sp * stk.stride] = (uint32_t)ch[3];
      sp += p3 ? 1 : 0;
      stk.lds[sp * stk.stride] = (uint32_t)ch[2];
      sp += p2 ? 1 : 0;
      stk.lds[sp * stk.stride] = (uint32_t)ch[1];
      sp += p1 ? 1 : 0;
    } else {
      if (p3) stack_push(S, stk, sp, (uint32_t)ch[3]);
      if (p2) stack_push(S, stk, sp, (uint32_t)ch[2]);
      if (p1) stack_push(S, stk, sp, (uint32_t)ch[1]);
    }
    if (h0 || h1 || h2 || h3) cur = enter;
    else if (sp == 0) cur = MIPT_TRAV_DONE;
    else cur = (int)stack_pop(S, stk, sp);
    // few lanes left descending: let them pause so the others can go on with their leaves
    constexpr int kMinDescending = ANY ? MIPT_MIN_DESC_ANY : MIPT_MIN_DESC_CLOSEST;
    if (kMinDescending > 0 && (int)__popcll(__ballot(cur >= 0)) < kMinDescending) break;
  }
  if (cur < 0 && cur != MIPT_TRAV_DONE) {
    uint32_t code = (uint32_t)~cur;
    uint32_t first = code >> 3, count = (code & 7u) + 1u;
    bool found = false;
    for (uint32_t k = 0; k < count; k++) {
      if (TRI_ONLY && first + k < (uint32_t)S.n_analytic) continue;
#ifdef MIPT_LANE_STATS
      if (COUNT) { unsigned long long m_ = __ballot(1); if (__lane_id() == (unsigned)__ffsll((long long)m_) - 1u) tc.prims += 64; }
#else
      if (COUNT) tc.prims++;
#endif
      if (prim_test<ANY, LOCAL, TRI_ONLY, NO_TRI>(T, S, first + k) && ANY) { found = true; break; }
    }
    if (found || sp == 0) cur = MIPT_TRAV_DONE;
    else cur = (int)stack_pop(S, stk, sp);
  }
  T.cur = cur;
  T.sp = sp;
}

// The whole walk of one ray.
template <bool ANY, bool COUNT, bool LOCAL>
__device__ __forceinline__ Hit traverse(const SceneDev &S, d3 org, d3 dir, double tmax, const Stack &stk, TravCounters &tc) {
  Trav T;
  trav_init(T, S, org, dir, tmax);
  while (T.cur != MIPT_TRAV_DONE) trav_round<ANY, COUNT, LOCAL>(T, S, stk, tc);
  return T.best;
}

// World-space unit normal of a hit, as isect::getN() holds it after
// Geometry::intersect (scene.cpp:30-34).
__device__ __forceinline__ d3 hit_normal(const SceneDev &S, int prim, d3 org, d3 dir) {
  uint32_t info = S.prim_info[prim];
  uint32_t kind = info >> 30;
  const ObjRec *o = S.objects + (info & 0x3fffffffu);
  if (kind == KIND_TRI) {
    const Tri *tr = S.tris + prim;
    if (o->smooth && S.tri_normals) {
      double t, w1, w2;
      if (tri_hit(tr, org, dir, t, w1, w2)) {
        const TriNormals *vn = S.tri_normals + prim;
        double w0 = 1 - w1 - w2;
        d3 n = (ld3(vn->na) * w0 + ld3(vn->nb) * w1) + ld3(vn->nc) * w2;
        return normalize(n);
      }
    }
    return ld3(tr->n);
  }
  double t;
  d3 Nl = d3{0, 0, 1};
  analytic_hit(o, kind, org, dir, t, Nl);
  return normalize(mul3(o->normi, Nl));
}

// UV coordinates of a hit, as isect::getUVCoordinates() holds them: Square.cpp:41,
// Box.cpp:56-71, trimesh.cpp:165-172; (0, 0) for spheres and meshes without UVs (ray.h:71).
__device__ __forceinline__ void hit_uv(const SceneDev &S, int prim, d3 org, d3 dir, double &u, double &v) {
  u = 0.0;
  v = 0.0;
  uint32_t info = S.prim_info[prim];
  uint32_t kind = info >> 30;
  const ObjRec *o = S.objects + (info & 0x3fffffffu);
  if (kind == KIND_TRI) {
    double t, w1, w2;
    if (o->has_uv && S.tri_uvs && tri_hit(S.tris + prim, org, dir, t, w1, w2)) {
      const TriUV *q = S.tri_uvs + prim;
      double w0 = 1 - w1 - w2;
      u = (q->ua * w0 + q->ub * w1) + q->uc * w2;
      v = (q->va * w0 + q->vb * w1) + q->vc * w2;
    }
    return;
  }
  if (kind == KIND_SPHERE) return;
  d3 p = xform_point(o->inv, org);
  d3 dl = normalize(xform_point(o->inv, org + dir) - p);
  double tl;
  d3 Nl;
  if (kind == KIND_SQUARE) {
    if (!square_local(p, dl, tl, Nl)) return;
    d3 P = p + tl * dl;
    u = P.x + 0.5;
    v = P.y + 0.5;
  } else {
    int face = 0;
    if (!box_local(p, dl, tl, Nl, &face)) return;
    d3 P = p + tl * dl;
    int i1 = (face + 1) % 3, i2 = (face + 2) % 3;
    double lo = comp(P, i1 < i2 ? i1 : i2), hi = comp(P, i1 < i2 ? i2 : i1);
    u = face < 3 ? 0.5 - lo : 0.5 + lo;
    v = 0.5 + hi;
  }
}

// Diffuse colour of a hit on a mesh that shades with per-vertex colours
// (trimesh.cpp:179-189: a copy of the mesh's material with kd replaced); false elsewhere.
__device__ __forceinline__ bool hit_vertex_color(const SceneDev &S, int prim, d3 org, d3 dir, d3 &kd) {
  if (!S.tri_colors) return false;
  uint32_t info = S.prim_info[prim];
  if ((info >> 30) != KIND_TRI) return false;
  const ObjRec *o = S.objects + (info & 0x3fffffffu);
  double t, w1, w2;
  if (!o->has_colors || !tri_hit(S.tris + prim, org, dir, t, w1, w2)) return false;
  const TriColor *q = S.tri_colors + prim;
  double w0 = 1 - w1 - w2;
  kd = (ld3(q->ca) * w0 + ld3(q->cb) * w1) + ld3(q->cc) * w2;
  return true;
}

}  // namespace mipt
#endif
