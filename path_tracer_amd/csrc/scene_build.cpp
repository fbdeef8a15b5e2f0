// scene_build.cpp — host-side scene flattening for libmipt.
//
// What it produces (layout.h):
//  * every primitive of the scene (analytic objects and non-degenerate mesh
//    triangles) numbered in the order the reference's two BVH levels would visit
//    it — that number breaks exact ties in t the way the reference's "first
//    visited wins" rule does (scene/bvh.h:160-185 with strict `<` at :172);
//  * triangles pre-transformed to world space in fp64;
//  * one flat binned-SAH BVH2 over all primitives with fp32 child boxes rounded
//    outward.  The tree shape is ours; only the tie-break order is the reference's.
#include "scene_build.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <limits>
#include <memory>

namespace mipt {
namespace {

// Weights of a box's x-, y- and z-facing sides in the builder's surface-area measure: (1, 1, 1) is
// the plain surface-area heuristic (rays from all directions alike).
double g_face_weight[3] = {1.0, 1.0, 1.0};

struct D3 {
  double x, y, z;
};
inline D3 operator-(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline D3 operator+(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline D3 operator*(D3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot(D3 a, D3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline D3 cross(D3 a, D3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline double len(D3 a) { return std::sqrt(dot(a, a)); }
inline D3 unit(D3 a) { return a * (1.0 / std::sqrt(dot(a, a))); }

// Column-major 4x4 as in the ABI: m[4*c + r].
inline double M(const double *m, int r, int c) { return m[4 * c + r]; }

inline D3 apply_point(const double *m, D3 v) {  // fourth row dropped, scene.h:59-63
  return {M(m, 0, 0) * v.x + M(m, 0, 1) * v.y + M(m, 0, 2) * v.z + M(m, 0, 3),
          M(m, 1, 0) * v.x + M(m, 1, 1) * v.y + M(m, 1, 2) * v.z + M(m, 1, 3),
          M(m, 2, 0) * v.x + M(m, 2, 1) * v.y + M(m, 2, 2) * v.z + M(m, 2, 3)};
}

// Inverse of a general 4x4 (Gauss-Jordan with partial pivoting, fp64).
bool invert4(const double *a, double *out) {
  double w[4][8];
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) {
      w[r][c] = M(a, r, c);
      w[r][c + 4] = (r == c) ? 1.0 : 0.0;
    }
  for (int col = 0; col < 4; col++) {
    int piv = col;
    for (int r = col + 1; r < 4; r++)
      if (std::fabs(w[r][col]) > std::fabs(w[piv][col])) piv = r;
    if (w[piv][col] == 0.0) return false;
    if (piv != col)
      for (int c = 0; c < 8; c++) std::swap(w[piv][c], w[col][c]);
    double d = 1.0 / w[col][col];
    for (int c = 0; c < 8; c++) w[col][c] *= d;
    for (int r = 0; r < 4; r++) {
      if (r == col) continue;
      double f = w[r][col];
      if (f == 0.0) continue;
      for (int c = 0; c < 8; c++) w[r][c] -= f * w[col][c];
    }
  }
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) out[4 * c + r] = w[r][c + 4];
  return true;
}

// transpose(inverse(upper-left 3x3)), row-major out[3*r + c].
bool normal_matrix(const double *m, double *out) {
  double a = M(m, 0, 0), b = M(m, 0, 1), c = M(m, 0, 2);
  double d = M(m, 1, 0), e = M(m, 1, 1), f = M(m, 1, 2);
  double g = M(m, 2, 0), h = M(m, 2, 1), i = M(m, 2, 2);
  double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
  if (det == 0.0) return false;
  double k = 1.0 / det;
  // cofactor matrix / det == transpose(inverse)
  out[0] = (e * i - f * h) * k; out[1] = (f * g - d * i) * k; out[2] = (d * h - e * g) * k;
  out[3] = (c * h - b * i) * k; out[4] = (a * i - c * g) * k; out[5] = (b * g - a * h) * k;
  out[6] = (b * f - c * e) * k; out[7] = (c * d - a * f) * k; out[8] = (a * e - b * d) * k;
  return true;
}

inline D3 apply3(const double *n, D3 v) {  // row-major 3x3
  return {n[0] * v.x + n[1] * v.y + n[2] * v.z, n[3] * v.x + n[4] * v.y + n[5] * v.z, n[6] * v.x + n[7] * v.y + n[8] * v.z};
}

struct Box6 {
  double mn[3], mx[3];
  void reset() {
    for (int a = 0; a < 3; a++) {
      mn[a] = std::numeric_limits<double>::infinity();
      mx[a] = -std::numeric_limits<double>::infinity();
    }
  }
  void grow(D3 p) {
    mn[0] = std::min(mn[0], p.x); mx[0] = std::max(mx[0], p.x);
    mn[1] = std::min(mn[1], p.y); mx[1] = std::max(mx[1], p.y);
    mn[2] = std::min(mn[2], p.z); mx[2] = std::max(mx[2], p.z);
  }
  void grow(const Box6 &b) {
    for (int a = 0; a < 3; a++) {
      mn[a] = std::min(mn[a], b.mn[a]);
      mx[a] = std::max(mx[a], b.mx[a]);
    }
  }
  double area() const {
    double dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    if (dx < 0) return 0;
    return 2.0 * (g_face_weight[2] * (dx * dy) + g_face_weight[0] * (dy * dz) + g_face_weight[1] * (dz * dx));
  }
};

// Largest and smallest singular value of a row-major 3x3 matrix: square roots of the extreme
// eigenvalues of A^T A, by cyclic Jacobi rotations (a 3x3 symmetric matrix converges to
// rounding in a few sweeps).
void singular_range3(const double *a, double *smax, double *smin) {
  double s[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) s[i][j] = a[i] * a[j] + a[3 + i] * a[3 + j] + a[6 + i] * a[6 + j];
  for (int sweep = 0; sweep < 32; sweep++) {
    double off = std::fabs(s[0][1]) + std::fabs(s[0][2]) + std::fabs(s[1][2]);
    if (off <= 1e-300 || off <= 1e-18 * (std::fabs(s[0][0]) + std::fabs(s[1][1]) + std::fabs(s[2][2]))) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        if (s[p][q] == 0.0) continue;
        double theta = (s[q][q] - s[p][p]) / (2.0 * s[p][q]);
        double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < 3; k++) {  // columns p, q
          double kp = s[k][p], kq = s[k][q];
          s[k][p] = c * kp - sn * kq;
          s[k][q] = sn * kp + c * kq;
        }
        for (int k = 0; k < 3; k++) {  // rows p, q
          double pk = s[p][k], qk = s[q][k];
          s[p][k] = c * pk - sn * qk;
          s[q][k] = sn * pk + c * qk;
        }
      }
  }
  double hi = std::max(s[0][0], std::max(s[1][1], s[2][2])), lo = std::min(s[0][0], std::min(s[1][1], s[2][2]));
  *smax = std::sqrt(std::max(hi, 0.0));
  *smin = std::sqrt(std::max(lo, 0.0));
}

inline float round_down(double v) {
  float f = (float)v;
  if ((double)f > v) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
  return f;
}
inline float round_up(double v) {
  float f = (float)v;
  if ((double)f < v) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
  return f;
}

// ------------------------------------------------------------- SAH builder ----
struct BuildPrim {
  Box6 box;
  double c[3];
  uint32_t id;  // reference-order index
};

constexpr int kLeafMax = 2;    // preferred leaf size (wide quantised nodes make small leaves pay)
constexpr int kLeafHard = 8;   // encoding limit (3 bits)
constexpr int kBins = 32;

// Binary node of the SAH build, before the collapse to four-wide nodes.
struct BinNode {
  Box6 box[2];
  int32_t child[2];
};

struct Builder {
  std::vector<BuildPrim> &prims;
  std::vector<BinNode> nodes;
  int max_depth = 0;
  int leaf_max = kLeafMax;
  int forks = 0;  // levels of the recursion that still build their two halves on two threads
  explicit Builder(std::vector<BuildPrim> &p) : prims(p) {}

  // Appends the nodes of a finished sub-builder; returns the offset its indices moved by.
  int32_t splice(const Builder &sub) {
    int32_t off = (int32_t)nodes.size();
    for (BinNode nd : sub.nodes) {
      if (nd.child[0] >= 0) nd.child[0] += off;
      if (nd.child[1] >= 0) nd.child[1] += off;
      nodes.push_back(nd);
    }
    max_depth = std::max(max_depth, sub.max_depth);
    return off;
  }

  static int32_t leaf_code(size_t first, size_t count) { return ~(int32_t)((first << 3) | (count - 1)); }

  // Builds [b, e); returns the child reference and fills `bounds`.
  int32_t build(size_t b, size_t e, int depth, Box6 *bounds) {
    size_t n = e - b;
    // large ranges near the root are scanned by several threads, each on a slice of its own
    const int slices = (forks > 0 && n >= ((size_t)1 << 17)) ? 8 : 1;
    auto slice_of = [&](int k, size_t *sb, size_t *se) {
      *sb = b + n * (size_t)k / (size_t)slices;
      *se = b + n * (size_t)(k + 1) / (size_t)slices;
    };
    struct Bounds { Box6 bb, cb; };
    auto scan_bounds = [&](size_t sb, size_t se) {
      Bounds r;
      r.bb.reset();
      r.cb.reset();
      for (size_t i = sb; i < se; i++) {
        r.bb.grow(prims[i].box);
        r.cb.grow(D3{prims[i].c[0], prims[i].c[1], prims[i].c[2]});
      }
      return r;
    };
    Box6 bb, cb;
    bb.reset();
    cb.reset();
    {
      std::vector<std::future<Bounds>> parts;
      for (int k = 1; k < slices; k++) {
        size_t sb, se;
        slice_of(k, &sb, &se);
        parts.push_back(std::async(std::launch::async, scan_bounds, sb, se));
      }
      size_t sb, se;
      slice_of(0, &sb, &se);
      Bounds r = scan_bounds(sb, se);
      bb = r.bb;
      cb = r.cb;
      for (auto &f : parts) {
        Bounds q = f.get();
        bb.grow(q.bb);
        cb.grow(q.cb);
      }
    }
    *bounds = bb;
    if (n <= (size_t)leaf_max) return leaf_code(b, n);

    // binned SAH over the three axes, one pass over the primitives
    struct Bins { Box6 box[3][kBins]; size_t cnt[3][kBins]; };
    double lo3[3], scale3[3];
    bool use[3];
    for (int axis = 0; axis < 3; axis++) {
      lo3[axis] = cb.mn[axis];
      use[axis] = cb.mx[axis] > cb.mn[axis];
      scale3[axis] = use[axis] ? kBins / (cb.mx[axis] - cb.mn[axis]) : 0.0;
    }
    auto scan_bins = [&](size_t sb, size_t se) {
      std::unique_ptr<Bins> r(new Bins);
      for (int axis = 0; axis < 3; axis++)
        for (int k = 0; k < kBins; k++) { r->box[axis][k].reset(); r->cnt[axis][k] = 0; }
      for (size_t i = sb; i < se; i++)
        for (int axis = 0; axis < 3; axis++) {
          if (!use[axis]) continue;
          int k = (int)((prims[i].c[axis] - lo3[axis]) * scale3[axis]);
          if (k < 0) k = 0;
          if (k >= kBins) k = kBins - 1;
          r->box[axis][k].grow(prims[i].box);
          r->cnt[axis][k]++;
        }
      return r;
    };
    std::unique_ptr<Bins> bins;
    {
      std::vector<std::future<std::unique_ptr<Bins>>> parts;
      for (int k = 1; k < slices; k++) {
        size_t sb, se;
        slice_of(k, &sb, &se);
        parts.push_back(std::async(std::launch::async, scan_bins, sb, se));
      }
      size_t sb, se;
      slice_of(0, &sb, &se);
      bins = scan_bins(sb, se);
      for (auto &f : parts) {
        std::unique_ptr<Bins> q = f.get();
        for (int axis = 0; axis < 3; axis++)
          for (int k = 0; k < kBins; k++) {
            if (q->cnt[axis][k]) bins->box[axis][k].grow(q->box[axis][k]);
            bins->cnt[axis][k] += q->cnt[axis][k];
          }
      }
    }
    double bestCost = std::numeric_limits<double>::infinity();
    int bestAxis = -1, bestSplit = -1;
    for (int axis = 0; axis < 3; axis++) {
      if (!use[axis]) continue;
      const Box6 *binBox = bins->box[axis];
      const size_t *binCnt = bins->cnt[axis];
      double rightArea[kBins];
      size_t rightCnt[kBins];
      Box6 acc;
      acc.reset();
      size_t cnt = 0;
      for (int k = kBins - 1; k >= 1; k--) {
        acc.grow(binBox[k]);
        cnt += binCnt[k];
        rightArea[k] = acc.area();
        rightCnt[k] = cnt;
      }
      acc.reset();
      cnt = 0;
      for (int k = 0; k < kBins - 1; k++) {
        acc.grow(binBox[k]);
        cnt += binCnt[k];
        if (cnt == 0 || rightCnt[k + 1] == 0) continue;
        double cost = acc.area() * (double)cnt + rightArea[k + 1] * (double)rightCnt[k + 1];
        if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestSplit = k; }
      }
    }
    size_t mid;
    if (bestAxis < 0) {
      // all centroids coincide: split by position in the array
      if (n <= (size_t)kLeafHard) return leaf_code(b, n);
      mid = b + n / 2;
    } else {
      double leafCost = bb.area() * (double)n;
      if (n <= (size_t)kLeafHard && leafCost <= 0.125 * bb.area() + bestCost) return leaf_code(b, n);
      double lo = cb.mn[bestAxis], hi = cb.mx[bestAxis];
      double scale = kBins / (hi - lo);
      auto it = std::partition(prims.begin() + b, prims.begin() + e, [&](const BuildPrim &p) {
        int k = (int)((p.c[bestAxis] - lo) * scale);
        if (k < 0) k = 0;
        if (k >= kBins) k = kBins - 1;
        return k <= bestSplit;
      });
      mid = (size_t)(it - prims.begin());
      if (mid == b || mid == e) mid = b + n / 2;
    }
    int32_t me = (int32_t)nodes.size();
    nodes.emplace_back();
    max_depth = std::max(max_depth, depth + 1);
    Box6 lb, rb;
    int32_t l, r;
    if (forks > 0 && n >= ((size_t)1 << 13)) {
      // the halves work on disjoint ranges of `prims`; each builds into a node array of its own,
      // spliced in afterwards in the order the sequential recursion allocates (left, then right)
      Builder L(prims), R(prims);
      L.leaf_max = R.leaf_max = leaf_max;
      L.forks = R.forks = forks - 1;
      auto left = std::async(std::launch::async, [&]() { return L.build(b, mid, depth + 1, &lb); });
      r = R.build(mid, e, depth + 1, &rb);
      l = left.get();
      int32_t offL = splice(L);
      if (l >= 0) l += offL;
      int32_t offR = splice(R);
      if (r >= 0) r += offR;
    } else {
      l = build(b, mid, depth + 1, &lb);
      r = build(mid, e, depth + 1, &rb);
    }
    BinNode &nd = nodes[(size_t)me];
    nd.box[0] = lb;
    nd.box[1] = rb;
    nd.child[0] = l;
    nd.child[1] = r;
    return me;
  }
};

// ---------------------------------------------------- collapse to BVH4 ----
// Quantises up to four fp64 child boxes into the frame of one Node4 (see layout.h).
void quantise(Node4 &nd, const Box6 *boxes, const int32_t *refs, int n) {
  std::memset(&nd, 0, sizeof(nd));
  nd.n_children = (uint8_t)n;
  for (int k = 0; k < 4; k++) nd.child[k] = k < n ? refs[k] : kNoChild;
  for (int a = 0; a < 3; a++) {
    double lo = std::numeric_limits<double>::infinity(), hi = -lo;
    for (int k = 0; k < n; k++) {
      lo = std::min(lo, boxes[k].mn[a]);
      hi = std::max(hi, boxes[k].mx[a]);
    }
    if (!(hi >= lo)) { lo = 0; hi = 0; }  // only empty boxes (an empty mesh)
    float O = round_down(lo);
    double extent = hi - (double)O;
    int e = 127 - 60;
    if (extent > 0) {
      int ex;
      std::frexp(extent / 255.0, &ex);  // extent/255 = m * 2^ex, m in [0.5, 1)
      e = std::max(e, 127 + ex);
    }
    for (;; e++) {
      double step = std::ldexp(1.0, e - 127);
      bool fits = true;
      uint32_t lo_w = 0, hi_w = 0;
      for (int k = 0; k < n && fits; k++) {
        double mn = boxes[k].mn[a], mx = boxes[k].mx[a];
        if (!(mx >= mn)) { mn = lo; mx = lo; }
        double ql = std::floor((mn - (double)O) / step);
        double qh = std::ceil((mx - (double)O) / step);
        if ((double)O + ql * step > mn) ql -= 1;
        if ((double)O + qh * step < mx) qh += 1;
        if (ql < 0) ql = 0;
        if (qh > 255) { fits = false; break; }
        lo_w |= (uint32_t)ql << (8 * k);
        hi_w |= (uint32_t)qh << (8 * k);
      }
      if (fits) {
        nd.origin[a] = O;
        nd.exp[a] = (uint8_t)e;
        nd.lo[a] = lo_w;
        nd.hi[a] = hi_w;
        break;
      }
    }
  }
}

struct Collapser {
  const std::vector<BinNode> &bin;
  std::vector<Node4> out;
  int max_depth = 0;
  explicit Collapser(const std::vector<BinNode> &b) : bin(b) {}

  // Emits the tree below binary node `n2`; returns the index of its root (always the first
  // node emitted) and sets *need to the most stack entries a walk below it can hold at once.
  int32_t emit(int32_t n2, int depth, int *need) {
    int32_t me = (int32_t)out.size();
    out.emplace_back();
    emit_into(me, n2, depth, need);
    return me;
  }

  // Fills slot `me` with the four-wide node that replaces binary node `n2`.  The inner
  // children of a node get adjacent slots (siblings are visited one after the other, so
  // they share cache lines), each child's own children follow as it is filled.
  void emit_into(int32_t me, int32_t n2, int depth, int *need) {
    Box6 boxes[4];
    int32_t refs[4];
    int n = 2;
    boxes[0] = bin[(size_t)n2].box[0]; refs[0] = bin[(size_t)n2].child[0];
    boxes[1] = bin[(size_t)n2].box[1]; refs[1] = bin[(size_t)n2].child[1];
    while (n < 4) {
      // open the inner child with the largest surface: its two children take its place
      int best = -1;
      double bestArea = -1;
      for (int k = 0; k < n; k++)
        if (refs[k] >= 0 && boxes[k].area() > bestArea) { bestArea = boxes[k].area(); best = k; }
      if (best < 0) break;
      const BinNode &b = bin[(size_t)refs[best]];
      boxes[best] = b.box[0]; refs[best] = b.child[0];
      boxes[n] = b.box[1]; refs[n] = b.child[1];
      n++;
    }
    // larger boxes first: an any-hit walk enters the first child it hits
    for (int i = 1; i < n; i++)
      for (int j = i; j > 0 && boxes[j].area() > boxes[j - 1].area(); j--) {
        std::swap(boxes[j], boxes[j - 1]);
        std::swap(refs[j], refs[j - 1]);
      }
    max_depth = std::max(max_depth, depth + 1);
    int32_t slot = (int32_t)out.size();
    int inner = 0;
    for (int k = 0; k < n; k++) inner += refs[k] >= 0 ? 1 : 0;
    out.resize(out.size() + (size_t)inner);
    int below = 0;
    for (int k = 0; k < n; k++)
      if (refs[k] >= 0) {
        int nb = 0;
        emit_into(slot, refs[k], depth + 1, &nb);
        refs[k] = slot++;
        below = std::max(below, nb);
      }
    quantise(out[(size_t)me], boxes, refs, n);
    *need = (n - 1) + below;
  }
};

}  // namespace

namespace {
// One range of BVH<T>::makeBVH's in-place partition (bvh.h:40-110) on order[begin, begin + amt).
// The two halves touch disjoint parts of `order`, so large ones go to another thread.
void visit_order_range(const std::vector<double> &boxes, std::vector<uint32_t> &order, size_t begin, size_t amt, int forks) {
  auto bmin = [&](size_t k, int a) { return boxes[6 * (size_t)order[k] + a]; };
  auto bmax = [&](size_t k, int a) { return boxes[6 * (size_t)order[k] + 3 + a]; };
  while (true) {
    if (amt <= 2) return;  // bvh.h:49-52
    size_t end = begin + amt - 1;
    double mn[3], mx[3];
    for (int a = 0; a < 3; a++) { mn[a] = bmin(begin, a); mx[a] = bmax(begin, a); }
    for (size_t k = begin + 1; k <= end; k++)
      for (int a = 0; a < 3; a++) {
        if (bmin(k, a) < mn[a]) mn[a] = bmin(k, a);
        if (bmax(k, a) > mx[a]) mx[a] = bmax(k, a);
      }
    double lengths[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
    int axis = 0;  // bvh.h:58-64
    if (lengths[1] > lengths[axis]) axis = 1;
    if (lengths[2] > lengths[axis]) axis = 2;
    double half = lengths[axis] / 2 + mn[axis];  // bvh.h:66
    // two-pointer partition, bvh.h:67-80 (signed indices: j may pass below begin)
    int64_t i = (int64_t)begin, j = (int64_t)end;
    while (i <= j) {
      double mid = (bmin((size_t)i, axis) + bmax((size_t)i, axis)) / 2;
      if (mid >= half) {
        std::swap(order[(size_t)i], order[(size_t)j]);
        j--;
      } else {
        i++;
      }
    }
    size_t leftCount = (size_t)i - begin;
    if (leftCount == 0 || leftCount == amt) return;  // bvh.h:97-100: stays one leaf
    size_t rightBegin = (size_t)i, rightCount = amt - leftCount;
    if (forks > 0 && amt >= ((size_t)1 << 15)) {
      auto left = std::async(std::launch::async, [&, begin, leftCount, forks]() { visit_order_range(boxes, order, begin, leftCount, forks - 1); });
      visit_order_range(boxes, order, rightBegin, rightCount, forks - 1);
      left.get();
      return;
    }
    visit_order_range(boxes, order, begin, leftCount, 0);
    begin = rightBegin;  // the right half in this frame: no deep recursion on lopsided splits
    amt = rightCount;
  }
}
}  // namespace

void reference_visit_order(const std::vector<double> &boxes, std::vector<uint32_t> *order_out) {
  size_t n = boxes.size() / 6;
  std::vector<uint32_t> &order = *order_out;
  order.resize(n);
  for (size_t i = 0; i < n; i++) order[i] = (uint32_t)i;
  if (n == 0) return;
  visit_order_range(boxes, order, 0, n, 5);
}

namespace {
// MIPT_BUILD_TIMING=1 prints the time of each build phase to stderr.
struct PhaseTimer {
  bool on = debug_env("MIPT_BUILD_TIMING") != nullptr;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void lap(const char *what) {
    auto n = std::chrono::steady_clock::now();
    if (on) std::fprintf(stderr, "[mipt build] %-28s %8.3f s\n", what, std::chrono::duration<double>(n - t).count());
    t = n;
  }
};
}  // namespace

int build_host_scene(const mipt_scene_desc *d, HostScene *out, std::string *err, bool build_tree) {
  PhaseTimer timer;
  HostScene &S = *out;
  S = HostScene();
  auto fail = [&](int code, const char *msg) { *err = msg; return code; };
  if (!d) return fail(MIPT_ERR_INVALID, "null scene description");
  if (d->n_objects < 0 || d->n_materials < 0 || d->n_meshes < 0 || d->n_lights < 0)
    return fail(MIPT_ERR_INVALID, "negative count in scene description");
  if ((d->n_objects && !d->objects) || (d->n_materials && !d->materials) || (d->n_meshes && !d->meshes) || (d->n_lights && !d->lights))
    return fail(MIPT_ERR_INVALID, "null array in scene description");
  if (d->n_objects >= (1 << 30)) return fail(MIPT_ERR_LIMIT, "too many objects (limit 2^30-1)");

  g_face_weight[0] = g_face_weight[1] = g_face_weight[2] = 1.0;
  if (const char *e = debug_env("MIPT_SAH_WEIGHTS")) {  // tuning knob: "wx,wy,wz"
    double w[3];
    if (std::sscanf(e, "%lf%*[,:]%lf%*[,:]%lf", &w[0], &w[1], &w[2]) == 3 && w[0] > 0 && w[1] > 0 && w[2] > 0)
      for (int a = 0; a < 3; a++) g_face_weight[a] = w[a];
  }
  auto finite3 = [](const double *v) { return std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]); };
  if (!finite3(d->camera.eye) || !finite3(d->camera.look) || !finite3(d->camera.u) || !finite3(d->camera.v))
    return fail(MIPT_ERR_INVALID, "camera vectors are not finite");
  std::memcpy(S.camera.eye, d->camera.eye, sizeof(double) * 3);
  std::memcpy(S.camera.look, d->camera.look, sizeof(double) * 3);
  std::memcpy(S.camera.u, d->camera.u, sizeof(double) * 3);
  std::memcpy(S.camera.v, d->camera.v, sizeof(double) * 3);
  std::memcpy(S.ambient, d->ambient, sizeof(double) * 3);

  auto lum = [](double v) { return (0.299 * v) + (0.587 * v) + (0.114 * v); };  // material.cpp:240
  for (int k = 0; k < d->n_materials; k++) {
    const mipt_material &m = d->materials[k];
    MatRec r;
    for (int a = 0; a < 3; a++) { r.kd[a] = m.kd[a]; r.ka[a] = m.ka[a]; r.kt[a] = m.kt[a]; }
    r.roughness = lum(m.roughness);
    r.metallic = lum(m.metallic);
    r.index = lum(m.index);
    if (m.kd_texture < 0 || m.kd_texture > d->n_textures || m.ka_texture < 0 || m.ka_texture > d->n_textures)
      return fail(MIPT_ERR_INVALID, "material texture index out of range");
    r.kd_tex = m.kd_texture;
    r.ka_tex = m.ka_texture;
    if (m.kt[0] != 0.0 || m.kt[1] != 0.0 || m.kt[2] != 0.0) S.translucent = true;
    S.materials.push_back(r);
  }

  // ---- lights --------------------------------------------------------------
  for (int k = 0; k < d->n_lights; k++) {
    const mipt_light &src = d->lights[k];
    if (src.type < MIPT_LIGHT_DIRECTIONAL || src.type > MIPT_LIGHT_RECT_AREA) return fail(MIPT_ERR_INVALID, "unknown light type");
    LightRec L;
    std::memset(&L, 0, sizeof(L));
    L.type = src.type;
    for (int a = 0; a < 3; a++) {
      L.color[a] = src.color[a];
      L.position[a] = src.position[a];
      L.u[a] = src.u[a];
      L.v[a] = src.v[a];
    }
    if (src.type == MIPT_LIGHT_DIRECTIONAL) {
      D3 o = unit(D3{src.direction[0], src.direction[1], src.direction[2]});  // light.h:46
      L.orientation[0] = o.x; L.orientation[1] = o.y; L.orientation[2] = o.z;
    }
    L.uL = src.uL;
    L.vL = src.vL;
    for (int a = 0; a < 3; a++)  // light.h:124: uL/2*uVec + vL/2*vVec + corner
      L.center[a] = (src.uL / 2 * src.u[a] + src.vL / 2 * src.v[a]) + src.position[a];
    L.a = (double)src.atten[0];
    L.b = (double)src.atten[1];
    L.c = (double)src.atten[2];
    // The per-sample attenuation of an area light, min(1, 1 / (a + b d + c d^2)) (light.cpp:139-142), is the
    // same for every sample when b = c = 0: the products are zeros at any finite distance d, evaluated here
    // with the expression the kernels use.  Point and directional shadow rays carry no such factor.
    L.flat = 1;
    L.flat_scale = 1.0;
    if (src.type == MIPT_LIGHT_RECT_AREA) {
      const double d1 = 1.0;
      L.flat = (L.b == 0.0 && L.c == 0.0) ? 1 : 0;
      L.flat_scale = std::fmin(1.0, 1 / (L.a + L.b * d1 + L.c * (d1 * d1)));
    }
    if (!L.flat) S.flat_lights = false;
    S.lights.push_back(L);
  }

  // ---- objects: transforms and world boxes (scene.cpp:54-92) ---------------
  size_t nObj = (size_t)d->n_objects;
  std::vector<double> objBoxes(6 * nObj);
  S.objects.resize(nObj);
  for (size_t k = 0; k < nObj; k++) {
    const mipt_object &src = d->objects[k];
    if (src.type < MIPT_OBJ_SPHERE || src.type > MIPT_OBJ_TRIMESH) return fail(MIPT_ERR_INVALID, "unknown object type");
    if (src.material < 0 || src.material >= d->n_materials) return fail(MIPT_ERR_INVALID, "object material index out of range");
    if (src.type == MIPT_OBJ_TRIMESH && (src.mesh < 0 || src.mesh >= d->n_meshes)) return fail(MIPT_ERR_INVALID, "object mesh index out of range");
    ObjRec &o = S.objects[k];
    std::memset(&o, 0, sizeof(o));
    for (int e = 0; e < 16; e++)
      if (!std::isfinite(src.xform[e])) return fail(MIPT_ERR_INVALID, "object transform is not finite");
    double inv[16];
    if (!invert4(src.xform, inv)) return fail(MIPT_ERR_INVALID, "singular object transform");
    for (int e = 0; e < 16; e++)
      if (!std::isfinite(inv[e])) return fail(MIPT_ERR_INVALID, "singular object transform");
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 4; c++) o.inv[4 * r + c] = M(inv, r, c);
    if (!normal_matrix(src.xform, o.normi)) return fail(MIPT_ERR_INVALID, "singular object transform");
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) o.fwdt[3 * r + c] = M(src.xform, c, r);
    if (src.type == MIPT_OBJ_TRIMESH) {
      // the reference's two face thresholds live in the mesh's local units (trimesh.cpp:85-86,
      // 107-109); they coincide with their world-space forms unless the inverse transform
      // stretches a direction (local t = world t * |inv3 d| can pass t_max first) or is
      // anisotropic (the angle between ray and face differs between the spaces)
      const double inv3[9] = {o.inv[0], o.inv[1], o.inv[2], o.inv[4], o.inv[5], o.inv[6], o.inv[8], o.inv[9], o.inv[10]};
      double smax, smin;
      singular_range3(inv3, &smax, &smin);
      if (smax > 1.0 + 1e-9 || smax > smin * (1.0 + 1e-9)) {
        o.local_units = 1;
        S.local_meshes = true;
      }
    }
    o.kind = src.type;
    o.material = src.material;
    D3 mn{0, 0, 0}, mx{0, 0, 0};
    bool have = true;
    if (src.type == MIPT_OBJ_SPHERE) { mn = {-1, -1, -1}; mx = {1, 1, 1}; }            // Sphere.h:13-18
    else if (src.type == MIPT_OBJ_BOX) { mn = {-0.5, -0.5, -0.5}; mx = {0.5, 0.5, 0.5}; }  // Box.h:13-18
    else if (src.type == MIPT_OBJ_SQUARE) { mn = {-0.5, -0.5, -1e-8}; mx = {0.5, 0.5, 1e-8}; }  // Square.h:13-18
    else {
      const mipt_mesh &m = d->meshes[src.mesh];
      if (m.n_vertices < 0 || m.n_faces < 0 || (m.n_vertices && !m.vertices) || (m.n_faces && !m.faces))
        return fail(MIPT_ERR_INVALID, "bad mesh arrays");
      o.smooth = m.normals ? 1 : 0;
      o.has_uv = m.uvs ? 1 : 0;
      o.has_colors = (m.colors && !m.uvs) ? 1 : 0;
      if (m.n_vertices == 0) have = false;
      else {
        mn = mx = D3{m.vertices[0], m.vertices[1], m.vertices[2]};  // trimesh.h:75-90
        for (int64_t v = 0; v < m.n_vertices; v++) {
          D3 p{m.vertices[3 * v], m.vertices[3 * v + 1], m.vertices[3 * v + 2]};
          if (!(std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z))) return fail(MIPT_ERR_INVALID, "mesh vertex is not finite");
          mn = {std::min(mn.x, p.x), std::min(mn.y, p.y), std::min(mn.z, p.z)};
          mx = {std::max(mx.x, p.x), std::max(mx.y, p.y), std::max(mx.z, p.z)};
        }
      }
    }
    Box6 wb;
    wb.reset();
    if (have) {
      for (int corner = 0; corner < 8; corner++) {
        D3 p{(corner & 1) ? mx.x : mn.x, (corner & 2) ? mx.y : mn.y, (corner & 4) ? mx.z : mn.z};
        wb.grow(apply_point(src.xform, p));
      }
    } else {
      // an empty mesh has an empty box in the reference; it can never be hit
      for (int a = 0; a < 3; a++) { wb.mn[a] = 0; wb.mx[a] = 0; }
    }
    // the builder's binning and the 8-bit box quantisation need coordinates they can do arithmetic on
    // (a NaN vertex makes every comparison false; products of coordinates beyond 1e100 overflow)
    for (int a = 0; a < 3; a++)
      if (!(std::fabs(wb.mn[a]) <= 1e100 && std::fabs(wb.mx[a]) <= 1e100))
        return fail(MIPT_ERR_INVALID, "object coordinates are not finite (or beyond 1e100)");
    for (int a = 0; a < 3; a++) { objBoxes[6 * k + a] = wb.mn[a]; objBoxes[6 * k + 3 + a] = wb.mx[a]; }
  }

  // ---- reference visiting order: objects, then faces inside each mesh --------
  std::vector<uint32_t> objOrder;
  reference_visit_order(objBoxes, &objOrder);

  std::vector<BuildPrim> prims;
  std::vector<Tri> trisRef;          // by reference-order index
  std::vector<TriNormals> normsRef;  // by reference-order index (if any smooth mesh)
  std::vector<uint32_t> infoRef;
  std::vector<TriUV> uvsRef;         // by reference-order index (if any mesh has UV coordinates)
  std::vector<TriColor> colorsRef;   // by reference-order index (if any mesh shades with vertex colours)
  bool anySmooth = false, anyUV = false, anyColor = false;
  for (size_t k = 0; k < nObj; k++) {
    if (S.objects[k].smooth) anySmooth = true;
    if (S.objects[k].has_uv) anyUV = true;
    if (S.objects[k].has_colors) anyColor = true;
  }

  for (size_t ord = 0; ord < nObj; ord++) {
    uint32_t oi = objOrder[ord];
    const mipt_object &src = d->objects[oi];
    const ObjRec &o = S.objects[oi];
    if (src.type != MIPT_OBJ_TRIMESH) {
      BuildPrim bp;
      for (int a = 0; a < 3; a++) {
        bp.box.mn[a] = objBoxes[6 * (size_t)oi + a];
        bp.box.mx[a] = objBoxes[6 * (size_t)oi + 3 + a];
        bp.c[a] = 0.5 * (bp.box.mn[a] + bp.box.mx[a]);
      }
      bp.id = (uint32_t)prims.size();
      prims.push_back(bp);
      Tri t;
      std::memset(&t, 0, sizeof(t));
      trisRef.push_back(t);
      if (anySmooth) { TriNormals tn; std::memset(&tn, 0, sizeof(tn)); normsRef.push_back(tn); }
      if (anyUV) uvsRef.push_back(TriUV{0, 0, 0, 0, 0, 0});
      if (anyColor) { TriColor tc; std::memset(&tc, 0, sizeof(tc)); colorsRef.push_back(tc); }
      infoRef.push_back(make_prim_info((uint32_t)src.type, oi));
      continue;
    }
    const mipt_mesh &m = d->meshes[src.mesh];
    // non-degenerate faces in addFace order (trimesh.cpp:32-46, trimesh.h:139-141)
    std::vector<int64_t> faces;
    std::vector<double> fboxes;
    faces.reserve((size_t)m.n_faces);
    for (int64_t f = 0; f < m.n_faces; f++) {
      int ia = m.faces[3 * f], ib = m.faces[3 * f + 1], ic = m.faces[3 * f + 2];
      if (ia < 0 || ib < 0 || ic < 0 || ia >= m.n_vertices || ib >= m.n_vertices || ic >= m.n_vertices)
        return fail(MIPT_ERR_INVALID, "mesh face index out of range");
      D3 A{m.vertices[3 * ia], m.vertices[3 * ia + 1], m.vertices[3 * ia + 2]};
      D3 B{m.vertices[3 * ib], m.vertices[3 * ib + 1], m.vertices[3 * ib + 2]};
      D3 Cc{m.vertices[3 * ic], m.vertices[3 * ic + 1], m.vertices[3 * ic + 2]};
      if (len(B - A) == 0.0 || len(Cc - A) == 0.0 || len(B - Cc) == 0.0) continue;
      faces.push_back(f);
      // local face box, trimesh.h:175-188
      fboxes.push_back(std::min(Cc.x, std::min(A.x, B.x)));
      fboxes.push_back(std::min(Cc.y, std::min(A.y, B.y)));
      fboxes.push_back(std::min(Cc.z, std::min(A.z, B.z)));
      fboxes.push_back(std::max(Cc.x, std::max(A.x, B.x)));
      fboxes.push_back(std::max(Cc.y, std::max(A.y, B.y)));
      fboxes.push_back(std::max(Cc.z, std::max(A.z, B.z)));
    }
    std::vector<uint32_t> fOrder;
    reference_visit_order(fboxes, &fOrder);
    if (prims.size() + faces.size() >= ((size_t)1 << 28)) return fail(MIPT_ERR_LIMIT, "too many primitives (limit 2^28-1)");
    // every face fills its own slots: slices of the faces go to threads
    const size_t base = prims.size(), nf = fOrder.size();
    prims.resize(base + nf);
    trisRef.resize(base + nf);
    if (anySmooth) normsRef.resize(base + nf);
    if (anyUV) uvsRef.resize(base + nf);
    if (anyColor) colorsRef.resize(base + nf);
    infoRef.resize(base + nf);
    auto do_faces = [&](size_t q0, size_t q1) {
    for (size_t q = q0; q < q1; q++) {
      int64_t f = faces[fOrder[q]];
      int ia = m.faces[3 * f], ib = m.faces[3 * f + 1], ic = m.faces[3 * f + 2];
      D3 A{m.vertices[3 * ia], m.vertices[3 * ia + 1], m.vertices[3 * ia + 2]};
      D3 B{m.vertices[3 * ib], m.vertices[3 * ib + 1], m.vertices[3 * ib + 2]};
      D3 Cc{m.vertices[3 * ic], m.vertices[3 * ic + 1], m.vertices[3 * ic + 2]};
      D3 nLocal = unit(cross(B - A, Cc - A));           // trimesh.h:145-146
      D3 nWorld = unit(apply3(o.normi, nLocal));        // scene.h:92-94
      D3 a = apply_point(src.xform, A), b = apply_point(src.xform, B), c = apply_point(src.xform, Cc);
      D3 e1 = b - a, e2 = c - a;
      double g11 = dot(e1, e1), g12 = dot(e1, e2), g22 = dot(e2, e2);
      double ood = 1.0 / (g11 * g22 - g12 * g12);       // trimesh.h:152-157 in world space
      double A00 = g22 * ood, A01 = -g12 * ood, A11 = g11 * ood;
      D3 q1 = e1 * A00 + e2 * A01;
      D3 q2 = e1 * A01 + e2 * A11;
      Tri t;
      t.a[0] = a.x; t.a[1] = a.y; t.a[2] = a.z;
      t.n[0] = nWorld.x; t.n[1] = nWorld.y; t.n[2] = nWorld.z;
      t.q1[0] = q1.x; t.q1[1] = q1.y; t.q1[2] = q1.z;
      t.q2[0] = q2.x; t.q2[1] = q2.y; t.q2[2] = q2.z;
      BuildPrim bp;
      bp.box.reset();
      bp.box.grow(a); bp.box.grow(b); bp.box.grow(c);
      for (int ax = 0; ax < 3; ax++) bp.c[ax] = 0.5 * (bp.box.mn[ax] + bp.box.mx[ax]);
      bp.id = (uint32_t)(base + q);
      prims[base + q] = bp;
      trisRef[base + q] = t;
      if (anySmooth) {
        TriNormals tn;
        std::memset(&tn, 0, sizeof(tn));
        if (m.normals) {
          D3 na = apply3(o.normi, D3{m.normals[3 * ia], m.normals[3 * ia + 1], m.normals[3 * ia + 2]});
          D3 nb = apply3(o.normi, D3{m.normals[3 * ib], m.normals[3 * ib + 1], m.normals[3 * ib + 2]});
          D3 nc = apply3(o.normi, D3{m.normals[3 * ic], m.normals[3 * ic + 1], m.normals[3 * ic + 2]});
          tn.na[0] = na.x; tn.na[1] = na.y; tn.na[2] = na.z;
          tn.nb[0] = nb.x; tn.nb[1] = nb.y; tn.nb[2] = nb.z;
          tn.nc[0] = nc.x; tn.nc[1] = nc.y; tn.nc[2] = nc.z;
        }
        normsRef[base + q] = tn;
      }
      if (anyUV) {
        TriUV tu{0, 0, 0, 0, 0, 0};
        if (m.uvs) tu = TriUV{m.uvs[2 * ia], m.uvs[2 * ia + 1], m.uvs[2 * ib], m.uvs[2 * ib + 1], m.uvs[2 * ic], m.uvs[2 * ic + 1]};
        uvsRef[base + q] = tu;
      }
      if (anyColor) {
        TriColor tc;
        std::memset(&tc, 0, sizeof(tc));
        if (o.has_colors)
          for (int ch = 0; ch < 3; ch++) {
            tc.ca[ch] = m.colors[3 * ia + ch];
            tc.cb[ch] = m.colors[3 * ib + ch];
            tc.cc[ch] = m.colors[3 * ic + ch];
          }
        colorsRef[base + q] = tc;
      }
      infoRef[base + q] = make_prim_info(KIND_TRI, oi);
    }
    };
    {
      const size_t slices = nf >= ((size_t)1 << 16) ? 16 : 1;
      std::vector<std::future<void>> parts;
      for (size_t k = 1; k < slices; k++) parts.push_back(std::async(std::launch::async, do_faces, nf * k / slices, nf * (k + 1) / slices));
      do_faces(0, nf / slices);
      for (auto &f : parts) f.get();
    }
  }

  timer.lap("primitives, reference order");
  // ---- BVH over all primitives ---------------------------------------------------
  size_t nPrims = prims.size();
  Box6 rootBox;
  rootBox.reset();
  if (!build_tree) {
    // flat mode: arrays stay in reference order, the tree is built on the GPU
    S.tris = std::move(trisRef);
    S.prim_info = std::move(infoRef);
    if (anySmooth) S.tri_normals = std::move(normsRef);
    if (anyUV) S.tri_uvs = std::move(uvsRef);
    if (anyColor) S.tri_colors = std::move(colorsRef);
    S.ref_order.resize(nPrims);
    S.prim_boxes.resize(6 * nPrims);
    for (size_t s = 0; s < nPrims; s++) {
      S.ref_order[s] = (uint32_t)s;
      bool empty = prims[s].box.mx[0] < prims[s].box.mn[0];
      for (int a = 0; a < 3; a++) {
        S.prim_boxes[6 * s + a] = empty ? 0.0f : round_down(prims[s].box.mn[a]);
        S.prim_boxes[6 * s + 3 + a] = empty ? 0.0f : round_up(prims[s].box.mx[a]);
      }
      if (!empty) rootBox.grow(prims[s].box);
      if ((S.prim_info[s] >> 30) != KIND_TRI) S.n_analytic++;
    }
    timer.lap("flat arrays");
  } else {
  size_t nAnalytic = (size_t)(std::stable_partition(prims.begin(), prims.end(), [&](const BuildPrim &p) { return (infoRef[p.id] >> 30) != KIND_TRI; }) -
                              prims.begin());
  S.n_analytic = (int)nAnalytic;
  Builder builder(prims);
  if (const char *lm = debug_env("MIPT_LEAF_MAX")) {  // tuning knob: preferred primitives per leaf
    int v = std::atoi(lm);
    if (v >= 1 && v <= kLeafHard) builder.leaf_max = v;
  }
  builder.nodes.reserve(nPrims / 2 + 16);
  builder.forks = 6;  // up to 64 tasks: SAH splits are uneven, many small tasks balance the threads
  Collapser wide(builder.nodes);
  timer.lap("partition by kind");
  int need = 0;
  if (nPrims == 0) {
    Node4 nd;
    quantise(nd, nullptr, nullptr, 0);
    wide.out.push_back(nd);
    wide.max_depth = 1;
  } else {
    // analytic primitives and triangles get subtrees of their own, joined at the root: the
    // kernels then know a primitive's kind from its storage index alone
    int32_t root;
    if (nAnalytic == 0 || nAnalytic == nPrims) {
      root = builder.build(0, nPrims, 0, &rootBox);
    } else {
      Box6 ba, bt;
      int32_t me = (int32_t)builder.nodes.size();
      builder.nodes.emplace_back();
      builder.max_depth = 1;
      int32_t ra = builder.build(0, nAnalytic, 1, &ba);
      int32_t rt = builder.build(nAnalytic, nPrims, 1, &bt);
      BinNode &nd = builder.nodes[(size_t)me];
      nd.box[0] = ba; nd.box[1] = bt;
      nd.child[0] = ra; nd.child[1] = rt;
      rootBox = ba;
      rootBox.grow(bt);
      root = me;
    }
    timer.lap("binary SAH build");
    if (root < 0) {
      // the whole scene fits one leaf: a root node with that single child
      Node4 nd;
      quantise(nd, &rootBox, &root, 1);
      wide.out.push_back(nd);
      wide.max_depth = 1;
    } else {
      wide.out.reserve(builder.nodes.size() / 2 + 16);
      wide.emit(root, 0, &need);
    }
  }
  timer.lap("collapse + quantise");
  S.nodes = std::move(wide.out);
  S.max_depth = wide.max_depth;
  S.stack_need = need + 1;
  S.tris.resize(nPrims);
  S.prim_info.resize(nPrims);
  S.ref_order.resize(nPrims);
  if (anySmooth) S.tri_normals.resize(nPrims);
  if (anyUV) S.tri_uvs.resize(nPrims);
  if (anyColor) S.tri_colors.resize(nPrims);
  for (size_t s = 0; s < nPrims; s++) {
    uint32_t id = prims[s].id;
    S.tris[s] = trisRef[id];
    S.prim_info[s] = infoRef[id];
    S.ref_order[s] = id;
    if (anySmooth) S.tri_normals[s] = normsRef[id];
    if (anyUV) S.tri_uvs[s] = uvsRef[id];
    if (anyColor) S.tri_colors[s] = colorsRef[id];
  }
  timer.lap("leaf-order arrays");
  }
  for (int a = 0; a < 3; a++) {
    S.bounds_min[a] = nPrims ? rootBox.mn[a] : 0.0;
    S.bounds_max[a] = nPrims ? rootBox.mx[a] : 0.0;
  }
  double R = 0;
  if (nPrims)
    for (int a = 0; a < 3; a++) R = std::max(R, std::max(std::fabs(rootBox.mn[a]), std::fabs(rootBox.mx[a])));
  S.radius = R;
  // world boxes of the objects, a little too large: a ray that an analytic primitive's test accepts passes
  // through its box (the shadow kernel tests the few analytic primitives of a mesh scene ray by ray, box first)
  S.object_boxes.resize(6 * nObj);
  {
    const double grow = 1e-9 * (R + 1.0);
    for (size_t k = 0; k < nObj; k++)
      for (int a = 0; a < 3; a++) {
        S.object_boxes[6 * k + a] = objBoxes[6 * k + a] - grow;
        S.object_boxes[6 * k + 3 + a] = objBoxes[6 * k + 3 + a] + grow;
      }
  }

  if (d->cubemap) {
    S.has_cubemap = true;
    for (int f = 0; f < 6; f++) {
      const mipt_image &im = d->cubemap[f];
      if (im.width <= 0 || im.height <= 0 || !im.rgb) return fail(MIPT_ERR_INVALID, "bad cubemap face");
      S.cube_w[f] = im.width;
      S.cube_h[f] = im.height;
      S.cube_rgb[f].assign(im.rgb, im.rgb + (size_t)3 * im.width * im.height);
    }
  }
  if (d->n_textures < 0 || (d->n_textures && !d->textures)) return fail(MIPT_ERR_INVALID, "bad texture array");
  for (int k = 0; k < d->n_textures; k++) {
    const mipt_image &im = d->textures[k];
    if (im.width <= 0 || im.height <= 0 || !im.rgb) return fail(MIPT_ERR_INVALID, "bad texture image");
    S.tex_w.push_back(im.width);
    S.tex_h.push_back(im.height);
    S.tex_rgb.emplace_back(im.rgb, im.rgb + (size_t)3 * im.width * im.height);
  }
  return MIPT_OK;
}

}  // namespace mipt
