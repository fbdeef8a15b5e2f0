// gpu_build.hip — linear BVH on the GPU: Morton codes, radix sort, the parallel radix-tree
// construction of Karras (one thread per inner node, no atomics), a bottom-up pass for the
// boxes, and a level-by-level collapse into four-wide nodes with 8-bit quantised child boxes
// (the format of layout.h; scene_build.cpp holds the host builder it mirrors).
#include "gpu_build.h"

#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

namespace mipt {

namespace {

constexpr int kB = 256;
static inline unsigned blocks(uint64_t n) { return (unsigned)((n + kB - 1) / kB); }

struct F6 { float mn[3], mx[3]; };

__device__ __forceinline__ uint64_t spread21(uint64_t v) {  // 21 bits -> every third bit
  v &= 0x1fffffull;
  v = (v | v << 32) & 0x1f00000000ffffull;
  v = (v | v << 16) & 0x1f0000ff0000ffull;
  v = (v | v << 8) & 0x100f00f00f00f00full;
  v = (v | v << 4) & 0x10c30c30c30c30c3ull;
  v = (v | v << 2) & 0x1249249249249249ull;
  return v;
}

// key = is_triangle << 63 | 63-bit Morton code of the box centre: analytic primitives sort
// first, so the root separates them from the triangles and their storage indices are [0, n_analytic)
__global__ void k_keys(const F6 *boxes, const uint32_t *info, uint32_t n, double3 lo, double3 inv_extent, uint64_t *keys, uint32_t *ids,
                       uint32_t *n_analytic) {
  uint32_t i = blockIdx.x * kB + threadIdx.x;
  if (i >= n) return;
  F6 b = boxes[i];
  double c[3] = {0.5 * ((double)b.mn[0] + b.mx[0]), 0.5 * ((double)b.mn[1] + b.mx[1]), 0.5 * ((double)b.mn[2] + b.mx[2])};
  double l[3] = {lo.x, lo.y, lo.z}, s[3] = {inv_extent.x, inv_extent.y, inv_extent.z};
  uint64_t q[3];
  for (int a = 0; a < 3; a++) {
    double f = (c[a] - l[a]) * s[a] * 2097152.0;
    if (!(f > 0.0)) f = 0.0;
    if (f > 2097151.0) f = 2097151.0;
    q[a] = (uint64_t)f;
  }
  bool tri = (info[i] >> 30) == KIND_TRI;
  keys[i] = ((uint64_t)(tri ? 1 : 0) << 63) | (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
  ids[i] = i;
  if (!tri) atomicAdd(n_analytic, 1u);
}

// length of the common prefix of keys i and j, ties broken by the index (Karras 2012)
__device__ __forceinline__ int delta(const uint64_t *keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  uint64_t a = keys[i], b = keys[j];
  if (a == b) return 64 + __clz((unsigned)(i ^ j));
  return __clzll((long long)(a ^ b));
}

// inner node i of the radix tree over the sorted keys: children (>= 0 inner node, < 0 = ~leaf
// position), the range of sorted positions it covers, parent links
__global__ void k_radix_tree(const uint64_t *keys, int n, int *left, int *right, int *first, int *last, int *parent_inner, int *parent_leaf) {
  int i = blockIdx.x * kB + threadIdx.x;
  if (i >= n - 1) return;
  int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  int dmin = delta(keys, n, i, i - d);
  int lmax = 2;
  while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
  int l = 0;
  for (int t = lmax / 2; t >= 1; t /= 2)
    if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  int j = i + l * d;
  int dnode = delta(keys, n, i, j);
  int s = 0, t = l;
  do {
    t = (t + 1) / 2;
    if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
  } while (t > 1);
  int g = i + s * d + (d < 0 ? d : 0);
  int lo = i < j ? i : j, hi = i < j ? j : i;
  int L, R;
  if (lo == g) { L = ~g; parent_leaf[g] = i; } else { L = g; parent_inner[g] = i; }
  if (hi == g + 1) { R = ~(g + 1); parent_leaf[g + 1] = i; } else { R = g + 1; parent_inner[g + 1] = i; }
  left[i] = L;
  right[i] = R;
  first[i] = lo;
  last[i] = hi;
  if (i == 0) parent_inner[0] = -1;
}

__device__ __forceinline__ F6 merge(F6 a, F6 b) {
  F6 r;
  for (int k = 0; k < 3; k++) { r.mn[k] = fminf(a.mn[k], b.mn[k]); r.mx[k] = fmaxf(a.mx[k], b.mx[k]); }
  return r;
}

// boxes of the inner nodes, bottom-up: the second thread to arrive at a node computes it
__global__ void k_refit(const F6 *boxes, const uint32_t *ids, int n, const int *left, const int *right, const int *parent_inner,
                        const int *parent_leaf, unsigned *arrived, F6 *inner_box) {
  int p = blockIdx.x * kB + threadIdx.x;
  if (p >= n) return;
  int node = parent_leaf[p];
  while (node >= 0) {
    // release what this thread wrote below `node`, take a ticket, and — as the second to
    // arrive — acquire what the first one wrote (agent scope: the other thread may sit on
    // another CU or XCD)
    __threadfence();
    if (atomicAdd(&arrived[node], 1u) == 0u) return;
    __threadfence();
    int L = left[node], R = right[node];
    F6 a = L < 0 ? boxes[ids[~L]] : inner_box[L];
    F6 b = R < 0 ? boxes[ids[~R]] : inner_box[R];
    inner_box[node] = merge(a, b);
    node = parent_inner[node];
  }
}

__device__ __forceinline__ float area(const F6 &b) {
  float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
  return 2.0f * (dx * dy + dy * dz + dz * dx);
}

// Quantises up to four child boxes into the frame of one Node4 (the same rules as `quantise`
// in scene_build.cpp: origin and lower planes rounded down, upper planes up).
__device__ void quantise_node(Node4 &nd, const F6 *boxes, const int32_t *refs, int n) {
  nd.n_children = (uint8_t)n;
  nd.pad[0] = nd.pad[1] = 0;
  for (int k = 0; k < 4; k++) nd.child[k] = k < n ? refs[k] : kNoChild;
  for (int a = 0; a < 3; a++) {
    float lo = boxes[0].mn[a], hi = boxes[0].mx[a];
    for (int k = 1; k < n; k++) { lo = fminf(lo, boxes[k].mn[a]); hi = fmaxf(hi, boxes[k].mx[a]); }
    float O = lo;  // already a float: no rounding needed
    double extent = (double)hi - (double)O;
    int e = 127 - 60;
    if (extent > 0) {
      int ex;
      frexp(extent / 255.0, &ex);
      e = e > 127 + ex ? e : 127 + ex;
    }
    for (;; e++) {
      double step = ldexp(1.0, e - 127);
      bool fits = true;
      uint32_t lo_w = 0, hi_w = 0;
      for (int k = 0; k < n; k++) {
        double mn = boxes[k].mn[a], mx = boxes[k].mx[a];
        double ql = floor((mn - (double)O) / step), qh = ceil((mx - (double)O) / step);
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

struct Front { int node2; int slot; };

// One level of the collapse: every frontier entry (an inner node of the binary tree and the
// slot of the four-wide node that replaces it) opens its largest inner child until it has four
// children, writes its Node4 and queues its inner children for the next level.  A binary node
// that covers at most leaf_max primitives becomes a leaf (its range is contiguous).
__global__ void k_collapse(const Front *in, uint32_t n_in, Front *out, unsigned *n_out, unsigned *n_nodes, Node4 *nodes4, const int *left,
                           const int *right, const int *first, const int *last, const F6 *inner_box, const F6 *boxes, const uint32_t *ids,
                           int leaf_max) {
  uint32_t t = blockIdx.x * kB + threadIdx.x;
  if (t >= n_in) return;
  Front f = in[t];
  int32_t refs[4];
  F6 bx[4];
  auto box_of = [&](int ref) { return ref < 0 ? boxes[ids[~ref]] : inner_box[ref]; };
  auto is_leaf = [&](int ref) { return ref < 0 || last[ref] - first[ref] + 1 <= leaf_max; };
  int n = 2;
  refs[0] = left[f.node2]; refs[1] = right[f.node2];
  bx[0] = box_of(refs[0]); bx[1] = box_of(refs[1]);
  while (n < 4) {
    int best = -1;
    float bestA = -1.f;
    for (int k = 0; k < n; k++)
      if (!is_leaf(refs[k]) && area(bx[k]) > bestA) { bestA = area(bx[k]); best = k; }
    if (best < 0) break;
    int r = refs[best];
    refs[best] = left[r]; bx[best] = box_of(left[r]);
    refs[n] = right[r]; bx[n] = box_of(right[r]);
    n++;
  }
  for (int i = 1; i < n; i++)  // larger boxes first, as the host builder orders them
    for (int j = i; j > 0 && area(bx[j]) > area(bx[j - 1]); j--) {
      F6 tb = bx[j]; bx[j] = bx[j - 1]; bx[j - 1] = tb;
      int tr = refs[j]; refs[j] = refs[j - 1]; refs[j - 1] = tr;
    }
  int inner = 0;
  for (int k = 0; k < n; k++) inner += is_leaf(refs[k]) ? 0 : 1;
  unsigned slot = inner ? atomicAdd(n_nodes, (unsigned)inner) : 0u;  // siblings get adjacent slots
  unsigned fo = inner ? atomicAdd(n_out, (unsigned)inner) : 0u;
  int32_t child[4];
  for (int k = 0; k < n; k++) {
    int r = refs[k];
    if (r < 0) child[k] = ~(int32_t)(((uint32_t)(~r) << 3) | 0u);
    else if (is_leaf(r)) child[k] = ~(int32_t)(((uint32_t)first[r] << 3) | (uint32_t)(last[r] - first[r]));
    else {
      out[fo++] = Front{r, (int)slot};
      child[k] = (int32_t)slot++;
    }
  }
  Node4 nd;
  quantise_node(nd, bx, child, n);
  nodes4[f.slot] = nd;
}

__global__ void k_gather(uint32_t *dst, const uint32_t *src, const uint32_t *order, uint32_t n, uint32_t words) {
  uint64_t i = (uint64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= (uint64_t)n * words) return;
  uint32_t s = (uint32_t)(i / words), w = (uint32_t)(i - (uint64_t)s * words);
  dst[i] = src[(uint64_t)order[s] * words + w];
}

}  // namespace

void gpu_gather_records(void *dst, const void *src, const uint32_t *order, uint32_t n, uint32_t bytes, hipStream_t st) {
  if (!n) return;
  uint32_t words = bytes / 4;
  hipLaunchKernelGGL(k_gather, dim3(blocks((uint64_t)n * words)), dim3(kB), 0, st, (uint32_t *)dst, (const uint32_t *)src, order, n, words);
}

hipError_t gpu_build_bvh(const float *d_boxes, const uint32_t *d_prim_info, uint32_t n, const double bmin[3], const double bmax[3], int leaf_max,
                         GpuBvh *out, hipStream_t st) {
  *out = GpuBvh();
  hipError_t e = hipSuccess;
  std::vector<void *> tmp;
  auto alloc = [&](size_t bytes) -> void * {
    void *p = nullptr;
    if (e == hipSuccess) e = hipMalloc(&p, bytes ? bytes : 16);
    if (p) tmp.push_back(p);
    return p;
  };
  auto cleanup = [&]() { for (void *p : tmp) (void)hipFree(p); };
  const F6 *boxes = reinterpret_cast<const F6 *>(d_boxes);
  uint64_t *keys = (uint64_t *)alloc((size_t)n * 8), *keys2 = (uint64_t *)alloc((size_t)n * 8);
  uint32_t *ids = (uint32_t *)alloc((size_t)n * 4);
  unsigned *counters = (unsigned *)alloc(64);  // [0] n_analytic, [1] n_nodes, [2], [3] frontier sizes
  int *left = (int *)alloc((size_t)n * 4), *right = (int *)alloc((size_t)n * 4), *first = (int *)alloc((size_t)n * 4), *last = (int *)alloc((size_t)n * 4);
  int *parent_inner = (int *)alloc((size_t)n * 4), *parent_leaf = (int *)alloc((size_t)n * 4);
  unsigned *arrived = (unsigned *)alloc((size_t)n * 4);
  F6 *inner_box = (F6 *)alloc((size_t)n * sizeof(F6));
  Front *fr[2] = {(Front *)alloc((size_t)n * sizeof(Front)), (Front *)alloc((size_t)n * sizeof(Front))};
  uint32_t *order = nullptr;
  Node4 *nodes4 = nullptr;
  if (e == hipSuccess) e = hipMalloc((void **)&order, (size_t)n * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&nodes4, (size_t)n * sizeof(Node4));
  if (e != hipSuccess) { cleanup(); (void)hipFree(order); (void)hipFree(nodes4); return e; }
  (void)hipMemsetAsync(counters, 0, 64, st);
  (void)hipMemsetAsync(arrived, 0, (size_t)n * 4, st);

  double3 lo{bmin[0], bmin[1], bmin[2]};
  double3 inv{bmax[0] > bmin[0] ? 1.0 / (bmax[0] - bmin[0]) : 0.0, bmax[1] > bmin[1] ? 1.0 / (bmax[1] - bmin[1]) : 0.0,
              bmax[2] > bmin[2] ? 1.0 / (bmax[2] - bmin[2]) : 0.0};
  hipLaunchKernelGGL(k_keys, dim3(blocks(n)), dim3(kB), 0, st, boxes, d_prim_info, n, lo, inv, keys, ids, counters);
  size_t sort_bytes = 0;
  e = rocprim::radix_sort_pairs(nullptr, sort_bytes, keys, keys2, ids, order, n, 0, 64, st);
  void *sort_tmp = alloc(sort_bytes);
  if (e == hipSuccess) e = rocprim::radix_sort_pairs(sort_tmp, sort_bytes, keys, keys2, ids, order, n, 0, 64, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_radix_tree, dim3(blocks(n - 1)), dim3(kB), 0, st, keys2, (int)n, left, right, first, last, parent_inner, parent_leaf);
    hipLaunchKernelGGL(k_refit, dim3(blocks(n)), dim3(kB), 0, st, boxes, order, (int)n, left, right, parent_inner, parent_leaf, arrived, inner_box);
  }
  // collapse, level by level from the root
  unsigned h_counts[4] = {0, 1, 1, 0};  // node 0 = the root, one frontier entry
  Front root{0, 0};
  if (e == hipSuccess) e = hipMemcpyAsync(fr[0], &root, sizeof(root), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(counters + 1, h_counts + 1, 4, hipMemcpyHostToDevice, st);
  uint32_t n_in = 1;
  int levels = 0, cur = 0;
  while (e == hipSuccess && n_in > 0) {
    (void)hipMemsetAsync(counters + 2, 0, 4, st);
    hipLaunchKernelGGL(k_collapse, dim3(blocks(n_in)), dim3(kB), 0, st, fr[cur], n_in, fr[cur ^ 1], counters + 2, counters + 1, nodes4, left, right, first,
                       last, inner_box, boxes, order, leaf_max);
    e = hipMemcpyAsync(h_counts, counters, 16, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    n_in = h_counts[2];
    cur ^= 1;
    levels++;
  }
  if (e == hipSuccess) e = hipGetLastError();
  cleanup();
  if (e != hipSuccess) { (void)hipFree(order); (void)hipFree(nodes4); return e; }
  out->nodes = nodes4;
  out->order = order;
  out->n_nodes = (int)h_counts[1];
  out->levels = levels;
  out->n_analytic = (int)h_counts[0];
  return hipSuccess;
}

}  // namespace mipt
