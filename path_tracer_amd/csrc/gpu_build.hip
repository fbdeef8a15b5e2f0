// gpu_build.hip — BVH construction on the GPU: Morton codes, radix sort, a binary tree by parallel
// locally-ordered clustering (nearest-neighbour merging in Morton order), and a level-by-level
// collapse into four-wide nodes with 8-bit quantised child boxes (the format of layout.h;
// scene_build.cpp holds the host builder it mirrors).
#include "gpu_build.h"

#include <algorithm>
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

__device__ __forceinline__ F6 merge(F6 a, F6 b) {
  F6 r;
  for (int k = 0; k < 3; k++) { r.mn[k] = fminf(a.mn[k], b.mn[k]); r.mx[k] = fmaxf(a.mx[k], b.mx[k]); }
  return r;
}

__device__ __forceinline__ float area(const F6 &b) {
  float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
  return 2.0f * (dx * dy + dy * dz + dz * dx);
}

// ---- binary tree by parallel locally-ordered clustering (PLOC, Meister & Bittner 2018) ----
// The clusters start as the primitives in Morton order.  In every round each cluster looks at the
// kRadius clusters either side of it in the array and picks the one it forms the smallest box with;
// two clusters that pick each other are merged into a node that takes the place of the first, and
// the array is compacted.  Merging nearest neighbours bottom-up gives trees close to a top-down
// surface-area build (a radix tree over the Morton codes splits where the codes do, not where the
// geometry does: 2.7 times the node visits of the host builder on the benchmark scene).
// Node ids: [0, n) the primitives in sorted order, [n, 2n - 1) merged nodes in the order they
// were made.  Analytic primitives and triangles never merge with one another before each kind is
// a single cluster, so each kind ends up in a subtree of its own under the root.
#ifndef MIPT_PLOC_RADIUS
#define MIPT_PLOC_RADIUS 64
#endif
constexpr int kRadius = MIPT_PLOC_RADIUS;

__global__ void __launch_bounds__(kB) k_nearest(const int *cl, uint32_t m, const F6 *nbox, const uint8_t *node_kind, int *nn) {
  __shared__ F6 tile[kB + 2 * kRadius];
  __shared__ int tkind[kB + 2 * kRadius];  // -1: no cluster at that position
  const int base = (int)(blockIdx.x * kB) - kRadius;
  for (int t = threadIdx.x; t < kB + 2 * kRadius; t += kB) {
    int g = base + t;
    if (g >= 0 && g < (int)m) {
      int node = cl[g];
      tkind[t] = (int)node_kind[node];
      tile[t] = nbox[node];
    } else {
      tkind[t] = -1;
    }
  }
  __syncthreads();
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i >= (int)m) return;
  const int me = threadIdx.x + kRadius;
  const F6 b = tile[me];
  const int kind = tkind[me];
  // the partner with the smallest joint box among the clusters of the same kind in the window; of equal
  // ones the first.  (The pair with the smallest joint box of all picks each other, so every round
  // merges something as long as some kind has two clusters: they are neighbours in the array.)
  float best = 3.0e38f;
  int best_j = -1;
  for (int d = -kRadius; d <= kRadius; d++) {
    if (d == 0 || tkind[me + d] != kind) continue;
    float a = area(merge(b, tile[me + d]));
    if (a < best) { best = a; best_j = i + d; }
  }
  nn[i] = best_j;
}

// Two clusters that picked each other merge: the first keeps its place as the new node, the
// second leaves the array.
__global__ void __launch_bounds__(kB) k_pairs(const int *nn, uint32_t m, unsigned long long *flags) {
  const uint32_t i = blockIdx.x * kB + threadIdx.x;
  if (i >= m) return;
  const int j = nn[i];
  bool mutual = j >= 0 && nn[j] == (int)i;
  // low word: this position keeps a cluster; high word: that cluster is a new node
  unsigned long long f = 1ull;
  if (mutual) f = (int)i < j ? (1ull | (1ull << 32)) : 0ull;
  flags[i] = f;
}

__global__ void __launch_bounds__(kB) k_merge(const int *cl, const int *nn, uint32_t m, const unsigned long long *flags, const unsigned long long *scan,
                                              uint32_t next_node, F6 *nbox, int *nleft, int *nright, uint32_t *nsize, uint8_t *node_kind, int *cl_out) {
  const uint32_t i = blockIdx.x * kB + threadIdx.x;
  if (i >= m) return;
  const unsigned long long f = flags[i];
  if (!(f & 1ull)) return;
  const uint32_t pos = (uint32_t)(scan[i] & 0xffffffffull);
  int node = cl[i];
  if (f >> 32) {
    const int other = cl[nn[i]];
    const uint32_t id = next_node + (uint32_t)(scan[i] >> 32);
    nbox[id] = merge(nbox[node], nbox[other]);
    nleft[id] = node;
    nright[id] = other;
    nsize[id] = nsize[node] + nsize[other];
    node_kind[id] = node_kind[node];
    node = (int)id;
  }
  cl_out[pos] = node;
}

__global__ void __launch_bounds__(kB) k_init_clusters(const F6 *boxes, const uint32_t *sorted_ids, const uint64_t *sorted_keys, uint32_t n, F6 *nbox, uint32_t *nsize,
                                                      uint8_t *node_kind, int *cl) {
  const uint32_t i = blockIdx.x * kB + threadIdx.x;
  if (i >= n) return;
  nbox[i] = boxes[sorted_ids[i]];
  nsize[i] = 1u;
  node_kind[i] = (uint8_t)(sorted_keys[i] >> 63);
  cl[i] = (int)i;
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

struct Front { int node2; int slot; uint32_t first; };

// The primitives under binary node `ref`, left to right, into order[first...] (a leaf of the
// four-wide tree: at most eight of them).
__device__ void write_leaf_order(int ref, uint32_t first, uint32_t n_leaves, const int *nleft, const int *nright, const uint32_t *sorted_ids, uint32_t *order) {
  int stack[16];
  int sp = 0;
  stack[sp++] = ref;
  while (sp > 0) {
    int r = stack[--sp];
    if ((uint32_t)r < n_leaves) {
      order[first++] = sorted_ids[r];
    } else {
      stack[sp++] = nright[r];  // popped after the left one
      stack[sp++] = nleft[r];
    }
  }
}

// One level of the collapse: every frontier entry (an inner node of the binary tree, the slot of
// the four-wide node that replaces it, and the position of its first primitive in the final
// order) opens its largest inner child until it has four children, writes its Node4 and queues
// its inner children for the next level.  The final order of the primitives is the binary tree's
// left-to-right order, so every node covers a contiguous range and a binary node that covers at
// most leaf_max primitives becomes a leaf.
__global__ void k_collapse(const Front *in, uint32_t n_in, Front *out, unsigned *n_out, unsigned *n_nodes, Node4 *nodes4, const int *nleft,
                           const int *nright, const uint32_t *nsize, const F6 *nbox, uint32_t n_leaves, const uint32_t *sorted_ids, uint32_t *order,
                           int leaf_max) {
  uint32_t t = blockIdx.x * kB + threadIdx.x;
  if (t >= n_in) return;
  Front f = in[t];
  int32_t refs[4];
  uint32_t firsts[4];
  F6 bx[4];
  auto is_leaf = [&](int ref) { return nsize[ref] <= (uint32_t)leaf_max; };
  int n = 2;
  refs[0] = nleft[f.node2]; refs[1] = nright[f.node2];
  firsts[0] = f.first; firsts[1] = f.first + nsize[refs[0]];
  bx[0] = nbox[refs[0]]; bx[1] = nbox[refs[1]];
  while (n < 4) {
    int best = -1;
    float bestA = -1.f;
    for (int k = 0; k < n; k++)
      if (!is_leaf(refs[k]) && area(bx[k]) > bestA) { bestA = area(bx[k]); best = k; }
    if (best < 0) break;
    int r = refs[best];
    refs[best] = nleft[r]; bx[best] = nbox[nleft[r]];
    refs[n] = nright[r]; bx[n] = nbox[nright[r]];
    firsts[n] = firsts[best] + nsize[nleft[r]];
    n++;
  }
  for (int i = 1; i < n; i++)  // larger boxes first, as the host builder orders them
    for (int j = i; j > 0 && area(bx[j]) > area(bx[j - 1]); j--) {
      F6 tb = bx[j]; bx[j] = bx[j - 1]; bx[j - 1] = tb;
      int tr = refs[j]; refs[j] = refs[j - 1]; refs[j - 1] = tr;
      uint32_t tf = firsts[j]; firsts[j] = firsts[j - 1]; firsts[j - 1] = tf;
    }
  int inner = 0;
  for (int k = 0; k < n; k++) inner += is_leaf(refs[k]) ? 0 : 1;
  unsigned slot = inner ? atomicAdd(n_nodes, (unsigned)inner) : 0u;  // siblings get adjacent slots
  unsigned fo = inner ? atomicAdd(n_out, (unsigned)inner) : 0u;
  int32_t child[4];
  for (int k = 0; k < n; k++) {
    int r = refs[k];
    if (is_leaf(r)) {
      child[k] = ~(int32_t)((firsts[k] << 3) | (nsize[r] - 1u));
      write_leaf_order(r, firsts[k], n_leaves, nleft, nright, sorted_ids, order);
    } else {
      out[fo++] = Front{r, (int)slot, firsts[k]};
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
  if (n < 2) return hipErrorInvalidValue;  // a single primitive has no tree: its "root" would be a leaf without children
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
  const size_t n2 = 2 * (size_t)n;
  uint64_t *keys = (uint64_t *)alloc((size_t)n * 8), *keys2 = (uint64_t *)alloc((size_t)n * 8);
  uint32_t *ids = (uint32_t *)alloc((size_t)n * 4), *sorted_ids = (uint32_t *)alloc((size_t)n * 4);
  unsigned *counters = (unsigned *)alloc(64);  // [0] n_analytic, [1] n_nodes, [2], [3] frontier sizes
  F6 *nbox = (F6 *)alloc(n2 * sizeof(F6));
  int *nleft = (int *)alloc(n2 * 4), *nright = (int *)alloc(n2 * 4);
  uint32_t *nsize = (uint32_t *)alloc(n2 * 4);
  uint8_t *node_kind = (uint8_t *)alloc(n2);
  int *cl[2] = {(int *)alloc((size_t)n * 4), (int *)alloc((size_t)n * 4)};
  int *nn = (int *)alloc((size_t)n * 4);
  unsigned long long *flags = (unsigned long long *)alloc((size_t)n * 8), *scan = (unsigned long long *)alloc((size_t)n * 8);
  Front *fr[2] = {(Front *)alloc((size_t)n * sizeof(Front)), (Front *)alloc((size_t)n * sizeof(Front))};
  uint32_t *order = nullptr;
  Node4 *nodes4 = nullptr;
  if (e == hipSuccess) e = hipMalloc((void **)&order, (size_t)n * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&nodes4, (size_t)n * sizeof(Node4));
  if (e != hipSuccess) { cleanup(); (void)hipFree(order); (void)hipFree(nodes4); return e; }
  (void)hipMemsetAsync(counters, 0, 64, st);

  double3 lo{bmin[0], bmin[1], bmin[2]};
  double3 inv{bmax[0] > bmin[0] ? 1.0 / (bmax[0] - bmin[0]) : 0.0, bmax[1] > bmin[1] ? 1.0 / (bmax[1] - bmin[1]) : 0.0,
              bmax[2] > bmin[2] ? 1.0 / (bmax[2] - bmin[2]) : 0.0};
  hipLaunchKernelGGL(k_keys, dim3(blocks(n)), dim3(kB), 0, st, boxes, d_prim_info, n, lo, inv, keys, ids, counters);
  size_t sort_bytes = 0, scan_bytes = 0;
  e = rocprim::radix_sort_pairs(nullptr, sort_bytes, keys, keys2, ids, sorted_ids, n, 0, 64, st);
  if (e == hipSuccess) e = rocprim::exclusive_scan(nullptr, scan_bytes, flags, scan, 0ull, (size_t)n, rocprim::plus<unsigned long long>(), st);
  void *sort_tmp = alloc(sort_bytes);
  void *scan_tmp = alloc(scan_bytes);
  if (e == hipSuccess) e = rocprim::radix_sort_pairs(sort_tmp, sort_bytes, keys, keys2, ids, sorted_ids, n, 0, 64, st);
  if (e == hipSuccess) hipLaunchKernelGGL(k_init_clusters, dim3(blocks(n)), dim3(kB), 0, st, boxes, sorted_ids, keys2, n, nbox, nsize, node_kind, cl[0]);

  // PLOC rounds until each kind of primitive is one cluster
  uint32_t m = n, next_node = n;
  int cur = 0, rounds = 0;
  while (e == hipSuccess && m > 1) {
    hipLaunchKernelGGL(k_nearest, dim3(blocks(m)), dim3(kB), 0, st, cl[cur], m, nbox, node_kind, nn);
    hipLaunchKernelGGL(k_pairs, dim3(blocks(m)), dim3(kB), 0, st, nn, m, flags);
    e = rocprim::exclusive_scan(scan_tmp, scan_bytes, flags, scan, 0ull, (size_t)m, rocprim::plus<unsigned long long>(), st);
    if (e != hipSuccess) break;
    hipLaunchKernelGGL(k_merge, dim3(blocks(m)), dim3(kB), 0, st, cl[cur], nn, m, flags, scan, next_node, nbox, nleft, nright, nsize, node_kind, cl[cur ^ 1]);
    unsigned long long last_flag = 0, last_scan = 0;
    e = hipMemcpyAsync(&last_flag, flags + (m - 1), 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&last_scan, scan + (m - 1), 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) break;
    const unsigned long long total = last_scan + last_flag;
    const uint32_t kept = (uint32_t)(total & 0xffffffffull), made = (uint32_t)(total >> 32);
    cur ^= 1;
    rounds++;
    if (made == 0) {
      // nothing paired: what is left is one cluster of analytic primitives and one of triangles
      // (clusters of one kind always hold a mutual pair); the root joins them, analytic side first
      if (kept != 2) { e = hipErrorUnknown; break; }  // no mutual pair among > 2 clusters (non-finite box areas): the caller falls back
      int two[2];
      e = hipMemcpy(two, cl[cur], 8, hipMemcpyDeviceToHost);
      if (e != hipSuccess) break;
      const uint32_t id = next_node++;
      uint32_t sizes[2];
      F6 bx[2];
      for (int k = 0; k < 2 && e == hipSuccess; k++) {
        e = hipMemcpy(&sizes[k], nsize + two[k], 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(&bx[k], nbox + two[k], sizeof(F6), hipMemcpyDeviceToHost);
      }
      if (e != hipSuccess) break;
      F6 rb;
      for (int k = 0; k < 3; k++) { rb.mn[k] = std::min(bx[0].mn[k], bx[1].mn[k]); rb.mx[k] = std::max(bx[0].mx[k], bx[1].mx[k]); }
      const uint32_t total_size = sizes[0] + sizes[1];
      e = hipMemcpy(nbox + id, &rb, sizeof(F6), hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMemcpy(nleft + id, &two[0], 4, hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMemcpy(nright + id, &two[1], 4, hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMemcpy(nsize + id, &total_size, 4, hipMemcpyHostToDevice);
      const int root_id = (int)id;
      if (e == hipSuccess) e = hipMemcpy(cl[cur], &root_id, 4, hipMemcpyHostToDevice);
      m = 1;
      break;
    }
    next_node += made;
    m = kept;
  }
  int root = 0;
  if (e == hipSuccess) e = hipMemcpy(&root, cl[cur], 4, hipMemcpyDeviceToHost);

  // collapse, level by level from the root
  unsigned h_counts[4] = {0, 1, 1, 0};  // slot 0 = the root, one frontier entry
  Front rootf{root, 0, 0u};
  if (e == hipSuccess) e = hipMemcpyAsync(fr[0], &rootf, sizeof(rootf), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(counters + 1, h_counts + 1, 4, hipMemcpyHostToDevice, st);
  uint32_t n_in = 1;
  int levels = 0;
  cur = 0;
  while (e == hipSuccess && n_in > 0) {
    (void)hipMemsetAsync(counters + 2, 0, 4, st);
    hipLaunchKernelGGL(k_collapse, dim3(blocks(n_in)), dim3(kB), 0, st, fr[cur], n_in, fr[cur ^ 1], counters + 2, counters + 1, nodes4, nleft, nright, nsize,
                       nbox, n, sorted_ids, order, leaf_max);
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
  out->rounds = rounds;
  out->n_analytic = (int)h_counts[0];
  return hipSuccess;
}

}  // namespace mipt
