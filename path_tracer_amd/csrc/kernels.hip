// kernels.hip — the wavefront path-tracing kernels for gfx950.
//
//   generate  : camera rays                    (RayTracer::trace, RayTracer.cpp:46-52; camera.cpp:20-29)
//   intersect : closest hit per queued ray     (Scene::intersect, scene.cpp:114-138)
//   shade     : Russian roulette, bounce and mirror rays, BRDF terms that need no
//               shadow ray, queue compaction   (RayTracer::tracePath, RayTracer.cpp:221-306;
//                                               Material::shadeBRDF, material.cpp:88-170)
//   shadow    : one lane per shadow ray, one result per ray; a second kernel sums the
//               rays of each hit                (Light::shadowAttenuation, light.cpp:15-147)
//   resolve   : per-sample clamp and per-pixel sum (RayTracer.cpp:54,75)
//   finalize  : mean and 8-bit truncation      (RayTracer.cpp:75,97-99)
//
// The recursion of tracePath is unrolled level by level.  A path-tree node returns
//   (A + 2*pi*B*L1) / (1-q)            without a mirror ray
//   (A + 2*pi*B*L1 + L2) / (2*(1-q))   with one
// which is affine in the child radiances L1, L2, so every edge of the tree carries a
// throughput T and every node adds T*A/norm to its sample's accumulator.  The
// per-sample clamp (RayTracer.cpp:54) is applied by `resolve` once the whole tree of
// the sample has been summed.
#include "kernels.h"

#include <hip/hip_runtime.h>

namespace mipt {

constexpr int kBlock = 256;
#ifndef MIPT_TRAV_WAVES
#define MIPT_TRAV_WAVES 4  // waves per SIMD the traversal kernels are compiled for
#endif
#ifndef MIPT_LEAN_WAVES
#define MIPT_LEAN_WAVES MIPT_TRAV_WAVES  // the same for the shadow kernel without the analytic tests (it needs fewer registers)
#endif
constexpr int kMaxTravWaves = MIPT_LEAN_WAVES > MIPT_TRAV_WAVES ? MIPT_LEAN_WAVES : MIPT_TRAV_WAVES;

// ------------------------------------------------------------------ helpers ----
__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }

// Block-aggregated append of up to three output streams: every thread of the block calls it
// once.  Each wave counts its emitters with ballots, the block's first thread reserves
// space with ONE atomic per counter (a single word takes only ~90 atomics per microsecond,
// so per-wave atomics on the queue counters would bound the whole kernel), and every
// emitter gets base + waves-before + lanes-before.  Streams a and b share counter `ca`.
template <int BLOCK>
__device__ __forceinline__ void block_append3(bool a, bool b, bool c, uint32_t *ca, uint32_t *cc, uint32_t &ia, uint32_t &ib, uint32_t &ic) {
  constexpr int W = BLOCK / 64;
  __shared__ uint32_t cnt[3][W];
  __shared__ uint32_t base[2];
  const uint32_t lane = lane_id(), w = threadIdx.x / 64;
  unsigned long long ma = __ballot(a), mb = __ballot(b), mc = __ballot(c);
  if (lane == 0) {
    cnt[0][w] = (uint32_t)__popcll(ma);
    cnt[1][w] = (uint32_t)__popcll(mb);
    cnt[2][w] = (uint32_t)__popcll(mc);
  }
  __syncthreads();
  static_assert(BLOCK >= 128, "the two reservations are made by the first lanes of two waves");
  if (threadIdx.x == 0) {
    uint32_t ta = 0, tb = 0;
    for (int k = 0; k < W; k++) {
      uint32_t x = cnt[0][k]; cnt[0][k] = ta; ta += x;
      uint32_t y = cnt[1][k]; cnt[1][k] = tb; tb += y;
    }
    for (int k = 0; k < W; k++) cnt[1][k] += ta;  // stream b follows stream a in the shared region
    base[0] = (ta + tb) ? atomicAdd(ca, ta + tb) : 0u;
  } else if (threadIdx.x == 64) {  // the other counter from another wave: the two round trips overlap
    uint32_t tc = 0;
    for (int k = 0; k < W; k++) {
      uint32_t z = cnt[2][k]; cnt[2][k] = tc; tc += z;
    }
    base[1] = tc ? atomicAdd(cc, tc) : 0u;
  }
  __syncthreads();
  unsigned long long below = (1ull << lane) - 1ull;
  ia = base[0] + cnt[0][w] + (uint32_t)__popcll(ma & below);
  ib = base[0] + cnt[1][w] + (uint32_t)__popcll(mb & below);
  ic = base[1] + cnt[2][w] + (uint32_t)__popcll(mc & below);
}

__device__ __forceinline__ void wave_count(unsigned long long *dst, uint32_t v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  if (lane_id() == 0 && v) atomicAdd(dst, (unsigned long long)v);
}

// The same addition where no other thread of the launch, and no other launch running beside it, touches
// the slot: the camera rays of a batch (one per sample) and their shaded hits.  A plain read-add-write
// gives the bits the atomic would (one addition to the same value); the atomics of the fp64 adder in L2
// are what bounds k_shadow_sum (31 -> 22 ms per benchmark step without them).
__device__ __forceinline__ void deposit_alone(double *accum, uint32_t slot, d3 v) {
  double *a = accum + 3ull * slot;
  if (v.x != 0.0) a[0] = a[0] + v.x;
  if (v.y != 0.0) a[1] = a[1] + v.y;
  if (v.z != 0.0) a[2] = a[2] + v.z;
}
__device__ __forceinline__ void deposit(double *accum, uint32_t slot, d3 v) {
  double *a = accum + 3ull * slot;
  if (v.x != 0.0) unsafeAtomicAdd(a + 0, v.x);
  if (v.y != 0.0) unsafeAtomicAdd(a + 1, v.y);
  if (v.z != 0.0) unsafeAtomicAdd(a + 2, v.z);
}

__device__ __forceinline__ d3 pow5v(d3 F0, double x) {  // F0 + (1-F0)*x^5, material.cpp:75-78
  double x2 = x * x;
  double x5 = x2 * x2 * x;
  return d3{F0.x + (1.0 - F0.x) * x5, F0.y + (1.0 - F0.y) * x5, F0.z + (1.0 - F0.z) * x5};
}
__device__ __forceinline__ double ggx_geometry(d3 n, d3 x, double alpha) {  // material.cpp:68-73
  double nDotX = fabs(dot(n, x));
  double denom = nDotX + sqrt((alpha * alpha) + (1 - (alpha * alpha)) * (nDotX * nDotX));
  return (2 * nDotX) / denom;
}
__device__ __forceinline__ double ndf(double alpha, d3 N, d3 H) {  // material.cpp:80-86
  double a2 = alpha * alpha;
  double nDotH = fabs(dot(N, H));
  double inner = (nDotH * nDotH) * (a2 - 1) + 1;
  return a2 / ((inner * inner) * M_PI);
}

__device__ __forceinline__ d3 light_direction(const LightRec &L, d3 P) {  // light.cpp:39-41,55-57,96-98
  if (L.type == MIPT_LIGHT_DIRECTIONAL) return -ld3(L.orientation);
  if (L.type == MIPT_LIGHT_POINT) return normalize(ld3(L.position) - P);
  return normalize(ld3(L.center) - P);
}
__device__ __forceinline__ double distance_attenuation(const LightRec &L, d3 P) {  // light.cpp:10-13,47-51,88-90
  if (L.type != MIPT_LIGHT_POINT) return 1.0;
  double d = length(P - ld3(L.position));
  double denom = L.a + L.b * d + L.c * (d * d);
  return fmin(1.0, 1 / denom);
}

// TextureMap::getPixelAt / getMappedValue, material.cpp:186-222.
__device__ __forceinline__ d3 tex_pixel(const uint8_t *data, int w, int h, int x, int y) {
  if (x < 0 || x >= w || y < 0 || y >= h) return d3{0, 0, 0};
  const uint8_t *p = data + ((size_t)x + (size_t)y * w) * 3;
  return d3{(double)p[0] / 255.0, (double)p[1] / 255.0, (double)p[2] / 255.0};
}
__device__ __forceinline__ d3 tex_mapped(const uint8_t *data, int w, int h, double cu, double cv) {
  double x = cu * (double)(w - 1), y = cv * (double)(h - 1);
  double x1 = (double)(int)x, y1 = (double)(int)y;
  double x2 = x1 + 1, y2 = y1 + 1;
  d3 c11 = tex_pixel(data, w, h, (int)x1, (int)y1);
  d3 c12 = tex_pixel(data, w, h, (int)x1, (int)y2);
  d3 c21 = tex_pixel(data, w, h, (int)x2, (int)y1);
  d3 c22 = tex_pixel(data, w, h, (int)x2, (int)y2);
  return ((c11 * (x2 - x) * (y2 - y) + c21 * (x - x1) * (y2 - y)) + c12 * (x2 - x) * (y - y1)) + c22 * (x - x1) * (y - y1);
}
// CubeMap::getColor, cubeMap.cpp:7-64 (independent ifs: the last match wins).
__device__ __forceinline__ d3 cube_color(const SceneDev &S, d3 dirIn) {
  d3 dir = normalize(dirIn);
  double absX = fabs(dir.x), absY = fabs(dir.y), absZ = fabs(dir.z);
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
  return tex_mapped(S.cube[idx], S.cube_w[idx], S.cube_h[idx], cu, cv);
}

// ---------------------------------------------------------------- generate ----
// Camera ray `idx` of a batch: direction and the random-stream key of its sample.  (Origin = the eye,
// throughput 1, accumulator slot idx, depth 0.)
__device__ __forceinline__ void camera_ray(const SceneDev &S, const RenderConst &rc, const CameraRays &g, uint32_t idx, d3 &dir, uint64_t &key) {
  uint32_t lp = idx / (uint32_t)g.n_samples;
  uint32_t k = idx - lp * (uint32_t)g.n_samples;
  const uint32_t op = (uint32_t)g.first_pixel + lp;  // owned pixels are fewer than 2^30 (the frame is): 32-bit divisions
  const uint32_t row = op / (uint32_t)rc.width;
  int j = g.rows[row];
  int i = (int)(op - row * (uint32_t)rc.width);
  // tracePixel: x = i/w, y = j/h (RayTracer.cpp:70-71); rayThrough (camera.cpp:20-29):
  // normalize((look + x u) + y v).  Every operation is rounded by itself (no fused multiply-adds): the
  // kernel that traces a camera ray and the kernel that shades it compute the same bits, whatever the
  // compiler would contract in either, and they are the bits the expression has on a CPU.
  const double x = __dadd_rn((double)i / (double)rc.width, -0.5);
  const double y = __dadd_rn((double)j / (double)rc.height, -0.5);
  const double ax = __dadd_rn(__dadd_rn(S.camera.look[0], __dmul_rn(x, S.camera.u[0])), __dmul_rn(y, S.camera.v[0]));
  const double ay = __dadd_rn(__dadd_rn(S.camera.look[1], __dmul_rn(x, S.camera.u[1])), __dmul_rn(y, S.camera.v[1]));
  const double az = __dadd_rn(__dadd_rn(S.camera.look[2], __dmul_rn(x, S.camera.u[2])), __dmul_rn(y, S.camera.v[2]));
  const double inv = 1.0 / sqrt(__dadd_rn(__dadd_rn(__dmul_rn(ax, ax), __dmul_rn(ay, ay)), __dmul_rn(az, az)));
  dir = d3{__dmul_rn(ax, inv), __dmul_rn(ay, inv), __dmul_rn(az, inv)};
  key = sample_key(rc.seed, (uint64_t)i + (uint64_t)j * (uint64_t)rc.width, (uint64_t)(g.sample0 + (int)k));
}

__global__ void __launch_bounds__(kBlock) k_generate(SceneDev S, RenderConst rc, CameraRays g, RayQueue q) {
  uint32_t idx = blockIdx.x * kBlock + threadIdx.x;
  uint32_t total = (uint32_t)g.n_pixels * (uint32_t)g.n_samples;
  if (idx >= total) return;
  d3 dir;
  uint64_t key;
  camera_ray(S, rc, g, idx, dir, key);
  q.ox[idx] = S.camera.eye[0]; q.oy[idx] = S.camera.eye[1]; q.oz[idx] = S.camera.eye[2];
  q.dx[idx] = dir.x; q.dy[idx] = dir.y; q.dz[idx] = dir.z;
  q.tx[idx] = 1.0; q.ty[idx] = 1.0; q.tz[idx] = 1.0;
  q.key[idx] = key;
  q.slot[idx] = idx;
  q.depth[idx] = 0;
}

// --------------------------------------------------------------- intersect ----
// Persistent waves with ray replacement.  Rays differ wildly in how many nodes they
// visit, and a wave lasts as long as its slowest lane; so a lane whose ray is finished
// does not wait for the others: once kRefillIdle lanes of a wave are idle they are given
// new rays.  For most of a launch every wave owns a fixed, interleaved sequence of 64-ray
// chunks (chunk numbers wave, wave + W, wave + 2W, ...), so there is no work counter to contend
// for and rays that are neighbours in the queue still enter a wave together; the last quarter
// of the rounds comes from per-XCD counters so that the waves finish together (WorkFeed).
#ifndef MIPT_REFILL_IDLE
#define MIPT_REFILL_IDLE 40  // measured 24 / 32 / 40 / 48: 602 / 582 / 577 / 606 ms of k_shadow_any per benchmark step
#endif
#ifndef MIPT_REFILL_IDLE_CLOSEST
#define MIPT_REFILL_IDLE_CLOSEST 32
#endif
#ifndef MIPT_OWN_PRIM_FIRST
#define MIPT_OWN_PRIM_FIRST 1  // any-hit shadow rays that start into their own surface test the triangle they start on first
#endif
#ifndef MIPT_REREFILL_IDLE
#define MIPT_REREFILL_IDLE 8  // (measured 8 / 16 / 24 / 40: 571.5 / 574.6 / 576.8 / 575.6 ms) lanes idle again right after a refill (rays ended in the set-up) that trigger another one at once
#endif
constexpr uint32_t kRefillIdle = MIPT_REFILL_IDLE;                  // shadow rays
constexpr uint32_t kRefillIdleClosest = MIPT_REFILL_IDLE_CLOSEST;  // path rays

// Workgroups go to the eight XCDs round-robin (block b runs on XCD b % 8), and every XCD has
// an L2 of its own.  Numbering the blocks XCD-major makes the waves of one XCD take
// neighbouring chunks of the item sequence at any moment, so an XCD's L2 holds the part of
// the scene its rays are in rather than a share of everything.
#ifndef MIPT_XCD_MAJOR
#define MIPT_XCD_MAJOR 1
#endif
__device__ __forceinline__ uint32_t xcd_major_block() {
  if (!MIPT_XCD_MAJOR || (gridDim.x & 7u) != 0u) return blockIdx.x;
  return (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
}

// Work feed of a persistent wave: 64-item chunks, drawn a few at a time from a counter.
// The six waves of a SIMD do not run at the same pace: the one in the lowest wave slot issues first whenever it can, and
// measured on k_shadow_any (a build with -DMIPT_ANY_DEBUG, tools/experiments/any_wave_stats.py) the waves in slot 0 get
// through the same number of chunks in 18.2 ms that those in slot 5 need 27.0 ms for.  A share of the chunks that is a wave's
// own (chunk numbers wave, wave + W, ...: no counter) therefore ends at very different times, and with three quarters of the
// launch dealt out that way and a quarter drawn from counters 7 % of the launch's wave-slot time idled at its end.  With every
// chunk drawn from a counter the slow slots simply draw fewer (3500 to 6700 per wave in one launch): 0.15 % idle, the camera
// level's shadow launch of the benchmark step 220.7 -> 191.9 ms, the step 708 -> 664 ms.  What made "more from the counters"
// slower when it was first tried (2/8 / 3/8 of the rounds: 498.5 / 502.1 ms) was not the counters but where their chunks lay:
// a contiguous run of the launch per XCD, so that the eight XCDs worked in eight distant parts of the item sequence; the
// groups now alternate between the XCDs (below) and the chip moves through the items as one.
// MIPT_DYNAMIC_TAIL_EIGHTHS < 8 keeps that share of own chunks at the front of the launch (A/B builds).
#ifndef MIPT_DYNAMIC_TAIL_EIGHTHS
#define MIPT_DYNAMIC_TAIL_EIGHTHS 8  // share of the rounds handed out dynamically, in eighths (0: none)
#endif
struct WorkFeed {
  uint64_t base;   // first item of the current chunk
  uint32_t used;   // items of it already handed out
  uint32_t round;  // chunks this wave has started
  uint32_t static_rounds, wave, n_waves;
  uint32_t tail_first, tail_end;  // chunk numbers of the dynamic part this wave's XCD draws from
  uint32_t part;                  // which of the (up to eight) runs of the dynamic part that is
  uint32_t grp_next, grp_left;    // the chunks of the group this wave last drew that it has not started
  uint32_t parts;                 // runs (XCDs) the dynamic part is dealt to
  uint32_t group;                 // chunks a wave draws at a time
};
#ifndef MIPT_DYN_INTERLEAVE
#define MIPT_DYN_INTERLEAVE 1  // 0: a contiguous run of the launch per XCD (A/B builds)
#endif
#ifndef MIPT_GROUP_CLOSEST
#define MIPT_GROUP_CLOSEST 4  // most chunks a wave of k_intersect draws from its XCD's counter at a time
#endif
#ifndef MIPT_GROUP_DIV
#define MIPT_GROUP_DIV 8      // ... and at least this many groups per wave and launch
#endif
#ifndef MIPT_GROUP_ANY
#define MIPT_GROUP_ANY 6      // the same for k_shadow_any
#endif
// One counter per XCD (workgroups go to the XCDs round-robin: blockIdx % 8): one word takes ~90 atomics per microsecond
// and 6144 waves finish a chunk every ~5 us, so a wave draws a group of chunks per atomic (k_shadow_any alone, groups of
// 1 / 2 / 4 / 8 chunks: 489 / 458 / 454 / 455 ms per step; k_intersect prefers 2-4: 113-115 against 119 ms at 8) and the
// counters are eight.  Group v of XCD x is group 8 v + x of the launch.  (Smaller groups towards the end of a launch, so that
// it ends within a chunk's time rather than a group's, halve what idles at the end of a mid-sized launch, 5.5 -> 2.3 % of the
// wave slots at 65 chunks per wave, and cost k_intersect 4 % as a whole: not kept.)
__device__ __forceinline__ uint64_t feed_chunk(WorkFeed &f, unsigned int *counters) {
  if (f.round < f.static_rounds) return ((uint64_t)f.round * f.n_waves + f.wave) * 64ull;
  if (f.grp_left == 0u) {
    uint32_t v = 0;
    if (lane_id() == 0) v = atomicAdd(counters + f.part, 1u);
    v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
#if MIPT_DYN_INTERLEAVE
    const uint64_t first = (uint64_t)f.tail_first + ((uint64_t)v * f.parts + f.part) * (uint64_t)f.group;
#else
    const uint64_t first = (uint64_t)f.tail_first + (uint64_t)v * (uint64_t)f.group;
#endif
    f.grp_next = (uint32_t)min(first, (uint64_t)f.tail_end);
    f.grp_left = f.group;
  }
  const uint64_t c = f.grp_next;
  f.grp_next++;
  f.grp_left--;
  return c < f.tail_end ? c * 64ull : ~0ull;
}
__device__ __forceinline__ void feed_init(WorkFeed &f, uint64_t n_items, uint32_t wave, uint32_t n_waves, unsigned int *counters, uint32_t max_group) {
  const uint64_t chunks = (n_items + 63ull) / 64ull;
  // groups as large as allowed while every wave still draws MIPT_GROUP_DIV or more of them (the launch ends within a group's time)
  f.group = (uint32_t)min((uint64_t)max_group, max((uint64_t)1, chunks / ((uint64_t)n_waves * (uint64_t)MIPT_GROUP_DIV)));
  const uint64_t rounds = chunks / n_waves;  // whole rounds of static chunks
  f.static_rounds = MIPT_DYNAMIC_TAIL_EIGHTHS > 0 ? (uint32_t)(rounds - rounds * MIPT_DYNAMIC_TAIL_EIGHTHS / 8) : 0xffffffffu;
  f.wave = wave;
  f.n_waves = n_waves;
  if (MIPT_DYNAMIC_TAIL_EIGHTHS > 0) {
    const uint64_t t0 = (uint64_t)f.static_rounds * n_waves;
    const uint32_t parts = min(8u, gridDim.x);  // a launch of fewer than eight blocks does not reach every XCD
    const uint64_t len = (chunks - t0 + parts - 1u) / parts;
    f.part = blockIdx.x % parts;
    f.parts = parts;
#if MIPT_DYN_INTERLEAVE
    f.tail_first = (uint32_t)t0;
    f.tail_end = (uint32_t)chunks;
    (void)len;
#else
    f.tail_first = (uint32_t)min(t0 + f.part * len, chunks);
    f.tail_end = (uint32_t)min(t0 + (f.part + 1ull) * len, chunks);
#endif
  } else {
    f.tail_first = f.tail_end = 0;
    f.part = 0;
    f.parts = 1;
  }
  f.round = 0;
  f.used = 0;
  f.grp_next = f.grp_left = 0;
  f.base = feed_chunk(f, counters);
}
// Takes up to `want` (<= 64) items: the `got` items handed out are first .. first + n1 - 1 and,
// when the current chunk ran out, first2 .. of the next one; got = 0 when the launch's items are
// all handed out.  rank r of the takers gets feed_item(r, ...).
struct FeedTake { uint64_t first, first2; uint32_t n1, got; };
__device__ __forceinline__ FeedTake feed_take(WorkFeed &f, uint32_t want, uint64_t n_items, unsigned int *counter) {
  FeedTake t{0, 0, 0, 0};
  if (f.used == 64u) {
    f.round++;
    f.used = 0;
    f.base = feed_chunk(f, counter);
  }
  if (f.base >= n_items) return t;
  t.first = f.base + f.used;
  t.n1 = min(want, 64u - f.used);
  f.used += t.n1;
  t.got = t.n1;
  if (t.n1 < want) {  // the rest from the next chunk
    f.round++;
    f.base = feed_chunk(f, counter);
    f.used = 0;
    if (f.base < n_items) {
      t.first2 = f.base;
      f.used = want - t.n1;
      t.got = want;
    } else {
      f.used = 64u;  // stays past the end: the next call finds nothing either
    }
  }
  return t;
}
__device__ __forceinline__ uint64_t feed_item(const FeedTake &t, uint32_t r) { return r < t.n1 ? t.first + r : t.first2 + (r - t.n1); }

__device__ __forceinline__ uint32_t rank_below(unsigned long long mask, uint32_t lane) {
  return (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

#ifdef MIPT_ANY_DEBUG
__device__ unsigned long long g_isect_dbg[16384 * 4];  // per wave of the last k_intersect launch: start, end, XCD | chunks << 8 | hw id << 40, rays
extern "C" int mipt_debug_isect(unsigned long long *out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_isect_dbg), sizeof(unsigned long long) * (size_t)n);
}
#endif
// `dup` > 1: the queue holds runs of `dup` identical rays (the camera rays of one pixel's
// samples, RayTracer.cpp:72-75 calls trace(x, y) N times with the same x, y); only the first
// ray of every run is traced, k_shade reads its result for the whole run.  n = rays to trace.
// CAM: the level of the camera rays, computed from their index (`cam`) instead of read from `q`.
// LEAN (SceneDev::lean_any, never with CAM): the walk passes the few analytic primitives beside the meshes over — with their
// tests compiled in the kernel needs 96 registers and 68 bytes of scratch per lane, without them 80 and 12 — and k_shade,
// which reads the ray and its hit anyway, lets those primitives improve on the hit (analytic_override).
template <bool COUNT, bool LOCAL, bool CAM, bool LEAN = false>
#ifndef MIPT_LEAN_CLOSEST_WAVES
#define MIPT_LEAN_CLOSEST_WAVES MIPT_TRAV_WAVES  // waves per SIMD of the triangles-only closest-hit kernel; at 6 (80 registers, 12 B of scratch) 98.4 against 96.8 ms per benchmark step
#endif
__global__ void __launch_bounds__(kBlock, (LEAN ? MIPT_LEAN_CLOSEST_WAVES : MIPT_TRAV_WAVES)) k_intersect(SceneDev S, RenderConst rc, RayQueue q, uint32_t n, uint32_t dup,
                                                                        double *__restrict__ hit_t, int *__restrict__ hit_prim, DeviceCounters *dc,
                                                                        CameraRays cam) {
  extern __shared__ uint32_t lds_stack[];
#ifdef MIPT_ANY_DEBUG
  const unsigned long long dbg_t0 = wall_clock64();
#endif
  const Stack stack{lds_stack + threadIdx.x, kBlock, blockIdx.x * kBlock + threadIdx.x};
  const uint32_t lane = lane_id();
  const uint32_t wave = xcd_major_block() * (kBlock / 64) + threadIdx.x / 64;
  const uint32_t n_waves = gridDim.x * (kBlock / 64);
  TravCounters tc{0, 0};
  Trav T;
  T.cur = MIPT_TRAV_DONE;
  T.sp = 0;
  bool idle = true;
  uint32_t ray = 0;
  WorkFeed feed;
  feed_init(feed, n, wave, n_waves, dc->dyn_next, MIPT_GROUP_CLOSEST);
  bool exhausted = false;  // wave-uniform
  while (true) {
    unsigned long long idle_mask = __ballot(idle);
    if (idle_mask != 0ull && !exhausted) {
      const FeedTake take = feed_take(feed, (uint32_t)__popcll(idle_mask), n, dc->dyn_next);
      exhausted = take.got == 0u;
      const uint32_t rnk = rank_below(idle_mask, lane);
      if (idle && rnk < take.got) {
        uint64_t item = feed_item(take, rnk);
        if (item < n) {
          ray = (uint32_t)item * dup;
          d3 o, d;
          if (CAM) {
            uint64_t key_unused;
            o = ld3(S.camera.eye);
            camera_ray(S, rc, cam, ray, d, key_unused);
          } else {
            o = d3{q.ox[ray], q.oy[ray], q.oz[ray]};
            d = d3{q.dx[ray], q.dy[ray], q.dz[ray]};
          }
          trav_init(T, S, o, d, rc.t_max);
          idle = false;
        }
      }
    }
    if (__ballot(!idle) == 0ull) break;
    const uint32_t want_idle = exhausted ? 64u : kRefillIdleClosest;
    do {
      trav_round<false, COUNT, LOCAL, LEAN>(T, S, stack, tc);
      if (!idle && T.cur == MIPT_TRAV_DONE) {
        hit_t[ray] = T.best.t;
        hit_prim[ray] = T.best.prim;
        idle = true;
      }
    } while ((uint32_t)__popcll(__ballot(idle)) < want_idle);
  }
  if (COUNT) {
    wave_count(&dc->closest_nodes, tc.nodes);
    wave_count(&dc->closest_prims, tc.prims);
  }
#ifdef MIPT_ANY_DEBUG
  if (lane == 0) {
    const uint32_t w = blockIdx.x * (kBlock / 64) + threadIdx.x / 64;
    if (w < 16384u) {
      g_isect_dbg[4ull * w + 0] = dbg_t0;
      g_isect_dbg[4ull * w + 1] = wall_clock64();
      g_isect_dbg[4ull * w + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0xf) | ((unsigned long long)feed.round << 8) |
                                  ((unsigned long long)(__builtin_amdgcn_s_getreg(4 | (31 << 11)) & 0xffffu) << 40);
      g_isect_dbg[4ull * w + 3] = n;
    }
  }
#endif
}

// ------------------------------------------------------------------- shade ----
#ifndef MIPT_SHADE_BLOCK
#define MIPT_SHADE_BLOCK 512
#endif
constexpr int kShadeBlock = MIPT_SHADE_BLOCK;
// One path-tree node (RayTracer::tracePath, RayTracer.cpp:221-306) given the closest hit of its ray:
// the miss branch, Russian roulette, the bounce and mirror children and the terms of shadeBRDF that
// need no shadow ray (deposited here); what needs shadow rays is described by (P, N, W).
struct NodeOut {
  bool emitBounce, emitMirror, emitShade, shaded;
  d3 P, N, W;      // hit point, normal, T * kd / pi * (1 - metallic) / norm
  d3 wdir, rdir;   // bounce and mirror directions
  d3 Tb, Tm;       // their throughputs
  d3 dep;          // ALONE: what the node adds to its sample's sum (the caller stores it); else deposited here
};
template <bool ALONE = false>  // ALONE: the ray is the first of its sample (a camera ray): its term is handed back in r.dep, not added
__device__ __forceinline__ void path_node(const SceneDev &S, const RenderConst &rc, d3 o, d3 d, d3 T, uint64_t key, uint32_t slot, int depth, int prim,
                                          double t, double *accum, NodeOut &r) {
  r.emitBounce = r.emitMirror = r.emitShade = r.shaded = false;
  r.P = r.N = r.W = r.wdir = r.rdir = r.Tb = r.Tm = r.dep = d3{0, 0, 0};
  if (prim < 0) {
    // miss branch, RayTracer.cpp:279-304
    if (S.has_cubemap) { if (ALONE) r.dep = T * cube_color(S, d); else deposit(accum, slot, T * cube_color(S, d)); }
    return;
  }
  double u0 = u01(mix64(key, SLOT_RR));
  if (u0 < rc.rr_q) return;  // RayTracer.cpp:228-231
  r.shaded = true;
  d3 N = hit_normal(S, prim, o, d);
  d3 P = o + t * d;  // r.at(i)
  const ObjRec *ob = S.objects + (S.prim_info[prim] & 0x3fffffffu);
  const MatRec *m = S.materials + ob->material;
  d3 kd = ld3(m->kd), ka = ld3(m->ka);
  if (S.n_textures > 0 && (m->kd_tex | m->ka_tex) != 0) {
    // texture-mapped colours: MaterialParameter::value at the hit's UV, material.cpp:224-230
    double tu, tv;
    hit_uv(S, prim, o, d, tu, tv);
    if (m->kd_tex > 0) { const TexRec &tx = S.textures[m->kd_tex - 1]; kd = tex_mapped(tx.data, tx.w, tx.h, tu, tv); }
    if (m->ka_tex > 0) { const TexRec &tx = S.textures[m->ka_tex - 1]; ka = tex_mapped(tx.data, tx.w, tx.h, tu, tv); }
  }
  if (S.tri_colors) hit_vertex_color(S, prim, o, d, kd);  // per-vertex colours replace kd, trimesh.cpp:179-189
  // tangent frame, RayTracer.cpp:236-241
  d3 Nt;
  if (fabs(N.x) > fabs(N.y)) Nt = d3{N.z, 0, -N.x} / sqrt(N.x * N.x + N.z * N.z);
  else Nt = d3{0, -N.z, N.y} / sqrt(N.y * N.y + N.z * N.z);
  d3 Nb = cross(N, Nt);
  // hemisphere sample, RayTracer.cpp:244-257 (x and z pass through float)
  double r1 = u01(mix64(key, SLOT_COS));
  double sinTheta = sqrt(1 - r1 * r1);
  double phi = u01(mix64(key, SLOT_PHI)) * 2 * M_PI;
  double sphi, cphi;
  sincos(phi, &sphi, &cphi);
  float xf = (float)(sinTheta * cphi);
  float zf = (float)(sinTheta * sphi);
  double rx = (double)xf, rz = (double)zf;
  d3 wdir = normalize(d3{rx * Nb.x + r1 * N.x + rz * Nt.x, rx * Nb.y + r1 * N.y + rz * Nt.y, rx * Nb.z + r1 * N.z + rz * Nt.z});
  d3 V = normalize(-d);
  // shadeBRDF set-up, material.cpp:99-109
  double roughness = m->roughness;
  double rB = roughness == 0 ? 0.001 : roughness;
  double alpha = rB * rB;
  double f0s = (1.0 - m->index) / (1.0 + m->index);
  f0s = f0s * f0s;
  d3 F0{f0s, f0s, f0s};
  double metallic = m->metallic;
  if (metallic > 0) F0 = F0 * (1.0 - metallic) + kd * metallic;
  double nV = dot(N, V);
  double Gv = ggx_geometry(N, V, alpha);
  // mirror decision, RayTracer.cpp:268-269 (raw roughness)
  double u3 = u01(mix64(key, SLOT_MIRROR));
  bool mirror = roughness < u3;
  double norm = (1.0 - rc.rr_q) * (mirror ? 2.0 : 1.0);
  // terms that need no shadow ray: ambient + direct specular, material.cpp:96,132-143
  d3 A = ka * ld3(S.ambient);
  double lsum = 0.0;  // sum over lights of the brightest colour channel
  for (int li = 0; li < S.n_lights; li++) {
    const LightRec &L = S.lights[li];
    lsum += fmax(fabs(L.color[0]), fmax(fabs(L.color[1]), fabs(L.color[2])));
    d3 Ld = light_direction(L, P);
    d3 H = normalize(Ld + V);
    d3 F = pow5v(F0, 1 - fabs(dot(V, H)));
    double D = ndf(alpha, N, H);
    double G = ggx_geometry(N, Ld, alpha) * Gv;
    double nl = fabs(dot(N, Ld));
    d3 spec = ((F * D) * G) / (4 * nl * nV);
    A = A + (spec * nl) * distance_attenuation(L, P);
  }
  if (ALONE) r.dep = (T * A) / norm;
  else deposit(accum, slot, (T * A) / norm);
  // indirect coefficient, material.cpp:146-163: B = kd*|n.w|/pi*(1-m) + F*G*D/(4*|n.w|*(n.V))*|n.w|
  {
    d3 H = normalize(wdir + V);
    d3 F = pow5v(F0, 1 - fabs(dot(V, H)));
    double D = ndf(alpha, N, H);
    double G = ggx_geometry(N, wdir, alpha) * Gv;
    double nl = fabs(dot(N, wdir));
    d3 B = (kd * nl) / M_PI * (1 - metallic) + (((F * G) * D) / (4 * nl * nV)) * nl;
    r.Tb = (T * B) * (2 * M_PI) / norm;  // pdf = 1/(2*pi), RayTracer.cpp:261-263
  }
  r.Tm = T / norm;
  d3 W = ((T * kd) / M_PI * (1 - metallic)) / norm;
  bool deeper = depth + 1 < rc.max_depth;
  r.emitBounce = deeper;
  r.emitMirror = deeper && mirror;
  r.rdir = d - N * dot(N, d) * 2.0;  // glm::reflect, RayTracer.cpp:270
  r.emitShade = S.n_lights > 0 && (W.x != 0.0 || W.y != 0.0 || W.z != 0.0);
  if (r.emitShade && (rc.flags & MIPT_FLAG_CULL_NEGLIGIBLE)) {
    // every shadow ray of this hit only scales a term bounded by max|W| * sum of the
    // lights' brightest channels (attenuations and |n.L| are <= 1); below 2^-52 it
    // cannot move a sample, which lives in [0, 1], by one unit in the last place of 1
    double wmax = fmax(fabs(W.x), fmax(fabs(W.y), fabs(W.z)));
    r.emitShade = !(wmax * lsum < 2.220446049250313e-16);
  }
  r.P = P;
  r.N = N;
  r.W = W;
  r.wdir = wdir;
}

// The analytic primitives' share of a closest hit found by a walk over the triangles only (k_intersect<LEAN>): the same
// update rule as inside the walk (prim_test: smaller t, ties by the reference's visiting order), so the result is the
// one the full walk returns.  A primitive whose world box the ray cannot reach before its present hit is not tested.
__device__ __forceinline__ uint32_t analytic_override(const SceneDev &S, d3 o, d3 d, int &prim, double &t) {
  Trav T;
  T.org = o;
  T.dir = d;
  T.best.t = t;
  T.best.prim = prim;
  T.tlimit = 0.0f;
  T.cur = MIPT_TRAV_DONE;
  T.sp = 0;
  uint32_t tested = 0;
  for (int k = 0; k < S.n_analytic; k++) {
    const double *b = S.object_boxes + 6u * (S.prim_info[k] & 0x3fffffffu);
    // first without a division: on some axis the box lies beside the whole segment up to the present hit (the boxes are
    // grown by 1e-9 of the scene's size, far more than the rounding of the segment's end)
    const d3 e = o + T.best.t * d;
    if (fmax(o.x, e.x) < b[0] || fmin(o.x, e.x) > b[3] || fmax(o.y, e.y) < b[1] || fmin(o.y, e.y) > b[4] || fmax(o.z, e.z) < b[2] || fmin(o.z, e.z) > b[5]) continue;
    if (!ray_meets_box(o, d, T.best.t, b)) continue;
    tested++;
    prim_test<false, false, false, true>(T, S, (uint32_t)k);
  }
  prim = T.best.prim;
  t = T.best.t;
  return tested;
}

template <bool COUNT, bool CAM>
#ifdef MIPT_SHADE_WAVES
#define MIPT_SHADE_BOUNDS __launch_bounds__(kShadeBlock, MIPT_SHADE_WAVES)
#else
#define MIPT_SHADE_BOUNDS __launch_bounds__(kShadeBlock)
#endif
__global__ void MIPT_SHADE_BOUNDS k_shade(SceneDev S, RenderConst rc, RayQueue q, uint32_t n, uint32_t dup, const double *__restrict__ hit_t,
                                                   const int *__restrict__ hit_prim, RayQueue nq, ShadeList sl, BatchCounts *bc,
                                                   double *accum, DeviceCounters *dc, CameraRays cam) {
  uint32_t idx = blockIdx.x * kShadeBlock + threadIdx.x;
  bool live = idx < n;
  NodeOut r;
  r.emitBounce = r.emitMirror = r.emitShade = r.shaded = false;
  r.dep = d3{0, 0, 0};
  uint64_t key = 0;
  uint32_t slot = 0;
  int depth = 0;
  int hprim = -1;
  uint32_t analytic_tests = 0;
  if (live) {
    const uint32_t hidx = dup > 1u ? idx - idx % dup : idx;  // the run's traced ray
    int prim = hprim = hit_prim[hidx];
    d3 o, d, T;
    if (CAM) {
      o = ld3(S.camera.eye);
      camera_ray(S, rc, cam, idx, d, key);
      T = d3{1.0, 1.0, 1.0};
      slot = idx;
      depth = 0;
    } else {
      o = d3{q.ox[idx], q.oy[idx], q.oz[idx]};
      d = d3{q.dx[idx], q.dy[idx], q.dz[idx]};
      T = d3{q.tx[idx], q.ty[idx], q.tz[idx]};
      key = q.key[idx];
      slot = q.slot[idx];
      depth = (int)q.depth[idx];
    }
    double t_hit = hit_t[hidx];  // a miss holds the walk's limit
    if (!CAM && S.lean_any && S.n_analytic > 0) {
      analytic_tests = analytic_override(S, o, d, prim, t_hit);
      hprim = prim;
    }
    path_node<CAM>(S, rc, o, d, T, key, slot, depth, prim, prim >= 0 ? t_hit : 0.0, accum, r);  // CAM: one camera ray per sample
    if (CAM) {
      // The camera level visits every sample of the batch exactly once, before anything else adds to its sum: it
      // stores the sample's first term (zero for a sample that ends here) instead of adding it to a cleared array,
      // which saves clearing the sums (24 bytes per sample written) and reading them back here.  (+ 0.0: a -0.0 term
      // is stored as the +0.0 the cleared array would have kept.)
      double *a = accum + 3ull * slot;
      a[0] = r.dep.x + 0.0;
      a[1] = r.dep.y + 0.0;
      a[2] = r.dep.z + 0.0;
    }
  }
  const bool emitBounce = r.emitBounce, emitMirror = r.emitMirror, emitShade = r.emitShade, shaded = r.shaded;
  const d3 P = r.P, N = r.N, W = r.W, wdir = r.wdir, rdir = r.rdir, Tb = r.Tb, Tm = r.Tm;
  // ---- compaction: ballots per wave, one atomic per block and counter -----------
  uint32_t ib, im, is;
  block_append3<kShadeBlock>(emitBounce, emitMirror, emitShade, &bc->next_rays, &bc->shades, ib, im, is);
  if (emitBounce) {
    if (ib < rc.capacity) {
      d3 ro = P + wdir * MIPT_RAY_EPSILON;  // RayTracer.cpp:259
      nq.ox[ib] = ro.x; nq.oy[ib] = ro.y; nq.oz[ib] = ro.z;
      nq.dx[ib] = wdir.x; nq.dy[ib] = wdir.y; nq.dz[ib] = wdir.z;
      nq.tx[ib] = Tb.x; nq.ty[ib] = Tb.y; nq.tz[ib] = Tb.z;
      nq.key[ib] = mix64(key, SLOT_CHILD_BOUNCE);
      nq.slot[ib] = slot;
      nq.depth[ib] = (uint8_t)(depth + 1);
    } else {
      bc->overflow = 1;
    }
  }
  if (emitMirror) {
    if (im < rc.capacity) {
      d3 ro = P + MIPT_RAY_EPSILON * N;  // RayTracer.cpp:271
      nq.ox[im] = ro.x; nq.oy[im] = ro.y; nq.oz[im] = ro.z;
      nq.dx[im] = rdir.x; nq.dy[im] = rdir.y; nq.dz[im] = rdir.z;
      nq.tx[im] = Tm.x; nq.ty[im] = Tm.y; nq.tz[im] = Tm.z;
      nq.key[im] = mix64(key, SLOT_CHILD_MIRROR);
      nq.slot[im] = slot;
      nq.depth[im] = (uint8_t)(depth + 1);
    } else {
      bc->overflow = 1;
    }
  }
  if (emitShade) {
    if (is < rc.shade_capacity) {
      sl.px[is] = P.x; sl.py[is] = P.y; sl.pz[is] = P.z;
      sl.nx[is] = N.x; sl.ny[is] = N.y; sl.nz[is] = N.z;
      sl.wx[is] = W.x; sl.wy[is] = W.y; sl.wz[is] = W.z;
      sl.key[is] = key;
      sl.slot[is] = slot;
      sl.prim[is] = (uint32_t)hprim;
    } else {
      bc->overflow = 1;
    }
  }
  if (COUNT) {
    wave_count(&dc->shaded_hits, shaded ? 1u : 0u);
    wave_count(&dc->closest_prims, analytic_tests);
  }
}

// ------------------------------------------------------------------ shadow ----
// The occluder walk of light.cpp:21-33 / 70-83 / 125-138 for translucent scenes.
template <bool COUNT, bool LOCAL>
__device__ d3 shadow_walk(const SceneDev &S, d3 light, d3 p, d3 d, bool has_target, d3 target, double lightT, double tmax,
                          const Stack &stack, TravCounters &tc) {
  int mat = -1;
  Hit h = traverse<false, COUNT, LOCAL>(S, p, d, tmax, stack, tc);
  double pt = h.t;
  if (h.prim >= 0) mat = S.objects[S.prim_info[h.prim] & 0x3fffffffu].material;
  int guard = 0;
  while (pt < lightT && guard++ < 4096) {
    d3 entry = p + pt * d;
    p = p + (pt + MIPT_RAY_EPSILON) * d;
    h = traverse<false, COUNT, LOCAL>(S, p, d, tmax, stack, tc);
    pt = h.t;
    if (h.prim >= 0) mat = S.objects[S.prim_info[h.prim] & 0x3fffffffu].material;
    d3 exitp = p + pt * d;
    double dist = length(exitp - entry);
    d3 kt = mat >= 0 ? ld3(S.materials[mat].kt) : d3{0, 0, 0};
    light = light * d3{pow(kt.x, dist), pow(kt.y, dist), pow(kt.z, dist)};
    p = p + (pt + MIPT_RAY_EPSILON) * d;
    h = traverse<false, COUNT, LOCAL>(S, p, d, tmax, stack, tc);
    pt = h.t;
    if (h.prim >= 0) mat = S.objects[S.prim_info[h.prim] & 0x3fffffffu].material;
    if (has_target) lightT = sqrt(dot(target - p, target - p));
  }
  return light;
}

// Geometry of one shadow ray of light `L` from firePos `fire`: sample `s` of an area light
// or the single ray of a point / directional light.  `scale` is the area sample's
// min(1, 1/denom) (light.cpp:139-142), 1 otherwise.
struct ShadowSetup {
  d3 dir, target;
  double lightT, scale;
  bool has_target;
};
__device__ __forceinline__ ShadowSetup shadow_setup(const RenderConst &rc, const LightRec &L, int li, int s, d3 fire, uint64_t key) {
  ShadowSetup r;
  r.target = d3{0, 0, 0};
  r.lightT = rc.t_max;
  r.has_target = true;
  r.scale = 1.0;
  if (L.type == MIPT_LIGHT_DIRECTIONAL) {  // light.cpp:15-35
    r.dir = -ld3(L.orientation);
    r.has_target = false;
  } else if (L.type == MIPT_LIGHT_POINT) {  // light.cpp:63-85
    r.target = ld3(L.position);
    d3 to = r.target - fire;
    r.lightT = sqrt(dot(to, to));
    r.dir = to * (1.0 / r.lightT);  // normalize(to)
  } else {  // light.cpp:105-147
    double uI, vI;
    if (rc.flags & MIPT_FLAG_FIXED_AREA_PATTERN) {
      uI = L.uL * (((double)s + 0.5) / (double)rc.area_samples);
      vI = L.vL * radical_inverse2((uint32_t)s);
    } else {
      uint64_t sl = SLOT_AREA + 2ull * ((uint64_t)li * (uint64_t)rc.area_samples + (uint64_t)s);
      uI = L.uL * u01(mix64(key, sl));
      vI = L.vL * u01(mix64(key, sl + 1));
    }
    // samplePoint: corner + uVec*uI + vVec + vI — vVec whole, vI on every component (light.cpp:109)
    d3 cu = (ld3(L.position) + ld3(L.u) * uI) + ld3(L.v);
    r.target = d3{cu.x + vI, cu.y + vI, cu.z + vI};
    d3 to = r.target - fire;
    r.lightT = sqrt(dot(to, to));
    r.dir = to * (1.0 / r.lightT);  // normalize(to)
    double dd = r.lightT;           // length(fire - target): the same squares, the same sum
    double denom = L.a + L.b * dd + L.c * (dd * dd);
    r.scale = fmin(1.0, 1 / denom);
  }
  return r;
}

// One whole shadow ray (test hook and the translucent kernel): Light::shadowAttenuation's
// per-ray value, area 1/S excluded.
template <bool COUNT, bool WALK, bool LOCAL>
__device__ d3 shadow_ray(const SceneDev &S, const RenderConst &rc, const LightRec &L, int li, int s, d3 fire, uint64_t key,
                         const Stack &stack, TravCounters &tc) {
  ShadowSetup su = shadow_setup(rc, L, li, s, fire, key);
  d3 color = ld3(L.color);
  d3 light;
  if (WALK) {
    light = shadow_walk<COUNT, LOCAL>(S, color, fire, su.dir, su.has_target, su.target, su.lightT, rc.t_max, stack, tc);
  } else {
    // every kt is 0: the walk multiplies by pow(0, dist > 0) = 0 on the first occluder,
    // so "any hit closer than the light" decides it
    Hit h = traverse<true, COUNT, LOCAL>(S, fire, su.dir, fmin(su.lightT, rc.t_max), stack, tc);
    light = h.prim >= 0 ? d3{0, 0, 0} : color;
  }
  return light * su.scale;
}

// Shadow ray `r` of a hit belongs to the light whose block of rays contains it.
__device__ __forceinline__ int light_of_ray(const SceneDev &S, uint32_t r) {
  if (S.n_lights > 4) return (int)S.ray_light[r];  // many lights: a table, one cached load
  int li = 0;
  for (int k = 1; k < S.n_lights; k++)
    if ((uint32_t)S.lights[k].first_ray <= r) li = k;
  return li;
}

// Any-hit shadow rays for opaque scenes, persistent waves with ray replacement (see
// k_intersect).  Work item w = hit * rays_per_hit + ray, hits counted from `first_hit`: the
// rays of one hit are adjacent and enter a wave together, so they share origin and target
// region.  Result per ray: vis[w] = 0 if occluded, else the sample's attenuation scale; in scenes whose lights
// all have a sample-independent scale (SceneDev::vis_bytes) one byte per ray instead, 0 or 1, and the sum
// kernels put the light's scale in: an eighth of the bytes written here and read there.
// LEAN (SceneDev::lean_any: meshes with at most a few analytic primitives beside them, a floor square say): this
// kernel passes the analytic primitives over and k_shadow_analytic, launched after it, clears the records of
// the rays they block — any primitive closer than the light blocks a ray, whichever kernel finds it.  With the
// sphere, box and square tests compiled into it, anywhere, this kernel is 7 % slower on the benchmark scene
// (registers), whose one analytic primitive hardly any shadow ray meets.
constexpr int kLeanAnalyticMax = 8;
#ifdef MIPT_ANY_DEBUG
__device__ unsigned long long g_any_dbg[16384 * 4];  // per wave of the last k_shadow_any launch: start, end, XCD, items of the launch
extern "C" int mipt_debug_any(unsigned long long *out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_any_dbg), sizeof(unsigned long long) * (size_t)n);
}
#endif
// NO_TRI: the scene has no triangle at all (the reference's demo scenes): the triangle test and the own-triangle
// shortcut are not compiled in either (28 B less scratch per lane: spheres.json 674 -> 702 Msamples/s).
template <bool COUNT, bool LOCAL, bool VIS8, bool LEAN, bool NO_TRI = false>  // VIS8: one-byte records (a kernel of its own: the fp64 one keeps its code and registers)
__global__ void __launch_bounds__(kBlock, (LEAN ? MIPT_LEAN_WAVES : MIPT_TRAV_WAVES)) k_shadow_any(SceneDev S, RenderConst rc, ShadeList sl, uint32_t first_hit, uint32_t n_items,
                                                                         double *__restrict__ vis, DeviceCounters *dc) {
  extern __shared__ uint32_t lds_stack[];
#ifdef MIPT_ANY_DEBUG
  const unsigned long long dbg_t0 = wall_clock64();
  unsigned long long dbg_tdyn = 0;
#endif
  const Stack stack{lds_stack + threadIdx.x, kBlock, blockIdx.x * kBlock + threadIdx.x};
  const uint32_t lane = lane_id();
  const uint32_t wave = xcd_major_block() * (kBlock / 64) + threadIdx.x / 64;
  const uint32_t n_waves = gridDim.x * (kBlock / 64);
  const uint32_t R = (uint32_t)S.rays_per_hit;
  TravCounters tc{0, 0};
  Trav T;
  T.cur = MIPT_TRAV_DONE;
  T.sp = 0;
  bool idle = true;
  uint32_t item = 0;
  double scale = 0.0;
  WorkFeed feed;
  feed_init(feed, n_items, wave, n_waves, dc->dyn_next, MIPT_GROUP_ANY);
  bool exhausted = false;
  while (true) {
    unsigned long long idle_mask = __ballot(idle);
    if (idle_mask != 0ull && !exhausted) {
      const FeedTake take = feed_take(feed, (uint32_t)__popcll(idle_mask), n_items, dc->dyn_next);
#ifdef MIPT_ANY_DEBUG
      if (dbg_tdyn == 0 && feed.round >= feed.static_rounds) dbg_tdyn = wall_clock64();
#endif
      exhausted = take.got == 0u;
      const uint32_t rnk = rank_below(idle_mask, lane);
      if (idle && rnk < take.got) {
        uint64_t it = feed_item(take, rnk);
        if (it < n_items) {
          item = (uint32_t)it;
          uint32_t hl = item / R;  // hit within this launch
          uint32_t r = item - hl * R;
          uint32_t h = first_hit + hl;
          int li = light_of_ray(S, r);
          const LightRec &L = S.lights[li];
          d3 P{sl.px[h], sl.py[h], sl.pz[h]};
          d3 N{sl.nx[h], sl.ny[h], sl.nz[h]};
          d3 fire = P + (N * MIPT_RAY_EPSILON) * 3.0;  // material.cpp:117
          ShadowSetup su = shadow_setup(rc, L, li, (int)r - L.first_ray, fire, sl.key[h]);
          scale = su.scale;
          trav_init(T, S, fire, su.dir, fmin(su.lightT, rc.t_max));
          idle = false;
#if MIPT_OWN_PRIM_FIRST
          // A ray that leaves toward the far side of its own surface starts 3e-8 above it and
          // crosses it at once: the triangle the hit lies on is its occluder.  Testing that
          // triangle first ends such a ray without a walk; any primitive closer than the light
          // blocks the ray, so which one is found does not matter.  A ray it does not block
          // (a hit near an edge, a smooth normal on the other side) walks as usual.  Hits on
          // analytic primitives walk: their test costs more than the short walks of the small
          // scenes they make up (spheres.json 665 -> 634 Msamples/s with it, 661 without).
          if (!NO_TRI && dot(N, su.dir) < 0.0 && sl.prim[h] >= (uint32_t)S.n_analytic) {
            if (COUNT) tc.prims++;
            if (prim_test<true, LOCAL, true>(T, S, sl.prim[h])) {
              if (VIS8) reinterpret_cast<uint8_t *>(vis)[item] = 0;
              else vis[item] = 0.0;
              T.cur = MIPT_TRAV_DONE;
              idle = true;
            }
          }
#endif
        }
      }
#if MIPT_OWN_PRIM_FIRST
      // the lanes whose rays ended in the set-up are free again: when enough of them are, take more items first
      if (!exhausted && (uint32_t)__popcll(__ballot(idle)) >= (uint32_t)MIPT_REREFILL_IDLE) continue;
#endif
    }
    if (__ballot(!idle) == 0ull) break;
    const uint32_t want_idle = exhausted ? 64u : kRefillIdle;
    do {
      trav_round<true, COUNT, LOCAL, LEAN, NO_TRI>(T, S, stack, tc);
      if (!idle && T.cur == MIPT_TRAV_DONE) {
        if (VIS8) reinterpret_cast<uint8_t *>(vis)[item] = T.best.prim >= 0 ? 0 : 1;
        else vis[item] = T.best.prim >= 0 ? 0.0 : scale;
        idle = true;
      }
    } while ((uint32_t)__popcll(__ballot(idle)) < want_idle);
  }
  if (COUNT) {
    wave_count(&dc->shadow_nodes, tc.nodes);
    wave_count(&dc->shadow_prims, tc.prims);
  }
#ifdef MIPT_ANY_DEBUG
  if (lane == 0) {
    const uint32_t w = blockIdx.x * (kBlock / 64) + threadIdx.x / 64;
    if (w < 16384u) {
      g_any_dbg[4ull * w + 0] = dbg_t0;
      g_any_dbg[4ull * w + 1] = wall_clock64();
      g_any_dbg[4ull * w + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0xf) | ((unsigned long long)(feed.round - min(feed.round, feed.static_rounds)) << 8) |
                                ((unsigned long long)(__builtin_amdgcn_s_getreg(4 | (31 << 11)) & 0xffffu) << 40);
      g_any_dbg[4ull * w + 3] = dbg_tdyn;
    }
  }
#endif
}

// A light record read through the scalar data path: `p` must be the same for every lane of the wave.  The
// light table does not change while a kernel runs, which a load through the constant address space tells the
// compiler (after the first store of a kernel it would otherwise reload the fields lane by lane).
__device__ __forceinline__ LightRec load_light_uniform(const LightRec *p) {
  typedef const __attribute__((address_space(4))) uint64_t *ConstWords;
  ConstWords c = (ConstWords)(uintptr_t)p;
  LightRec L;
  uint64_t *d = reinterpret_cast<uint64_t *>(&L);
  static_assert(sizeof(LightRec) % 8 == 0, "LightRec is copied in 8-byte words");
#pragma unroll
  for (int i = 0; i < (int)(sizeof(LightRec) / 8); i++) d[i] = c[i];
  return L;
}

// The analytic primitives' share of a LEAN shadow launch.  Every shadow ray of a (hit, light) pair runs from the hit's
// firing point to a point of the light's box of sample points (light.cpp:109), so it stays inside the shaft between the
// two — at parameter t the box [F + t (B.lo - F), F + t (B.hi - F)] — and a primitive whose (slightly grown) world box the
// shaft misses blocks none of them: almost every pair ends there, most of them without a division.
// analytic_candidates: the primitives (bit k = primitive k, at most kLeanAnalyticMax of them) the pair's rays can meet.
// analytic_blocks: does one of those block the pair's sample `sidx`?  (The ray is set up as in k_shadow_any.)
__device__ __forceinline__ uint32_t analytic_candidates(const SceneDev &S, const RenderConst &rc, const LightRec &L, d3 fire, d3 N, uint32_t own) {
  const double f[3] = {fire.x, fire.y, fire.z}, nn[3] = {N.x, N.y, N.z};
  const bool directional = L.type == MIPT_LIGHT_DIRECTIONAL;
  double dl[3], dh[3];  // the light's box of sample points relative to the firing point (directional: the ray's far end)
#pragma unroll
  for (int a = 0; a < 3; a++) {
    double lo, hi;
    if (directional) {
      lo = hi = f[a] - L.orientation[a] * rc.t_max;
    } else if (L.type == MIPT_LIGHT_RECT_AREA) {
      const double base = L.position[a] + L.v[a], du = L.u[a] * L.uL;
      lo = base + fmin(0.0, du) + fmin(0.0, L.vL);
      hi = base + fmax(0.0, du) + fmax(0.0, L.vL);
    } else {
      lo = hi = L.position[a];
    }
    const double grow = 1e-9 * (S.radius + 1.0 + fabs(lo) + fabs(hi));  // far more than the rounding of a sample point
    dl[a] = (lo - grow) - f[a];
    dh[a] = (hi + grow) - f[a];
  }
  // Do all rays of the pair leave the surface the hit lies on (N.dir > 1e-6 for every sample point)?  Then a
  // convex primitive — sphere, box or square — cannot block the rays that start 3e-8 above it, by a margin far
  // beyond the rounding of its test: the pair skips the primitive its hit lies on, whose box the shaft always
  // meets (the hits on a floor square would otherwise set up all their rays to learn nothing).
  double least = 0.0, far2 = 0.0;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    least += fmin(nn[a] * dl[a], nn[a] * dh[a]);
    far2 += fmax(dl[a] * dl[a], dh[a] * dh[a]);
  }
  const bool leaving = least > 1e-6 * sqrt(far2);
  uint32_t mask = 0;
  for (int k = 0; k < S.n_analytic; k++) {
    if (leaving && own == (uint32_t)k) continue;
    const double *b = S.object_boxes + 6u * (S.prim_info[k] & 0x3fffffffu);
    // first without a division: on some axis the box lies beside everything between the firing point and the light
    bool apart = false;
#pragma unroll
    for (int a = 0; a < 3; a++) apart = apart || b[3 + a] - f[a] < fmin(0.0, dl[a]) || b[a] - f[a] > fmax(0.0, dh[a]);
    if (apart) continue;
    // some t in [0, 1] with F + t dl <= b.hi and F + t dh >= b.lo on every axis: two half-lines in t per axis
    double t0 = 0.0, t1 = 1.0;
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const double up = b[3 + a] - f[a], dn = b[a] - f[a];
      if (dl[a] > 0.0) t1 = fmin(t1, up / dl[a]);
      else if (dl[a] < 0.0) t0 = fmax(t0, up / dl[a]);
      else ok = ok && up >= 0.0;
      if (dh[a] > 0.0) t0 = fmax(t0, dn / dh[a]);
      else if (dh[a] < 0.0) t1 = fmin(t1, dn / dh[a]);
      else ok = ok && dn <= 0.0;
    }
    if (ok && t0 <= t1 * (1.0 + 1e-12) + 1e-12) mask |= 1u << k;
  }
  return mask;
}
__device__ __forceinline__ bool analytic_blocks(const SceneDev &S, const RenderConst &rc, const LightRec &L, int li, int sidx, d3 fire, uint64_t key, uint32_t mask) {
  const ShadowSetup su = shadow_setup(rc, L, li, sidx, fire, key);
  Trav T;
  T.org = fire;
  T.dir = su.dir;
  T.best.t = fmin(su.lightT, rc.t_max);
  T.best.prim = -1;
  T.tlimit = 0.0f;
  T.cur = MIPT_TRAV_DONE;
  T.sp = 0;
  for (uint32_t m = mask; m != 0u; m &= m - 1u) {
    const uint32_t k = (uint32_t)__ffs((int)m) - 1u;
    if (!ray_meets_box(T.org, T.dir, T.best.t, S.object_boxes + 6u * (S.prim_info[k] & 0x3fffffffu))) continue;
    if (prim_test<true, false, false, true>(T, S, k)) return true;
  }
  return false;
}

// One lane per hit, light after light: the records of the rays the analytic primitives block are cleared (this launch
// follows k_shadow_any's on the same stream).  Renders whose sum kernel looks at those primitives itself
// (shadow_sum_takes_analytic) do not launch it.
template <bool LOCAL>
__global__ void __launch_bounds__(kBlock) k_shadow_analytic(SceneDev S, RenderConst rc, ShadeList sl, uint32_t first_hit, uint32_t n_hits, double *vis) {
  const uint32_t hl = blockIdx.x * kBlock + threadIdx.x;
  if (hl >= n_hits) return;
  const uint32_t h = first_hit + hl;
  const d3 P{sl.px[h], sl.py[h], sl.pz[h]};
  const d3 N{sl.nx[h], sl.ny[h], sl.nz[h]};
  const uint32_t own = sl.prim[h];
  const d3 fire = P + (N * MIPT_RAY_EPSILON) * 3.0;  // material.cpp:117
  for (int li = 0; li < S.n_lights; li++) {
    const LightRec L = load_light_uniform(S.lights + li);
    const uint32_t mask = analytic_candidates(S, rc, L, fire, N, own);
    if (!mask) continue;
    for (int sidx = 0; sidx < L.samples; sidx++) {
      if (!analytic_blocks(S, rc, L, li, sidx, fire, sl.key[h], mask)) continue;
      const uint64_t item = (uint64_t)hl * (uint32_t)S.rays_per_hit + (uint32_t)(L.first_ray + sidx);
      if (S.vis_bytes) reinterpret_cast<uint8_t *>(vis)[item] = 0;
      else vis[item] = 0.0;
    }
  }
}

// Translucent scenes: the full occluder walk, one lane per ray; vis3 holds the walk's colour
// times the sample scale.
template <bool COUNT, bool LOCAL>
__global__ void __launch_bounds__(kBlock, 2) k_shadow_walk(SceneDev S, RenderConst rc, ShadeList sl, uint64_t first_item, uint32_t n_items,
                                                            double *__restrict__ vis3, DeviceCounters *dc) {
  extern __shared__ uint32_t lds_stack[];
  const Stack stack{lds_stack + threadIdx.x, kBlock, blockIdx.x * kBlock + threadIdx.x};
  TravCounters tc{0, 0};
  const uint32_t R = (uint32_t)S.rays_per_hit;
  // a wave takes 64 items at a time from the work feed (its lanes walk in step: no ray replacement here)
  WorkFeed feed;
  feed_init(feed, n_items, blockIdx.x * (kBlock / 64) + threadIdx.x / 64, gridDim.x * (kBlock / 64), dc->dyn_next, MIPT_GROUP_CLOSEST);
  for (;;) {
    const FeedTake take = feed_take(feed, 64u, n_items, dc->dyn_next);
    if (take.got == 0u) break;
    const uint64_t it = feed_item(take, lane_id());
    if (lane_id() < take.got && it < n_items) {  // (the feed's calls above are made by the whole wave)
      uint64_t w = first_item + it;
      uint32_t h = (uint32_t)(w / R);
      uint32_t r = (uint32_t)(w - (uint64_t)h * R);
      int li = light_of_ray(S, r);
      const LightRec &L = S.lights[li];
      d3 P{sl.px[h], sl.py[h], sl.pz[h]};
      d3 N{sl.nx[h], sl.ny[h], sl.nz[h]};
      d3 fire = P + (N * MIPT_RAY_EPSILON) * 3.0;
      d3 light = shadow_ray<COUNT, true, LOCAL>(S, rc, L, li, (int)r - L.first_ray, fire, sl.key[h], stack, tc);
      vis3[3ull * it + 0] = light.x;
      vis3[3ull * it + 1] = light.y;
      vis3[3ull * it + 2] = light.z;
    }
  }
  if (COUNT) {
    wave_count(&dc->shadow_nodes, tc.nodes);
    wave_count(&dc->shadow_prims, tc.prims);
  }
}

// Direct diffuse term of one shaded hit from its shadow-ray results (material.cpp:111-130):
// sum over lights of shadowAttenuation * distAtten(P) * kd * |n.L(P)| / pi * (1 - metallic),
// added to the sample accumulator with the hit's throughput (folded into W).
template <bool WALK>
__global__ void __launch_bounds__(kBlock) k_shadow_sum(SceneDev S, ShadeList sl, uint32_t first_hit, uint32_t n_hits, const double *__restrict__ vis,
                                                        double *accum) {
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_hits) return;
  uint32_t h = first_hit + i;
  const uint32_t R = (uint32_t)S.rays_per_hit;
  d3 P{sl.px[h], sl.py[h], sl.pz[h]};
  d3 N{sl.nx[h], sl.ny[h], sl.nz[h]};
  d3 Wt{sl.wx[h], sl.wy[h], sl.wz[h]};
  const double *v = vis + (WALK ? 3ull : 1ull) * (uint64_t)i * R;
  d3 total{0, 0, 0};
  for (int li = 0; li < S.n_lights; li++) {
    const LightRec &L = S.lights[li];
    d3 sum{0, 0, 0};
    for (int s = 0; s < L.samples; s++) {
      uint32_t r = (uint32_t)(L.first_ray + s);
      if (WALK) sum = sum + d3{v[3 * r], v[3 * r + 1], v[3 * r + 2]};
      else sum = sum + ld3(L.color) * v[r];
    }
    if (L.type == MIPT_LIGHT_RECT_AREA) sum = sum / (double)L.samples;  // light.cpp:145
    double f = distance_attenuation(L, P) * fabs(dot(N, light_direction(L, P)));
    total = total + (sum * f) * Wt;
  }
  deposit(accum, sl.slot[h], total);
}

// The same sum with the block's visibility values staged through LDS.  A lane that walks its own
// hit's rays_per_hit doubles reads 8 bytes of a different cache line with every load, and with
// tens of waves per CU doing so the lines are evicted between a lane's loads (L1 miss rate 46 %,
// 1 TB/s).  The values of 256 consecutive hits are one contiguous run of 256 * rays_per_hit
// doubles: the block copies it to LDS with fully coalesced loads, then every lane sums its hit's
// values from there.  Opaque scenes with at most four lights and rays_per_hit <= kSumTileRays.
constexpr int kSumTileRays = 32;
#ifndef MIPT_SUM_UNTILED
#define MIPT_SUM_UNTILED 0  // 1: the lane-per-hit kernel for every scene (A/B builds)
#endif
template <bool ALONE>  // ALONE: every hit of the launch belongs to another sample (the hits of the camera rays)
__global__ void __launch_bounds__(kBlock) k_shadow_sum_tiled(SceneDev S, ShadeList sl, uint32_t first_hit, uint32_t n_hits,
                                                              const double *__restrict__ vis, double *accum) {
  extern __shared__ double vis_tile[];
  const uint32_t R = (uint32_t)S.rays_per_hit;
  const uint32_t i0 = blockIdx.x * kBlock;
  const uint32_t hits_here = min((uint32_t)kBlock, n_hits - i0);
  const uint32_t n_vals = hits_here * R;
  const double *src = vis + (uint64_t)i0 * R;
  for (uint32_t e = threadIdx.x; e < n_vals; e += kBlock) vis_tile[e] = src[e];
  __syncthreads();
  if (threadIdx.x >= hits_here) return;
  const uint32_t h = first_hit + i0 + threadIdx.x;
  d3 P{sl.px[h], sl.py[h], sl.pz[h]};
  d3 N{sl.nx[h], sl.ny[h], sl.nz[h]};
  d3 Wt{sl.wx[h], sl.wy[h], sl.wz[h]};
  const double *v = vis_tile + threadIdx.x * R;
  d3 total{0, 0, 0};
  for (int li = 0; li < S.n_lights; li++) {
    const LightRec &L = S.lights[li];
    d3 sum{0, 0, 0};
    for (int s = 0; s < L.samples; s++) sum = sum + ld3(L.color) * v[L.first_ray + s];
    if (L.type == MIPT_LIGHT_RECT_AREA) sum = sum / (double)L.samples;  // light.cpp:145
    double f = distance_attenuation(L, P) * fabs(dot(N, light_direction(L, P)));
    total = total + (sum * f) * Wt;
  }
  if (ALONE) deposit_alone(accum, sl.slot[h], total);
  else deposit(accum, sl.slot[h], total);
}

// The same sum from one-byte visibility records (SceneDev::vis_bytes): a lane reads its hit's rays_per_hit
// bytes, which lie side by side with its neighbours' (no staging through LDS), and adds the light's scale
// for every ray that reached it: the additions, and their order, of the fp64 records' sum.
// ANALYTIC (lean scenes): this kernel also looks at the few analytic primitives the shadow walk passed over — a ray one of them
// blocks counts as a ray that did not arrive — instead of a pass of its own over the same records clearing them first.
template <bool ALONE, bool ANALYTIC>
__global__ void __launch_bounds__(kBlock) k_shadow_sum_bytes(SceneDev S, RenderConst rc, ShadeList sl, uint32_t first_hit, uint32_t n_hits,
                                                              const uint8_t *__restrict__ vis8, double *accum) {
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_hits) return;
  const uint32_t h = first_hit + i;
  d3 P{sl.px[h], sl.py[h], sl.pz[h]};
  d3 N{sl.nx[h], sl.ny[h], sl.nz[h]};
  d3 Wt{sl.wx[h], sl.wy[h], sl.wz[h]};
  const uint8_t *v = vis8 + (uint64_t)i * (uint32_t)S.rays_per_hit;
  const d3 fire = P + (N * MIPT_RAY_EPSILON) * 3.0;  // material.cpp:117
  d3 total{0, 0, 0};
  for (int li = 0; li < S.n_lights; li++) {
    const LightRec &L = S.lights[li];
    const uint32_t mask = ANALYTIC ? analytic_candidates(S, rc, L, fire, N, sl.prim[h]) : 0u;
    d3 sum{0, 0, 0};
    for (int s = 0; s < L.samples; s++) {
      bool arrived = v[L.first_ray + s] != 0;
      if (ANALYTIC && mask != 0u && arrived) arrived = !analytic_blocks(S, rc, L, li, s, fire, sl.key[h], mask);
      sum = sum + ld3(L.color) * (arrived ? L.flat_scale : 0.0);
    }
    if (L.type == MIPT_LIGHT_RECT_AREA) sum = sum / (double)L.samples;  // light.cpp:145
    double f = distance_attenuation(L, P) * fabs(dot(N, light_direction(L, P)));
    total = total + (sum * f) * Wt;
  }
  if (ALONE) deposit_alone(accum, sl.slot[h], total);
  else deposit(accum, sl.slot[h], total);
}

// The same sum for scenes with many lights, one lane per (hit, light): a lane-per-hit loop walks
// rays_per_hit doubles at a stride of rays_per_hit doubles between lanes, which wastes most of
// every cache line when there are hundreds of rays per hit.  Here adjacent lanes read adjacent
// runs of `samples` doubles, evaluate their light once, and the lanes of one hit (adjacent, a
// run may straddle waves) are added with a segmented shuffle reduction before the deposit.
__global__ void __launch_bounds__(kBlock) k_shadow_sum_lights(SceneDev S, ShadeList sl, uint32_t first_hit, uint32_t n_hits,
                                                               const double *__restrict__ vis, double *accum) {
  const uint64_t item = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const uint32_t nL = (uint32_t)S.n_lights;
  const bool live = item < (uint64_t)n_hits * nL;
  const uint32_t i = live ? (uint32_t)(item / nL) : 0xffffffffu;  // hit within this launch
  const uint32_t h = first_hit + (live ? i : 0u);
  d3 c{0, 0, 0};
  if (live) {
    const uint32_t li = (uint32_t)(item - (uint64_t)i * nL);
    const LightRec &L = S.lights[li];
    double sum = 0.0;
    if (S.vis_bytes) {
      const uint8_t *v8 = reinterpret_cast<const uint8_t *>(vis) + (uint64_t)i * (uint32_t)S.rays_per_hit + (uint32_t)L.first_ray;
      for (int s = 0; s < L.samples; s++) sum += v8[s] ? L.flat_scale : 0.0;
    } else {
      const double *v = vis + (uint64_t)i * (uint32_t)S.rays_per_hit + (uint32_t)L.first_ray;
      for (int s = 0; s < L.samples; s++) sum += v[s];
    }
    d3 col = ld3(L.color) * sum;
    if (L.type == MIPT_LIGHT_RECT_AREA) col = col / (double)L.samples;  // light.cpp:145
    d3 P{sl.px[h], sl.py[h], sl.pz[h]};
    d3 N{sl.nx[h], sl.ny[h], sl.nz[h]};
    d3 Wt{sl.wx[h], sl.wy[h], sl.wz[h]};
    double f = distance_attenuation(L, P) * fabs(dot(N, light_direction(L, P)));
    c = (col * f) * Wt;
  }
  const uint32_t lane = lane_id();
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t oi = (uint32_t)__shfl_down((int)i, off);
    const double ox = __shfl_down(c.x, off), oy = __shfl_down(c.y, off), oz = __shfl_down(c.z, off);
    if (lane + (uint32_t)off < 64u && oi == i) c = c + d3{ox, oy, oz};
  }
  const uint32_t pi = (uint32_t)__shfl_up((int)i, 1);
  if (live && (lane == 0u || pi != i)) deposit(accum, sl.slot[h], c);
}

// -------------------------------------------------------------------- tail ----
// The deep levels of a path tree hold few rays: a level of a few thousand rays costs four
// launches whose duration is the latency of their longest walk, and there can be fifty such
// levels.  When a level is down to `tail` rays the rest of the tree is finished by ONE launch:
// every wave follows its rays depth first.  A lane takes a ray, finds its closest hit, evaluates
// the path node (path_node), traces the node's shadow rays itself, one after the other, and
// hands the node's children to its wave's stack in LDS, from which the lanes of that wave take
// their next rays (last in, first out, so the stack stays shallow).  No wave waits for another.
// A wave that has worked through kTailBudget rounds, or whose stack is full, hands what it holds
// (its stack, its lanes' rays, its untouched input) to the output queue and ends; the host then
// goes on with those (another tail launch, or ordinary levels).  A wave hands back at most
// kTailStack + 128 rays beyond its input, so a launch of n rays emits fewer than 5 n + 2048.
// A sample's direct term is added ray by ray here, hit by hit in k_shadow_sum: the sums differ by
// the rounding of fp64 additions in another order.
#ifndef MIPT_TAIL_READY_DIV
#define MIPT_TAIL_READY_DIV 2  // a wave turns to its path nodes when this share of its walks has ended (1 / 2 / 3 / 4 / 8: 5.9 / 4.3 / 4.3 / 4.8 / 5.8 ms for the launch that starts from 2^17 rays of the benchmark scene)
#endif
#ifndef MIPT_TAIL_ROUNDS
#define MIPT_TAIL_ROUNDS (64 * 4096)  // rounds after which a wave of an exporting launch hands its rays back
#endif
#ifndef MIPT_TAIL_CHUNK
#define MIPT_TAIL_CHUNK 16
#endif
constexpr uint32_t kTailChunk = MIPT_TAIL_CHUNK;  // input rays a wave takes at a time (<= 64)
constexpr int kTailBudget = 4096;    // rounds (nodes per lane) a wave works before it hands its stack back
// STACK: pending rays per wave; HITS: room for the shaded hits of a round (64) where the wave traces their shadow
// rays itself, else none
template <int STACK, int HITS>
struct TailWaveStack {
  double v[9][STACK];  // origin, direction, throughput
  uint64_t key[STACK];
  uint32_t slot[STACK];
  uint32_t depth[STACK];
  // the shaded hits of the current round, whose shadow rays the wave's lanes share out
  double hit[9][HITS];  // P, N, W
  uint64_t hit_key[HITS];
  uint32_t hit_slot[HITS];
};
#ifndef MIPT_TAIL_EXPORT_WAVES
#define MIPT_TAIL_EXPORT_WAVES 2  // waves per SIMD the exporting form is compiled for
#endif
#ifndef MIPT_TAIL_EXPORT_STACK
#define MIPT_TAIL_EXPORT_STACK 96
#endif

// EXPORT: the shaded hits are appended to the shade list `sl` (counted in bc->shades) instead of
// having their shadow rays traced here; the host then runs the shadow kernels once over all of
// them.  A wave appends at most 64 x `budget` records, the launch waves x 64 x budget.  The lanes
// of such a launch do not walk in step: a launch that starts from 2^17 rays of the benchmark scene
// has chains of fifty and more bounces to follow, and its duration is that of the slowest chain;
// with every round waiting for the longest of the wave's walks a chain pays the longest of 64 walks
// per bounce instead of its own.
#ifdef MIPT_TAIL_DEBUG
__device__ unsigned long long g_tail_dbg[8192 * 8];
extern "C" int mipt_debug_tail(unsigned long long *out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tail_dbg), sizeof(unsigned long long) * (size_t)n);
}
#endif
template <bool WALK, bool LOCAL, bool EXPORT>
__global__ void __launch_bounds__(kBlock, (EXPORT ? MIPT_TAIL_EXPORT_WAVES : 2)) k_tail(SceneDev S, RenderConst rc, RayQueue q, uint32_t n, RayQueue nq, ShadeList sl, uint32_t budget,
                                                     BatchCounts *bc, double *accum) {
  extern __shared__ uint32_t lds_stack[];
  constexpr int kTailStack = EXPORT ? MIPT_TAIL_EXPORT_STACK : 96;  // pending rays per wave
  typedef TailWaveStack<kTailStack, (EXPORT ? 1 : 64)> WaveStack;
  __shared__ WaveStack pend_all[kBlock / 64];
  const Stack stack{lds_stack + threadIdx.x, kBlock, blockIdx.x * kBlock + threadIdx.x};
  WaveStack &pend = pend_all[threadIdx.x / 64];
  const uint32_t lane = lane_id();
#ifdef MIPT_TAIL_DEBUG
  const uint32_t wave = blockIdx.x * (kBlock / 64) + threadIdx.x / 64;
#endif
  TravCounters tc{0, 0};
  uint32_t top = 0;                 // entries on the wave's stack (wave-uniform)
  bool stop = false;                // wave-uniform: out of budget or out of stack, hand everything back
  // The input is handed out in chunks of kTailChunk rays from one counter, to whichever wave has lanes free and
  // nothing waiting of its own: rays that are neighbours in the queue are neighbours in the scene, and where
  // hits keep spawning two rays (smooth surfaces) a wave that owned 64 such neighbours had five times the
  // average wave's work (the launch lasts as long as its slowest wave).
  uint64_t next_in = 0;             // next input ray of this wave's current chunk
  uint32_t in_left = 0;             // rays of that chunk not yet taken
  bool input_done = false;          // the counter has passed the end of the input (wave-uniform)
  auto next_chunk = [&]() {         // taken when the wave has lanes free for it, not before
    uint32_t c = 0;
    if (lane == 0) c = atomicAdd(&bc->tail_next, 1u);
    c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
    next_in = (uint64_t)c * (uint64_t)kTailChunk;
    input_done = next_in >= (uint64_t)n;
    in_left = input_done ? 0u : min(kTailChunk, (uint32_t)((uint64_t)n - next_in));
  };
  bool active = false;   // the lane holds a ray
  bool walking = false;  // ... whose closest hit is still being looked for
  Trav walk;
  walk.cur = MIPT_TRAV_DONE;
  walk.sp = 0;
  uint32_t appended = 0;  // shade records this wave has exported (wave-uniform)
  d3 o{0, 0, 0}, d{0, 0, 1}, T{0, 0, 0};
  uint64_t key = 0;
  uint32_t slot = 0;
  int depth = 0;
  // appends this lane's ray to the output queue (a wave-aggregated reservation)
  auto emit_out = [&](bool mine, d3 eo, d3 ed, d3 eT, uint64_t ekey, uint32_t eslot, int edepth) {
    unsigned long long m = __ballot(mine);
    if (m == 0ull) return;
    uint32_t base = 0;
    if (lane == (uint32_t)__ffsll((long long)m) - 1u) base = atomicAdd(&bc->next_rays, (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl((int)base, __ffsll((long long)m) - 1));
    if (mine) {
      uint32_t i = base + rank_below(m, lane);
      if (i < rc.capacity) {
        nq.ox[i] = eo.x; nq.oy[i] = eo.y; nq.oz[i] = eo.z;
        nq.dx[i] = ed.x; nq.dy[i] = ed.y; nq.dz[i] = ed.z;
        nq.tx[i] = eT.x; nq.ty[i] = eT.y; nq.tz[i] = eT.z;
        nq.key[i] = ekey;
        nq.slot[i] = eslot;
        nq.depth[i] = (uint8_t)edepth;
      } else {
        bc->overflow = 1;
      }
    }
  };
  // pushes this lane's ray on the wave's stack, or to the output queue when the stack is full
  auto push = [&](bool mine, d3 eo, d3 ed, d3 eT, uint64_t ekey, uint32_t eslot, int edepth) {
    unsigned long long m = __ballot(mine);
    if (m == 0ull) return;
    uint32_t i = top + rank_below(m, lane);
    bool fits = mine && i < (uint32_t)kTailStack;
    if (fits) {
      pend.v[0][i] = eo.x; pend.v[1][i] = eo.y; pend.v[2][i] = eo.z;
      pend.v[3][i] = ed.x; pend.v[4][i] = ed.y; pend.v[5][i] = ed.z;
      pend.v[6][i] = eT.x; pend.v[7][i] = eT.y; pend.v[8][i] = eT.z;
      pend.key[i] = ekey;
      pend.slot[i] = eslot;
      pend.depth[i] = (uint32_t)edepth;
    }
    emit_out(mine && !fits, eo, ed, eT, ekey, eslot, edepth);
    if (top + (uint32_t)__popcll(m) > (uint32_t)kTailStack) stop = true;  // at most 64 rays per wave leave this way: the wave ends its work
    top = min(top + (uint32_t)__popcll(m), (uint32_t)kTailStack);
  };
#ifdef MIPT_TAIL_DEBUG
  const unsigned long long dbg_t0 = wall_clock64();
  unsigned long long dbg_walk = 0, dbg_node = 0, dbg_nodes = 0, dbg_rounds = 0, dbg_trounds = 0;
#endif
  for (int round = 0;; round++) {
    // ---- lanes without a ray take one: from the wave's stack first, then from the input ----
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    unsigned long long idle = __ballot(!active);
    if (idle != 0ull && top > 0 && !stop) {
      uint32_t take = min((uint32_t)__popcll(idle), top);
      uint32_t r = rank_below(idle, lane);
      if (!active && r < take) {
        uint32_t i = top - 1u - r;
        o = d3{pend.v[0][i], pend.v[1][i], pend.v[2][i]};
        d = d3{pend.v[3][i], pend.v[4][i], pend.v[5][i]};
        T = d3{pend.v[6][i], pend.v[7][i], pend.v[8][i]};
        key = pend.key[i];
        slot = pend.slot[i];
        depth = (int)pend.depth[i];
        active = true;
        if (EXPORT) {
          trav_init(walk, S, o, d, rc.t_max);
          walking = true;
        }
      }
      top -= take;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      idle = __ballot(!active);
    }
    while (idle != 0ull && !input_done && !stop) {
      if (in_left == 0u) {
        next_chunk();
        if (input_done) break;
      }
      uint32_t take = min((uint32_t)__popcll(idle), in_left);
      uint32_t r = rank_below(idle, lane);
      if (!active && r < take) {
        uint64_t i = next_in + r;
        o = d3{q.ox[i], q.oy[i], q.oz[i]};
        d = d3{q.dx[i], q.dy[i], q.dz[i]};
        T = d3{q.tx[i], q.ty[i], q.tz[i]};
        key = q.key[i];
        slot = q.slot[i];
        depth = (int)q.depth[i];
        active = true;
        if (EXPORT) {
          trav_init(walk, S, o, d, rc.t_max);
          walking = true;
        }
      }
      next_in += take;
      in_left -= take;
      idle = __ballot(!active);
    }
    if (__ballot(active) == 0ull) {
#ifdef MIPT_TAIL_DEBUG
      if (lane == 0 && wave < 8192u) {
        unsigned long long *g = g_tail_dbg + 8ull * wave;
        g[0] = dbg_t0; g[1] = wall_clock64(); g[2] = dbg_rounds; g[3] = dbg_nodes; g[4] = dbg_walk; g[5] = dbg_node; g[6] = dbg_trounds; g[7] = stop ? 1 : 0;
      }
#endif
      if (stop) {
        // out of budget with rays left: hand the stack and this wave's untouched input back
        for (uint32_t b = 0; b < top; b += 64u) {
          uint32_t i = b + lane;
          bool mine = i < top;
          uint32_t k = mine ? i : 0u;
          emit_out(mine, d3{pend.v[0][k], pend.v[1][k], pend.v[2][k]}, d3{pend.v[3][k], pend.v[4][k], pend.v[5][k]},
                   d3{pend.v[6][k], pend.v[7][k], pend.v[8][k]}, pend.key[k], pend.slot[k], (int)pend.depth[k]);
        }
        while (!input_done) {  // what is left of this wave's chunk, then the chunks nobody has taken
          if (in_left == 0u) {
            next_chunk();
            if (input_done) break;
          }
          bool mine = lane < in_left;
          uint64_t k = mine ? next_in + lane : 0ull;
          emit_out(mine, d3{q.ox[k], q.oy[k], q.oz[k]}, d3{q.dx[k], q.dy[k], q.dz[k]}, d3{q.tx[k], q.ty[k], q.tz[k]}, q.key[k], q.slot[k], (int)q.depth[k]);
          in_left = 0;
        }
      }
      break;
    }
    // ---- closest hits -------------------------------------------------------------------------
    // EXPORT: the walks go on until half of them have ended.  A lane whose walk has ended need not wait for the
    // longest walk of the wave, only for some more to end with it (the path node costs the wave the same for one
    // lane as for 64, ~5 us, so turning to it for every single lane serialises the wave's chains instead).
    // Without EXPORT every lane walks to the end: the lanes then share out the shadow rays of the round's hits,
    // which use the lanes' traversal stacks.
    bool ready = active;
    int hprim = -1;
    double ht = 0.0;
    if (EXPORT) {
      const uint32_t n_walking = (uint32_t)__popcll(__ballot(walking));
      const uint32_t want = max(1u, n_walking / (uint32_t)MIPT_TAIL_READY_DIV);
#ifdef MIPT_TAIL_DEBUG
      const unsigned long long dbg_a = wall_clock64();
#endif
      do {
        trav_round<false, false, LOCAL>(walk, S, stack, tc);
#ifdef MIPT_TAIL_DEBUG
        dbg_trounds++;
#endif
      } while ((uint32_t)__popcll(__ballot(walking && walk.cur == MIPT_TRAV_DONE)) < want);
#ifdef MIPT_TAIL_DEBUG
      dbg_walk += wall_clock64() - dbg_a;
#endif
      ready = walking && walk.cur == MIPT_TRAV_DONE;
      if (ready) {
        hprim = walk.best.prim;
        ht = walk.best.t;
        walking = false;
      }
    } else if (active) {
      Hit h = traverse<false, false, LOCAL>(S, o, d, rc.t_max, stack, tc);
      hprim = h.prim;
      ht = h.t;
    }
    // ---- one path node per lane whose walk has ended ---------------------------------------
    NodeOut r;
    r.emitBounce = r.emitMirror = r.emitShade = r.shaded = false;
#ifdef MIPT_TAIL_DEBUG
    const unsigned long long dbg_b = wall_clock64();
    dbg_rounds++;
    dbg_nodes += (unsigned long long)__popcll(__ballot(ready));
#endif
    if (ready) path_node(S, rc, o, d, T, key, slot, depth, hprim, hprim >= 0 ? ht : 0.0, accum, r);
#ifdef MIPT_TAIL_DEBUG
    dbg_node += wall_clock64() - dbg_b;
#endif
    // ---- the direct diffuse term (material.cpp:111-130): the shadow rays of the round's shaded hits,
    // dealt out over all 64 lanes (a wave that is down to a few long paths would otherwise trace each
    // hit's rays one after the other on one lane) ----
    if (EXPORT) {
      const bool shade = ready && r.emitShade;
      const unsigned long long sm = __ballot(shade);
      appended += (uint32_t)__popcll(sm);
      if (sm != 0ull) {
        const uint32_t first = (uint32_t)__ffsll((long long)sm) - 1u;
        uint32_t base = 0;
        if (lane == first) base = atomicAdd(&bc->shades, (uint32_t)__popcll(sm));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl((int)base, (int)first));
        if (shade) {
          const uint32_t i = base + rank_below(sm, lane);
          if (i < rc.shade_capacity) {
            sl.px[i] = r.P.x; sl.py[i] = r.P.y; sl.pz[i] = r.P.z;
            sl.nx[i] = r.N.x; sl.ny[i] = r.N.y; sl.nz[i] = r.N.z;
            sl.wx[i] = r.W.x; sl.wy[i] = r.W.y; sl.wz[i] = r.W.z;
            sl.key[i] = key;
            sl.slot[i] = slot;
            sl.prim[i] = (uint32_t)hprim;
          } else {
            bc->overflow = 1;
          }
        }
      }
    } else {
      const bool shade = ready && r.emitShade;
      const unsigned long long sm = __ballot(shade);
      const uint32_t n_shade = (uint32_t)__popcll(sm);
      if (n_shade) {
        if (shade) {
          const uint32_t k = rank_below(sm, lane);
          pend.hit[0][k] = r.P.x; pend.hit[1][k] = r.P.y; pend.hit[2][k] = r.P.z;
          pend.hit[3][k] = r.N.x; pend.hit[4][k] = r.N.y; pend.hit[5][k] = r.N.z;
          pend.hit[6][k] = r.W.x; pend.hit[7][k] = r.W.y; pend.hit[8][k] = r.W.z;
          pend.hit_key[k] = key;
          pend.hit_slot[k] = slot;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t R = (uint32_t)S.rays_per_hit, items = n_shade * R;
        for (uint32_t it = lane; it < items; it += 64u) {
          const uint32_t k = it / R, ray = it - k * R;
          const int li = light_of_ray(S, ray);
          const LightRec &L = S.lights[li];
          d3 P{pend.hit[0][k], pend.hit[1][k], pend.hit[2][k]};
          d3 N{pend.hit[3][k], pend.hit[4][k], pend.hit[5][k]};
          d3 W{pend.hit[6][k], pend.hit[7][k], pend.hit[8][k]};
          d3 fire = P + (N * MIPT_RAY_EPSILON) * 3.0;  // material.cpp:117
          d3 v = shadow_ray<false, WALK, LOCAL>(S, rc, L, li, (int)ray - L.first_ray, fire, pend.hit_key[k], stack, tc);
          if (L.type == MIPT_LIGHT_RECT_AREA) v = v / (double)L.samples;  // light.cpp:145
          double f = distance_attenuation(L, P) * fabs(dot(N, light_direction(L, P)));
          deposit(accum, pend.hit_slot[k], (v * f) * W);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
    // ---- children: the bounce ray stays with the lane, the mirror ray goes to the stack ----
    const bool bounce = ready && r.emitBounce, mirror = ready && r.emitMirror;
    push(mirror, r.P + MIPT_RAY_EPSILON * r.N, r.rdir, r.Tm, mix64(key, SLOT_CHILD_MIRROR), slot, depth + 1);  // RayTracer.cpp:270-271
    if (bounce) {
      o = r.P + r.wdir * MIPT_RAY_EPSILON;  // RayTracer.cpp:259
      d = r.wdir;
      T = r.Tb;
      key = mix64(key, SLOT_CHILD_BOUNCE);
      depth = depth + 1;
      if (EXPORT) {
        trav_init(walk, S, o, d, rc.t_max);
        walking = true;
      }
    }
    if (ready) active = bounce;
    // EXPORT: `budget` bounds the records a wave appends (64 per round of whole waves); a round may add 64 more.
    // A bound on the rounds themselves, far beyond what any tree needs, lets every wave reach its end whatever happens.
    if (EXPORT ? (appended + 64u > budget * 64u || round + 1 >= MIPT_TAIL_ROUNDS) : ((uint32_t)round + 1u >= budget)) stop = true;
    if (stop) {  // the lanes' own rays go back too, walked or not
      emit_out(active, o, d, T, key, slot, depth);
      active = false;
      walking = false;
      walk.cur = MIPT_TRAV_DONE;
    }
  }
}

// ----------------------------------------------------------------- resolve ----
__global__ void __launch_bounds__(kBlock) k_resolve(RenderConst rc, const double *__restrict__ accum, const int *__restrict__ rows,
                                                     int64_t first_pixel, int n_pixels, int n_samples, double *d_sum, DeviceCounters *dc) {
  uint32_t lp = blockIdx.x * kBlock + threadIdx.x;
  uint32_t nans = 0;
  if (lp < (uint32_t)n_pixels) {
    int64_t op = first_pixel + lp;
    int j = rows[op / rc.width];
    int i = (int)(op % rc.width);
    double s[3] = {0, 0, 0};
    const double *a = accum + 3ull * (uint64_t)lp * (uint64_t)n_samples;
    for (int k = 0; k < n_samples; k++) {
      bool nan = false;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        double v = a[3 * k + c];
        if (v != v) { nan = true; v = 0.0; }
        v = fmin(fmax(v, 0.0), 1.0);  // glm::clamp, RayTracer.cpp:54
        s[c] += v;
      }
      if (nan) nans++;
    }
    double *dst = d_sum + 3ull * ((uint64_t)i + (uint64_t)j * (uint64_t)rc.width);
    dst[0] += s[0];
    dst[1] += s[1];
    dst[2] += s[2];
  }
  wave_count(&dc->nan_samples, nans);
}

__global__ void __launch_bounds__(kBlock) k_finalize(const double *__restrict__ d_sum, int64_t n, int spp, float *rgb_f32, uint8_t *rgb8) {
  int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= n) return;
  double c = d_sum[idx] / (double)spp;  // RayTracer.cpp:75
  if (rgb_f32) rgb_f32[idx] = (float)c;
  if (rgb8) rgb8[idx] = (uint8_t)(int)(255.0 * c);  // RayTracer.cpp:97-99
}

// -------------------------------------------------------------- test hooks ----
template <bool LOCAL>
__global__ void __launch_bounds__(kBlock) k_intersect_rays(SceneDev S, int64_t n, const double *__restrict__ org, const double *__restrict__ dir,
                                                            double t_max, double *t, int *prim, double *normal) {
  extern __shared__ uint32_t lds_stack[];
  const Stack stack{lds_stack + threadIdx.x, kBlock, blockIdx.x * kBlock + threadIdx.x};
  TravCounters tc{0, 0};
  for (int64_t idx = stack.gtid; idx < n; idx += (int64_t)gridDim.x * kBlock) {
    d3 o{org[3 * idx], org[3 * idx + 1], org[3 * idx + 2]};
    d3 d{dir[3 * idx], dir[3 * idx + 1], dir[3 * idx + 2]};
    Hit h = traverse<false, false, LOCAL>(S, o, d, t_max, stack, tc);
    t[idx] = h.t;
    prim[idx] = h.prim >= 0 ? (int)S.ref_order[h.prim] : -1;
    d3 N{0, 0, 0};
    if (h.prim >= 0) N = hit_normal(S, h.prim, o, d);
    normal[3 * idx] = N.x;
    normal[3 * idx + 1] = N.y;
    normal[3 * idx + 2] = N.z;
  }
}

template <bool LOCAL>
__global__ void __launch_bounds__(kBlock) k_hit_material(SceneDev S, int64_t n, const double *__restrict__ org, const double *__restrict__ dir,
                                                          double t_max, double *uv, double *kd_out, double *ka_out) {
  extern __shared__ uint32_t lds_stack[];
  const Stack stack{lds_stack + threadIdx.x, kBlock, blockIdx.x * kBlock + threadIdx.x};
  TravCounters tc{0, 0};
  for (int64_t idx = stack.gtid; idx < n; idx += (int64_t)gridDim.x * kBlock) {
    d3 o{org[3 * idx], org[3 * idx + 1], org[3 * idx + 2]};
    d3 d{dir[3 * idx], dir[3 * idx + 1], dir[3 * idx + 2]};
    Hit h = traverse<false, false, LOCAL>(S, o, d, t_max, stack, tc);
    double u = 0, v = 0;
    d3 kd{0, 0, 0}, ka{0, 0, 0};
    if (h.prim >= 0) {
      const MatRec *m = S.materials + S.objects[S.prim_info[h.prim] & 0x3fffffffu].material;
      kd = ld3(m->kd);
      ka = ld3(m->ka);
      hit_uv(S, h.prim, o, d, u, v);
      if (m->kd_tex > 0) { const TexRec &tx = S.textures[m->kd_tex - 1]; kd = tex_mapped(tx.data, tx.w, tx.h, u, v); }
      if (m->ka_tex > 0) { const TexRec &tx = S.textures[m->ka_tex - 1]; ka = tex_mapped(tx.data, tx.w, tx.h, u, v); }
      hit_vertex_color(S, h.prim, o, d, kd);
    }
    uv[2 * idx] = u;
    uv[2 * idx + 1] = v;
    kd_out[3 * idx] = kd.x; kd_out[3 * idx + 1] = kd.y; kd_out[3 * idx + 2] = kd.z;
    ka_out[3 * idx] = ka.x; ka_out[3 * idx + 1] = ka.y; ka_out[3 * idx + 2] = ka.z;
  }
}

template <bool WALK, bool LOCAL>
__global__ void __launch_bounds__(kBlock) k_shadow_attenuation(SceneDev S, RenderConst rc, int li, int64_t n, const double *__restrict__ fire,
                                                                const uint64_t *__restrict__ key, double *out) {
  extern __shared__ uint32_t lds_stack[];
  const Stack stack{lds_stack + threadIdx.x, kBlock, blockIdx.x * kBlock + threadIdx.x};
  TravCounters tc{0, 0};
  const LightRec &L = S.lights[li];
  for (int64_t idx = stack.gtid; idx < n; idx += (int64_t)gridDim.x * kBlock) {
    d3 f{fire[3 * idx], fire[3 * idx + 1], fire[3 * idx + 2]};
    d3 sum{0, 0, 0};
    for (int s = 0; s < L.samples; s++) sum = sum + shadow_ray<false, WALK, LOCAL>(S, rc, L, li, s, f, key[idx], stack, tc);
    if (L.type == MIPT_LIGHT_RECT_AREA) sum = sum / (double)L.samples;
    out[3 * idx] = sum.x;
    out[3 * idx + 1] = sum.y;
    out[3 * idx + 2] = sum.z;
  }
}

// ---------------------------------------------------------------- launchers ----
static inline unsigned grid_for(uint64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }
static inline size_t stack_bytes(const SceneDev &S) { return (size_t)S.stack_entries * kBlock * sizeof(uint32_t); }
// Blocks of a traversal kernel one CU holds: the occupancy the kernels are compiled for,
// or what fits its 160 KB of LDS when the stacks need more.
int traversal_blocks_per_cu(const SceneDev &S) {
  size_t per_block = stack_bytes(S) + 256;  // + the kernels' small static LDS
  int by_lds = (int)((160u * 1024u) / per_block);
  int n = kMaxTravWaves < by_lds ? kMaxTravWaves : by_lds;  // the most any of the traversal kernels keeps resident (sizes the stack overflow area)
  return n < 1 ? 1 : n;
}

// Grid of a grid-stride traversal kernel: never more threads than the stack overflow area holds.
static inline unsigned bounded_grid(const SceneDev &S, uint64_t n) {
  uint64_t full = (n + kBlock - 1) / kBlock, cap = S.ovf_stride / kBlock;
  return (unsigned)(full < cap ? full : cap);
}

void launch_generate(const SceneDev &S, const RenderConst &rc, const int *rows, int64_t first_pixel, int n_pixels, int sample0,
                     int n_samples, RayQueue q, hipStream_t st) {
  uint64_t total = (uint64_t)n_pixels * (uint64_t)n_samples;
  if (!total) return;
  hipLaunchKernelGGL(k_generate, dim3(grid_for(total)), dim3(kBlock), 0, st, S, rc, CameraRays{rows, first_pixel, n_pixels, sample0, n_samples}, q);
}

// Grid of a persistent traversal kernel: enough blocks to fill every CU at the compiled
// occupancy, fewer when there is less work than that.
// `waves`: the waves per SIMD (= blocks per CU) the kernel to launch is compiled for.
static inline unsigned persistent_grid(const SceneDev &S, const RenderConst &rc, uint64_t n, int waves = MIPT_TRAV_WAVES) {
  uint64_t full = (n + kBlock - 1) / kBlock;
  const int per_cu_max = traversal_blocks_per_cu(S);  // rc.persist_blocks = CUs x this (the most any traversal kernel keeps resident)
  uint64_t cap = (uint64_t)rc.persist_blocks / (uint64_t)per_cu_max * (uint64_t)(waves < per_cu_max ? waves : per_cu_max);
  return (unsigned)(full < cap ? full : cap);
}

void launch_intersect(const SceneDev &S, const RenderConst &rc, RayQueue q, uint32_t n, uint32_t dup, double *hit_t, int *hit_prim,
                      DeviceCounters *dc, bool count, const CameraRays *cam, hipStream_t st) {
  if (!n) return;
  if (dup < 1u || n % dup != 0u) dup = 1u;
  uint32_t traced = n / dup;
  auto kern = S.local_meshes ? (count ? k_intersect<true, true, false> : k_intersect<false, true, false>)
                             : (count ? k_intersect<true, false, false> : k_intersect<false, false, false>);
  const bool lean = !cam && S.lean_any != 0 && S.n_analytic <= kLeanAnalyticMax;
  if (lean)  // triangles only; k_shade looks at the analytic primitives
    kern = S.local_meshes ? (count ? k_intersect<true, true, false, true> : k_intersect<false, true, false, true>)
                          : (count ? k_intersect<true, false, false, true> : k_intersect<false, false, false, true>);
  if (cam)
    kern = S.local_meshes ? (count ? k_intersect<true, true, true> : k_intersect<false, true, true>)
                          : (count ? k_intersect<true, false, true> : k_intersect<false, false, true>);
  (void)hipMemsetAsync(dc->dyn_next, 0, sizeof(dc->dyn_next), st);
  hipLaunchKernelGGL(kern, dim3(persistent_grid(S, rc, traced, lean ? MIPT_LEAN_CLOSEST_WAVES : MIPT_TRAV_WAVES)), dim3(kBlock), stack_bytes(S), st, S, rc, q, traced, dup, hit_t, hit_prim, dc,
                     cam ? *cam : CameraRays{nullptr, 0, 0, 0, 1});
}

void launch_shade(const SceneDev &S, const RenderConst &rc, RayQueue q, uint32_t n, uint32_t dup, const double *hit_t, const int *hit_prim,
                  RayQueue next, ShadeList sl, BatchCounts *bc, double *accum, DeviceCounters *dc, bool count, const CameraRays *cam, hipStream_t st) {
  if (!n) return;
  if (dup < 1u || n % dup != 0u) dup = 1u;
  auto kern = cam ? (count ? k_shade<true, true> : k_shade<false, true>) : (count ? k_shade<true, false> : k_shade<false, false>);
  hipLaunchKernelGGL(kern, dim3((n + kShadeBlock - 1) / kShadeBlock), dim3(kShadeBlock), 0, st, S, rc, q, n, dup, hit_t, hit_prim, next, sl, bc, accum, dc,
                     cam ? *cam : CameraRays{nullptr, 0, 0, 0, 1});
}

// Moves `n` ray records between queues (spill to / refill from the overflow stack).
__global__ void __launch_bounds__(kBlock) k_copy_rays(RayQueue src, uint64_t src_off, RayQueue dst, uint64_t dst_off, uint32_t n) {
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  uint64_t a = src_off + i, b = dst_off + i;
  dst.ox[b] = src.ox[a]; dst.oy[b] = src.oy[a]; dst.oz[b] = src.oz[a];
  dst.dx[b] = src.dx[a]; dst.dy[b] = src.dy[a]; dst.dz[b] = src.dz[a];
  dst.tx[b] = src.tx[a]; dst.ty[b] = src.ty[a]; dst.tz[b] = src.tz[a];
  dst.key[b] = src.key[a];
  dst.slot[b] = src.slot[a];
  dst.depth[b] = src.depth[a];
}

void launch_copy_rays(RayQueue src, uint64_t src_off, RayQueue dst, uint64_t dst_off, uint32_t n, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_copy_rays, dim3(grid_for(n)), dim3(kBlock), 0, st, src, src_off, dst, dst_off, n);
}

typedef void (*ShadowAnyKernel)(SceneDev, RenderConst, ShadeList, uint32_t, uint32_t, double *, DeviceCounters *);
template <bool COUNT, bool LOCAL>
static ShadowAnyKernel shadow_any_kernel2(bool vis8, bool lean) {
  if (vis8) return lean ? k_shadow_any<COUNT, LOCAL, true, true> : k_shadow_any<COUNT, LOCAL, true, false>;
  return lean ? k_shadow_any<COUNT, LOCAL, false, true> : k_shadow_any<COUNT, LOCAL, false, false>;
}
static ShadowAnyKernel shadow_any_kernel(bool count, bool local, bool vis8, bool lean, bool no_tri) {
  if (no_tri && !local && !lean) {
    if (count) return vis8 ? k_shadow_any<true, false, true, false, true> : k_shadow_any<true, false, false, false, true>;
    return vis8 ? k_shadow_any<false, false, true, false, true> : k_shadow_any<false, false, false, false, true>;
  }
  if (count) return local ? shadow_any_kernel2<true, true>(vis8, lean) : shadow_any_kernel2<true, false>(vis8, lean);
  return local ? shadow_any_kernel2<false, true>(vis8, lean) : shadow_any_kernel2<false, false>(vis8, lean);
}

void launch_shadow_trace(const SceneDev &S, const RenderConst &rc, ShadeList sl, uint64_t first_hit, uint32_t n_hits, double *vis,
                         DeviceCounters *dc, bool count, hipStream_t st, bool analytic_in_sum) {
  uint64_t R = (uint64_t)S.rays_per_hit;
  if (!n_hits || !R) return;
  uint32_t items = (uint32_t)(n_hits * R);
  if (S.translucent) {
    auto kern = S.local_meshes ? (count ? k_shadow_walk<true, true> : k_shadow_walk<false, true>) : (count ? k_shadow_walk<true, false> : k_shadow_walk<false, false>);
    (void)hipMemsetAsync(dc->dyn_next, 0, sizeof(dc->dyn_next), st);
    hipLaunchKernelGGL(kern, dim3(persistent_grid(S, rc, items)), dim3(kBlock), stack_bytes(S), st, S, rc, sl, first_hit * R, items, vis, dc);
  } else {
    const bool lean = S.lean_any != 0 && S.n_analytic <= kLeanAnalyticMax;
    auto kern = shadow_any_kernel(count, S.local_meshes != 0, S.vis_bytes != 0, lean, S.n_analytic == S.n_prims);
    (void)hipMemsetAsync(dc->dyn_next, 0, sizeof(dc->dyn_next), st);
    hipLaunchKernelGGL(kern, dim3(persistent_grid(S, rc, items, lean ? MIPT_LEAN_WAVES : MIPT_TRAV_WAVES)), dim3(kBlock), stack_bytes(S), st, S, rc, sl,
                       (uint32_t)first_hit, items, vis, dc);
    if (lean && S.n_analytic > 0 && !analytic_in_sum)  // what the analytic primitives block, which the walk passed over
      hipLaunchKernelGGL(S.local_meshes ? k_shadow_analytic<true> : k_shadow_analytic<false>, dim3(grid_for(n_hits)), dim3(kBlock), 0, st, S, rc, sl,
                         (uint32_t)first_hit, n_hits, vis);
  }
}

// Lean scenes whose sum kernel looks at the analytic primitives itself.
bool shadow_sum_takes_analytic(const SceneDev &S) {
  // (the lane-per-hit kernel of byte records only: in the kernel of many-light scenes, a lane per (hit, light), the same questions cost
  // more than the pass they replace: config 5's sums 136 -> 247 ms per step for 37 ms less in the shadow launches)
  const bool lean = !S.translucent && S.lean_any != 0 && S.n_analytic <= kLeanAnalyticMax && S.n_analytic > 0;
  return lean && S.n_lights <= 4 && S.vis_bytes != 0;
}

void launch_shadow_sum(const SceneDev &S, const RenderConst &rc, ShadeList sl, uint64_t first_hit, uint32_t n_hits, const double *vis, double *accum, bool alone,
                       hipStream_t st, bool analytic_in_sum) {
  if (!n_hits || !S.rays_per_hit) return;
  const bool analytic = analytic_in_sum && shadow_sum_takes_analytic(S);
  if (!S.translucent && S.n_lights > 4) {
    hipLaunchKernelGGL(k_shadow_sum_lights, dim3(grid_for((uint64_t)n_hits * (uint64_t)S.n_lights)), dim3(kBlock), 0, st, S, sl, (uint32_t)first_hit,
                       n_hits, vis, accum);
    return;
  }
  if (S.vis_bytes) {
    auto kern = analytic ? (alone ? k_shadow_sum_bytes<true, true> : k_shadow_sum_bytes<false, true>) : (alone ? k_shadow_sum_bytes<true, false> : k_shadow_sum_bytes<false, false>);
    hipLaunchKernelGGL(kern, dim3(grid_for(n_hits)), dim3(kBlock), 0, st, S, rc, sl, (uint32_t)first_hit, n_hits, reinterpret_cast<const uint8_t *>(vis), accum);
    return;
  }
  if (!S.translucent && S.rays_per_hit <= kSumTileRays && !MIPT_SUM_UNTILED) {
    hipLaunchKernelGGL(alone ? k_shadow_sum_tiled<true> : k_shadow_sum_tiled<false>, dim3(grid_for(n_hits)), dim3(kBlock),
                       (size_t)kBlock * S.rays_per_hit * sizeof(double), st, S, sl, (uint32_t)first_hit, n_hits, vis, accum);
    return;
  }
  auto kern = S.translucent ? k_shadow_sum<true> : k_shadow_sum<false>;
  hipLaunchKernelGGL(kern, dim3(grid_for(n_hits)), dim3(kBlock), 0, st, S, sl, (uint32_t)first_hit, n_hits, vis, accum);
}

// Grid of the tail kernel: one lane per input ray, at most the two blocks per CU it is compiled for.
// sl == null: the shadow rays of the shaded hits are traced inside the launch; else the hits are
// appended to *sl (the caller has checked that waves x 64 x budget records fit it).
void launch_tail(const SceneDev &S, const RenderConst &rc, RayQueue q, uint32_t n, RayQueue next, const ShadeList *sl, uint32_t budget, BatchCounts *bc,
                 double *accum, hipStream_t st) {
  if (!n) return;
  const unsigned grid = tail_grid(rc, n);
  if (sl) {
    auto kern = S.local_meshes ? k_tail<false, true, true> : k_tail<false, false, true>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), stack_bytes(S), st, S, rc, q, n, next, *sl, budget, bc, accum);
    return;
  }
  auto kern = S.local_meshes ? (S.translucent ? k_tail<true, true, false> : k_tail<false, true, false>) : (S.translucent ? k_tail<true, false, false> : k_tail<false, false, false>);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), stack_bytes(S), st, S, rc, q, n, next, ShadeList{}, (uint32_t)kTailBudget, bc, accum);
}
unsigned tail_grid(const RenderConst &rc, uint32_t n) {
  uint64_t full = ((uint64_t)n + kBlock - 1) / kBlock, cap = (uint64_t)rc.persist_blocks;  // persist_blocks >= CUs, within the stack overflow area
  return (unsigned)(full < cap ? full : cap);
}

void launch_resolve(const RenderConst &rc, const double *accum, const int *rows, int64_t first_pixel, int n_pixels, int n_samples,
                    double *d_sum, DeviceCounters *dc, hipStream_t st) {
  if (!n_pixels) return;
  hipLaunchKernelGGL(k_resolve, dim3(grid_for((uint64_t)n_pixels)), dim3(kBlock), 0, st, rc, accum, rows, first_pixel, n_pixels, n_samples, d_sum, dc);
}

void launch_finalize(const double *d_sum, int64_t n_pixels, int spp, float *rgb_f32, uint8_t *rgb8, hipStream_t st) {
  int64_t n = 3 * n_pixels;
  if (!n) return;
  hipLaunchKernelGGL(k_finalize, dim3(grid_for((uint64_t)n)), dim3(kBlock), 0, st, d_sum, n, spp, rgb_f32, rgb8);
}

void launch_intersect_rays(const SceneDev &S, int64_t n, const double *org, const double *dir, double t_max, double *t, int *prim,
                           double *normal, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(S.local_meshes ? k_intersect_rays<true> : k_intersect_rays<false>, dim3(bounded_grid(S, (uint64_t)n)), dim3(kBlock), stack_bytes(S), st, S, n, org, dir, t_max, t, prim, normal);
}

void launch_hit_material(const SceneDev &S, int64_t n, const double *org, const double *dir, double t_max, double *uv, double *kd, double *ka,
                         hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(S.local_meshes ? k_hit_material<true> : k_hit_material<false>, dim3(bounded_grid(S, (uint64_t)n)), dim3(kBlock), stack_bytes(S), st, S, n, org, dir, t_max, uv, kd, ka);
}

void launch_shadow_attenuation(const SceneDev &S, const RenderConst &rc, int light, int64_t n, const double *fire, const uint64_t *key,
                               double *out, hipStream_t st) {
  if (!n) return;
  auto kern = S.local_meshes ? (S.translucent ? k_shadow_attenuation<true, true> : k_shadow_attenuation<false, true>) : (S.translucent ? k_shadow_attenuation<true, false> : k_shadow_attenuation<false, false>);
  hipLaunchKernelGGL(kern, dim3(bounded_grid(S, (uint64_t)n)), dim3(kBlock), stack_bytes(S), st, S, rc, light, n, fire, key, out);
}

}  // namespace mipt
