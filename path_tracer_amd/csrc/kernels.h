// kernels.h — launch interface between the C ABI (api.cpp) and the HIP kernels
// (kernels.hip).  Device pointers only.
#ifndef MIPT_KERNELS_H_
#define MIPT_KERNELS_H_

#include <hip/hip_runtime.h>

#include "../../include/mipt.h"
#include "intersect.h"

namespace mipt {

// Ray queue, structure of arrays: one path-tree edge per entry.
struct RayQueue {
  double *ox, *oy, *oz;  // origin
  double *dx, *dy, *dz;  // unit direction
  double *tx, *ty, *tz;  // throughput of the edge (what its radiance is multiplied by)
  uint64_t *key;         // random-stream key of the path node this ray arrives at
  uint32_t *slot;        // sample accumulator the edge deposits into
  uint8_t *depth;        // tree level of the edge (camera ray = 0)
};

// Shaded hits waiting for their shadow rays.
struct ShadeList {
  double *px, *py, *pz;  // hit point P
  double *nx, *ny, *nz;  // normal
  double *wx, *wy, *wz;  // T * kd / pi * (1 - metallic) / norm
  uint64_t *key;
  uint32_t *slot;
  uint32_t *prim;        // the primitive the hit lies on (storage index)
};

struct DeviceCounters {
  unsigned long long shaded_hits, nan_samples;
  unsigned long long closest_nodes, closest_prims;  // intersect kernel
  unsigned long long shadow_nodes, shadow_prims;    // shadow kernel
  unsigned int dyn_next[8];                         // per-XCD work counters of the running persistent kernel's dynamic tail (not a statistic)
};

// The two append counters take one atomic each from every block of k_shade; they sit in cache lines of
// their own, 256 bytes apart, so that the two streams of atomics do not queue behind each other.
struct BatchCounts {
  uint32_t next_rays;  // entries appended to the next queue
  uint32_t pad0[31];
  uint32_t tail_next;  // the finishing launch's input: chunks of it handed out so far
  uint32_t pad2[31];
  uint32_t shades;     // entries appended to the shade list
  uint32_t overflow;   // a queue was full
  uint32_t pad1[62];
};

struct RenderConst {
  int width, height;
  int area_samples;
  int max_depth;
  double rr_q;
  double t_max;
  uint64_t seed;
  uint32_t flags;
  uint32_t capacity;  // entries per queue
  uint32_t shade_capacity;  // entries of the shade list
  int persist_blocks; // grid of the persistent traversal kernels (CUs x resident blocks per CU)
};

// The camera rays of a batch: entry idx is sample sample0 + idx % n_samples of owned pixel first_pixel + idx / n_samples.
struct CameraRays {
  const int *rows;  // frame rows the call owns, bottom-up (device)
  int64_t first_pixel;
  int n_pixels, sample0, n_samples;
};
void launch_generate(const SceneDev &S, const RenderConst &rc, const int *rows, int64_t first_pixel, int n_pixels,
                     int sample0, int n_samples, RayQueue q, hipStream_t st);
// dup > 1: the queue holds runs of `dup` identical rays; one ray per run is traced and the
// whole run is shaded with its result (n must be a multiple of dup, else dup is ignored).
// cam != null (the level of the camera rays): the rays are computed from their index as k_generate would
// write them, and `q` is not read: a batch of several samples per pixel never stores its camera rays.
void launch_intersect(const SceneDev &S, const RenderConst &rc, RayQueue q, uint32_t n, uint32_t dup, double *hit_t, int *hit_prim,
                      DeviceCounters *dc, bool count, const CameraRays *cam, hipStream_t st);
void launch_shade(const SceneDev &S, const RenderConst &rc, RayQueue q, uint32_t n, uint32_t dup, const double *hit_t,
                  const int *hit_prim, RayQueue next, ShadeList sl, BatchCounts *bc, double *accum,
                  DeviceCounters *dc, bool count, const CameraRays *cam, hipStream_t st);
// Finishes the path trees of the `n` rays of `q` depth first, wave by wave; rays it does not get to (a wave's
// budget of rounds, a full wave stack) are appended to `next` and counted in bc->next_rays.
// sl == null: the launch traces the shadow rays of its shaded hits itself.  Else it appends the hits to *sl
// (bc->shades counts them) for the shadow kernels, every wave appending at most 64 x `budget` records: the caller
// makes sure tail_grid(rc, n) x 4 waves x 64 x budget records fit the list.
unsigned tail_grid(const RenderConst &rc, uint32_t n);
void launch_tail(const SceneDev &S, const RenderConst &rc, RayQueue q, uint32_t n, RayQueue next, const ShadeList *sl, uint32_t budget, BatchCounts *bc,
                 double *accum, hipStream_t st);
void launch_copy_rays(RayQueue src, uint64_t src_off, RayQueue dst, uint64_t dst_off, uint32_t n, hipStream_t st);
// vis: scratch for the per-ray shadow results, room for vis_rays rays (1 double each, 3 in
// translucent scenes); the hits are processed in chunks that fit it.
int traversal_blocks_per_cu(const SceneDev &S);
// analytic_in_sum: the caller's launch_shadow_sum takes the analytic primitives' share of a lean launch (it must pass the same flag,
// and shadow_sum_takes_analytic(S) must hold); else a pass behind the walk clears the records of the rays they block.
bool shadow_sum_takes_analytic(const SceneDev &S);
void launch_shadow_trace(const SceneDev &S, const RenderConst &rc, ShadeList sl, uint64_t first_hit, uint32_t n_hits, double *vis,
                         DeviceCounters *dc, bool count, hipStream_t st, bool analytic_in_sum = false);
// alone: every hit belongs to another sample and no other launch that adds to the samples' sums runs beside this
// one (the hits of a batch's camera rays on the caller's stream): plain additions instead of atomics.
void launch_shadow_sum(const SceneDev &S, const RenderConst &rc, ShadeList sl, uint64_t first_hit, uint32_t n_hits, const double *vis, double *accum, bool alone,
                       hipStream_t st, bool analytic_in_sum = false);
void launch_resolve(const RenderConst &rc, const double *accum, const int *rows, int64_t first_pixel, int n_pixels,
                    int n_samples, double *d_sum, DeviceCounters *dc, hipStream_t st);
void launch_finalize(const double *d_sum, int64_t n_pixels, int spp, float *rgb_f32, uint8_t *rgb8, hipStream_t st);

// Test hooks.
void launch_intersect_rays(const SceneDev &S, int64_t n, const double *org, const double *dir, double t_max,
                           double *t, int *prim, double *normal, hipStream_t st);
void launch_hit_material(const SceneDev &S, int64_t n, const double *org, const double *dir, double t_max, double *uv, double *kd,
                         double *ka, hipStream_t st);
void launch_shadow_attenuation(const SceneDev &S, const RenderConst &rc, int light, int64_t n, const double *fire,
                               const uint64_t *key, double *out, hipStream_t st);

}  // namespace mipt
#endif
