/*
 * mipt_dist.h — one frame on several GPUs: C ABI of libmiptdist.so.
 *
 * The reference splits traceImage over CPU threads by column chunks inside the call and joins them before it
 * returns (RayTracer.cpp:482-489).  Here the frame is split over GPUs, one process (rank) per GPU, scene
 * replicated, and the ranks' per-pixel radiance sums are brought together with ONE collective over RCCL (xGMI
 * between the GPUs of a node): a sum-reduce when the ranks split the sample range, a gather of disjoint bands of
 * rows when they split the frame.  Per-sample clamping (RayTracer.cpp:54) happens on the rank that owns the
 * sample, before any sum, so both give the one-GPU frame (bands: bit for bit; samples: to the order of fp64
 * additions).  Plain C, no HIP or RCCL types in the signatures; the library links libmipt.so and librccl.so.
 */
#ifndef MIPT_DIST_H_
#define MIPT_DIST_H_

#include "mipt.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MIPT_DIST_ID_BYTES 128 /* an RCCL unique id */
#define MIPT_SHARD_SAMPLES 0   /* every rank renders the whole frame for its share of [sample_begin, sample_end) */
#define MIPT_SHARD_TILES 1     /* every rank renders all samples of the bands of rows it owns (band b -> rank b % n) */

typedef struct mipt_dist mipt_dist;

/* Rank 0 makes the id of a new group of ranks and hands it to the others by any means (a file, a pipe, MPI). */
int mipt_dist_new_id(unsigned char id[MIPT_DIST_ID_BYTES]);

/* Join the group as `rank` of `n_ranks` on HIP device `device` (collective: every rank calls it). */
int mipt_dist_init(mipt_dist **out, int rank, int n_ranks, const unsigned char id[MIPT_DIST_ID_BYTES], int device);
void mipt_dist_destroy(mipt_dist *d);

/* The share of the job `whole` that `rank` of `n_ranks` renders (no device needed; `whole` must be unsharded for
 * MIPT_SHARD_TILES).  The ranks' shares are disjoint and cover the job.  Returns MIPT_OK or MIPT_ERR_INVALID. */
int mipt_dist_shard_params(const mipt_params *whole, int rank, int n_ranks, int mode, mipt_params *mine);

/* RayTracer::traceImage + waitRender + getBuffer (RayTracer.cpp:462-491, 514-526, 319-324) for the group: every
 * rank calls it with the same `whole` and `mode` and its own context (same scene uploaded).  Each renders its
 * share into a device buffer of 3*w*h doubles, the collective brings the sums to rank 0, and rank 0 resolves them:
 * rgb_f32 / rgb8 as in mipt_render, filled on rank 0 only (other ranks may pass NULL). */
int mipt_dist_render(mipt_dist *d, mipt_ctx *ctx, const mipt_params *whole, int mode, float *rgb_f32, uint8_t *rgb8);

int mipt_dist_rank(const mipt_dist *d);
int mipt_dist_size(const mipt_dist *d);
const char *mipt_dist_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MIPT_DIST_H_ */
