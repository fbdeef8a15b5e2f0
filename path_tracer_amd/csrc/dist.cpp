// dist.cpp — libmiptdist.so: the multi-GPU render of include/mipt_dist.h.  One collective per frame on the
// per-pixel fp64 sums (49.8 MB at 1920 x 1080, 199 MB at 3840 x 2160): ncclReduce(SUM) to rank 0 when the ranks
// split the sample range, grouped ncclSend / ncclRecv of the bands each rank owns when they split the frame (every
// peer's bands travel over its own xGMI link into rank 0).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include "../../include/mipt_dist.h"

struct mipt_dist {
  int rank = 0, size = 1, device = 0;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
};

namespace {
thread_local std::string g_err;
int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIP_OK(call)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? MIPT_ERR_NOMEM : MIPT_ERR_DEVICE, "%s: %s", #call, hipGetErrorString(e_)); \
  } while (0)
#define NCCL_OK(call)                                                                         \
  do {                                                                                        \
    ncclResult_t r_ = (call);                                                                 \
    if (r_ != ncclSuccess) return fail(MIPT_ERR_DEVICE, "%s: %s", #call, ncclGetErrorString(r_)); \
  } while (0)
static_assert(sizeof(ncclUniqueId) <= MIPT_DIST_ID_BYTES, "an RCCL unique id must fit MIPT_DIST_ID_BYTES");
}  // namespace

extern "C" {

const char *mipt_dist_last_error(void) { return g_err.c_str(); }
int mipt_dist_rank(const mipt_dist *d) { return d ? d->rank : -1; }
int mipt_dist_size(const mipt_dist *d) { return d ? d->size : 0; }

int mipt_dist_new_id(unsigned char id[MIPT_DIST_ID_BYTES]) {
  if (!id) return fail(MIPT_ERR_INVALID, "null id");
  ncclUniqueId u;
  NCCL_OK(ncclGetUniqueId(&u));
  std::memset(id, 0, MIPT_DIST_ID_BYTES);
  std::memcpy(id, &u, sizeof(u));
  return MIPT_OK;
}

int mipt_dist_init(mipt_dist **out, int rank, int n_ranks, const unsigned char id[MIPT_DIST_ID_BYTES], int device) {
  if (!out) return fail(MIPT_ERR_INVALID, "null out pointer");
  *out = nullptr;
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(MIPT_ERR_INVALID, "rank %d of %d", rank, n_ranks);
  if (n_ranks > 1 && !id) return fail(MIPT_ERR_INVALID, "null id");
  mipt_dist *d = new (std::nothrow) mipt_dist();
  if (!d) return fail(MIPT_ERR_NOMEM, "out of host memory");
  d->rank = rank;
  d->size = n_ranks;
  d->device = device;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete d;
    return fail(MIPT_ERR_DEVICE, "device %d: %s", device, hipGetErrorString(e));
  }
  if (n_ranks > 1) {  // a group of one needs no communicator: its collective is the identity
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    ncclResult_t r = ncclCommInitRank(&d->comm, n_ranks, u, rank);
    if (r != ncclSuccess) {
      (void)hipStreamDestroy(d->stream);
      delete d;
      return fail(MIPT_ERR_DEVICE, "ncclCommInitRank(rank %d of %d): %s", rank, n_ranks, ncclGetErrorString(r));
    }
  }
  *out = d;
  return MIPT_OK;
}

void mipt_dist_destroy(mipt_dist *d) {
  if (!d) return;
  (void)hipSetDevice(d->device);
  if (d->comm) (void)ncclCommDestroy(d->comm);
  if (d->stream) (void)hipStreamDestroy(d->stream);
  delete d;
}

int mipt_dist_shard_params(const mipt_params *whole, int rank, int n_ranks, int mode, mipt_params *mine) {
  if (!whole || !mine || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(MIPT_ERR_INVALID, "bad arguments");
  *mine = *whole;
  if (mode == MIPT_SHARD_SAMPLES) {
    const int64_t n = (int64_t)whole->sample_end - whole->sample_begin;
    mine->sample_begin = whole->sample_begin + (int32_t)(n * rank / n_ranks);
    mine->sample_end = whole->sample_begin + (int32_t)(n * (rank + 1) / n_ranks);
  } else if (mode == MIPT_SHARD_TILES) {
    if (whole->shard_count != 1) return fail(MIPT_ERR_INVALID, "MIPT_SHARD_TILES splits an unsharded frame");
    mine->shard_index = rank;
    mine->shard_count = n_ranks;
  } else {
    return fail(MIPT_ERR_INVALID, "mode must be MIPT_SHARD_SAMPLES or MIPT_SHARD_TILES");
  }
  return MIPT_OK;
}

int mipt_dist_render(mipt_dist *d, mipt_ctx *ctx, const mipt_params *whole, int mode, float *rgb_f32, uint8_t *rgb8) {
  if (!d || !ctx || !whole) return fail(MIPT_ERR_INVALID, "null argument");
  mipt_params mine;
  int rc = mipt_dist_shard_params(whole, d->rank, d->size, mode, &mine);
  if (rc != MIPT_OK) return rc;
  if (whole->width <= 0 || whole->height <= 0 || whole->band_rows < 1) return fail(MIPT_ERR_INVALID, "bad frame");
  HIP_OK(hipSetDevice(d->device));
  const size_t npx = (size_t)whole->width * whole->height, n = 3 * npx;
  double *d_sum = nullptr;
  float *d_f = nullptr;
  uint8_t *d_b = nullptr;
  HIP_OK(hipMalloc((void **)&d_sum, n * sizeof(double)));
  struct Free {
    double *&s; float *&f; uint8_t *&b;
    ~Free() { if (s) (void)hipFree(s); if (f) (void)hipFree(f); if (b) (void)hipFree(b); }
  } guard{d_sum, d_f, d_b};
  HIP_OK(hipMemsetAsync(d_sum, 0, n * sizeof(double), d->stream));  // traceSetup zero-fills, RayTracer.cpp:427
  rc = mipt_render_accumulate(ctx, &mine, d_sum, d->stream);
  if (rc != MIPT_OK) return fail(rc, "rank %d: %s", d->rank, mipt_last_error(ctx));
  if (d->size > 1) {
    if (mode == MIPT_SHARD_SAMPLES) {
      NCCL_OK(ncclReduce(d_sum, d_sum, n, ncclDouble, ncclSum, 0, d->comm, d->stream));
    } else {
      // band b = rows [b * band_rows, ...) belongs to rank b % size: a contiguous run of the buffer per band
      const int bands = (whole->height + whole->band_rows - 1) / whole->band_rows;
      NCCL_OK(ncclGroupStart());
      for (int b = 0; b < bands; b++) {
        const int owner = b % d->size;
        if (owner == 0) continue;
        const int row0 = b * whole->band_rows, rows = std::min(whole->band_rows, whole->height - row0);
        double *at = d_sum + (size_t)3 * whole->width * row0;
        const size_t count = (size_t)3 * whole->width * rows;
        if (d->rank == 0) NCCL_OK(ncclRecv(at, count, ncclDouble, owner, d->comm, d->stream));
        else if (d->rank == owner) NCCL_OK(ncclSend(at, count, ncclDouble, 0, d->comm, d->stream));
      }
      NCCL_OK(ncclGroupEnd());
    }
  }
  if (d->rank == 0) {
    if (rgb_f32) HIP_OK(hipMalloc((void **)&d_f, n * sizeof(float)));
    if (rgb8) HIP_OK(hipMalloc((void **)&d_b, n));
    rc = mipt_resolve(ctx, d_sum, whole->width, whole->height, whole->spp, d_f, d_b, d->stream);
    if (rc != MIPT_OK) return fail(rc, "%s", mipt_last_error(ctx));
    if (rgb_f32) HIP_OK(hipMemcpyAsync(rgb_f32, d_f, n * sizeof(float), hipMemcpyDeviceToHost, d->stream));
    if (rgb8) HIP_OK(hipMemcpyAsync(rgb8, d_b, n, hipMemcpyDeviceToHost, d->stream));
  }
  HIP_OK(hipStreamSynchronize(d->stream));
  return MIPT_OK;
}

}  // extern "C"
