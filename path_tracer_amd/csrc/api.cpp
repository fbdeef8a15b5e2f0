// api.cpp — the C ABI of libmipt (include/mipt.h): context, scene upload, and the
// host loop that drives the wavefront kernels level by level.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/mipt.h"
#include "gpu_build.h"
#include "kernels.h"
#include "scene_build.h"

using namespace mipt;

namespace {
thread_local std::string g_thread_err;

struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
};
}  // namespace

// ---- static workspace of mipt_render_accumulate_static -------------------------------
// One file-scope device array per GPU the library is loaded on, held until the process ends.
// Host code never uses the array's name as an address: mipt_create asks the runtime for the
// array's device address on its own device (hipGetSymbolAddress) and keeps that pointer in
// the context; kernels, memsets and copies get the pointer.
namespace {
constexpr int64_t kStaticBatch = 1 << 14;                            // path samples per wavefront batch
constexpr uint64_t kStaticSpill = 1 << 18;                           // ray records the fixed overflow stack holds
constexpr uint32_t kStaticCap = (uint32_t)(2 * kStaticBatch + 1024);  // queue entries: every ray has <= 2 children
constexpr uint32_t kStaticShadeCap = (uint32_t)(kStaticBatch + 1024);
constexpr int kStaticMaxRows = 16384;                                // frame height limit
constexpr uint32_t kStaticMaxRaysPerHit = 65536;                     // shadow rays per shaded hit (light table entries)
constexpr size_t kStaticEntryBytes = 10 * sizeof(double) + sizeof(uint32_t) + 1;  // = kEntryBytes below
constexpr size_t kStaticShadeEntryBytes = 10 * sizeof(double) + 2 * sizeof(uint32_t);  // = kShadeEntryBytes below
constexpr size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
constexpr size_t kOffQueue0 = 0;
constexpr size_t kOffQueue1 = kOffQueue0 + align256((size_t)kStaticCap * kStaticEntryBytes + 64);
constexpr size_t kOffShade = kOffQueue1 + align256((size_t)kStaticCap * kStaticEntryBytes + 64);
constexpr size_t kOffHits = kOffShade + align256((size_t)kStaticShadeCap * kStaticShadeEntryBytes + 64);
constexpr size_t kOffAccum = kOffHits + align256((size_t)kStaticCap * (sizeof(double) + sizeof(int)) + 64);
constexpr size_t kOffRows = kOffAccum + align256((size_t)kStaticBatch * 3 * sizeof(double));
constexpr size_t kOffCounts = kOffRows + align256((size_t)kStaticMaxRows * sizeof(int));
constexpr size_t kOffDcount = kOffCounts + align256(sizeof(mipt::BatchCounts));
constexpr size_t kOffRayLight = kOffDcount + align256(sizeof(mipt::DeviceCounters));
constexpr size_t kOffSpill = kOffRayLight + align256((size_t)kStaticMaxRaysPerHit * sizeof(uint16_t));
constexpr size_t kOffVis = kOffSpill + align256((size_t)kStaticSpill * kStaticEntryBytes + 64);
constexpr size_t kStaticBytes = (size_t)48 << 20;                    // 48 MiB in all
constexpr uint64_t kStaticVisDoubles = (kStaticBytes - kOffVis) / sizeof(double);
static_assert(kOffVis + ((size_t)16 << 20) <= kStaticBytes, "the visibility buffer needs at least 16 MiB of the static workspace");
}  // namespace
static __device__ __align__(256) unsigned char scratch[kStaticBytes];

struct mipt_ctx {
  int device = 0;
  void *static_ws = nullptr;  // device address of `scratch` on this context's GPU
  std::string err;
  bool have_scene = false;
  HostScene host;  // kept only for its small tables (lights, stats); big arrays are freed after upload
  SceneDev dev;
  std::vector<void *> scene_allocs;
  int64_t scene_bytes = 0;
  int64_t n_prims = 0, n_nodes = 0;
  std::vector<LightRec> lights_host;
  int lights_area_samples = -1;  // area_samples the light records (host copy and device table) are laid out for
  int table_area_samples = -1;   // area_samples the ray -> light table in `ray_light` is built for
  uint64_t scene_serial = 0;     // distinguishes this context's uploads (the static workspace's table remembers whose it holds)

  // render workspace
  uint32_t capacity = 0;
  DevBuf queue[2], shade, hits, accum, rows, counts, dcount;
  DevBuf vis;                // per-shadow-ray results of one chunk of hits
  DevBuf ray_light;          // light index of every shadow ray of a hit
  int n_cus = 256;
  int persist_blocks = 1536; // CUs x blocks per CU for the persistent traversal kernels
  uint32_t tail_rays = 0;        // a level with at most this many rays is handed to the tail launch (0: never)
  long tail_rays_env = -1;       // MIPT_TAIL_RAYS, or -1: chosen per scene at upload
  DevBuf spill;              // overflow stack of ray records (allocated on first use)
  uint64_t spill_cap = 0;    // entries
  // Second launch stream: the shadow rays of a level run on it beside the next level's closest-hit
  // launch on the caller's stream.  Kernels on it get their own stack overflow area and their own
  // persistent-kernel work counters (the second DeviceCounters of `dcount`).
  hipStream_t shadow_stream = nullptr;
  hipEvent_t summed[2] = {nullptr, nullptr};  // per half of the shade list: its last reader on the shadow stream is done
  uint32_t *stack_ovf_b = nullptr;
  bool overlap = true;       // MIPT_OVERLAP=0 keeps every launch on the caller's stream
  uint32_t overlap_rays = 1u << 18;  // levels with at most this many rays use the second stream (MIPT_OVERLAP_RAYS)
  uint32_t fuse_rays = 0;            // mesh scenes: a level with at most this many rays is followed to the end by one launch that
  long fuse_rays_env = -1;           // exports its shaded hits to the shadow kernels (MIPT_FUSE_RAYS, or -1: chosen at upload)
  uint32_t fuse_budget_cap = 0;      // MIPT_FUSE_BUDGET: that launch's waves append at most 64 x this many records each, however much room the
                                     // list has, and the launch is used with any budget >= 1 (tests: waves that stop early and hand rays back)
  mipt_counters totals{};
  // mipt_set_option; -1 = decided by the library (per scene at upload)
  struct Options {
    int builder_gpu = 0;   // "builder": 1 = the BVH is built on the device
    int leaf_max = 2;      // "leaf_max": primitives per leaf the device builder aims for
    int lds_stack = 24;    // "lds_stack": traversal stack entries per lane kept in LDS
    int lean_any = -1;     // "lean_any": 0 keeps the analytic primitives' tests inside the shadow walk
    int vis_bytes = -1;    // "vis_bytes": 0 = fp64 visibility records, 1 = one byte wherever that is exact
  } opt;

  // MIPT_FLAG_PROFILE: event pairs recorded around kernel launches, resolved after
  // the stream has been synchronised
  struct Span { int kind; hipEvent_t a, b; uint64_t items; };
  std::vector<hipEvent_t> event_pool;
  size_t events_used = 0;
  std::vector<Span> spans;
  mipt_kernel_times ktimes{};
};

namespace {

int set_err(mipt_ctx *c, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  g_thread_err = buf;
  return code;
}

#define HIP_TRY(ctx, call)                                                                       \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      return set_err(ctx, e_ == hipErrorOutOfMemory ? MIPT_ERR_NOMEM : MIPT_ERR_DEVICE, "%s: %s", #call, hipGetErrorString(e_)); \
  } while (0)

int ensure(mipt_ctx *c, DevBuf &b, size_t bytes) {
  if (b.bytes >= bytes && b.p) return MIPT_OK;
  if (b.p) {
    HIP_TRY(c, hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
  }
  if (bytes == 0) bytes = 16;
  HIP_TRY(c, hipMalloc(&b.p, bytes));
  b.bytes = bytes;
  return MIPT_OK;
}

void release(DevBuf &b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.bytes = 0;
}

template <typename T>
int upload(mipt_ctx *c, const std::vector<T> &v, const T **dst) {
  *dst = nullptr;
  size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);
  void *p = nullptr;
  HIP_TRY(c, hipMalloc(&p, bytes));
  c->scene_allocs.push_back(p);
  c->scene_bytes += (int64_t)bytes;
  if (!v.empty()) HIP_TRY(c, hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *dst = reinterpret_cast<const T *>(p);
  return MIPT_OK;
}

hipEvent_t take_event(mipt_ctx *c) {
  if (c->events_used == c->event_pool.size()) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    c->event_pool.push_back(e);
  }
  return c->event_pool[c->events_used++];
}

// Brackets one kernel launch with events on its own stream.
struct ProfScope {
  mipt_ctx *c;
  hipStream_t st;
  bool on;
  mipt_ctx::Span span;
  ProfScope(mipt_ctx *ctx, bool enabled, int kind, uint64_t items, hipStream_t stream) : c(ctx), st(stream), on(enabled) {
    if (!on) return;
    span.kind = kind;
    span.items = items;
    span.a = take_event(c);
    span.b = take_event(c);
    if (!span.a || !span.b) { on = false; return; }
    (void)hipEventRecord(span.a, st);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(span.b, st);
    c->spans.push_back(span);
  }
};

void collect_spans(mipt_ctx *c) {
  for (const auto &sp : c->spans) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
      c->ktimes.ms[sp.kind] += (double)ms;
      c->ktimes.launches[sp.kind] += 1;
      c->ktimes.items[sp.kind] += sp.items;
    }
  }
  c->spans.clear();
  c->events_used = 0;
}

void free_scene(mipt_ctx *c) {
  for (void *p : c->scene_allocs) (void)hipFree(p);
  c->scene_allocs.clear();
  c->scene_bytes = 0;
  c->have_scene = false;
}

RayQueue carve_queue(void *base, uint32_t cap) {
  RayQueue q;
  double *d = reinterpret_cast<double *>(base);
  q.ox = d + 0ull * cap; q.oy = d + 1ull * cap; q.oz = d + 2ull * cap;
  q.dx = d + 3ull * cap; q.dy = d + 4ull * cap; q.dz = d + 5ull * cap;
  q.tx = d + 6ull * cap; q.ty = d + 7ull * cap; q.tz = d + 8ull * cap;
  q.key = reinterpret_cast<uint64_t *>(d + 9ull * cap);
  q.slot = reinterpret_cast<uint32_t *>(d + 10ull * cap);
  q.depth = reinterpret_cast<uint8_t *>(q.slot + cap);
  return q;
}
ShadeList carve_shade(void *base, uint32_t cap) {
  ShadeList s;
  double *d = reinterpret_cast<double *>(base);
  s.px = d + 0ull * cap; s.py = d + 1ull * cap; s.pz = d + 2ull * cap;
  s.nx = d + 3ull * cap; s.ny = d + 4ull * cap; s.nz = d + 5ull * cap;
  s.wx = d + 6ull * cap; s.wy = d + 7ull * cap; s.wz = d + 8ull * cap;
  s.key = reinterpret_cast<uint64_t *>(d + 9ull * cap);
  s.slot = reinterpret_cast<uint32_t *>(d + 10ull * cap);
  s.prim = s.slot + cap;
  return s;
}
ShadeList offset_shade(ShadeList s, uint32_t first) {  // the list that starts at entry `first` of `s`
  s.px += first; s.py += first; s.pz += first;
  s.nx += first; s.ny += first; s.nz += first;
  s.wx += first; s.wy += first; s.wz += first;
  s.key += first;
  s.slot += first;
  s.prim += first;
  return s;
}
constexpr size_t kEntryBytes = 10 * sizeof(double) + sizeof(uint32_t) + 1;  // per queue entry
constexpr size_t kShadeEntryBytes = 10 * sizeof(double) + 2 * sizeof(uint32_t);  // per shade-list entry
static_assert(kEntryBytes == kStaticEntryBytes, "the static workspace is laid out for this entry size");
static_assert(kShadeEntryBytes == kStaticShadeEntryBytes, "the static workspace is laid out for this entry size");

int check_params(mipt_ctx *c, const mipt_params *p) {
  if (!p) return set_err(c, MIPT_ERR_INVALID, "null params");
  if (p->width <= 0 || p->height <= 0) return set_err(c, MIPT_ERR_INVALID, "width and height must be positive");
  if ((int64_t)p->width * p->height > (1ll << 30)) return set_err(c, MIPT_ERR_LIMIT, "frame larger than 2^30 pixels");
  if (p->spp <= 0) return set_err(c, MIPT_ERR_INVALID, "spp must be positive");
  if (p->area_samples <= 0 || p->area_samples > 4096) return set_err(c, MIPT_ERR_INVALID, "area_samples must be in 1..4096");
  if (!(p->rr_q >= 0.0 && p->rr_q < 1.0)) return set_err(c, MIPT_ERR_INVALID, "rr_q must be in [0, 1)");
  if (!(p->t_max > 0.0)) return set_err(c, MIPT_ERR_INVALID, "t_max must be positive");
  if (p->max_depth <= 0 || p->max_depth > 255) return set_err(c, MIPT_ERR_INVALID, "max_depth must be in 1..255");
  if (p->shard_count < 1 || p->shard_index < 0 || p->shard_index >= p->shard_count || p->band_rows < 1)
    return set_err(c, MIPT_ERR_INVALID, "bad shard parameters");
  if (p->sample_begin < 0 || p->sample_end > p->spp || p->sample_begin > p->sample_end)
    return set_err(c, MIPT_ERR_INVALID, "bad sample range");
  if (p->batch_samples < 0) return set_err(c, MIPT_ERR_INVALID, "batch_samples must be >= 0");
  return MIPT_OK;
}

// Makes room for `extra` more entries on the overflow stack that holds `used`.
int ensure_spill(mipt_ctx *c, uint64_t used, uint64_t extra, hipStream_t st) {
  uint64_t need = used + extra;
  if (need <= c->spill_cap) return MIPT_OK;
  uint64_t cap = std::max<uint64_t>(2 * need, 1u << 20);
  if (cap > 0x7fffff00ull || cap * kEntryBytes > (64ull << 30))
    return set_err(c, MIPT_ERR_LIMIT,
                   "path tree outgrew the overflow stack (%llu rays): the scene multiplies rays faster than they die; lower max_depth",
                   (unsigned long long)need);
  void *np = nullptr;
  HIP_TRY(c, hipMalloc(&np, cap * kEntryBytes + 64));
  if (used) {
    launch_copy_rays(carve_queue(c->spill.p, (uint32_t)c->spill_cap), 0, carve_queue(np, (uint32_t)cap), 0, (uint32_t)used, st);
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
      (void)hipFree(np);
      return set_err(c, MIPT_ERR_DEVICE, "growing the overflow stack: %s", hipGetErrorString(e));
    }
  }
  if (c->spill.p) HIP_TRY(c, hipFree(c->spill.p));
  c->spill.p = np;
  c->spill.bytes = cap * kEntryBytes + 64;
  c->spill_cap = cap;
  return MIPT_OK;
}

RenderConst make_const(const mipt_ctx *c, const mipt_params *p, uint32_t capacity, uint32_t shade_capacity = 0) {
  RenderConst rc;
  rc.persist_blocks = c->persist_blocks;
  rc.width = p->width;
  rc.height = p->height;
  rc.area_samples = p->area_samples;
  rc.max_depth = p->max_depth;
  rc.rr_q = p->rr_q;
  rc.t_max = p->t_max;
  rc.seed = p->seed;
  rc.flags = p->flags;
  rc.capacity = capacity;
  rc.shade_capacity = shade_capacity ? shade_capacity : capacity;
  return rc;
}

// The per-hit shadow-ray block depends on area_samples: every light's ray count and first ray.  Lays the light
// records out for `area_samples` (host copy and device table) if they are not; returns the rays per hit.
int layout_lights(mipt_ctx *c, int area_samples) {
  if (c->lights_area_samples != area_samples) {
    int first = 0;
    for (LightRec &L : c->lights_host) {
      L.samples = (L.type == MIPT_LIGHT_RECT_AREA) ? area_samples : 1;
      L.first_ray = first;
      first += L.samples;
    }
    c->lights_area_samples = -1;  // until the device has them
    if (!c->lights_host.empty())
      HIP_TRY(c, hipMemcpy(const_cast<LightRec *>(c->dev.lights), c->lights_host.data(), c->lights_host.size() * sizeof(LightRec), hipMemcpyHostToDevice));
    c->lights_area_samples = area_samples;
  }
  return MIPT_OK;
}
int rays_per_hit_for(const mipt_ctx *c, int area_samples) {
  int64_t n = 0;
  for (const LightRec &L : c->lights_host) n += (L.type == MIPT_LIGHT_RECT_AREA) ? area_samples : 1;
  return n > 0x7fffffff ? 0x7fffffff : (int)n;
}

// The allocating entry points' view of the lights: records laid out, and the context's own ray -> light table.
int sync_lights(mipt_ctx *c, int area_samples) {
  int r = layout_lights(c, area_samples);
  if (r != MIPT_OK) return r;
  const int first = rays_per_hit_for(c, area_samples);
  if (c->table_area_samples != area_samples) {
    std::vector<uint16_t> table((size_t)std::max(first, 1));
    for (size_t k = 0; k < c->lights_host.size(); k++)
      for (int s = 0; s < c->lights_host[k].samples; s++) table[(size_t)(c->lights_host[k].first_ray + s)] = (uint16_t)k;
    if ((r = ensure(c, c->ray_light, table.size() * sizeof(uint16_t))) != MIPT_OK) return r;
    c->table_area_samples = -1;
    HIP_TRY(c, hipMemcpy(c->ray_light.p, table.data(), table.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    c->table_area_samples = area_samples;
  }
  c->dev.rays_per_hit = first;
  c->dev.ray_light = reinterpret_cast<const uint16_t *>(c->ray_light.p);
  return MIPT_OK;
}

// Whose ray -> light table the static workspace of each GPU holds (every context on a GPU shares that array).
struct StaticTableOwner {
  const mipt_ctx *ctx = nullptr;
  uint64_t scene_serial = 0;
  int area_samples = -1;
};
std::mutex g_static_owner_lock;
StaticTableOwner g_static_owner[64];
std::atomic<uint64_t> g_scene_serial{0};

}  // namespace

extern "C" {

int mipt_version(void) { return MIPT_VERSION; }

const char *mipt_last_error(const mipt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_thread_err.c_str(); }

void mipt_default_params(mipt_params *p, int width, int height) {
  if (!p) return;
  std::memset(p, 0, sizeof(*p));
  p->width = width;
  p->height = height;
  p->spp = 100;          // RayTracer.cpp:61
  p->area_samples = 10;  // light.cpp:118
  p->rr_q = 0.1;         // RayTracer.cpp:229
  p->t_max = 1000.0;     // scene.cpp:129
  p->max_depth = 64;
  p->flags = 0;
  p->seed = 1;
  p->shard_index = 0;
  p->shard_count = 1;
  p->band_rows = 8;
  p->sample_begin = 0;
  p->sample_end = p->spp;
  p->batch_samples = 0;
}

int mipt_create(mipt_ctx **out, int device) {
  if (!out) return set_err(nullptr, MIPT_ERR_INVALID, "null out pointer");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return set_err(nullptr, MIPT_ERR_DEVICE, "no HIP device available (%s)", hipGetErrorString(e));
  if (device < 0 || device >= n) return set_err(nullptr, MIPT_ERR_INVALID, "device %d out of range (have %d)", device, n);
  mipt_ctx *c = new (std::nothrow) mipt_ctx();
  if (!c) return set_err(nullptr, MIPT_ERR_NOMEM, "out of host memory");
  c->device = device;
  std::memset(&c->dev, 0, sizeof(c->dev));
  if (const char *e = debug_env("MIPT_TAIL_RAYS")) {  // tuning knob (MIPT_DEBUG=1 only, like the others)
    long v = std::atol(e);
    if (v >= 0 && v <= (1l << 24)) c->tail_rays_env = v;
  }
  // the static workspace of this GPU: every device holds its own copy of the array at its own address
  {
    void *ptr = nullptr;
    hipError_t se = hipSetDevice(device);
    if (se == hipSuccess) se = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (se != hipSuccess) {
      delete c;
      return set_err(nullptr, MIPT_ERR_DEVICE, "hipGetSymbolAddress(static workspace) on device %d: %s", device, hipGetErrorString(se));
    }
    c->static_ws = ptr;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
    c->n_cus = prop.multiProcessorCount;
  if (const char *e = debug_env("MIPT_OVERLAP")) c->overlap = std::atoi(e) != 0;  // measurement knobs
  if (const char *e = debug_env("MIPT_FUSE_RAYS")) {
    long v = std::atol(e);
    if (v >= 0 && v <= (1l << 24)) c->fuse_rays_env = v;
  }
  if (const char *e = debug_env("MIPT_FUSE_BUDGET")) {
    long v = std::atol(e);
    if (v >= 1 && v <= 4096) c->fuse_budget_cap = (uint32_t)v;
  }
  if (const char *e = debug_env("MIPT_OVERLAP_RAYS")) {
    long v = std::atol(e);
    if (v >= 0 && v <= (1l << 30)) c->overlap_rays = (uint32_t)v;
  }
  if (c->overlap) {
    hipError_t se = hipStreamCreateWithFlags(&c->shadow_stream, hipStreamNonBlocking);
    for (int k = 0; k < 2 && se == hipSuccess; k++) se = hipEventCreateWithFlags(&c->summed[k], hipEventDisableTiming);
    if (se != hipSuccess) {
      mipt_destroy(c);
      return set_err(nullptr, MIPT_ERR_DEVICE, "second launch stream on device %d: %s", device, hipGetErrorString(se));
    }
  }
  *out = c;
  return MIPT_OK;
}

void mipt_destroy(mipt_ctx *c) {
  if (!c) return;
  {
    std::lock_guard<std::mutex> lock(g_static_owner_lock);  // a later context may get this one's address
    for (StaticTableOwner &own : g_static_owner)
      if (own.ctx == c) own = StaticTableOwner{};
  }
  (void)hipSetDevice(c->device);
  free_scene(c);
  release(c->queue[0]);
  release(c->queue[1]);
  release(c->shade);
  release(c->hits);
  release(c->accum);
  release(c->rows);
  release(c->counts);
  release(c->dcount);
  release(c->spill);
  release(c->vis);
  release(c->ray_light);
  for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->summed)
    if (e) (void)hipEventDestroy(e);
  if (c->shadow_stream) (void)hipStreamDestroy(c->shadow_stream);
  delete c;
}

int mipt_set_option(mipt_ctx *c, const char *name, int64_t value) {
  if (!c || !name) return set_err(c, MIPT_ERR_INVALID, "null argument");
  auto in = [&](int64_t lo, int64_t hi) { return value >= lo && value <= hi; };
  const std::string n(name);
  if (n == "builder" && in(0, 1)) c->opt.builder_gpu = (int)value;
  else if (n == "leaf_max" && in(1, 8)) c->opt.leaf_max = (int)value;
  else if (n == "lds_stack" && in(1, 24)) c->opt.lds_stack = (int)value;
  else if (n == "lean_any" && in(-1, 1)) c->opt.lean_any = (int)value;
  else if (n == "vis_bytes" && in(-1, 1)) c->opt.vis_bytes = (int)value;
  else if (n == "tail_rays" && in(-1, 1 << 24)) c->tail_rays_env = (long)value;
  else if (n == "fuse_rays" && in(-1, 1 << 24)) c->fuse_rays_env = (long)value;
  else if (n == "overlap_rays" && in(0, 1 << 30)) c->overlap_rays = (uint32_t)value;
  else return set_err(c, MIPT_ERR_INVALID, "unknown option or value out of range: %s = %lld", name, (long long)value);
  return MIPT_OK;
}

int mipt_upload_scene(mipt_ctx *c, const mipt_scene_desc *desc) {
  if (!c) return set_err(nullptr, MIPT_ERR_INVALID, "null context");
  HIP_TRY(c, hipSetDevice(c->device));
  free_scene(c);
  std::string err;
  HostScene &H = c->host;
  int rcode;
  // MIPT_BUILDER=gpu: build the BVH on the GPU (nearest-neighbour clustering, tens of milliseconds, ~1.15x slower to traverse)
  // instead of the host's binned SAH; small scenes always use the host builder
  const char *bsel = debug_env("MIPT_BUILDER");
  int64_t prim_guess = desc ? desc->n_objects : 0;
  if (desc && desc->meshes)
    for (int k = 0; k < desc->n_meshes; k++) prim_guess += desc->meshes[k].n_faces;
  const bool gpu_build = (bsel ? std::strcmp(bsel, "gpu") == 0 : c->opt.builder_gpu != 0) && prim_guess >= 4096;
  try {
    rcode = build_host_scene(desc, &H, &err, !gpu_build);
  } catch (const std::bad_alloc &) {
    return set_err(c, MIPT_ERR_NOMEM, "out of host memory while building the scene");
  }
  if (rcode != MIPT_OK) return set_err(c, rcode, "%s", err.c_str());
  SceneDev &D = c->dev;
  std::memset(&D, 0, sizeof(D));
  int r;
  bool gpu_fallback = false;
  if (gpu_build && H.prim_info.size() >= 3) {
    // flat arrays in reference order -> tree and leaf order on the device
    const uint32_t n = (uint32_t)H.prim_info.size();
    std::vector<void *> staging;
    auto stage = [&](const void *src, size_t bytes) -> void * {
      void *p = nullptr;
      if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
      staging.push_back(p);
      if (bytes && hipMemcpy(p, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
      return p;
    };
    auto drop_staging = [&]() { for (void *p : staging) (void)hipFree(p); };
    void *s_boxes = stage(H.prim_boxes.data(), H.prim_boxes.size() * sizeof(float));
    void *s_info = stage(H.prim_info.data(), (size_t)n * sizeof(uint32_t));
    void *s_tris = stage(H.tris.data(), (size_t)n * sizeof(Tri));
    void *s_norm = H.tri_normals.empty() ? nullptr : stage(H.tri_normals.data(), (size_t)n * sizeof(TriNormals));
    void *s_uv = H.tri_uvs.empty() ? nullptr : stage(H.tri_uvs.data(), (size_t)n * sizeof(TriUV));
    void *s_col = H.tri_colors.empty() ? nullptr : stage(H.tri_colors.data(), (size_t)n * sizeof(TriColor));
    if (!s_boxes || !s_info || !s_tris || (!H.tri_normals.empty() && !s_norm) || (!H.tri_uvs.empty() && !s_uv) ||
        (!H.tri_colors.empty() && !s_col)) {
      drop_staging();
      return set_err(c, MIPT_ERR_NOMEM, "out of device memory while staging the scene");
    }
    int leaf_max = c->opt.leaf_max;
    if (const char *lm = debug_env("MIPT_LEAF_MAX")) {
      int v = std::atoi(lm);
      if (v >= 1 && v <= 8) leaf_max = v;
    }
    GpuBvh g;
    auto t_build = std::chrono::steady_clock::now();
    hipError_t ge = gpu_build_bvh((const float *)s_boxes, (const uint32_t *)s_info, n, H.bounds_min, H.bounds_max, leaf_max, &g, nullptr);
    if (debug_env("MIPT_BUILD_TIMING"))
      std::fprintf(stderr, "[mipt build] %-28s %8.3f s (%d nodes, %d levels, %d clustering rounds)\n", "GPU clustering BVH",
                   std::chrono::duration<double>(std::chrono::steady_clock::now() - t_build).count(), g.n_nodes, g.levels, g.rounds);
    if (ge == hipErrorUnknown || ge == hipErrorInvalidValue) {
      // the clustering did not converge (box areas that overflow fp32 leave every cluster without a nearest neighbour):
      // the host builder, which works in fp64, takes the scene instead
      drop_staging();
      (void)hipGetLastError();
      gpu_fallback = true;
    } else if (ge != hipSuccess) {
      drop_staging();
      return set_err(c, ge == hipErrorOutOfMemory ? MIPT_ERR_NOMEM : MIPT_ERR_DEVICE, "GPU BVH build: %s", hipGetErrorString(ge));
    }
    if (!gpu_fallback) {
      c->scene_allocs.push_back(g.nodes);
      c->scene_allocs.push_back(g.order);
      c->scene_bytes += (int64_t)n * (int64_t)(sizeof(Node4) + sizeof(uint32_t));
      D.nodes = g.nodes;
      D.ref_order = g.order;  // the flat arrays were in reference order
      auto gathered = [&](const void *src, uint32_t bytes) -> void * {
        void *p = nullptr;
        if (hipMalloc(&p, (size_t)n * bytes) != hipSuccess) return nullptr;
        c->scene_allocs.push_back(p);
        c->scene_bytes += (int64_t)n * bytes;
        gpu_gather_records(p, src, g.order, n, bytes, nullptr);
        return p;
      };
      D.tris = (const Tri *)gathered(s_tris, sizeof(Tri));
      D.prim_info = (const uint32_t *)gathered(s_info, sizeof(uint32_t));
      if (s_norm) D.tri_normals = (const TriNormals *)gathered(s_norm, sizeof(TriNormals));
      if (s_uv) D.tri_uvs = (const TriUV *)gathered(s_uv, sizeof(TriUV));
      if (s_col) D.tri_colors = (const TriColor *)gathered(s_col, sizeof(TriColor));
      hipError_t se = hipDeviceSynchronize();
      drop_staging();
      if (!D.tris || !D.prim_info || (s_norm && !D.tri_normals) || (s_uv && !D.tri_uvs) || (s_col && !D.tri_colors)) return set_err(c, MIPT_ERR_NOMEM, "out of device memory for the scene");
      if (se != hipSuccess) return set_err(c, MIPT_ERR_DEVICE, "GPU BVH build: %s", hipGetErrorString(se));
      H.n_analytic = g.n_analytic;
      H.max_depth = g.levels;
      H.stack_need = 3 * g.levels + 1;  // at most three siblings wait per level
      c->n_nodes = g.n_nodes;
    }
  }
  if (!(gpu_build && H.prim_info.size() >= 3) || gpu_fallback) {
    if (gpu_build) {  // too few primitives after dropping degenerate faces, or the device builder gave up: host builder
      if ((rcode = build_host_scene(desc, &H, &err, true)) != MIPT_OK) return set_err(c, rcode, "%s", err.c_str());
    }
    if ((r = upload(c, H.nodes, &D.nodes)) != MIPT_OK) return r;
    if ((r = upload(c, H.tris, &D.tris)) != MIPT_OK) return r;
    if (!H.tri_normals.empty()) {
      if ((r = upload(c, H.tri_normals, &D.tri_normals)) != MIPT_OK) return r;
    }
    if (!H.tri_uvs.empty()) {
      if ((r = upload(c, H.tri_uvs, &D.tri_uvs)) != MIPT_OK) return r;
    }
    if (!H.tri_colors.empty()) {
      if ((r = upload(c, H.tri_colors, &D.tri_colors)) != MIPT_OK) return r;
    }
    if ((r = upload(c, H.prim_info, &D.prim_info)) != MIPT_OK) return r;
    if ((r = upload(c, H.ref_order, &D.ref_order)) != MIPT_OK) return r;
    c->n_nodes = (int64_t)H.nodes.size();
  }
  if (!H.tex_rgb.empty()) {
    std::vector<TexRec> table;
    for (size_t k = 0; k < H.tex_rgb.size(); k++) {
      const uint8_t *p = nullptr;
      if ((r = upload(c, H.tex_rgb[k], &p)) != MIPT_OK) return r;
      table.push_back(TexRec{p, H.tex_w[k], H.tex_h[k]});
    }
    if ((r = upload(c, table, &D.textures)) != MIPT_OK) return r;
    D.n_textures = (int)table.size();
  }
  if ((r = upload(c, H.objects, &D.objects)) != MIPT_OK) return r;
  if ((r = upload(c, H.materials, &D.materials)) != MIPT_OK) return r;
  if ((r = upload(c, H.lights, &D.lights)) != MIPT_OK) return r;
  D.n_lights = (int)H.lights.size();
  if (D.n_lights > 65535) return set_err(c, MIPT_ERR_LIMIT, "more than 65535 lights");
  D.ray_light = nullptr;  // filled by sync_lights once area_samples is known
  D.n_materials = (int)H.materials.size();
  D.n_prims = (int)H.prim_info.size();
  D.n_analytic = H.n_analytic;
  {
    // world boxes of the objects; in mesh scenes with at most a few analytic primitives the shadow kernel tests
    // those ahead of its walk, box first, and walks the triangles only
    const size_t bytes = std::max<size_t>(H.object_boxes.size() * sizeof(double), 16);
    void *p = nullptr;
    HIP_TRY(c, hipMalloc(&p, bytes));
    c->scene_allocs.push_back(p);
    c->scene_bytes += (int64_t)bytes;
    if (!H.object_boxes.empty()) HIP_TRY(c, hipMemcpy(p, H.object_boxes.data(), H.object_boxes.size() * sizeof(double), hipMemcpyHostToDevice));
    D.object_boxes = reinterpret_cast<const double *>(p);
    D.lean_any = (H.n_analytic <= 8 && (size_t)H.n_analytic < H.prim_info.size()) ? 1 : 0;
    if (c->opt.lean_any == 0) D.lean_any = 0;
    if (const char *e = debug_env("MIPT_LEAN_ANY")) {  // measurement knob: 0 keeps the analytic tests in the walk
      if (std::atoi(e) == 0) D.lean_any = 0;
    }
  }
  // traversal stacks: up to kLdsStackMax entries per lane in LDS (6 blocks x 256 lanes x 24
  // entries fill 144 of a CU's 160 KB), the rest of the worst case in global memory
  const int kLdsStackMax = 24;
  int lds_entries = std::min(kLdsStackMax, std::max(1, c->opt.lds_stack));
  if (const char *e = debug_env("MIPT_LDS_STACK")) {  // test and tuning knob
    int v = std::atoi(e);
    if (v >= 1 && v <= 60) lds_entries = v;
  }
  D.stack_entries = std::min(H.stack_need, lds_entries);
  c->persist_blocks = c->n_cus * traversal_blocks_per_cu(D);
  D.ovf_stride = (uint32_t)c->persist_blocks * 256u;
  {
    size_t extra = (size_t)std::max(0, H.stack_need - D.stack_entries);
    void *p = nullptr;
    size_t bytes = std::max<size_t>(extra * D.ovf_stride * sizeof(uint32_t), 16);
    HIP_TRY(c, hipMalloc(&p, bytes));
    c->scene_allocs.push_back(p);
    c->scene_bytes += (int64_t)bytes;
    D.stack_ovf = reinterpret_cast<uint32_t *>(p);
    c->stack_ovf_b = nullptr;
    if (c->overlap) {  // kernels of the two streams run side by side and number their threads alike
      HIP_TRY(c, hipMalloc(&p, bytes));
      c->scene_allocs.push_back(p);
      c->scene_bytes += (int64_t)bytes;
      c->stack_ovf_b = reinterpret_cast<uint32_t *>(p);
    }
  }
  D.translucent = H.translucent ? 1 : 0;
  // one-byte visibility records where they are exact (see LightRec::flat) and pay: in scenes of a few primitives a
  // shadow ray's walk is so short that its byte store, which shares a word with its neighbours', costs more than the
  // sums save (cornellBoxes with an area light: shadow kernel 202 -> 211 ms, sums 26 -> 24)
  D.vis_bytes = (!H.translucent && H.flat_lights && H.prim_info.size() > 4096) ? 1 : 0;
  if (c->opt.vis_bytes >= 0) D.vis_bytes = (c->opt.vis_bytes != 0 && !H.translucent && H.flat_lights) ? 1 : 0;
  if (const char *e = debug_env("MIPT_VIS_BYTES")) {  // measurement and test knob: 0 = fp64 records, 1 = bytes wherever they are exact
    D.vis_bytes = (std::atoi(e) != 0 && !H.translucent && H.flat_lights) ? 1 : 0;
  }
  D.local_meshes = H.local_meshes ? 1 : 0;
  D.t_cap = 1000.0;  // every entry point that traces sets it to its own t_max
  D.radius = H.radius;
  std::memcpy(D.ambient, H.ambient, sizeof(D.ambient));
  D.camera = H.camera;
  D.has_cubemap = H.has_cubemap ? 1 : 0;
  if (H.has_cubemap) {
    for (int f = 0; f < 6; f++) {
      const uint8_t *p = nullptr;
      if ((r = upload(c, H.cube_rgb[f], &p)) != MIPT_OK) return r;
      D.cube[f] = p;
      D.cube_w[f] = H.cube_w[f];
      D.cube_h[f] = H.cube_h[f];
    }
  }
  c->n_prims = (int64_t)H.prim_info.size();
  // The tail launch pays where walks are short and trees deep (the analytic demo scenes: spheres.json
  // 56.6 -> 51.5 ms per 128 spp, cornellBoxes 157.7 -> 154.1); in mesh scenes its one-ray-per-lane walks cost
  // more than the few small launches they replace (100 k triangles +1 %, 1 M +3.5 % at 262144 rays), so it is off there.
  c->tail_rays = c->tail_rays_env >= 0 ? (uint32_t)c->tail_rays_env : (c->n_prims <= 4096 ? (1u << 18) : 0u);
  c->fuse_rays = c->fuse_rays_env >= 0 ? (uint32_t)c->fuse_rays_env : (c->n_prims > 4096 ? (1u << 19) : 0u);
  c->lights_host = H.lights;
  c->lights_area_samples = -1;
  c->table_area_samples = -1;
  c->scene_serial = ++g_scene_serial;
  // drop the big host copies
  std::vector<Node4>().swap(H.nodes);
  std::vector<Tri>().swap(H.tris);
  std::vector<TriNormals>().swap(H.tri_normals);
  std::vector<TriUV>().swap(H.tri_uvs);
  std::vector<TriColor>().swap(H.tri_colors);
  std::vector<std::vector<uint8_t>>().swap(H.tex_rgb);
  std::vector<uint32_t>().swap(H.prim_info);
  std::vector<uint32_t>().swap(H.ref_order);
  for (int f = 0; f < 6; f++) std::vector<uint8_t>().swap(H.cube_rgb[f]);
  c->have_scene = true;
  return MIPT_OK;
}

}  // extern "C"

// Device memory one mipt_render_accumulate call works in.  The allocating entry point sizes it
// for the requested batch and grows the overflow stack on demand; the entry point that may not
// allocate carves it out of a static device array and fails where the other would grow.
struct Workspace {
  void *queue[2], *shade, *hits, *accum, *rows, *counts, *dcount, *vis;
  uint32_t cap;        // entries per queue / hit record array
  uint32_t shade_cap;  // entries of the shade list: a launch takes <= batch rays and lists each at most once
  int64_t batch;       // most samples per wavefront batch (also the most rays one launch takes)
  uint64_t vis_rays;   // shadow rays the visibility buffer holds (3 doubles each in translucent scenes, else 1)
  bool may_allocate;   // the overflow stack is the context's and grows with hipMalloc; else it is the fixed one below
  void *spill;         // fixed overflow stack of ray records (used when !may_allocate)
  uint64_t spill_cap;  // its entries
};

static int render_core(mipt_ctx *c, const mipt_params *p, double *d_sum, hipStream_t st, const Workspace &ws, const int *rows, size_t n_rows) {
  int r;
  bool count = (p->flags & MIPT_FLAG_COUNTERS) != 0;
  bool prof = (p->flags & MIPT_FLAG_PROFILE) != 0;
  // a call that fails half way leaves no half-recorded event pairs behind
  struct SpanGuard {
    mipt_ctx *c;
    bool done = false;
    ~SpanGuard() {
      if (done) return;
      c->spans.clear();
      c->events_used = 0;
    }
  } span_guard{c};

  int ns_total = p->sample_end - p->sample_begin;
  int64_t owned = (int64_t)n_rows * p->width;

  // batch geometry: SB samples of PB pixels per batch, PB*SB <= B
  const int64_t B = ws.batch;
  int SB = (int)std::min<int64_t>(ns_total, B);
  int64_t PB = std::max<int64_t>(1, B / SB);
  PB = std::min<int64_t>(PB, owned);
  const int64_t batch = PB * SB;
  const uint32_t cap = ws.cap;
  c->capacity = cap;
  uint64_t rays_per_hit = (uint64_t)std::max(1, c->dev.rays_per_hit);
  const uint64_t hits_per_chunk = std::min<uint64_t>(ws.vis_rays / rays_per_hit, 0xffffff00ull / rays_per_hit);
  HIP_TRY(c, hipMemcpyAsync(ws.rows, rows, n_rows * sizeof(int), hipMemcpyHostToDevice, st));
  HIP_TRY(c, hipStreamSynchronize(st));  // `rows` may be gone when the copy runs otherwise

  RenderConst rc = make_const(c, p, cap, ws.shade_cap);
  RayQueue q[2] = {carve_queue(ws.queue[0], cap), carve_queue(ws.queue[1], cap)};
  ShadeList sl = carve_shade(ws.shade, ws.shade_cap);

  // Two launch streams for the narrow levels.  A level with few rays cannot fill the chip: its
  // launches last as long as one ray's walk (50-150 us), whatever their size.  The shadow rays of
  // level k (shadow trace + sum) depend on k_shade(k) only, and the closest-hit launch of level
  // k + 1 does not depend on them, so on such levels they run on the context's second stream while
  // the caller's stream goes on with level k + 1.  What the two share is the shade list: these
  // levels alternate between its halves, and k_shade waits for the last shadow launch that read
  // the half it is about to fill; the per-sample accumulators take atomic adds from both sides.
  // Not with the work counters (one set), and not in the entry point that may not allocate (its
  // workspace has one stack overflow area).
  const bool overlap = c->overlap && c->shadow_stream && c->stack_ovf_b && ws.may_allocate && !count;
  SceneDev dev_b = c->dev;
  if (overlap) dev_b.stack_ovf = c->stack_ovf_b;
  const uint32_t half = (ws.shade_cap / 2) & ~63u;  // keeps the second half's arrays aligned like the first's
  bool reading[2] = {false, false};  // summed[k] marks the end of a shadow launch that reads half k
  int side = 0;
  struct ShadowStreamGuard {  // no return leaves work behind on the second stream
    hipStream_t s;
    bool on;
    ~ShadowStreamGuard() {
      if (on) (void)hipStreamSynchronize(s);
    }
  } shadow_guard{c->shadow_stream, overlap};
  double *hit_t = reinterpret_cast<double *>(ws.hits);
  int *hit_prim = reinterpret_cast<int *>(hit_t + cap);
  double *accum = reinterpret_cast<double *>(ws.accum);
  double *vis = reinterpret_cast<double *>(ws.vis);
  BatchCounts *bc = reinterpret_cast<BatchCounts *>(ws.counts);
  DeviceCounters *dc = reinterpret_cast<DeviceCounters *>(ws.dcount);
  const int *d_rows = reinterpret_cast<const int *>(ws.rows);
  mipt_counters &tot = c->totals;

  // lean scenes: the sum kernel looks at the few analytic primitives the shadow walk passes over, where it is one of the kernels that can
  const bool analytic_in_sum = shadow_sum_takes_analytic(c->dev);
  for (int64_t p0 = 0; p0 < owned; p0 += PB) {
    int npix = (int)std::min<int64_t>(PB, owned - p0);
    for (int s0 = p->sample_begin; s0 < p->sample_end; s0 += SB) {
      int ns = std::min(SB, p->sample_end - s0);
      uint32_t n = (uint32_t)((int64_t)npix * ns);
      // a batch of several samples per pixel never stores its camera rays: the level that traces and shades
      // them computes them from their index, and stores every sample's first term (k_shade<CAM>) so that the
      // sums need no clearing; a batch of one sample per pixel goes through the queue and adds to cleared sums
      const CameraRays cam{d_rows, p0, npix, s0, ns};
      if (ns == 1) {
        HIP_TRY(c, hipMemsetAsync(accum, 0, (size_t)n * 3 * sizeof(double), st));
        ProfScope ps(c, prof, MIPT_K_GENERATE, n, st);
        launch_generate(c->dev, rc, d_rows, p0, npix, s0, ns, q[0], st);
      }
      if (count) { tot.samples += n; tot.ray_stages += n; }
      int cur = 0;
      uint64_t spilled = 0;  // entries on the overflow stack
      const uint32_t chunk = (uint32_t)batch;  // rays per launch; children of one launch always fit a queue
      uint32_t dup = (uint32_t)ns;             // the camera rays of a pixel's samples are identical
      // What a launch left in q[cur] (m rays) becomes the next launch's input: at most `chunk` rays stay, the excess is
      // parked on the overflow stack (the tree is growing), and a short queue is topped up from that stack.
      auto settle = [&](uint32_t m) -> int {
        if (m > chunk) {
          uint32_t excess = m - chunk;
          if (ws.may_allocate) {
            int rs = ensure_spill(c, spilled, excess, st);
            if (rs != MIPT_OK) return rs;
          } else if (spilled + excess > ws.spill_cap) {
            return set_err(c, MIPT_ERR_LIMIT,
                           "path tree outgrew the static workspace (%llu rays waiting, room for %llu): use mipt_render_accumulate or lower max_depth",
                           (unsigned long long)(spilled + excess), (unsigned long long)ws.spill_cap);
          }
          RayQueue stack = ws.may_allocate ? carve_queue(c->spill.p, (uint32_t)c->spill_cap) : carve_queue(ws.spill, (uint32_t)ws.spill_cap);
          launch_copy_rays(q[cur], chunk, stack, spilled, excess, st);
          spilled += excess;
          m = chunk;
        } else if (m < chunk && spilled > 0) {
          uint32_t take = (uint32_t)std::min<uint64_t>(chunk - m, spilled);
          RayQueue stack = ws.may_allocate ? carve_queue(c->spill.p, (uint32_t)c->spill_cap) : carve_queue(ws.spill, (uint32_t)ws.spill_cap);
          launch_copy_rays(stack, spilled - take, q[cur], m, take, st);
          spilled -= take;
          m += take;
        }
        n = m;
        return MIPT_OK;
      };
      while (n > 0) {
        HIP_TRY(c, hipMemsetAsync(bc, 0, sizeof(BatchCounts), st));
        if (dup == 1 && n <= c->tail_rays && spilled == 0 && !count && (uint64_t)5 * n + 2048 <= cap) {  // the launch hands back < 5 n + 2048 rays
          // few rays left: one launch follows their trees to the end (or as far as its budget goes)
          {
            ProfScope ps(c, prof, MIPT_K_TAIL, n, st);
            launch_tail(c->dev, rc, q[cur], n, q[cur ^ 1], nullptr, 0, bc, accum, st);
          }
          HIP_TRY(c, hipGetLastError());
          BatchCounts ht;
          HIP_TRY(c, hipMemcpyAsync(&ht, bc, sizeof(ht), hipMemcpyDeviceToHost, st));
          HIP_TRY(c, hipStreamSynchronize(st));
          if (ht.overflow || ht.next_rays > cap)
            return set_err(c, MIPT_ERR_DEVICE, "internal error: ray queue overflow in the tail launch (%u rays, capacity %u)", ht.next_rays, cap);
          cur ^= 1;
          // a launch of n rays can hand back close to 5 n of them (every wave its stack and its lanes' rays): more than one
          // launch takes when n is near the entry limit, so the excess is parked like any other level's
          if ((r = settle(ht.next_rays)) != MIPT_OK) return r;
          continue;
        }
        // where this level's shaded hits go: the whole list, or (two streams, few enough rays) one half of it
        // (wide levels fill the chip by themselves and only lose by sharing it: they stay on one stream)
        const bool fusable = dup == 1 && spilled == 0 && n <= c->fuse_rays && (uint64_t)5 * n + 2048 <= cap;  // see below
        const bool halves = overlap && n <= half && (n <= c->overlap_rays || fusable);
        hipStream_t sb = halves ? c->shadow_stream : st;
        if (halves) side ^= 1;
        const uint32_t sl_off = halves && side ? half : 0u;
        RenderConst rc_l = rc;
        if (halves) rc_l.shade_capacity = half;
        ShadeList sl_l = offset_shade(sl, sl_off);
        // Few rays left in a mesh scene: one launch follows their trees to the end, closest hits and shading only,
        // and hands its shaded hits over as one level's worth would be; the shadow kernels then run once for all of
        // them.  Its waves work `fuse_budget` rounds at most, which bounds what they can append to the list.
        const bool cam_level = dup > 1;  // the camera rays of a batch of several samples per pixel: one ray and at most one hit per sample
        uint32_t fuse_budget = 0;
        if (halves && fusable)
          fuse_budget = (uint32_t)std::min<uint64_t>(c->fuse_budget_cap ? c->fuse_budget_cap : 4096, rc_l.shade_capacity / ((uint64_t)tail_grid(rc, n) * 256u));
        const bool fused = fuse_budget >= (c->fuse_budget_cap ? 1u : 48u);
        if (!fused) {
          ProfScope ps(c, prof, MIPT_K_INTERSECT, n, st);
          launch_intersect(c->dev, rc, q[cur], n, dup, hit_t, hit_prim, dc, count, dup > 1 ? &cam : nullptr, st);
        }
        if (overlap) {
          for (int k = 0; k < 2; k++)
            if (reading[k] && (!halves || k == side)) HIP_TRY(c, hipStreamWaitEvent(st, c->summed[k], 0));
        }
        if (fused) {
          ProfScope ps(c, prof, MIPT_K_TAIL, n, st);
          launch_tail(c->dev, rc_l, q[cur], n, q[cur ^ 1], &sl_l, fuse_budget, bc, accum, st);
        } else {
          ProfScope ps(c, prof, MIPT_K_SHADE, n, st);
          launch_shade(c->dev, rc_l, q[cur], n, dup, hit_t, hit_prim, q[cur ^ 1], sl_l, bc, accum, dc, count, dup > 1 ? &cam : nullptr, st);
          dup = 1;
        }
        HIP_TRY(c, hipGetLastError());  // a launch that failed must not read as "the tree ended here"
        BatchCounts h;
        HIP_TRY(c, hipMemcpyAsync(&h, bc, sizeof(h), hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        if (h.overflow || h.next_rays > cap || h.shades > rc_l.shade_capacity)
          return set_err(c, MIPT_ERR_DEVICE, "internal error: ray queue overflow (%u rays, capacity %u)", h.next_rays, cap);
        if (count) {
          tot.path_rays += n;
          tot.ray_stages += 2ull * n + h.shades;
          tot.shadow_rays += (uint64_t)h.shades * (uint64_t)c->dev.rays_per_hit;
        }
        // shadow rays of this level, in chunks of whole hits that fit the visibility buffer
        // (the caller's stream has been synchronised since k_shade: the second stream needs no event to start)
        if (overlap && !halves && h.shades) {  // back on one stream: the visibility buffer may still be in use on the other
          for (int k = 0; k < 2; k++)
            if (reading[k]) HIP_TRY(c, hipStreamWaitEvent(st, c->summed[k], 0));
          reading[0] = reading[1] = false;
        }
        for (uint64_t h0 = 0; h0 < h.shades; h0 += hits_per_chunk) {
          uint32_t nh = (uint32_t)std::min<uint64_t>(hits_per_chunk, h.shades - h0);
          {
            ProfScope ps(c, prof, MIPT_K_SHADOW, (uint64_t)nh * rays_per_hit, sb);
            launch_shadow_trace(halves ? dev_b : c->dev, rc, sl_l, h0, nh, vis, halves ? dc + 1 : dc, count, sb, analytic_in_sum);
          }
          {
            ProfScope ps(c, prof, MIPT_K_SHADOW_SUM, nh, sb);
            launch_shadow_sum(c->dev, rc, sl_l, h0, nh, vis, accum, cam_level && !halves, sb, analytic_in_sum);
          }
        }
        if (halves && h.shades) {
          HIP_TRY(c, hipEventRecord(c->summed[side], sb));
          reading[side] = true;
        }
        HIP_TRY(c, hipGetLastError());
        cur ^= 1;
        if ((r = settle(h.next_rays)) != MIPT_OK) return r;
      }
      if (overlap) {  // the batch's last shadow sums land before its samples are resolved
        for (int k = 0; k < 2; k++)
          if (reading[k]) HIP_TRY(c, hipStreamWaitEvent(st, c->summed[k], 0));
      }
      {
        ProfScope ps(c, prof, MIPT_K_RESOLVE, (uint64_t)npix, st);
        launch_resolve(rc, accum, d_rows, p0, npix, ns, d_sum, dc, st);
      }
    }
  }
  HIP_TRY(c, hipStreamSynchronize(st));
  if (prof) collect_spans(c);
  span_guard.done = true;
  HIP_TRY(c, hipGetLastError());
  return MIPT_OK;
}

// rows of the frame this call's shard owns
static std::vector<int> owned_rows(const mipt_params *p) {
  std::vector<int> rows;
  for (int j = 0; j < p->height; j++)
    if ((j / p->band_rows) % p->shard_count == p->shard_index) rows.push_back(j);
  return rows;
}

extern "C" {

int mipt_render_accumulate(mipt_ctx *c, const mipt_params *p, double *d_sum, void *stream) {
  if (!c) return set_err(nullptr, MIPT_ERR_INVALID, "null context");
  if (!c->have_scene) return set_err(c, MIPT_ERR_NOSCENE, "no scene uploaded");
  int r = check_params(c, p);
  if (r != MIPT_OK) return r;
  if (!d_sum) return set_err(c, MIPT_ERR_INVALID, "null d_sum");
  HIP_TRY(c, hipSetDevice(c->device));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if ((r = sync_lights(c, p->area_samples)) != MIPT_OK) return r;
  c->dev.t_cap = p->t_max;
  std::vector<int> rows = owned_rows(p);
  int ns_total = p->sample_end - p->sample_begin;
  int64_t owned = (int64_t)rows.size() * p->width;
  if (owned == 0 || ns_total == 0) return MIPT_OK;

  Workspace ws;
  ws.may_allocate = true;
  ws.spill = nullptr;
  ws.spill_cap = 0;
  if (p->batch_samples > 0) {
    ws.batch = p->batch_samples;
  } else {
    // As large as the job and the card allow: the deep, nearly empty levels of a path tree cost every batch the same
    // ~15 ms, so few large batches beat many small ones (config 4: 323 Msamples/s at 2^25 samples per batch, 372 at 2^28).
    // A sample takes two queue entries in each of the two queues and the hit records, a shade record, its sums and the
    // results of its shadow rays; 60 % of what is free now (plus what this context already holds) goes to the workspace,
    // the rest stays for the overflow stack of growing trees and for the caller.
    size_t free_b = 0, total_b = 0;
    HIP_TRY(c, hipMemGetInfo(&free_b, &total_b));
    const size_t held = c->queue[0].bytes + c->queue[1].bytes + c->shade.bytes + c->hits.bytes + c->accum.bytes + c->vis.bytes;
    const double per_sample = 4.0 * (double)kEntryBytes + 2.0 * (sizeof(double) + sizeof(int)) + (double)kShadeEntryBytes + 3.0 * sizeof(double) +
                              (double)std::max(1, c->dev.rays_per_hit) * (c->dev.translucent ? 24.0 : (c->dev.vis_bytes ? 1.0 : 8.0));
    const double fit = 0.6 * (double)(free_b + held) / per_sample;
    ws.batch = (int64_t)std::min<double>((double)(1ll << 28), std::max<double>((double)(1 << 20), fit));
  }
  // the batch the core will form (SB samples of PB pixels, PB*SB <= B) decides the sizes
  int64_t SB = std::min<int64_t>(ns_total, ws.batch);
  int64_t PB = std::min<int64_t>(std::max<int64_t>(1, ws.batch / SB), owned);
  int64_t batch = PB * SB;
  if (batch > (1ll << 28)) return set_err(c, MIPT_ERR_LIMIT, "batch_samples too large (limit 2^28)");
  ws.cap = (uint32_t)(2 * batch + 1024);  // a launch takes <= batch rays and every ray has <= 2 children
  ws.shade_cap = (uint32_t)(batch + 1024);  // the shade list and the visibility buffer are sized by the batch, not by the queues
  if ((r = ensure(c, c->queue[0], (size_t)ws.cap * kEntryBytes + 64)) != MIPT_OK) return r;
  if ((r = ensure(c, c->queue[1], (size_t)ws.cap * kEntryBytes + 64)) != MIPT_OK) return r;
  if ((r = ensure(c, c->shade, (size_t)ws.shade_cap * kShadeEntryBytes + 64)) != MIPT_OK) return r;
  if ((r = ensure(c, c->hits, (size_t)ws.cap * (sizeof(double) + sizeof(int)) + 64)) != MIPT_OK) return r;
  if ((r = ensure(c, c->accum, (size_t)batch * 3 * sizeof(double))) != MIPT_OK) return r;
  if ((r = ensure(c, c->rows, rows.size() * sizeof(int))) != MIPT_OK) return r;
  if ((r = ensure(c, c->counts, sizeof(BatchCounts))) != MIPT_OK) return r;
  // visibility scratch: all shadow rays of a level (at most `batch` hits) if they fit 2^31 rays, else chunks of whole hits
  uint64_t rays_per_hit = (uint64_t)std::max(1, c->dev.rays_per_hit);
  ws.vis_rays = std::max<uint64_t>(std::min<uint64_t>((uint64_t)(batch + 1024) * rays_per_hit, 1ull << 31), rays_per_hit);
  // per ray: one byte where a record says "reached its light or not" (SceneDev::vis_bytes), else a double, three in translucent scenes
  if ((r = ensure(c, c->vis, ws.vis_rays * (c->dev.translucent ? 3 * sizeof(double) : (c->dev.vis_bytes ? 1 : sizeof(double))) + 64)) != MIPT_OK) return r;
  if (!c->dcount.p) {
    if ((r = ensure(c, c->dcount, 2 * sizeof(DeviceCounters))) != MIPT_OK) return r;  // the second one: work counters of the shadow stream
    HIP_TRY(c, hipMemset(c->dcount.p, 0, 2 * sizeof(DeviceCounters)));
  }
  ws.queue[0] = c->queue[0].p;
  ws.queue[1] = c->queue[1].p;
  ws.shade = c->shade.p;
  ws.hits = c->hits.p;
  ws.accum = c->accum.p;
  ws.rows = c->rows.p;
  ws.counts = c->counts.p;
  ws.dcount = c->dcount.p;
  ws.vis = c->vis.p;
  return render_core(c, p, d_sum, st, ws, rows.data(), rows.size());
}

int mipt_render_accumulate_static(mipt_ctx *c, const mipt_params *p, double *d_sum, void *stream) {
  if (!c) return set_err(nullptr, MIPT_ERR_INVALID, "null context");
  if (!c->have_scene) return set_err(c, MIPT_ERR_NOSCENE, "no scene uploaded");
  int r = check_params(c, p);
  if (r != MIPT_OK) return r;
  if (!d_sum) return set_err(c, MIPT_ERR_INVALID, "null d_sum");
  if (!c->static_ws) return set_err(c, MIPT_ERR_DEVICE, "the static workspace was not set up for device %d", c->device);
  if (p->flags & (MIPT_FLAG_COUNTERS | MIPT_FLAG_PROFILE))
    return set_err(c, MIPT_ERR_INVALID, "MIPT_FLAG_COUNTERS and MIPT_FLAG_PROFILE need mipt_render_accumulate (they keep device counters and events)");
  if (p->height > kStaticMaxRows)
    return set_err(c, MIPT_ERR_LIMIT, "frame height %d beyond the static workspace's limit of %d rows", p->height, kStaticMaxRows);
  HIP_TRY(c, hipSetDevice(c->device));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  unsigned char *base = static_cast<unsigned char *>(c->static_ws);

  // the per-hit shadow-ray block for this area_samples: counted first (nothing is touched when it is beyond the
  // limits), then the light records are laid out and the ray -> light table goes into the workspace, unless this
  // context's table for this scene and area_samples is what the workspace of this GPU already holds
  const int first = rays_per_hit_for(c, p->area_samples);
  const uint64_t per_ray = c->dev.translucent ? 3 : 1;
  if ((uint32_t)first > kStaticMaxRaysPerHit || (uint64_t)std::max(first, 1) * per_ray > kStaticVisDoubles)
    return set_err(c, MIPT_ERR_LIMIT, "%d shadow rays per hit beyond the static workspace's limit of %u", first,
                   (unsigned)std::min<uint64_t>(kStaticMaxRaysPerHit, kStaticVisDoubles / per_ray));
  if (c->device < 0 || c->device >= 64) return set_err(c, MIPT_ERR_LIMIT, "the static workspace serves devices 0..63");
  if ((r = layout_lights(c, p->area_samples)) != MIPT_OK) return r;
  uint16_t *d_table = reinterpret_cast<uint16_t *>(base + kOffRayLight);
  {
    std::lock_guard<std::mutex> lock(g_static_owner_lock);
    StaticTableOwner &own = g_static_owner[c->device];
    if (own.ctx != c || own.scene_serial != c->scene_serial || own.area_samples != p->area_samples) {
      own.ctx = nullptr;  // until the whole table is there
      uint16_t table[256];  // staged through the stack in pieces: no host allocation either
      size_t filled = 0, sent = 0;
      for (size_t k = 0; k < c->lights_host.size(); k++)
        for (int s2 = 0; s2 < c->lights_host[k].samples; s2++) {
          table[filled++] = (uint16_t)k;
          if (filled == 256) {
            HIP_TRY(c, hipMemcpy(d_table + sent, table, filled * sizeof(uint16_t), hipMemcpyHostToDevice));
            sent += filled;
            filled = 0;
          }
        }
      if (filled) HIP_TRY(c, hipMemcpy(d_table + sent, table, filled * sizeof(uint16_t), hipMemcpyHostToDevice));
      own.ctx = c;
      own.scene_serial = c->scene_serial;
      own.area_samples = p->area_samples;
    }
  }
  c->dev.rays_per_hit = first;
  c->dev.ray_light = d_table;
  c->dev.t_cap = p->t_max;

  int rows[kStaticMaxRows];  // on the stack: no host allocation either
  size_t n_rows = 0;
  for (int j = 0; j < p->height; j++)
    if ((j / p->band_rows) % p->shard_count == p->shard_index) rows[n_rows++] = j;
  if (n_rows == 0 || p->sample_end == p->sample_begin) return MIPT_OK;

  Workspace ws;
  ws.may_allocate = false;
  ws.batch = kStaticBatch;
  ws.cap = kStaticCap;
  ws.shade_cap = kStaticShadeCap;
  ws.queue[0] = base + kOffQueue0;
  ws.queue[1] = base + kOffQueue1;
  ws.shade = base + kOffShade;
  ws.hits = base + kOffHits;
  ws.accum = base + kOffAccum;
  ws.rows = base + kOffRows;
  ws.counts = base + kOffCounts;
  ws.dcount = base + kOffDcount;
  ws.vis = base + kOffVis;
  ws.vis_rays = kStaticVisDoubles / per_ray;
  ws.spill = base + kOffSpill;
  ws.spill_cap = kStaticSpill;
  // the array keeps the last call's contents: the work counters of the persistent kernels and the
  // NaN tally start at zero; queues, hit records, sums and visibility are written before they are read
  HIP_TRY(c, hipMemsetAsync(ws.dcount, 0, sizeof(DeviceCounters), st));
  return render_core(c, p, d_sum, st, ws, rows, n_rows);
}

int mipt_resolve(mipt_ctx *c, const double *d_sum, int width, int height, int spp, float *d_rgb_f32, uint8_t *d_rgb8, void *stream) {
  if (!c) return set_err(nullptr, MIPT_ERR_INVALID, "null context");
  if (!d_sum || width <= 0 || height <= 0 || spp <= 0) return set_err(c, MIPT_ERR_INVALID, "bad resolve arguments");
  HIP_TRY(c, hipSetDevice(c->device));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  launch_finalize(d_sum, (int64_t)width * height, spp, d_rgb_f32, d_rgb8, st);
  HIP_TRY(c, hipGetLastError());
  return MIPT_OK;
}

int mipt_render(mipt_ctx *c, const mipt_params *p, float *rgb_f32, uint8_t *rgb8) {
  if (!c) return set_err(nullptr, MIPT_ERR_INVALID, "null context");
  if (!c->have_scene) return set_err(c, MIPT_ERR_NOSCENE, "no scene uploaded");
  int r = check_params(c, p);
  if (r != MIPT_OK) return r;
  HIP_TRY(c, hipSetDevice(c->device));
  size_t npx = (size_t)p->width * p->height;
  double *d_sum = nullptr;
  float *d_f = nullptr;
  uint8_t *d_b = nullptr;
  HIP_TRY(c, hipMalloc((void **)&d_sum, npx * 3 * sizeof(double)));
  auto cleanup = [&]() {
    if (d_sum) (void)hipFree(d_sum);
    if (d_f) (void)hipFree(d_f);
    if (d_b) (void)hipFree(d_b);
  };
  hipError_t e = hipMemset(d_sum, 0, npx * 3 * sizeof(double));  // traceSetup zero-fill, RayTracer.cpp:427
  if (e == hipSuccess && rgb_f32) e = hipMalloc((void **)&d_f, npx * 3 * sizeof(float));
  if (e == hipSuccess && rgb8) e = hipMalloc((void **)&d_b, npx * 3);
  if (e != hipSuccess) {
    cleanup();
    return set_err(c, MIPT_ERR_NOMEM, "device allocation failed: %s", hipGetErrorString(e));
  }
  r = mipt_render_accumulate(c, p, d_sum, nullptr);
  if (r == MIPT_OK) r = mipt_resolve(c, d_sum, p->width, p->height, p->spp, d_f, d_b, nullptr);
  if (r == MIPT_OK) {
    if (rgb_f32) e = hipMemcpy(rgb_f32, d_f, npx * 3 * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess && rgb8) e = hipMemcpy(rgb8, d_b, npx * 3, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) r = set_err(c, MIPT_ERR_DEVICE, "copy back failed: %s", hipGetErrorString(e));
  }
  cleanup();
  return r;
}

int mipt_trace_pixel(mipt_ctx *c, const mipt_params *p, int i, int j, double out[3]) {
  if (!c) return set_err(nullptr, MIPT_ERR_INVALID, "null context");
  if (!c->have_scene) {  // tracePixel returns black without a scene, RayTracer.cpp:63-64
    if (out) out[0] = out[1] = out[2] = 0.0;
    return set_err(c, MIPT_ERR_NOSCENE, "no scene uploaded");
  }
  int r = check_params(c, p);
  if (r != MIPT_OK) return r;
  if (!out || i < 0 || j < 0 || i >= p->width || j >= p->height) return set_err(c, MIPT_ERR_INVALID, "pixel out of range");
  // one-row frame view: shard the frame so that only row j is owned, then read pixel i
  mipt_params q = *p;
  q.shard_count = p->height;
  q.shard_index = j;
  q.band_rows = 1;
  q.sample_begin = 0;
  q.sample_end = p->spp;
  HIP_TRY(c, hipSetDevice(c->device));
  size_t npx = (size_t)p->width * p->height;
  double *d_sum = nullptr;
  HIP_TRY(c, hipMalloc((void **)&d_sum, npx * 3 * sizeof(double)));
  hipError_t e = hipMemset(d_sum, 0, npx * 3 * sizeof(double));
  if (e != hipSuccess) { (void)hipFree(d_sum); return set_err(c, MIPT_ERR_DEVICE, "memset failed"); }
  r = mipt_render_accumulate(c, &q, d_sum, nullptr);
  double px[3] = {0, 0, 0};
  if (r == MIPT_OK) {
    e = hipMemcpy(px, d_sum + 3 * ((size_t)i + (size_t)j * p->width), sizeof(px), hipMemcpyDeviceToHost);
    if (e != hipSuccess) r = set_err(c, MIPT_ERR_DEVICE, "copy back failed");
  }
  (void)hipFree(d_sum);
  if (r == MIPT_OK)
    for (int k = 0; k < 3; k++) out[k] = px[k] / (double)p->spp;
  return r;
}

int mipt_intersect(mipt_ctx *c, int64_t n, const double *org, const double *dir, double t_max, double *t, int32_t *prim, double *normal) {
  if (!c) return set_err(nullptr, MIPT_ERR_INVALID, "null context");
  if (!c->have_scene) return set_err(c, MIPT_ERR_NOSCENE, "no scene uploaded");
  if (n < 0 || (n > 0 && (!org || !dir || !t || !prim || !normal))) return set_err(c, MIPT_ERR_INVALID, "bad arguments");
  if (n == 0) return MIPT_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  double *d_org = nullptr, *d_dir = nullptr, *d_t = nullptr, *d_n = nullptr;
  int *d_p = nullptr;
  size_t v3 = (size_t)n * 3 * sizeof(double);
  hipError_t e = hipMalloc((void **)&d_org, v3);
  if (e == hipSuccess) e = hipMalloc((void **)&d_dir, v3);
  if (e == hipSuccess) e = hipMalloc((void **)&d_n, v3);
  if (e == hipSuccess) e = hipMalloc((void **)&d_t, (size_t)n * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void **)&d_p, (size_t)n * sizeof(int));
  if (e == hipSuccess) e = hipMemcpy(d_org, org, v3, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_dir, dir, v3, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    c->dev.t_cap = t_max;
    launch_intersect_rays(c->dev, n, d_org, d_dir, t_max, d_t, d_p, d_n, nullptr);
    e = hipDeviceSynchronize();
  }
  if (e == hipSuccess) e = hipMemcpy(t, d_t, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(prim, d_p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(normal, d_n, v3, hipMemcpyDeviceToHost);
  (void)hipFree(d_org); (void)hipFree(d_dir); (void)hipFree(d_n); (void)hipFree(d_t); (void)hipFree(d_p);
  if (e != hipSuccess) return set_err(c, MIPT_ERR_DEVICE, "mipt_intersect: %s", hipGetErrorString(e));
  return MIPT_OK;
}

int mipt_hit_material(mipt_ctx *c, int64_t n, const double *org, const double *dir, double t_max, double *uv, double *kd, double *ka) {
  if (!c) return set_err(nullptr, MIPT_ERR_INVALID, "null context");
  if (!c->have_scene) return set_err(c, MIPT_ERR_NOSCENE, "no scene uploaded");
  if (n < 0 || (n > 0 && (!org || !dir || !uv || !kd || !ka))) return set_err(c, MIPT_ERR_INVALID, "bad arguments");
  if (n == 0) return MIPT_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  double *d_org = nullptr, *d_dir = nullptr, *d_uv = nullptr, *d_kd = nullptr, *d_ka = nullptr;
  size_t v3 = (size_t)n * 3 * sizeof(double), v2 = (size_t)n * 2 * sizeof(double);
  hipError_t e = hipMalloc((void **)&d_org, v3);
  if (e == hipSuccess) e = hipMalloc((void **)&d_dir, v3);
  if (e == hipSuccess) e = hipMalloc((void **)&d_kd, v3);
  if (e == hipSuccess) e = hipMalloc((void **)&d_ka, v3);
  if (e == hipSuccess) e = hipMalloc((void **)&d_uv, v2);
  if (e == hipSuccess) e = hipMemcpy(d_org, org, v3, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_dir, dir, v3, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    c->dev.t_cap = t_max;
    launch_hit_material(c->dev, n, d_org, d_dir, t_max, d_uv, d_kd, d_ka, nullptr);
    e = hipDeviceSynchronize();
  }
  if (e == hipSuccess) e = hipMemcpy(uv, d_uv, v2, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(kd, d_kd, v3, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(ka, d_ka, v3, hipMemcpyDeviceToHost);
  (void)hipFree(d_org); (void)hipFree(d_dir); (void)hipFree(d_uv); (void)hipFree(d_kd); (void)hipFree(d_ka);
  if (e != hipSuccess) return set_err(c, MIPT_ERR_DEVICE, "mipt_hit_material: %s", hipGetErrorString(e));
  return MIPT_OK;
}

int mipt_shadow_attenuation(mipt_ctx *c, const mipt_params *p, int light, int64_t n, const double *fire, const uint64_t *key, double *out) {
  if (!c) return set_err(nullptr, MIPT_ERR_INVALID, "null context");
  if (!c->have_scene) return set_err(c, MIPT_ERR_NOSCENE, "no scene uploaded");
  int r = check_params(c, p);
  if (r != MIPT_OK) return r;
  if (light < 0 || light >= c->dev.n_lights) return set_err(c, MIPT_ERR_INVALID, "light index out of range");
  if (n < 0 || (n > 0 && (!fire || !key || !out))) return set_err(c, MIPT_ERR_INVALID, "bad arguments");
  if (n == 0) return MIPT_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  if ((r = sync_lights(c, p->area_samples)) != MIPT_OK) return r;
  c->dev.t_cap = p->t_max;
  RenderConst rc = make_const(c, p, 0);
  double *d_f = nullptr, *d_o = nullptr;
  uint64_t *d_k = nullptr;
  size_t v3 = (size_t)n * 3 * sizeof(double);
  hipError_t e = hipMalloc((void **)&d_f, v3);
  if (e == hipSuccess) e = hipMalloc((void **)&d_o, v3);
  if (e == hipSuccess) e = hipMalloc((void **)&d_k, (size_t)n * sizeof(uint64_t));
  if (e == hipSuccess) e = hipMemcpy(d_f, fire, v3, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_k, key, (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    launch_shadow_attenuation(c->dev, rc, light, n, d_f, d_k, d_o, nullptr);
    e = hipDeviceSynchronize();
  }
  if (e == hipSuccess) e = hipMemcpy(out, d_o, v3, hipMemcpyDeviceToHost);
  (void)hipFree(d_f); (void)hipFree(d_o); (void)hipFree(d_k);
  if (e != hipSuccess) return set_err(c, MIPT_ERR_DEVICE, "mipt_shadow_attenuation: %s", hipGetErrorString(e));
  return MIPT_OK;
}

int mipt_shadow_rays(mipt_ctx *c, const mipt_params *p, int64_t n, const double *P, const double *N, const uint64_t *key, const int32_t *prim,
                     double *out) {
  if (!c) return set_err(nullptr, MIPT_ERR_INVALID, "null context");
  if (!c->have_scene) return set_err(c, MIPT_ERR_NOSCENE, "no scene uploaded");
  int r = check_params(c, p);
  if (r != MIPT_OK) return r;
  if (n < 0 || (n > 0 && (!P || !N || !key || !prim || !out))) return set_err(c, MIPT_ERR_INVALID, "bad arguments");
  if (n == 0 || c->dev.n_lights == 0) return MIPT_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  if ((r = sync_lights(c, p->area_samples)) != MIPT_OK) return r;
  c->dev.t_cap = p->t_max;
  const uint64_t R = (uint64_t)c->dev.rays_per_hit;
  if ((uint64_t)n * R > 0x7fffff00ull || n > (1 << 24)) return set_err(c, MIPT_ERR_LIMIT, "more than 2^31 shadow rays in one call");
  // storage index of every primitive given by its reference-order index (what mipt_intersect reports)
  std::vector<uint32_t> ref((size_t)c->n_prims), where((size_t)c->n_prims);
  HIP_TRY(c, hipMemcpy(ref.data(), c->dev.ref_order, ref.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  for (size_t k = 0; k < ref.size(); k++)
    if (ref[k] < where.size()) where[ref[k]] = (uint32_t)k;
  // the shade records, as k_shade would have listed them (W does not matter to the rays)
  const uint32_t cap = (uint32_t)n;
  std::vector<unsigned char> host((size_t)cap * kShadeEntryBytes + 64, 0);
  ShadeList hs = carve_shade(host.data(), cap);
  for (int64_t k = 0; k < n; k++) {
    if (prim[k] < 0 || prim[k] >= c->n_prims) return set_err(c, MIPT_ERR_INVALID, "hit %lld: primitive %d out of range", (long long)k, prim[k]);
    hs.px[k] = P[3 * k]; hs.py[k] = P[3 * k + 1]; hs.pz[k] = P[3 * k + 2];
    hs.nx[k] = N[3 * k]; hs.ny[k] = N[3 * k + 1]; hs.nz[k] = N[3 * k + 2];
    hs.key[k] = key[k];
    hs.slot[k] = (uint32_t)k;
    hs.prim[k] = where[(size_t)prim[k]];
  }
  const size_t per_ray = c->dev.translucent ? 3 : 1;
  void *d_list = nullptr, *d_vis = nullptr, *d_dc = nullptr;
  hipError_t e = hipMalloc(&d_list, host.size());
  if (e == hipSuccess) e = hipMalloc(&d_vis, (size_t)n * R * per_ray * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&d_dc, sizeof(DeviceCounters));
  if (e == hipSuccess) e = hipMemset(d_dc, 0, sizeof(DeviceCounters));
  if (e == hipSuccess) e = hipMemset(d_vis, 0xff, (size_t)n * R * per_ray * sizeof(double));  // a ray the launch forgets reads as NaN / 255
  if (e == hipSuccess) e = hipMemcpy(d_list, host.data(), host.size(), hipMemcpyHostToDevice);
  std::vector<double> vis;
  if (e == hipSuccess) {
    RenderConst rc = make_const(c, p, cap, cap);
    launch_shadow_trace(c->dev, rc, carve_shade(d_list, cap), 0, (uint32_t)n, reinterpret_cast<double *>(d_vis), reinterpret_cast<DeviceCounters *>(d_dc), false, nullptr);
    e = hipDeviceSynchronize();
  }
  if (e == hipSuccess) {
    vis.resize((size_t)n * R * per_ray);
    e = hipMemcpy(vis.data(), d_vis, vis.size() * sizeof(double), hipMemcpyDeviceToHost);
  }
  (void)hipFree(d_list); (void)hipFree(d_vis); (void)hipFree(d_dc);
  if (e != hipSuccess) return set_err(c, MIPT_ERR_DEVICE, "mipt_shadow_rays: %s", hipGetErrorString(e));
  const uint8_t *vis8 = reinterpret_cast<const uint8_t *>(vis.data());
  for (int64_t k = 0; k < n; k++)
    for (const LightRec &L : c->lights_host)
      for (int sm = 0; sm < L.samples; sm++) {
        const size_t ray = (size_t)k * R + (size_t)(L.first_ray + sm);
        double *o = out + 3 * ray;
        if (c->dev.translucent) {
          for (int a = 0; a < 3; a++) o[a] = vis[3 * ray + a];
        } else {
          // one-byte records say "reached its light": the light's constant scale goes in as the sum kernels put it in
          const double v = c->dev.vis_bytes ? (vis8[ray] == 1 ? L.flat_scale : vis8[ray] == 0 ? 0.0 : std::nan("")) : vis[ray];
          for (int a = 0; a < 3; a++) o[a] = L.color[a] * v;
        }
      }
  return MIPT_OK;
}

int mipt_get_counters(mipt_ctx *c, mipt_counters *out) {
  if (!c || !out) return set_err(c, MIPT_ERR_INVALID, "null argument");
  *out = c->totals;
  if (c->dcount.p) {
    HIP_TRY(c, hipSetDevice(c->device));
    DeviceCounters d;
    HIP_TRY(c, hipMemcpy(&d, c->dcount.p, sizeof(d), hipMemcpyDeviceToHost));
    out->node_visits = d.closest_nodes + d.shadow_nodes;
    out->prim_tests = d.closest_prims + d.shadow_prims;
    out->shadow_node_visits = d.shadow_nodes;
    out->shadow_prim_tests = d.shadow_prims;
    out->shaded_hits = d.shaded_hits;
    out->nan_samples = d.nan_samples;
  }
  return MIPT_OK;
}

int mipt_reset_counters(mipt_ctx *c) {
  if (!c) return set_err(nullptr, MIPT_ERR_INVALID, "null context");
  c->totals = mipt_counters{};
  c->ktimes = mipt_kernel_times{};
  if (c->dcount.p) {
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemset(c->dcount.p, 0, sizeof(DeviceCounters)));
  }
  return MIPT_OK;
}

int mipt_get_kernel_times(mipt_ctx *c, mipt_kernel_times *out) {
  if (!c || !out) return set_err(c, MIPT_ERR_INVALID, "null argument");
  *out = c->ktimes;
  return MIPT_OK;
}

int mipt_scene_stats(mipt_ctx *c, int64_t *n_prims, int64_t *n_nodes, int64_t *device_bytes) {
  if (!c) return set_err(nullptr, MIPT_ERR_INVALID, "null context");
  if (!c->have_scene) return set_err(c, MIPT_ERR_NOSCENE, "no scene uploaded");
  if (n_prims) *n_prims = c->n_prims;
  if (n_nodes) *n_nodes = c->n_nodes;
  if (device_bytes) *device_bytes = c->scene_bytes;
  return MIPT_OK;
}

}  // extern "C"
