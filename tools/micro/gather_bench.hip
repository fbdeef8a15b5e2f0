// Micro-benchmark of the vector-L1 path for the BVH gather pattern: every lane walks a
// dependent chain of random 64-byte records, loading NL x 16 bytes of each.
// Prints cycles per wave-step per CU for several footprints and load widths.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NL, int ACTIVE>
__global__ void __launch_bounds__(256, 6) gather(const float4 *__restrict__ nodes, uint32_t mask, int steps, float *out) {
  uint32_t tid = blockIdx.x * 256 + threadIdx.x;
  uint32_t idx = tid * 2654435761u + 12345u;
  float acc = 0.f;
  if ((threadIdx.x & 63) < ACTIVE) {
    for (int s = 0; s < steps; s++) {
      const float4 *p = nodes + (size_t)(idx & mask) * 4;
      float4 a = p[0];
      acc += a.x;
      uint32_t dep = __float_as_uint(a.w);
      if (NL > 1) { float4 b = p[1]; acc += b.y; dep += __float_as_uint(b.w); }
      if (NL > 2) { float4 c = p[2]; acc += c.z; dep += __float_as_uint(c.w); }
      if (NL > 3) { float4 d = p[3]; acc += d.x; dep += __float_as_uint(d.w); }
      idx = idx * 1664525u + 1013904223u + dep;
      idx ^= idx >> 15;
    }
  }
  out[tid] = acc + (float)idx;
}

// Width / sharing variants: every lane loads ONE item of W bytes per step; groups of G
// adjacent lanes follow the same chain (same address).
// G < 0: groups of -G lanes that share an address are strided over the wave (lanes l, l + 64/G', ...).
template <int W, int G>
__global__ void __launch_bounds__(256, 6) gather_w(const float4 *__restrict__ nodes, uint32_t mask, int steps, float *out) {
  uint32_t tid = blockIdx.x * 256 + threadIdx.x;
  uint32_t chain = G > 0 ? tid / G : (tid / 64) * (64 / -G) + (tid % 64) % (64 / -G);
  uint32_t idx = chain * 2654435761u + 12345u;
  float acc = 0.f;
  for (int s = 0; s < steps; s++) {
    const float4 *p = nodes + (size_t)(idx & mask) * 4;
    uint32_t dep;
    if (W == 16) { float4 a = p[0]; acc += a.x; dep = __float_as_uint(a.w); }
    else if (W == 8) { float2 a = *reinterpret_cast<const float2 *>(p); acc += a.x; dep = __float_as_uint(a.y) & 0u; }
    else { float a = *reinterpret_cast<const float *>(p); acc += a; dep = __float_as_uint(a) & 0u; }
    idx = idx * 1664525u + 1013904223u + dep;
    idx ^= idx >> 15;
  }
  out[tid] = acc + (float)idx;
}

// The shadow kernel's sharing pattern: runs of G adjacent lanes (the samples of one area light from one
// hit) follow the same chain, every lane loading all NL x 16 bytes of the 64-byte record itself.
template <int NL, int G>
__global__ void __launch_bounds__(256, 6) gather_runs(const float4 *__restrict__ nodes, uint32_t mask, int steps, float *out) {
  uint32_t tid = blockIdx.x * 256 + threadIdx.x;
  uint32_t idx = (tid / G) * 2654435761u + 12345u;
  float acc = 0.f;
  for (int s = 0; s < steps; s++) {
    const float4 *p = nodes + (size_t)(idx & mask) * 4;
    float4 a = p[0];
    acc += a.x;
    uint32_t dep = __float_as_uint(a.w);
    if (NL > 1) { float4 b = p[1]; acc += b.y; dep += __float_as_uint(b.w); }
    if (NL > 2) { float4 c = p[2]; acc += c.z; dep += __float_as_uint(c.w); }
    if (NL > 3) { float4 d = p[3]; acc += d.x; dep += __float_as_uint(d.w); }
    idx = idx * 1664525u + 1013904223u + dep;
    idx ^= idx >> 15;
  }
  out[tid] = acc + (float)idx;
}

// Cooperative form: the four lanes of a quad follow one chain and each loads its own 16-byte
// quarter of the 64-byte record (NCH chains per quad and step = NCH loads per lane).
template <int NCH>
__global__ void __launch_bounds__(256, 6) gather_quad(const float4 *__restrict__ nodes, uint32_t mask, int steps, float *out) {
  uint32_t tid = blockIdx.x * 256 + threadIdx.x;
  uint32_t idx[NCH];
  for (int c = 0; c < NCH; c++) idx[c] = ((tid / 4) * NCH + c) * 2654435761u + 12345u;
  float acc = 0.f;
  for (int s = 0; s < steps; s++) {
    float4 a[NCH];
    for (int c = 0; c < NCH; c++) a[c] = nodes[(size_t)(idx[c] & mask) * 4 + (tid & 3)];
    for (int c = 0; c < NCH; c++) {
      acc += a[c].x;
      idx[c] = idx[c] * 1664525u + 1013904223u + (__float_as_uint(a[c].w) & 0u);
      idx[c] ^= idx[c] >> 15;
    }
  }
  out[tid] = acc + (float)idx[0];
}

// LDS gather: every lane reads NL x 16 bytes of a random 64-byte record held in LDS.
template <int NL>
__global__ void __launch_bounds__(256, 6) gather_lds(int records, int steps, float *out) {
  extern __shared__ __attribute__((aligned(16))) float4 lds[];
  for (int i = threadIdx.x; i < records * 4; i += 256) lds[i] = float4{1.f, 2.f, 3.f, 0.f};
  __syncthreads();
  uint32_t tid = blockIdx.x * 256 + threadIdx.x;
  uint32_t idx = tid * 2654435761u + 12345u;
  float acc = 0.f;
  for (int s = 0; s < steps; s++) {
    const float4 *p = lds + (idx % (uint32_t)records) * 4;
    float4 a = p[0];
    acc += a.x;
    uint32_t dep = __float_as_uint(a.w);
    if (NL > 1) { float4 b = p[1]; acc += b.y; dep += __float_as_uint(b.w); }
    if (NL > 2) { float4 c = p[2]; acc += c.z; dep += __float_as_uint(c.w); }
    if (NL > 3) { float4 d = p[3]; acc += d.x; dep += __float_as_uint(d.w); }
    idx = idx * 1664525u + 1013904223u + dep;
    idx ^= idx >> 15;
  }
  out[tid] = acc + (float)idx;
}

template <typename F>
static double time_cycles(F launch, int steps, double mhz) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch(200);
  CHECK(hipEventRecord(e0));
  launch(steps);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e-3 * mhz * 1e6 / (24.0 * steps);
}

template <int NL, int ACTIVE>
static void run(const float4 *d_nodes, uint32_t mask, float *d_out, int cus, double mhz) {
  const int steps = 2000;
  int grid = cus * 6;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((gather<NL, ACTIVE>), dim3(grid), dim3(256), 0, 0, d_nodes, mask, 200, d_out);
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((gather<NL, ACTIVE>), dim3(grid), dim3(256), 0, 0, d_nodes, mask, steps, d_out);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double wave_steps_per_cu = 24.0 * steps;
  double cyc = ms * 1e-3 * mhz * 1e6 / wave_steps_per_cu;
  printf("  loads %d x16B  active lanes %2d  footprint %8.1f KiB : %7.3f ms  %6.1f cycles per wave-step per CU  (%.1f per lane-load)\n", NL, ACTIVE,
         (mask + 1) * 64.0 / 1024, ms, cyc, cyc / (NL * ACTIVE));
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount; double mhz = prop.clockRate / 1000.0;
  printf("%s, %d CUs, %.0f MHz\n", prop.name, cus, mhz);
  size_t n = 1u << 22;  // 256 MiB of records
  std::vector<float4> h(n * 4, float4{1.f, 2.f, 3.f, 0.f});
  float4 *d; float *out;
  CHECK(hipMalloc(&d, n * 64)); CHECK(hipMalloc(&out, (size_t)cus * 6 * 256 * 4));
  CHECK(hipMemcpy(d, h.data(), n * 64, hipMemcpyHostToDevice));
  for (uint32_t recs : {128u, 4096u, 1u << 16, 1u << 18, 1u << 21}) {
    uint32_t mask = recs - 1;
    run<1, 64>(d, mask, out, cus, mhz);
    run<2, 64>(d, mask, out, cus, mhz);
    run<4, 64>(d, mask, out, cus, mhz);
    run<4, 26>(d, mask, out, cus, mhz);
    run<2, 26>(d, mask, out, cus, mhz);
  }
  const uint32_t mask = (1u << 16) - 1;  // 4 MiB footprint
  int grid = cus * 6;
#define RUNW(W, G) printf("  one %2d-byte load, %3d lanes per address (negative: strided over the wave): %6.1f cycles per wave-step per CU\n", W, G, \
    time_cycles([&](int st) { hipLaunchKernelGGL((gather_w<W, G>), dim3(grid), dim3(256), 0, 0, d, mask, st, out); }, 2000, mhz));
  RUNW(16, 1) RUNW(8, 1) RUNW(4, 1) RUNW(16, 2) RUNW(16, 4) RUNW(16, 16) RUNW(16, 64) RUNW(16, -2) RUNW(16, -4) RUNW(16, -16)
#define RUNR(NL, G) printf("  %d x16B loads of a 64-byte record, runs of %2d adjacent lanes per record: %6.1f cycles per wave-step per CU\n", NL, G, \
    time_cycles([&](int st) { hipLaunchKernelGGL((gather_runs<NL, G>), dim3(grid), dim3(256), 0, 0, d, mask, st, out); }, 2000, mhz));
  RUNR(4, 1) RUNR(4, 2) RUNR(4, 4) RUNR(4, 10) RUNR(4, 16) RUNR(4, 64) RUNR(1, 10)
#define RUNQ(NCH) printf("  quad-cooperative fetch of %d records per quad and step (64 B each, 16 B per lane): %6.1f cycles per wave-step per CU\n", NCH, \
    time_cycles([&](int st) { hipLaunchKernelGGL((gather_quad<NCH>), dim3(grid), dim3(256), 0, 0, d, mask, st, out); }, 2000, mhz));
  RUNQ(1) RUNQ(4)
#define RUNL(NL) printf("  LDS gather, %d x16B of a 64-byte record (320 records): %6.1f cycles per wave-step per CU\n", NL, \
    time_cycles([&](int st) { hipLaunchKernelGGL((gather_lds<NL>), dim3(grid), dim3(256), 320 * 64, 0, 320, st, out); }, 2000, mhz));
  RUNL(1) RUNL(2) RUNL(4)
  return 0;
}
