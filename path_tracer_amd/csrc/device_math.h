// device_math.h — fp64 vector helpers and the counter-based random stream used by
// every kernel.  The arithmetic mirrors what glm computes in the reference:
// normalize(v) = v * (1/sqrt(dot(v,v))), reflect(I,N) = I - N*dot(N,I)*2.
#ifndef MIPT_DEVICE_MATH_H_
#define MIPT_DEVICE_MATH_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mipt {

struct d3 {
  double x, y, z;
};

__device__ __forceinline__ d3 mk(double x, double y, double z) { return d3{x, y, z}; }
__device__ __forceinline__ d3 ld3(const double *p) { return d3{p[0], p[1], p[2]}; }
__device__ __forceinline__ d3 operator+(d3 a, d3 b) { return d3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ d3 operator-(d3 a, d3 b) { return d3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ d3 operator-(d3 a) { return d3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ d3 operator*(d3 a, d3 b) { return d3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ d3 operator*(d3 a, double s) { return d3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ d3 operator*(double s, d3 a) { return d3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ d3 operator/(d3 a, double s) { return d3{a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ double dot(d3 a, d3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ d3 cross(d3 a, d3 b) {
  return d3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
__device__ __forceinline__ double length(d3 a) { return sqrt(dot(a, a)); }
__device__ __forceinline__ d3 normalize(d3 a) { return a * (1.0 / sqrt(dot(a, a))); }
__device__ __forceinline__ double comp(d3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// Row-major 3x4 affine map (rows of the inverse transform): M * (v, 1).
__device__ __forceinline__ d3 xform_point(const double *m, d3 v) {
  return d3{m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3], m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7],
            m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11]};
}
__device__ __forceinline__ d3 mul3(const double *m, d3 v) {
  return d3{m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z};
}

// ---- counter-based random stream ------------------------------------------------
// mix(key, j): splitmix64 finaliser of key + (j+1)*golden.  u01 keeps 53 bits.
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t key, uint64_t j) {
  uint64_t z = key + (j + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ double u01(uint64_t z) { return (double)(z >> 11) * (1.0 / 9007199254740992.0); }
__host__ __device__ __forceinline__ uint64_t sample_key(uint64_t seed, uint64_t pixel, uint64_t sample) {
  return mix64(mix64(mix64(seed, 0x5EEDull), pixel), sample);
}
__host__ __device__ __forceinline__ double radical_inverse2(uint32_t v) {
  v = (v << 16) | (v >> 16);
  v = ((v & 0x00ff00ffu) << 8) | ((v & 0xff00ff00u) >> 8);
  v = ((v & 0x0f0f0f0fu) << 4) | ((v & 0xf0f0f0f0u) >> 4);
  v = ((v & 0x33333333u) << 2) | ((v & 0xccccccccu) >> 2);
  v = ((v & 0x55555555u) << 1) | ((v & 0xaaaaaaaau) >> 1);
  return (double)v * (1.0 / 4294967296.0);
}

}  // namespace mipt
#endif
