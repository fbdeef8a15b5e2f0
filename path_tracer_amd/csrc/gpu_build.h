// gpu_build.h — BVH construction on the GPU (nearest-neighbour clustering of the primitives in
// Morton order), the alternative to the host's binned-SAH builder of scene_build.cpp.  Same
// output format (layout.h: Node4 with 8-bit quantised child boxes, primitives in leaf order),
// built in tens of milliseconds instead of seconds.  Results of the renderer do not depend on
// the tree (the box tests only cull).
#ifndef MIPT_GPU_BUILD_H_
#define MIPT_GPU_BUILD_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "layout.h"

namespace mipt {

struct GpuBvh {
  Node4 *nodes = nullptr;     // device, n_nodes used (allocated for n_prims); caller frees with hipFree
  uint32_t *order = nullptr;  // device, n_prims: order[s] = input index of the primitive stored at s; caller frees
  int n_nodes = 0;
  int levels = 0;             // depth of the four-wide tree
  int rounds = 0;             // clustering rounds the binary tree took
  int n_analytic = 0;         // primitives [0, n_analytic) of the new order are not triangles
};

// d_boxes: n x 6 floats (min xyz, max xyz), already rounded outward from the fp64 boxes.
// d_prim_info: kind << 30 | object per primitive (kind 3 = triangle).  bounds_min/max: scene
// bounds.  n must be at least 3.  Returns hipSuccess or the first error.
hipError_t gpu_build_bvh(const float *d_boxes, const uint32_t *d_prim_info, uint32_t n, const double bounds_min[3], const double bounds_max[3],
                         int leaf_max, GpuBvh *out, hipStream_t st);

// dst[s] = src[order[s]] for records of `bytes` bytes (a multiple of 4).
void gpu_gather_records(void *dst, const void *src, const uint32_t *order, uint32_t n, uint32_t bytes, hipStream_t st);

}  // namespace mipt
#endif
