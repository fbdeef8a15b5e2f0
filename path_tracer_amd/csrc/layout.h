// layout.h — device data layout shared by the host-side scene builder and the HIP
// kernels of libmipt.  Plain structs, no HIP types.
#ifndef MIPT_LAYOUT_H_
#define MIPT_LAYOUT_H_

#include <stdint.h>

namespace mipt {

// Primitive kinds (top two bits of PrimInfo).
enum : uint32_t { KIND_SPHERE = 0, KIND_BOX = 1, KIND_SQUARE = 2, KIND_TRI = 3 };

// One BVH4 node, 64 bytes, fetched as four 16-byte loads.  The boxes of up to four
// children are 8-bit plane offsets in a frame of the node's own: on axis a child k spans
//   [origin[a] + lo_k * step_a, origin[a] + hi_k * step_a],   step_a = 2^(exp[a] - 127),
// with lo_k / hi_k the k-th bytes of lo[a] / hi[a].  The builder rounds the origin and
// every lower plane down and every upper plane up, so a quantised box contains the
// fp64 box of its subtree: the boxes only cull, every accepted primitive is tested in
// fp64, and a coarser box costs time, never a hit.
//   child >= 0          : index of an inner node
//   child <  0          : leaf, ~child = (first << 3) | (count - 1), count in 1..8,
//                         `first` indexes the primitive arrays (stored in leaf order)
//   child == kNoChild   : absent (children are packed to the front)
constexpr int32_t kNoChild = (int32_t)0x80000000;
struct alignas(16) Node4 {
  float origin[3];
  uint8_t exp[3];
  uint8_t n_children;
  uint32_t lo[3];
  uint32_t hi[3];
  uint32_t pad[2];
  int32_t child[4];
};
static_assert(sizeof(Node4) == 64, "Node4 must be 64 bytes");

// One triangle in world space, 96 bytes (six 16-byte loads):
//   a  : first vertex
//   n  : unit geometric normal as the reference delivers it
//        (normalize(normalMatrix * cross(b-a, c-a)), trimesh.h:145-146, scene.h:92-94)
//   q1, q2 : dual edge vectors, beta = (P-a).q1, gamma = (P-a).q2 — the inverse Gram
//        matrix of trimesh.h:152-157 folded into the edges
// The plane (n, a) fills the first three 16-byte words, the edge vectors the last three:
// a test fetches the first half, and the second only if the plane is hit within range.
struct alignas(16) Tri {
  double n[3];
  double a[3];
  double q1[3];
  double q2[3];
};
static_assert(sizeof(Tri) == 96, "Tri must be 96 bytes");

// Per-vertex normals of one triangle, already multiplied by the object's normal
// matrix (not normalised): N = normalize(w0*na + w1*nb + w2*nc), trimesh.cpp:154-163.
struct alignas(8) TriNormals {
  double na[3], nb[3], nc[3];
};

// Per-vertex UV coordinates of one triangle (the mesh's "UV" key), in the order of the
// barycentric weights: uv = (ua, va) * w0 + (ub, vb) * w1 + (uc, vc) * w2, trimesh.cpp:165-172.
struct alignas(8) TriUV {
  double ua, va, ub, vb, uc, vc;
};

// Per-vertex colours of one triangle (only an OBJ file can give a mesh colours,
// JsonParser.cpp:609-612): on a mesh without UV coordinates the hit's diffuse colour is
// ca * w0 + cb * w1 + cc * w2, trimesh.cpp:179-189.
struct alignas(8) TriColor {
  double ca[3], cb[3], cc[3];
};

// One material texture: 8-bit RGB, row 0 = bottom row.
struct TexRec {
  const uint8_t *data;
  int32_t w, h;
};

// One scene object (file order).  inv = rows 0..2 of the inverse transform,
// row-major 3x4 (MatrixTransform::globalToLocalCoords, scene.h:80-82); normi =
// transpose(inverse(mat3(xform))), row-major 3x3 (scene.h:76,92-94); fwdt =
// transpose(mat3(xform)), row-major 3x3: fwdt * (world normal of a face) is parallel to the
// face's local normal and |fwdt * n| is what the local unit normal shrinks by on its way to
// world space.
struct alignas(8) ObjRec {
  double inv[12];
  double normi[9];
  double fwdt[9];
  int32_t kind;
  int32_t material;
  int32_t smooth;  // mesh has per-vertex normals
  int32_t has_uv;  // mesh has per-vertex UV coordinates
  int32_t has_colors;  // mesh has per-vertex colours and no UV coordinates (trimesh.cpp:170-189)
  // Mesh whose local units differ from world units enough to matter: the reference tests
  // faces in the mesh's local space, where a world distance t reads t * |inv3 * d|
  // (scene.cpp:18-22,33), and two of its thresholds are therefore in local units: hits at
  // local t >= t_max are misses (trimesh.cpp:85-86) and rays with |n.d| < 1e-8 between the
  // local unit normal and the local unit direction miss the face (trimesh.cpp:107-109).
  // Set when the inverse transform stretches some direction (largest singular value > 1)
  // or is anisotropic (the two thresholds then differ between the spaces); triangles of
  // such an object are tested with both thresholds carried over to world space.
  int32_t local_units;
};

// Material record as shadeBRDF reads it (material.cpp:88-109).  Scalars already
// passed through the luminance weights of material.cpp:232-241.
struct alignas(8) MatRec {
  double kd[3];
  double ka[3];
  double kt[3];
  double roughness;  // as the mirror test reads it (RayTracer.cpp:269)
  double metallic;
  double index;
  int32_t kd_tex, ka_tex;  // 0 = the constant, k > 0 = texture k - 1 at the hit's UV (material.cpp:224-230)
};

struct alignas(8) LightRec {
  int32_t type;
  int32_t samples;      // shadow rays this light fires per shading point
  int32_t first_ray;    // offset of its first ray in the per-hit shadow-ray block
  int32_t flat;         // the attenuation of its shadow rays does not depend on the sample: flat_scale
  double color[3];
  double position[3];     // point: position; area: corner
  double orientation[3];  // directional: normalised orientation (light.h:46)
  double u[3], v[3];
  double center[3];       // area: light.h:124
  double uL, vL;
  double a, b, c;         // attenuation coefficients (float in the reference, widened)
  double flat_scale;      // flat: min(1, 1/denom) of light.cpp:139-142 with b = c = 0 (1 for point and directional lights)
};

struct CameraRec {
  double eye[3], look[3], u[3], v[3];
};

// PrimInfo: kind << 30 | object index (file order).
static inline uint32_t make_prim_info(uint32_t kind, uint32_t obj) { return (kind << 30) | obj; }

// Counter-based random stream: slot numbers of one path node.
enum : uint64_t {
  SLOT_RR = 0,
  SLOT_COS = 1,
  SLOT_PHI = 2,
  SLOT_MIRROR = 3,
  SLOT_CHILD_BOUNCE = 4,
  SLOT_CHILD_MIRROR = 5,
  SLOT_AREA = 16
};

}  // namespace mipt
#endif
