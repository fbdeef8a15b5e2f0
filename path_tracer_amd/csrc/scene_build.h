// scene_build.h — host side of mipt_upload_scene: turns the scene description into
// the flat arrays the kernels read (see layout.h).
#ifndef MIPT_SCENE_BUILD_H_
#define MIPT_SCENE_BUILD_H_

#include <string>
#include <vector>

#include "../../include/mipt.h"
#include "layout.h"

#include <cstdlib>
#include <cstring>

namespace mipt {

// Tuning and test knobs read from the environment (MIPT_TAIL_RAYS, MIPT_LDS_STACK, ...) count only in a process
// that sets MIPT_DEBUG=1: a library whose kernels are chosen by stray environment variables is a support hazard.
// What a caller may want to choose is in mipt_set_option (include/mipt.h).
inline const char *debug_env(const char *name) {
  const char *gate = std::getenv("MIPT_DEBUG");
  if (!gate || std::strcmp(gate, "0") == 0 || gate[0] == 0) return nullptr;
  return std::getenv(name);
}

struct HostScene {
  CameraRec camera;
  double ambient[3];
  std::vector<Node4> nodes;            // nodes[0] is the root
  std::vector<Tri> tris;               // per primitive (leaf order); unused slots for analytic prims
  std::vector<TriNormals> tri_normals; // per primitive, empty if no mesh has vertex normals
  std::vector<TriUV> tri_uvs;          // per primitive, empty if no mesh has UV coordinates
  std::vector<TriColor> tri_colors;    // per primitive, empty if no mesh shades with vertex colours
  std::vector<uint32_t> prim_info;     // kind << 30 | object
  int n_analytic = 0;                  // primitives [0, n_analytic) are spheres/boxes/squares, the rest triangles
  std::vector<uint32_t> ref_order;     // reference visiting order of each primitive (tie-break key)
  std::vector<float> prim_boxes;       // flat mode only: 6 floats per primitive (min, max), rounded outward
  double bounds_min[3] = {0, 0, 0}, bounds_max[3] = {0, 0, 0};  // scene bounds
  std::vector<ObjRec> objects;
  std::vector<double> object_boxes;    // per object: its world box (min xyz, max xyz), grown by 1e-9 of the scene's size
  std::vector<MatRec> materials;
  std::vector<LightRec> lights;
  int rays_per_hit = 0;                // shadow rays per shaded hit for a given area_samples
  int max_depth = 0;                   // deepest inner-node nesting
  int stack_need = 0;                  // most entries a traversal can hold on its stack at once
  double radius = 0;                   // max |coordinate| of the scene bounds
  bool translucent = false;            // some material has kt != 0
  bool flat_lights = true;             // every light's shadow rays have a sample-independent attenuation (LightRec::flat)
  bool local_meshes = false;           // some mesh object has ObjRec::local_units set
  bool has_cubemap = false;
  int cube_w[6], cube_h[6];
  std::vector<uint8_t> cube_rgb[6];
  std::vector<std::vector<uint8_t>> tex_rgb;  // material textures
  std::vector<int> tex_w, tex_h;
};

// Returns MIPT_OK or an error code with `err` set.  build_tree = false ("flat mode") leaves
// `nodes` empty and the primitive arrays in reference order (ref_order[s] = s), with
// prim_boxes filled: the input of the GPU builder (gpu_build.h).
int build_host_scene(const mipt_scene_desc *desc, HostScene *out, std::string *err, bool build_tree = true);

// Replays BVH<T>::makeBVH (scene/bvh.h:40-110) on a list of boxes and returns the
// final allNodes order: order[k] = index of the k-th item the reference's
// left-to-right traversal visits.  boxes: 6 doubles per item (min xyz, max xyz).
void reference_visit_order(const std::vector<double> &boxes, std::vector<uint32_t> *order);

}  // namespace mipt
#endif
