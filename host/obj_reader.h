// obj_reader.h — Wavefront OBJ / MTL ingest for `obj_mesh` geometry
// (parser/JsonParser.cpp:529-754).  The reference goes through tinyobjloader 2.0
// (ray/src/libs/tiny_obj_loader.h, doubles, triangulation on, vertex_color off) and then
// re-indexes each shape into a Trimesh (loadObjToTrimesh, JsonParser.cpp:556-690); this
// reader restates what a scene can observe of that:
//   * numbers assembled digit by digit as tiny_obj_loader.h:915-1083 does (so "0.35" is
//     3*0.1 + 5*0.01, not the correctly rounded double);
//   * v [r g b], vn, vt, f with i, i/j, i//k, i/j/k corners and relative indices; `g` / `o`
//     start a new shape, `usemtl` does not; colours only if every vertex has them;
//   * quads cut along the shorter diagonal (:1642-1743), larger polygons by the loader's
//     built-in ear clipping (:1886-2132);
//   * MTL: newmtl, Ka, Kd, Ks, Kt/Tf, Ke, Ns, Ni, map_Kd, map_Ks.
#ifndef MIPT_HOST_OBJ_READER_H_
#define MIPT_HOST_OBJ_READER_H_

#include <array>
#include <stdexcept>
#include <string>
#include <vector>

namespace host {

struct ObjError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

struct ObjMaterial {  // tinyobj::material_t after InitMaterial, tiny_obj_loader.h:1501-1540
  std::string name;
  double ambient[3] = {0, 0, 0}, diffuse[3] = {0, 0, 0}, specular[3] = {0, 0, 0};
  double transmittance[3] = {0, 0, 0}, emission[3] = {0, 0, 0};
  double shininess = 1.0, ior = 1.0;
  std::string diffuse_texname, specular_texname;
};

struct ObjCorner {
  int v, vt, vn;  // zero-based, -1 = absent
  bool operator==(const ObjCorner &o) const { return v == o.v && vt == o.vt && vn == o.vn; }
};

struct ObjShape {
  std::string name;
  std::vector<ObjCorner> corners;  // three per triangle
};

struct ObjData {
  std::vector<double> vertices, normals, texcoords, colors;  // colors empty unless every vertex has one
  std::vector<ObjShape> shapes;
  std::vector<ObjMaterial> materials;
};

// One blank-delimited token -> double; false if it does not start with a number.
bool obj_parse_double(const std::string &tok, double *out);

// Material libraries are looked up in mtl_dir (the scene's directory, JsonParser.cpp:702).
ObjData read_obj(const std::string &path, const std::string &mtl_dir);

// loadObjToTrimesh, JsonParser.cpp:556-625: one mesh vertex per distinct corner in order of
// first use, normals normalised.  Arrays the shape does not have stay empty.
struct ObjMesh {
  std::vector<double> vertices, normals, uvs, colors;
  std::vector<int32_t> faces;
};
ObjMesh obj_shape_to_mesh(const ObjData &data, const ObjShape &shape);

}  // namespace host
#endif
