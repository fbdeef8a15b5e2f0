// scene_loader.h — the host's scene ingest: reads the reference's JSON scene format
// (parser/JsonParser.cpp, parser/jsonformat.md) into the flat description libmipt takes.
#ifndef MIPT_HOST_SCENE_LOADER_H_
#define MIPT_HOST_SCENE_LOADER_H_

#include <array>
#include <string>
#include <vector>

#include "../include/mipt.h"
#include "json.h"

namespace host {

using Vec3 = std::array<double, 3>;

// Column-major 4x4, m[4*c + r] — glm::dmat4 memory order, as mipt_object.xform.
struct Mat4 {
  double m[16];
  static Mat4 identity();
  static Mat4 translate(const Vec3 &v);
  static Mat4 scale(const Vec3 &v);
  static Mat4 rotate(double angle, const Vec3 &axis);  // glm::rotate, radians
  Mat4 operator*(const Mat4 &o) const;
};

// scene/camera.{h,cpp}: the camera stays on the host.
class Camera {
 public:
  Camera();
  void setEye(const Vec3 &e) { eye = e; }
  void setLook(const Vec3 &viewDir, const Vec3 &upDir);  // camera.cpp:57-67
  void setFOV(double fovDegrees);                        // camera.cpp:69-75
  void setAspectRatio(double ar);                        // camera.cpp:77-82
  double getAspectRatio() const { return aspectRatio; }
  mipt_camera toAbi() const;
  Vec3 eye, look, u, v;

 private:
  void update();  // camera.cpp:84-88
  double m[9];    // column-major 3x3
  double normalizedHeight, aspectRatio;
};

struct MeshData {
  std::vector<double> vertices;
  std::vector<int32_t> faces;
  std::vector<double> normals;
  std::vector<double> uvs;  // 2 per vertex, or empty (the "UV" key)
  std::vector<double> colors;  // 3 per vertex, or empty (OBJ files with "v x y z r g b" only)
};

struct SceneData {
  Camera camera;
  Vec3 ambient{{0, 0, 0}};
  std::vector<mipt_material> materials;
  std::vector<mipt_object> objects;
  std::vector<MeshData> meshes;
  std::vector<mipt_light> lights;
  // cubemap faces (+x,-x,+y,-y,+z,-z), RGB8, row 0 = bottom; empty = none
  std::vector<uint8_t> cube_rgb[6];
  int cube_w[6] = {0, 0, 0, 0, 0, 0}, cube_h[6] = {0, 0, 0, 0, 0, 0};
  // material textures ({"mapped": file}), RGB8, row 0 = bottom
  struct Texture { std::string path; int w = 0, h = 0; std::vector<uint8_t> rgb; };
  std::vector<Texture> textures;

  // Views for mipt_upload_scene; valid while *this is alive and unchanged.
  mipt_scene_desc describe();

 private:
  std::vector<mipt_mesh> mesh_views_;
  mipt_image cube_views_[6];
  std::vector<mipt_image> texture_views_;
};

struct SceneError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// JsonParser::parseScene (JsonParser.cpp:447-515).  Throws SceneError / JsonError.
SceneData load_scene_file(const std::string &path);
// scene_dir: where {"mapped": file} textures are looked up (the scene file's directory,
// JsonParser.cpp:95-96).  This host reads binary PPM (P6) textures only.
SceneData parse_scene(const Json &root, const std::string &scene_dir = ".");

// One binary PPM (P6, maxval 255) -> RGB8 with rows bottom-up, as the reference's image
// readers return them (fileio/pngimage.cpp:210-221).
void read_ppm(const std::string &path, int *w, int *h, std::vector<uint8_t> *rgb);

// Six binary PPM (P6) files posx/negx/posy/negy/posz/negz.ppm in `dir` -> cubemap.
void load_cubemap_ppm(const std::string &dir, SceneData *scene);

}  // namespace host
#endif
