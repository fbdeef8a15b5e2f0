#include "scene_loader.h"

#include "obj_reader.h"

#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>

namespace host {

// --------------------------------------------------------------------- matrices
Mat4 Mat4::identity() {
  Mat4 r;
  std::memset(r.m, 0, sizeof(r.m));
  r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0;
  return r;
}
Mat4 Mat4::translate(const Vec3 &v) {
  Mat4 r = identity();
  r.m[12] = v[0];
  r.m[13] = v[1];
  r.m[14] = v[2];
  return r;
}
Mat4 Mat4::scale(const Vec3 &v) {
  Mat4 r = identity();
  r.m[0] = v[0];
  r.m[5] = v[1];
  r.m[10] = v[2];
  return r;
}
Mat4 Mat4::rotate(double angle, const Vec3 &axisIn) {
  double len = std::sqrt(axisIn[0] * axisIn[0] + axisIn[1] * axisIn[1] + axisIn[2] * axisIn[2]);
  double a[3] = {axisIn[0] * (1.0 / len), axisIn[1] * (1.0 / len), axisIn[2] * (1.0 / len)};
  double c = std::cos(angle), s = std::sin(angle);
  double t[3] = {(1.0 - c) * a[0], (1.0 - c) * a[1], (1.0 - c) * a[2]};
  Mat4 r = identity();
  // r.m[4*col + row]
  r.m[0] = c + t[0] * a[0];
  r.m[1] = t[0] * a[1] + s * a[2];
  r.m[2] = t[0] * a[2] - s * a[1];
  r.m[4] = t[1] * a[0] - s * a[2];
  r.m[5] = c + t[1] * a[1];
  r.m[6] = t[1] * a[2] + s * a[0];
  r.m[8] = t[2] * a[0] + s * a[1];
  r.m[9] = t[2] * a[1] - s * a[0];
  r.m[10] = c + t[2] * a[2];
  return r;
}
Mat4 Mat4::operator*(const Mat4 &o) const {
  Mat4 r;
  for (int c = 0; c < 4; c++)
    for (int row = 0; row < 4; row++) {
      double acc = 0;
      for (int k = 0; k < 4; k++) acc += m[4 * k + row] * o.m[4 * c + k];
      r.m[4 * c + row] = acc;
    }
  return r;
}

// ----------------------------------------------------------------------- camera
Camera::Camera() : normalizedHeight(1), aspectRatio(1) {
  eye = {{0, 0, 0}};
  std::memset(m, 0, sizeof(m));
  m[0] = m[4] = m[8] = 1.0;
  update();
}
void Camera::setLook(const Vec3 &viewDir, const Vec3 &upDir) {
  Vec3 z{{-viewDir[0], -viewDir[1], -viewDir[2]}};
  const Vec3 &y = upDir;
  Vec3 x{{y[1] * z[2] - z[1] * y[2], y[2] * z[0] - z[2] * y[0], y[0] * z[1] - z[0] * y[1]}};  // cross(y, z)
  for (int r = 0; r < 3; r++) {
    m[0 + r] = x[r];
    m[3 + r] = y[r];
    m[6 + r] = z[r];
  }
  update();
}
void Camera::setFOV(double fov) {
  fov /= (180.0 / 3.14159265359);  // camera.cpp:4
  normalizedHeight = 2 * std::tan(fov / 2);
  update();
}
void Camera::setAspectRatio(double ar) {
  aspectRatio = ar;
  update();
}
void Camera::update() {
  for (int r = 0; r < 3; r++) {
    u[r] = m[0 + r] * normalizedHeight * aspectRatio;
    v[r] = m[3 + r] * normalizedHeight;
    look[r] = -m[6 + r];
  }
}
mipt_camera Camera::toAbi() const {
  mipt_camera c;
  for (int k = 0; k < 3; k++) {
    c.eye[k] = eye[k];
    c.look[k] = look[k];
    c.u[k] = u[k];
    c.v[k] = v[k];
  }
  return c;
}

// ----------------------------------------------------------------------- parser
namespace {

Vec3 vec3(const Json &j) { return Vec3{{j.at(0).number(), j.at(1).number(), j.at(2).number()}}; }

mipt_material default_material() {  // material.h:152-158
  mipt_material m;
  std::memset(&m, 0, sizeof(m));
  m.roughness = 1.0;
  m.metallic = 0.0;
  m.index = 1.0;
  return m;
}

struct Parser {
  SceneData scene;
  mipt_material cur = default_material();
  std::vector<Mat4> stack;
  std::string scene_dir = ".";

  // Scene::getTexture, scene.cpp:140-147: one texture per path.  Returns index + 1.
  int textureIndex(const std::string &name) {
    std::string path = scene_dir + "/" + name;
    for (size_t k = 0; k < scene.textures.size(); k++)
      if (scene.textures[k].path == path) return (int)k + 1;
    SceneData::Texture t;
    t.path = path;
    if (path.size() < 4 || path.compare(path.size() - 4, 4, ".ppm") != 0)
      throw SceneError("Unable to load texture map '" + path + "': this host reads binary PPM textures only");
    read_ppm(path, &t.w, &t.h, &t.rgb);
    scene.textures.push_back(std::move(t));
    return (int)scene.textures.size();
  }

  Mat4 currentTransform() const {  // JsonParser.cpp:59-67
    Mat4 t = Mat4::identity();
    for (const Mat4 &s : stack) t = t * s;
    return t;
  }

  // parseMaterialParameter, JsonParser.cpp:86-111.  Returns false when the key is absent
  // or ill-typed (IGNORE_MISSING leaves the value unchanged).
  // `texture`: where a {"mapped": file} parameter puts its texture (index + 1); null for the
  // parameters shadeBRDF never reads (dropped) and for transmissive (refused).
  bool param(const Json &j, const char *key, double out[3], int32_t *texture = nullptr, bool droppable = true) {
    if (!j.has(key)) return false;
    const Json &p = j.at(key);
    if (p.has("constant")) {
      try {
        Vec3 v = vec3(p.at("constant"));
        out[0] = v[0]; out[1] = v[1]; out[2] = v[2];
        if (texture) *texture = 0;
        return true;
      } catch (const JsonError &) {
        return false;
      }
    }
    if (p.has("mapped")) {
      if (texture) { *texture = textureIndex(p.at("mapped").str); return true; }
      if (droppable) return false;
      throw SceneError("a texture-mapped transmissive colour is not supported");
    }
    throw SceneError("Material parameter must be either constant or mapped");
  }

  mipt_material parseMaterial(const Json &j) {  // JsonParser.cpp:113-131
    mipt_material m = cur;
    double tmp[3];
    param(j, "ambient", m.ka, &m.ka_texture);
    param(j, "diffuse", m.kd, &m.kd_texture);
    param(j, "specular", tmp);
    param(j, "reflective", tmp);
    param(j, "transmissive", m.kt, nullptr, false);
    param(j, "emissive", tmp);
    param(j, "normal", tmp);
    if (j.has("index") && j.at("index").is_number()) m.index = j.at("index").number();
    if (j.has("metallic") && j.at("metallic").is_number()) m.metallic = j.at("metallic").number();
    if (j.has("roughness") && j.at("roughness").is_number()) m.roughness = j.at("roughness").number();
    return m;
  }

  int materialIndex(const Json &body) {  // GET_MAT_W_CUR, JsonParser.cpp:40-41
    scene.materials.push_back(body.has("material") ? parseMaterial(body.at("material")) : cur);
    return (int)scene.materials.size() - 1;
  }

  void addObject(int type, int material, int mesh) {
    mipt_object o;
    std::memset(&o, 0, sizeof(o));
    o.type = type;
    o.material = material;
    o.mesh = mesh;
    Mat4 t = currentTransform();
    std::memcpy(o.xform, t.m, sizeof(o.xform));
    scene.objects.push_back(o);
  }

  void parseTrimesh(const Json &j) {  // JsonParser.cpp:231-311
    int mat = materialIndex(j);
    MeshData mesh;
    for (const Json &p : j.at("points").arr) {
      Vec3 v = vec3(p);
      mesh.vertices.insert(mesh.vertices.end(), v.begin(), v.end());
    }
    size_t nv = mesh.vertices.size() / 3;
    for (const Json &f : j.at("faces").arr) {
      std::vector<int> idx;
      for (const Json &k : f.arr) idx.push_back((int)k.number());
      auto tri = [&](int a, int b, int c) {
        if (a < 0 || b < 0 || c < 0 || (size_t)a >= nv || (size_t)b >= nv || (size_t)c >= nv)
          throw SceneError("Error while adding face. Maybe the point doesn't exist?");
        mesh.faces.push_back(a); mesh.faces.push_back(b); mesh.faces.push_back(c);
      };
      if (idx.size() == 3) tri(idx[0], idx[1], idx[2]);
      else if (idx.size() == 4) { tri(idx[0], idx[1], idx[2]); tri(idx[0], idx[2], idx[3]); }
      else throw SceneError("Got " + std::to_string(idx.size()) + " indices in a face: must be 3 or 4 indices");
    }
    if (j.has("normals")) {
      for (const Json &n : j.at("normals").arr) {
        Vec3 v = vec3(n);
        mesh.normals.insert(mesh.normals.end(), v.begin(), v.end());
      }
      if (mesh.normals.size() != mesh.vertices.size()) throw SceneError("Bad Trimesh: Wrong number of normals.");
    }
    if (j.has("UV")) {  // JsonParser.cpp:271-279
      for (const Json &t : j.at("UV").arr) {
        if (t.arr.size() < 2) throw SceneError("a UV coordinate needs two numbers");
        mesh.uvs.push_back(t.arr[0].number());
        mesh.uvs.push_back(t.arr[1].number());
      }
      if (!mesh.uvs.empty() && mesh.uvs.size() / 2 != nv) throw SceneError("Error in mesh: Bad Trimesh: Wrong number of UV coordinates.");
    }
    if (j.has("materials")) throw SceneError("\"materials\" key is no longer supported: use obj_mesh instead");
    bool gen = false;
    try { if (j.has("gennormals")) gen = j.at("gennormals").boolean(); } catch (const JsonError &) {}
    if (gen) generateNormals(&mesh);
    scene.meshes.push_back(std::move(mesh));
    addObject(MIPT_OBJ_TRIMESH, mat, (int)scene.meshes.size() - 1);
  }

  // parseObjmeshBody + loadObjToTrimesh, JsonParser.cpp:556-754: one Trimesh per shape,
  // each with a fresh Material filled from the FIRST material of the OBJ's library (the
  // scene's current material is not inherited).
  void parseObjmesh(const Json &j) {
    std::string path = scene_dir + "/" + j.at("objfile").str;
    bool gen = false;
    try { if (j.has("gennormals")) gen = j.at("gennormals").boolean(); } catch (const JsonError &) {}
    try {
      ObjData data = read_obj(path, scene_dir);
      for (const ObjShape &shape : data.shapes) {
        ObjMesh om = obj_shape_to_mesh(data, shape);
        mipt_material m = default_material();
        if (!data.materials.empty()) {
          const ObjMaterial &mtl = data.materials[0];
          for (int k = 0; k < 3; k++) { m.kd[k] = mtl.diffuse[k]; m.ka[k] = mtl.ambient[k]; m.kt[k] = mtl.transmittance[k]; }
          m.index = mtl.ior;
          if (!mtl.diffuse_texname.empty()) m.kd_texture = textureIndex(mtl.diffuse_texname);
          if (!mtl.specular_texname.empty()) textureFileCheck(mtl.specular_texname);  // loaded by the reference, never read by shadeBRDF
        }
        scene.materials.push_back(m);
        MeshData mesh;
        mesh.vertices = std::move(om.vertices);
        mesh.faces = std::move(om.faces);
        mesh.normals = std::move(om.normals);
        mesh.uvs = std::move(om.uvs);
        mesh.colors = std::move(om.colors);
        if (gen) generateNormals(&mesh);
        scene.meshes.push_back(std::move(mesh));
        addObject(MIPT_OBJ_TRIMESH, (int)scene.materials.size() - 1, (int)scene.meshes.size() - 1);
      }
    } catch (const ObjError &e) {
      throw SceneError(std::string("Error while parsing OBJ file: ") + e.what());
    }
  }

  void textureFileCheck(const std::string &name) {
    std::ifstream f(scene_dir + "/" + name, std::ios::binary);
    if (!f) throw SceneError("Unable to load texture map '" + scene_dir + "/" + name + "'.");
  }

  // Trimesh::generateNormals, trimesh.cpp:204-231.
  static void generateNormals(MeshData *mesh) {
    size_t nv = mesh->vertices.size() / 3;
    mesh->normals.resize(3 * nv, 0.0);
    std::vector<int> cnt(nv, 0);
    const double *V = mesh->vertices.data();
    for (size_t f = 0; f + 2 < mesh->faces.size(); f += 3) {
      const int *id = &mesh->faces[f];
      double e1[3], e2[3], e3[3];
      for (int k = 0; k < 3; k++) {
        e1[k] = V[3 * id[1] + k] - V[3 * id[0] + k];
        e2[k] = V[3 * id[2] + k] - V[3 * id[0] + k];
        e3[k] = V[3 * id[1] + k] - V[3 * id[2] + k];
      }
      auto len = [](const double *a) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); };
      if (len(e1) == 0.0 || len(e2) == 0.0 || len(e3) == 0.0) continue;  // degenerate faces were never added
      double n[3] = {e1[1] * e2[2] - e2[1] * e1[2], e1[2] * e2[0] - e2[2] * e1[0], e1[0] * e2[1] - e2[0] * e1[1]};
      double inv = 1.0 / len(n);
      for (int v = 0; v < 3; v++) {
        for (int k = 0; k < 3; k++) mesh->normals[3 * id[v] + k] += n[k] * inv;
        cnt[id[v]]++;
      }
    }
    for (size_t v = 0; v < nv; v++)
      if (cnt[v])
        for (int k = 0; k < 3; k++) mesh->normals[3 * v + k] /= cnt[v];
  }

  void parseGeometry(const Json &obj) {  // JsonParser.cpp:313-364
    if (obj.size() != 1) throw SceneError("Geometry must be a one-element object");
    const std::string &key = obj.first_key();
    const Json &val = obj.at(key);
    if (key == "sphere") addObject(MIPT_OBJ_SPHERE, materialIndex(val), -1);
    else if (key == "box") addObject(MIPT_OBJ_BOX, materialIndex(val), -1);
    else if (key == "square") addObject(MIPT_OBJ_SQUARE, materialIndex(val), -1);
    else if (key == "tri_mesh") parseTrimesh(val);
    else if (key == "obj_mesh") parseObjmesh(val);
    else if (key == "cylinder" || key == "cone") throw SceneError(key + " is outside the path-tracing hot path and not supported by this host");
    else throw SceneError("Unknown geometry type: " + key);
  }

  static bool isTransformKey(const std::string &k) { return k == "rotate" || k == "scale" || k == "translate" || k == "transform"; }

  void parseTransform(const Json &obj) {  // JsonParser.cpp:384-445
    const std::string &key = obj.first_key();
    const Json &val = obj.at(key);
    const Json *children = &val.at(1);
    if (key == "rotate") {
      stack.push_back(Mat4::rotate(val.at(1).number(), vec3(val.at(0))));
      children = &val.at(2);
    } else if (key == "scale") {
      stack.push_back(Mat4::scale(vec3(val.at(0))));
    } else if (key == "translate") {
      stack.push_back(Mat4::translate(vec3(val.at(0))));
    } else if (key == "transform") {
      Mat4 t;
      if (val.at(0).size() != 16) throw SceneError("transform needs 16 numbers");
      for (int k = 0; k < 16; k++) t.m[k] = val.at(0).at((size_t)k).number();  // glm::make_mat4: column-major
      stack.push_back(t);
    } else {
      throw SceneError("Unknown transform type: " + key);
    }
    for (const Json &child : children->arr) {
      if (isTransformKey(child.first_key())) parseTransform(child);
      else parseGeometry(child);
    }
    stack.pop_back();
  }

  void parseCamera(const Json &j) {  // JsonParser.cpp:69-84
    Camera c;
    try { c.setEye(vec3(j.at("position"))); } catch (const JsonError &) {}
    if (j.has("updir") || j.has("viewdir")) {
      Vec3 up = vec3(j.at("updir"));
      Vec3 view = vec3(j.at("viewdir"));
      c.setLook(view, up);
    }
    try { c.setFOV(j.at("fov").number()); } catch (const JsonError &) {}
    try { c.setAspectRatio(j.at("aspectRatio").number()); } catch (const JsonError &) {}  // "aspectratio" is a different key
    scene.camera = c;
  }

  void parseLight(const std::string &key, const Json &j) {  // JsonParser.cpp:133-171
    mipt_light L;
    std::memset(&L, 0, sizeof(L));
    Vec3 color = vec3(j.at("color"));
    for (int k = 0; k < 3; k++) L.color[k] = color[k];
    L.atten[0] = 0.0f; L.atten[1] = 0.0f; L.atten[2] = 1.0f;  // JsonParser.cpp:144-146 (unused by directional lights)
    if (key == "directional_light") {
      L.type = MIPT_LIGHT_DIRECTIONAL;
      Vec3 d = vec3(j.at("direction"));
      for (int k = 0; k < 3; k++) L.direction[k] = d[k];
    } else {
      Vec3 p = vec3(j.at("position"));
      for (int k = 0; k < 3; k++) L.position[k] = p[k];
      const char *names[3] = {"constant_attenuation_coeff", "linear_attenuation_coeff", "quadratic_attenuation_coeff"};
      for (int k = 0; k < 3; k++)
        if (j.has(names[k]) && j.at(names[k]).is_number()) L.atten[k] = (float)j.at(names[k]).number();
      if (key == "point_light") {
        L.type = MIPT_LIGHT_POINT;
      } else {
        L.type = MIPT_LIGHT_RECT_AREA;
        Vec3 u = vec3(j.at("u")), v = vec3(j.at("v"));
        for (int k = 0; k < 3; k++) { L.u[k] = u[k]; L.v[k] = v[k]; }
        L.uL = j.at("uL").number();
        L.vL = j.at("vL").number();
      }
    }
    scene.lights.push_back(L);
  }

  void parseScene(const Json &root) {  // JsonParser.cpp:447-515
    for (const Json &obj : root.arr) {
      const std::string &key = obj.first_key();
      const Json &val = obj.at(key);
      if (key == "camera") parseCamera(val);
      else if (key == "material") { cur = default_material(); cur = parseMaterial(val); }
      else if (key == "ambient_light") { Vec3 c = vec3(val.at("color")); for (int k = 0; k < 3; k++) scene.ambient[k] += c[k]; }
      else if (key == "directional_light" || key == "point_light" || key == "rectangle_area_light") parseLight(key, val);
      else if (isTransformKey(key)) parseTransform(obj);
      else if (key == "sphere" || key == "box" || key == "square" || key == "cylinder" || key == "cone" || key == "tri_mesh" || key == "obj_mesh") parseGeometry(obj);
      else throw SceneError("Unknown scene object type: " + key);
    }
  }
};

}  // namespace

SceneData parse_scene(const Json &root, const std::string &scene_dir) {
  if (!root.is_array()) throw SceneError("a scene is a JSON array");
  Parser p;
  p.scene_dir = scene_dir;
  p.parseScene(root);
  return std::move(p.scene);
}

SceneData load_scene_file(const std::string &path) {
  std::ifstream ifs(path);
  if (!ifs) throw SceneError("Error: couldn't read scene file " + path);  // RayTracer.cpp:336-342
  std::stringstream ss;
  ss << ifs.rdbuf();
  size_t slash = path.find_last_of('/');
  return parse_scene(Json::parse(ss.str()), slash == std::string::npos ? "." : path.substr(0, slash));
}

mipt_scene_desc SceneData::describe() {
  mipt_scene_desc d;
  std::memset(&d, 0, sizeof(d));
  d.camera = camera.toAbi();
  for (int k = 0; k < 3; k++) d.ambient[k] = ambient[k];
  d.materials = materials.data();
  d.n_materials = (int32_t)materials.size();
  d.objects = objects.data();
  d.n_objects = (int32_t)objects.size();
  mesh_views_.clear();
  for (MeshData &m : meshes) {
    mipt_mesh v;
    v.vertices = m.vertices.data();
    v.n_vertices = (int64_t)(m.vertices.size() / 3);
    v.faces = m.faces.data();
    v.n_faces = (int64_t)(m.faces.size() / 3);
    v.normals = m.normals.empty() ? nullptr : m.normals.data();
    v.uvs = m.uvs.empty() ? nullptr : m.uvs.data();
    v.colors = m.colors.empty() ? nullptr : m.colors.data();
    mesh_views_.push_back(v);
  }
  d.meshes = mesh_views_.data();
  d.n_meshes = (int32_t)mesh_views_.size();
  d.lights = lights.data();
  d.n_lights = (int32_t)lights.size();
  d.cubemap = nullptr;
  if (!cube_rgb[0].empty()) {
    for (int f = 0; f < 6; f++) {
      cube_views_[f].rgb = cube_rgb[f].data();
      cube_views_[f].width = cube_w[f];
      cube_views_[f].height = cube_h[f];
    }
    d.cubemap = cube_views_;
  }
  texture_views_.clear();
  for (const Texture &t : textures) texture_views_.push_back(mipt_image{t.rgb.data(), t.w, t.h});
  d.textures = texture_views_.empty() ? nullptr : texture_views_.data();
  d.n_textures = (int32_t)texture_views_.size();
  return d;
}

void read_ppm(const std::string &path, int *w, int *h, std::vector<uint8_t> *rgb) {
  std::ifstream in(path, std::ios::binary);
  if (!in) throw SceneError("Unable to load image '" + path + "'.");
  std::string magic;
  int maxv = 0;
  in >> magic >> *w >> *h >> maxv;
  in.get();
  if (magic != "P6" || *w <= 0 || *h <= 0 || maxv != 255) throw SceneError("unsupported PPM: " + path);
  std::vector<uint8_t> top((size_t)3 * *w * *h);
  in.read(reinterpret_cast<char *>(top.data()), (std::streamsize)top.size());
  if (!in) throw SceneError("truncated PPM: " + path);
  // image readers of the reference return rows bottom-up (fileio/pngimage.cpp:210-221)
  rgb->resize(top.size());
  for (int y = 0; y < *h; y++) std::memcpy(&(*rgb)[(size_t)3 * *w * y], &top[(size_t)3 * *w * (*h - 1 - y)], (size_t)3 * *w);
}

void load_cubemap_ppm(const std::string &dir, SceneData *scene) {
  static const char *names[6] = {"posx", "negx", "posy", "negy", "posz", "negz"};  // TraceUI.cc:72-75
  for (int f = 0; f < 6; f++) {
    std::string path = dir + "/" + names[f] + ".ppm";
    {
      std::ifstream probe(path, std::ios::binary);
      if (!probe) throw SceneError("Cannot locate all six cubemap files (missing " + path + ")");
    }
    int w = 0, h = 0;
    read_ppm(path, &w, &h, &scene->cube_rgb[f]);
    scene->cube_w[f] = w;
    scene->cube_h[f] = h;
  }
}

}  // namespace host
