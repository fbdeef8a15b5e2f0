// dump_scene — prints the flat scene description the host loader produces, one number
// per token, for comparison with the Python loader in tests/test_host_cpp.py.
#include <cstdio>

#include "scene_loader.h"

int main(int argc, char **argv) {
  if (argc != 2) { std::fprintf(stderr, "usage: dump_scene scene.json\n"); return 1; }
  try {
    host::SceneData s = host::load_scene_file(argv[1]);
    mipt_scene_desc d = s.describe();
    std::printf("camera");
    for (int k = 0; k < 3; k++) std::printf(" %.17g", d.camera.eye[k]);
    for (int k = 0; k < 3; k++) std::printf(" %.17g", d.camera.look[k]);
    for (int k = 0; k < 3; k++) std::printf(" %.17g", d.camera.u[k]);
    for (int k = 0; k < 3; k++) std::printf(" %.17g", d.camera.v[k]);
    std::printf("\nambient %.17g %.17g %.17g\n", d.ambient[0], d.ambient[1], d.ambient[2]);
    for (int i = 0; i < d.n_objects; i++) {
      const mipt_object &o = d.objects[i];
      const mipt_material &m = d.materials[o.material];
      std::printf("object %d %d", o.type, o.mesh);
      for (int k = 0; k < 16; k++) std::printf(" %.17g", o.xform[k]);
      std::printf(" mat %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", m.kd[0], m.kd[1], m.kd[2], m.ka[0], m.ka[1],
                  m.ka[2], m.kt[0], m.kt[1], m.kt[2], m.roughness, m.metallic, m.index);
      std::printf("mattex %d %d\n", m.kd_texture, m.ka_texture);
    }
    for (int i = 0; i < d.n_meshes; i++) {
      const mipt_mesh &m = d.meshes[i];
      std::printf("mesh %lld %lld %d", (long long)m.n_vertices, (long long)m.n_faces, m.normals ? 1 : 0);
      for (long long k = 0; k < 3 * m.n_vertices; k++) std::printf(" %.17g", m.vertices[k]);
      for (long long k = 0; k < 3 * m.n_faces; k++) std::printf(" %d", m.faces[k]);
      if (m.normals)
        for (long long k = 0; k < 3 * m.n_vertices; k++) std::printf(" %.17g", m.normals[k]);
      std::printf("\n");
      if (m.uvs) {
        std::printf("uv");
        for (long long k = 0; k < 2 * m.n_vertices; k++) std::printf(" %.17g", m.uvs[k]);
        std::printf("\n");
      }
      if (m.colors) {
        std::printf("color");
        for (long long k = 0; k < 3 * m.n_vertices; k++) std::printf(" %.17g", m.colors[k]);
        std::printf("\n");
      }
    }
    for (int i = 0; i < d.n_textures; i++) {
      const mipt_image &t = d.textures[i];
      unsigned long long sum = 0;
      for (long long k = 0; k < 3ll * t.width * t.height; k++) sum = sum * 31ull + t.rgb[k];
      std::printf("texture %d %d %llu %d %d %d\n", t.width, t.height, sum, t.rgb[0], t.rgb[1], t.rgb[2]);
    }
    for (int i = 0; i < d.n_lights; i++) {
      const mipt_light &L = d.lights[i];
      std::printf("light %d", L.type);
      for (int k = 0; k < 3; k++) std::printf(" %.17g", L.color[k]);
      for (int k = 0; k < 3; k++) std::printf(" %.17g", L.position[k]);
      for (int k = 0; k < 3; k++) std::printf(" %.17g", L.direction[k]);
      for (int k = 0; k < 3; k++) std::printf(" %.17g", L.u[k]);
      for (int k = 0; k < 3; k++) std::printf(" %.17g", L.v[k]);
      std::printf(" %.17g %.17g %.9g %.9g %.9g\n", L.uL, L.vL, L.atten[0], L.atten[1], L.atten[2]);
    }
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 2;
  }
  return 0;
}
