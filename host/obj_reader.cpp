#include "obj_reader.h"

#include <cmath>
#include <cstdint>
#include <fstream>
#include <map>
#include <set>
#include <sstream>
#include <tuple>

namespace host {

namespace {

bool is_digit(char c) { return c >= '0' && c <= '9'; }
bool is_blank(char c) { return c == ' ' || c == '\t'; }

std::vector<std::string> split_blank(const std::string &s) {
  std::vector<std::string> out;
  size_t k = 0, n = s.size();
  while (k < n) {
    while (k < n && (is_blank(s[k]) || s[k] == '\r')) k++;
    size_t e = k;
    while (e < n && !is_blank(s[e]) && s[e] != '\r') e++;
    if (e > k) out.push_back(s.substr(k, e - k));
    k = e;
  }
  return out;
}

void reals(const std::vector<std::string> &tok, size_t first, int count, double *out, double dflt = 0.0) {
  for (int k = 0; k < count; k++) {
    double v;
    out[k] = (first + k < tok.size() && obj_parse_double(tok[first + k], &v)) ? v : dflt;
  }
}

// C atoi on the start of the text.
int leading_int(const std::string &s) {
  size_t k = 0;
  int sign = 1;
  if (k < s.size() && (s[k] == '+' || s[k] == '-')) { sign = s[k] == '-' ? -1 : 1; k++; }
  long long v = 0;
  bool any = false;
  while (k < s.size() && is_digit(s[k])) { v = v * 10 + (s[k] - '0'); k++; any = true; }
  return any ? (int)(sign * v) : 0;
}

int fix_index(const std::string &text, int count) {  // tiny_obj_loader.h:842-868
  int i = leading_int(text);
  if (i == 0) throw ObjError("Failed parse `f' line(e.g. zero value for face index.");
  return i > 0 ? i - 1 : count + i;
}

ObjCorner parse_corner(const std::string &tok, int nv, int nvn, int nvt) {  // tiny_obj_loader.h:1253-1313
  std::vector<std::string> parts;
  size_t start = 0;
  for (size_t k = 0; k <= tok.size(); k++)
    if (k == tok.size() || tok[k] == '/') { parts.push_back(tok.substr(start, k - start)); start = k + 1; }
  ObjCorner c{fix_index(parts[0], nv), -1, -1};
  if (parts.size() == 2) c.vt = fix_index(parts[1], nvt);
  else if (parts.size() >= 3) {
    if (!parts[1].empty()) c.vt = fix_index(parts[1], nvt);
    c.vn = fix_index(parts[2], nvn);
  }
  return c;
}

// File name of a map_* statement (tiny_obj_loader.h:1352-1469): options and their arguments
// are skipped; the first other token starts the name, which runs to the end of the line.
std::string texture_name(const std::string &rest) {
  static const std::map<std::string, int> opts = {{"-blendu", 1}, {"-blendv", 1}, {"-clamp", 1}, {"-boost", 1}, {"-bm", 1},
                                                  {"-o", 3}, {"-s", 3}, {"-t", 3}, {"-type", 1}, {"-texres", 1},
                                                  {"-imfchan", 1}, {"-mm", 2}, {"-colorspace", 1}};
  size_t pos = 0, n = rest.size();
  while (pos < n) {
    while (pos < n && is_blank(rest[pos])) pos++;
    size_t end = pos;
    while (end < n && !is_blank(rest[end])) end++;
    auto it = opts.find(rest.substr(pos, end - pos));
    if (it != opts.end() && end < n) {
      pos = end;
      for (int a = 0; a < it->second; a++) {
        while (pos < n && is_blank(rest[pos])) pos++;
        while (pos < n && !is_blank(rest[pos])) pos++;
      }
      continue;
    }
    return rest.substr(pos);
  }
  return "";
}

std::string trim_line(std::string line, bool trailing_blanks) {
  while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
  if (trailing_blanks)
    while (!line.empty() && is_blank(line.back())) line.pop_back();
  size_t k = 0;
  while (k < line.size() && is_blank(line[k])) k++;
  return line.substr(k);
}

bool read_mtl(const std::string &path, std::vector<ObjMaterial> *out) {  // tiny_obj_loader.h:2263-2700
  std::ifstream in(path);
  if (!in) return false;
  ObjMaterial cur;
  std::string raw;
  while (std::getline(in, raw)) {
    std::string line = trim_line(raw, true);
    if (line.empty() || line[0] == '#') continue;
    size_t sp = 0;
    while (sp < line.size() && !is_blank(line[sp])) sp++;
    std::string key = line.substr(0, sp);
    size_t rs = sp;
    while (rs < line.size() && is_blank(line[rs])) rs++;
    std::string rest = line.substr(rs);
    std::vector<std::string> tok = split_blank(rest);
    if (key == "newmtl") {
      if (!cur.name.empty()) out->push_back(cur);
      cur = ObjMaterial();
      cur.name = line.size() > 7 ? line.substr(7) : "";
    } else if (key == "Ka") reals(tok, 0, 3, cur.ambient);
    else if (key == "Kd") reals(tok, 0, 3, cur.diffuse);
    else if (key == "Ks") reals(tok, 0, 3, cur.specular);
    else if (key == "Kt" || key == "Tf") reals(tok, 0, 3, cur.transmittance);
    else if (key == "Ke") reals(tok, 0, 3, cur.emission);
    else if (key == "Ni") reals(tok, 0, 1, &cur.ior);
    else if (key == "Ns") reals(tok, 0, 1, &cur.shininess);
    else if (key == "map_Kd") cur.diffuse_texname = texture_name(rest);
    else if (key == "map_Ks") cur.specular_texname = texture_name(rest);
  }
  if (!cur.name.empty()) out->push_back(cur);
  return true;
}

// Crossing-number test of (tx, ty) against a triangle, tiny_obj_loader.h:1557-1569.
bool inside_triangle(const double vx[3], const double vy[3], double tx, double ty) {
  bool c = false;
  for (int i = 0, j = 2; i < 3; j = i++)
    if (((vy[i] > ty) != (vy[j] > ty)) && (tx < (vx[j] - vx[i]) * (ty - vy[i]) / (vy[j] - vy[i]) + vx[i])) c = !c;
  return c;
}

// Polygons with five or more corners, cut the way the reference's loader does it (built-in
// ear clipping, tiny_obj_loader.h:1886-2132), quirks included: projection on two axes chosen
// from the first non-degenerate corner, a convexity test that multiplies the corner's cross
// product by (x0 * y1 - y0 * x1) / 2 of its first two points, and a walk that gives up after
// a full turn without an ear (what is left is dropped unless it is a triangle).
void ear_clip(const std::vector<ObjCorner> &face, const std::vector<double> &v, std::vector<ObjCorner> *out) {
  const size_t n = face.size();
  const double eps = 2.220446049250313e-16;
  int axes[2] = {1, 2};
  for (size_t k = 0; k < n; k++) {
    const double *p0 = &v[3 * (size_t)face[k % n].v], *p1 = &v[3 * (size_t)face[(k + 1) % n].v], *p2 = &v[3 * (size_t)face[(k + 2) % n].v];
    double e0[3], e1[3];
    for (int a = 0; a < 3; a++) { e0[a] = p1[a] - p0[a]; e1[a] = p2[a] - p1[a]; }
    double cx = std::fabs(e0[1] * e1[2] - e0[2] * e1[1]);
    double cy = std::fabs(e0[2] * e1[0] - e0[0] * e1[2]);
    double cz = std::fabs(e0[0] * e1[1] - e0[1] * e1[0]);
    if (cx > eps || cy > eps || cz > eps) {
      if (!(cx > cy && cx > cz)) {
        axes[0] = 0;
        if (cz > cx && cz > cy) axes[1] = 1;
      }
      break;
    }
  }
  std::vector<ObjCorner> rest = face;
  size_t guess = 0, budget = n, before = n;
  while (rest.size() > 3 && budget > 0) {
    const size_t m = rest.size();
    if (guess >= m) guess -= m;
    if (before != m) { before = m; budget = m; }
    else budget--;
    ObjCorner tri[3];
    double vx[3], vy[3];
    for (int k = 0; k < 3; k++) {
      tri[k] = rest[(guess + k) % m];
      vx[k] = v[3 * (size_t)tri[k].v + axes[0]];
      vy[k] = v[3 * (size_t)tri[k].v + axes[1]];
    }
    double cross = (vx[1] - vx[0]) * (vy[2] - vy[1]) - (vy[1] - vy[0]) * (vx[2] - vx[1]);
    double area = (vx[0] * vy[1] - vy[0] * vx[1]) * 0.5;
    if (cross * area < 0.0) { guess++; continue; }
    bool overlap = false;
    for (size_t other = 3; other < m && !overlap; other++) {
      const double *q = &v[3 * (size_t)rest[(guess + other) % m].v];
      overlap = inside_triangle(vx, vy, q[axes[0]], q[axes[1]]);
    }
    if (overlap) { guess++; continue; }
    out->insert(out->end(), tri, tri + 3);
    rest.erase(rest.begin() + (long)((guess + 1) % m));
  }
  if (rest.size() == 3) out->insert(out->end(), rest.begin(), rest.end());
}

}  // namespace

bool obj_parse_double(const std::string &tok, double *out) {  // tiny_obj_loader.h:915-1083
  static const double neg_pow10[8] = {1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001};
  const size_t n = tok.size();
  if (n == 0) return false;
  size_t k = 0;
  bool negative = false, dot_first = false;
  if (tok[0] == '+' || tok[0] == '-') {
    negative = tok[0] == '-';
    k = 1;
    dot_first = k < n && tok[k] == '.';
  } else if (tok[0] == '.') {
    dot_first = true;
  } else if (!is_digit(tok[0])) {
    return false;
  }
  double mant = 0.0;
  if (!dot_first) {
    size_t start = k;
    while (k < n && is_digit(tok[k])) { mant = mant * 10 + (tok[k] - '0'); k++; }
    if (k == start) return false;
  }
  int exp10 = 0;
  if (k < n && tok[k] == '.') {
    k++;
    int place = 1;
    while (k < n && is_digit(tok[k])) {
      mant += (tok[k] - '0') * (place < 8 ? neg_pow10[place] : std::pow(10.0, -place));
      place++;
      k++;
    }
  } else if (k < n && tok[k] != 'e' && tok[k] != 'E') {
    k = n;  // any other character ends the number
  }
  if (k < n && (tok[k] == 'e' || tok[k] == 'E')) {
    k++;
    bool exp_neg = false;
    if (k < n && (tok[k] == '+' || tok[k] == '-')) { exp_neg = tok[k] == '-'; k++; }
    else if (!(k < n && is_digit(tok[k]))) return false;
    size_t start = k;
    while (k < n && is_digit(tok[k])) {
      if (exp10 > 2147483647 / 10) return false;
      exp10 = exp10 * 10 + (tok[k] - '0');
      k++;
    }
    if (k == start) return false;
    if (exp_neg) exp10 = -exp10;
  }
  double value = exp10 ? std::ldexp(mant * std::pow(5.0, exp10), exp10) : mant;
  *out = negative ? -value : value;
  return true;
}

ObjData read_obj(const std::string &path, const std::string &mtl_dir) {  // tiny_obj_loader.h:2841-3486
  std::ifstream in(path);
  if (!in) throw ObjError("Cannot open file [" + path + "]");
  ObjData data;
  std::set<std::string> loaded_libs;
  std::map<std::string, int> material_ids;
  bool all_colored = true;
  ObjShape shape;
  std::vector<std::vector<ObjCorner>> faces;
  int material = -1;

  auto flush_faces = [&]() {  // exportGroupsToShape, tiny_obj_loader.h:1607-1754
    for (const auto &f : faces) {
      if (f.size() < 3) continue;
      if (f.size() == 3) {
        shape.corners.insert(shape.corners.end(), f.begin(), f.end());
      } else if (f.size() == 4) {
        bool ok = true;
        for (const ObjCorner &c : f) ok = ok && c.v >= 0 && (size_t)(3 * c.v + 2) < data.vertices.size();
        if (!ok) continue;
        const double *p[4];
        for (int k = 0; k < 4; k++) p[k] = &data.vertices[3 * (size_t)f[k].v];
        double e02[3], e13[3];
        for (int k = 0; k < 3; k++) { e02[k] = p[2][k] - p[0][k]; e13[k] = p[3][k] - p[1][k]; }
        double s02 = e02[0] * e02[0] + e02[1] * e02[1] + e02[2] * e02[2];
        double s13 = e13[0] * e13[0] + e13[1] * e13[1] + e13[2] * e13[2];
        static const int near02[6] = {0, 1, 2, 0, 2, 3}, near13[6] = {0, 1, 3, 1, 2, 3};
        const int *order = s02 < s13 ? near02 : near13;
        for (int k = 0; k < 6; k++) shape.corners.push_back(f[order[k]]);
      } else {
        for (const ObjCorner &c : f)
          if (c.v < 0 || (size_t)(3 * c.v + 2) >= data.vertices.size()) throw ObjError(path + ": vertex index out of range in a polygon");
        ear_clip(f, data.vertices, &shape.corners);
      }
    }
    faces.clear();
  };
  auto end_shape = [&](const std::string &name) {
    flush_faces();
    if (!shape.corners.empty()) data.shapes.push_back(shape);
    shape = ObjShape();
    shape.name = name;
  };

  std::string raw;
  while (std::getline(in, raw)) {
    std::string line = trim_line(raw, false);
    if (line.empty() || line[0] == '#') continue;
    const char c0 = line[0], c1 = line.size() > 1 ? line[1] : '\0', c2 = line.size() > 2 ? line[2] : '\0';
    if (c0 == 'v' && is_blank(c1)) {
      std::vector<std::string> tok = split_blank(line.substr(2));
      double xyz[3];
      reals(tok, 0, 3, xyz);
      data.vertices.insert(data.vertices.end(), xyz, xyz + 3);
      double rgb[3];
      bool found = true;
      for (int k = 0; k < 3 && found; k++) found = 3 + (size_t)k < tok.size() && obj_parse_double(tok[3 + k], &rgb[k]);
      all_colored = all_colored && found;
      if (all_colored) data.colors.insert(data.colors.end(), rgb, rgb + 3);
    } else if (c0 == 'v' && c1 == 'n' && is_blank(c2)) {
      double n[3];
      reals(split_blank(line.substr(3)), 0, 3, n);
      data.normals.insert(data.normals.end(), n, n + 3);
    } else if (c0 == 'v' && c1 == 't' && is_blank(c2)) {
      double t[2];
      reals(split_blank(line.substr(3)), 0, 2, t);
      data.texcoords.insert(data.texcoords.end(), t, t + 2);
    } else if (c0 == 'f' && is_blank(c1)) {
      int nv = (int)(data.vertices.size() / 3), nvn = (int)(data.normals.size() / 3), nvt = (int)(data.texcoords.size() / 2);
      std::vector<ObjCorner> f;
      for (const std::string &t : split_blank(line.substr(2))) f.push_back(parse_corner(t, nv, nvn, nvt));
      faces.push_back(std::move(f));
    } else if (line.compare(0, 6, "usemtl") == 0) {
      std::vector<std::string> tok = split_blank(line.substr(6));
      auto it = material_ids.find(tok.empty() ? "" : tok[0]);
      int id = it == material_ids.end() ? -1 : it->second;
      if (id != material) {
        flush_faces();
        material = id;
      }
    } else if (line.compare(0, 6, "mtllib") == 0 && line.size() > 6 && is_blank(line[6])) {
      std::stringstream names(line.substr(7));
      std::string name;
      while (std::getline(names, name, ' ')) {
        if (name.empty()) continue;
        if (loaded_libs.count(name)) break;
        std::vector<ObjMaterial> mats;
        if (read_mtl(mtl_dir + "/" + name, &mats)) {
          for (const ObjMaterial &m : mats) {
            material_ids.emplace(m.name, (int)data.materials.size());
            data.materials.push_back(m);
          }
          loaded_libs.insert(name);
          break;
        }
      }
    } else if (c0 == 'g' && is_blank(c1)) {
      std::vector<std::string> tok = split_blank(line.substr(2));
      std::string name;
      for (size_t k = 0; k < tok.size(); k++) name += (k ? " " : "") + tok[k];
      end_shape(name);
    } else if (c0 == 'o' && is_blank(c1)) {
      end_shape(line.substr(2));
    }
  }
  end_shape("");
  if (!all_colored) data.colors.clear();
  return data;
}

ObjMesh obj_shape_to_mesh(const ObjData &data, const ObjShape &shape) {
  ObjMesh m;
  std::map<std::tuple<int, int, int>, int> index;
  for (const ObjCorner &c : shape.corners) {
    auto key = std::make_tuple(c.v, c.vt, c.vn);
    auto it = index.find(key);
    int k;
    if (it == index.end()) {
      k = (int)index.size();
      index.emplace(key, k);
      if (c.v < 0 || (size_t)(3 * c.v + 2) >= data.vertices.size()) throw ObjError("Error while parsing OBJ file: vertex index out of range");
      m.vertices.insert(m.vertices.end(), &data.vertices[3 * (size_t)c.v], &data.vertices[3 * (size_t)c.v] + 3);
      if (c.vn != -1) {
        if (c.vn < 0 || (size_t)(3 * c.vn + 2) >= data.normals.size()) throw ObjError("Error while parsing OBJ file: normal index out of range");
        const double *nn = &data.normals[3 * (size_t)c.vn];
        double inv = 1.0 / std::sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);  // glm::normalize
        for (int a = 0; a < 3; a++) m.normals.push_back(nn[a] * inv);
      }
      if (c.vt != -1) {
        if (c.vt < 0 || (size_t)(2 * c.vt + 1) >= data.texcoords.size()) throw ObjError("Error while parsing OBJ file: texture coordinate index out of range");
        m.uvs.push_back(data.texcoords[2 * (size_t)c.vt]);
        m.uvs.push_back(data.texcoords[2 * (size_t)c.vt + 1]);
      }
      if (!data.colors.empty()) m.colors.insert(m.colors.end(), &data.colors[3 * (size_t)c.v], &data.colors[3 * (size_t)c.v] + 3);
    } else {
      k = it->second;
    }
    m.faces.push_back(k);
  }
  // Trimesh::doubleCheck, trimesh.cpp:50-60
  size_t nv = m.vertices.size() / 3;
  if (!m.colors.empty() && m.colors.size() / 3 != nv) throw ObjError("Error while parsing OBJ file: Bad Trimesh: Wrong number of vertex colors.");
  if (!m.uvs.empty() && m.uvs.size() / 2 != nv) throw ObjError("Error while parsing OBJ file: Bad Trimesh: Wrong number of UV coordinates.");
  if (!m.normals.empty() && m.normals.size() / 3 != nv) throw ObjError("Error while parsing OBJ file: Bad Trimesh: Wrong number of normals.");
  return m;
}

}  // namespace host
