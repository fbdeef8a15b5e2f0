#include "RayTracer.h"

#include <cstdio>
#include <cstring>
#include <fstream>

namespace host {

RayTracer::RayTracer(int device) : device_(device) {
  mipt_default_params(&params_, 0, 0);
  if (mipt_create(&ctx_, device) != MIPT_OK) {
    alert_ = mipt_last_error(nullptr);
    ctx_ = nullptr;
  }
}

RayTracer::~RayTracer() {
  mipt_dist_destroy(dist_);
  mipt_destroy(ctx_);
}

bool RayTracer::joinRanks(int rank, int n_ranks, const unsigned char id[MIPT_DIST_ID_BYTES], int mode) {
  mipt_dist_destroy(dist_);
  dist_ = nullptr;
  dist_mode_ = mode;
  if (mipt_dist_init(&dist_, rank, n_ranks, id, device_) != MIPT_OK) {
    alert_ = mipt_dist_last_error();
    return false;
  }
  return true;
}

bool RayTracer::upload() {
  if (!ctx_) return false;
  mipt_scene_desc d = scene_.describe();
  if (mipt_upload_scene(ctx_, &d) != MIPT_OK) {
    alert_ = mipt_last_error(ctx_);
    loaded_ = false;
    return false;
  }
  loaded_ = true;
  return true;
}

bool RayTracer::loadScene(const char *fn) {
  try {
    scene_ = load_scene_file(fn);
  } catch (const std::exception &e) {
    alert_ = e.what();
    loaded_ = false;
    return false;
  }
  return upload();
}

bool RayTracer::loadScene(SceneData &&scene) {
  scene_ = std::move(scene);
  return upload();
}

bool RayTracer::loadCubemap(const char *dir) {
  try {
    load_cubemap_ppm(dir, &scene_);
  } catch (const std::exception &e) {
    alert_ = e.what();
    return false;
  }
  return loaded_ ? upload() : true;
}

void RayTracer::setShard(int index, int count, int band_rows) {
  params_.shard_index = index;
  params_.shard_count = count;
  params_.band_rows = band_rows;
}

void RayTracer::traceSetup(int w, int h) {
  size_t n = (size_t)w * h * 3;
  buffer_.assign(n, 0);  // std::fill(buffer, 0), RayTracer.cpp:427
  buffer_f32_.assign(n, 0.f);
  buffer_width_ = w;
  buffer_height_ = h;
  params_.width = w;
  params_.height = h;
  if (params_.sample_end > params_.spp || params_.sample_end <= params_.sample_begin) {
    params_.sample_begin = 0;
    params_.sample_end = params_.spp;
  }
}

void RayTracer::traceImage(int w, int h) {
  traceSetup(w, h);
  if (!loaded_) return;
  params_.sample_begin = 0;
  params_.sample_end = params_.spp;
  if (dist_) {  // the split lives inside the call, like the reference's threads (RayTracer.cpp:482-489)
    const bool root = mipt_dist_rank(dist_) == 0;
    if (mipt_dist_render(dist_, ctx_, &params_, dist_mode_, root ? buffer_f32_.data() : nullptr, root ? buffer_.data() : nullptr) != MIPT_OK)
      alert_ = mipt_dist_last_error();
    return;
  }
  if (mipt_render(ctx_, &params_, buffer_f32_.data(), buffer_.data()) != MIPT_OK) alert_ = mipt_last_error(ctx_);
}

std::array<double, 3> RayTracer::tracePixel(int i, int j) {
  std::array<double, 3> color{{0, 0, 0}};
  if (!loaded_ || buffer_.empty()) return color;  // RayTracer.cpp:63-64
  if (mipt_trace_pixel(ctx_, &params_, i, j, color.data()) != MIPT_OK) {
    alert_ = mipt_last_error(ctx_);
    return std::array<double, 3>{{0, 0, 0}};
  }
  unsigned char *pixel = buffer_.data() + ((size_t)i + (size_t)j * buffer_width_) * 3;
  for (int k = 0; k < 3; k++) pixel[k] = (unsigned char)(int)(255.0 * color[k]);  // RayTracer.cpp:97-99
  return color;
}

void RayTracer::getBuffer(unsigned char *&buf, int &w, int &h) {
  buf = buffer_.data();
  w = buffer_width_;
  h = buffer_height_;
}

std::array<double, 3> RayTracer::getPixel(int i, int j) const {
  const unsigned char *p = buffer_.data() + ((size_t)i + (size_t)j * buffer_width_) * 3;
  return {{p[0] / 255.0, p[1] / 255.0, p[2] / 255.0}};
}

// ------------------------------------------------------------------ image writers
bool write_ppm(const std::string &path, int w, int h, const unsigned char *rgb) {
  std::ofstream out(path, std::ios::binary);
  if (!out) return false;
  out << "P6\n" << w << " " << h << "\n255\n";
  for (int y = h - 1; y >= 0; y--) out.write(reinterpret_cast<const char *>(rgb + (size_t)3 * w * y), (std::streamsize)3 * w);
  return (bool)out;
}

namespace {
uint32_t crc_table[256];
bool crc_ready = false;
uint32_t crc32(uint32_t crc, const unsigned char *p, size_t n) {
  if (!crc_ready) {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      crc_table[i] = c;
    }
    crc_ready = true;
  }
  crc = ~crc;
  for (size_t i = 0; i < n; i++) crc = crc_table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
  return ~crc;
}
void be32(std::vector<unsigned char> &v, uint32_t x) {
  v.push_back((unsigned char)(x >> 24)); v.push_back((unsigned char)(x >> 16)); v.push_back((unsigned char)(x >> 8)); v.push_back((unsigned char)x);
}
void chunk(std::vector<unsigned char> &png, const char *tag, const std::vector<unsigned char> &data) {
  be32(png, (uint32_t)data.size());
  size_t start = png.size();
  png.insert(png.end(), tag, tag + 4);
  png.insert(png.end(), data.begin(), data.end());
  be32(png, crc32(0, png.data() + start, png.size() - start));
}
}  // namespace

// PNG with stored (uncompressed) deflate blocks: no zlib needed.
bool write_png(const std::string &path, int w, int h, const unsigned char *rgb) {
  std::vector<unsigned char> raw;
  raw.reserve((size_t)h * (3 * w + 1));
  for (int y = h - 1; y >= 0; y--) {  // top row first in the file
    raw.push_back(0);
    raw.insert(raw.end(), rgb + (size_t)3 * w * y, rgb + (size_t)3 * w * (y + 1));
  }
  std::vector<unsigned char> z;
  z.push_back(0x78); z.push_back(0x01);
  uint32_t a = 1, b = 0;
  for (unsigned char c : raw) { a = (a + c) % 65521; b = (b + a) % 65521; }
  size_t pos = 0;
  while (pos < raw.size() || raw.empty()) {
    size_t n = std::min<size_t>(65535, raw.size() - pos);
    bool last = pos + n >= raw.size();
    z.push_back(last ? 1 : 0);
    z.push_back((unsigned char)(n & 0xFF)); z.push_back((unsigned char)(n >> 8));
    z.push_back((unsigned char)(~n & 0xFF)); z.push_back((unsigned char)((~n >> 8) & 0xFF));
    z.insert(z.end(), raw.begin() + (long)pos, raw.begin() + (long)(pos + n));
    pos += n;
    if (last) break;
  }
  be32(z, (b << 16) | a);
  std::vector<unsigned char> png = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  std::vector<unsigned char> ihdr;
  be32(ihdr, (uint32_t)w); be32(ihdr, (uint32_t)h);
  ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
  chunk(png, "IHDR", ihdr);
  chunk(png, "IDAT", z);
  chunk(png, "IEND", {});
  std::ofstream out(path, std::ios::binary);
  if (!out) return false;
  out.write(reinterpret_cast<const char *>(png.data()), (std::streamsize)png.size());
  return (bool)out;
}

bool write_image(const std::string &path, int w, int h, const unsigned char *rgb) {  // fileio/images.cc:56-64
  size_t dot = path.find_last_of('.');
  std::string ext = dot == std::string::npos ? "" : path.substr(dot);
  if (ext == ".png") return write_png(path, w, h, rgb);
  return write_ppm(path, w, h, rgb);
}

}  // namespace host
