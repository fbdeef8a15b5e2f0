// RayTracer.h — the host-side render driver with the reference's interface
// (ray/src/RayTracer.h:32-56): loadScene / traceSetup / traceImage / tracePixel /
// checkRender / waitRender / getBuffer.  The work behind traceImage and tracePixel is
// done by libmipt through the C ABI of include/mipt.h; this class owns the scene
// description, the output buffers and the render parameters.
#ifndef MIPT_HOST_RAYTRACER_H_
#define MIPT_HOST_RAYTRACER_H_

#include <array>
#include <memory>
#include <string>
#include <vector>

#include "../include/mipt.h"
#include "../include/mipt_dist.h"
#include "scene_loader.h"

namespace host {

class RayTracer {
 public:
  explicit RayTracer(int device = 0);
  ~RayTracer();
  RayTracer(const RayTracer &) = delete;
  RayTracer &operator=(const RayTracer &) = delete;

  // RayTracer.cpp:333-415.  Returns false and keeps the message in lastAlert() on failure
  // (the reference reports through traceUI->alert).
  bool loadScene(const char *fn);
  bool loadScene(SceneData &&scene);
  bool loadCubemap(const char *dir);  // the `-c` option (TraceUI::smartLoadCubemap)
  bool sceneLoaded() const { return loaded_; }
  double aspectRatio() const { return loaded_ ? scene_.camera.getAspectRatio() : 1; }  // RayTracer.cpp:327-330

  void traceSetup(int w, int h);             // RayTracer.cpp:417-442
  void traceImage(int w, int h);             // RayTracer.cpp:462-491 (returns after the image is complete)
  std::array<double, 3> tracePixel(int i, int j);  // RayTracer.cpp:59-101
  bool checkRender() { return true; }        // RayTracer.cpp:503-512
  void waitRender() {}                       // RayTracer.cpp:514-526
  void getBuffer(unsigned char *&buf, int &w, int &h);  // RayTracer.cpp:319-324
  std::array<double, 3> getPixel(int i, int j) const;   // RayTracer.cpp:529-534
  const std::vector<float> &floatBuffer() const { return buffer_f32_; }

  // One process per GPU: this instance renders the bands of rows shard `index` of `count` owns
  // (band b of `band_rows` rows belongs to shard b % count), as the reference's threads each take a
  // range of columns (RayTracer.cpp:482-487); the rows of other shards stay as traceSetup left them.
  void setShard(int index, int count, int band_rows = 8);

  // One process per GPU, joined over RCCL: this instance becomes rank `rank` of `n_ranks` (collective: every rank
  // calls it with the id rank 0 made with mipt_dist_new_id), and traceImage then renders this rank's share of the
  // frame — of the sample range (MIPT_SHARD_SAMPLES) or of the bands of rows (MIPT_SHARD_TILES) — and brings the
  // sums to rank 0 with one collective; the finished frame is in rank 0's buffer, the other ranks' stay zero.
  bool joinRanks(int rank, int n_ranks, const unsigned char id[MIPT_DIST_ID_BYTES], int mode = MIPT_SHARD_SAMPLES);
  int rank() const { return dist_ ? mipt_dist_rank(dist_) : 0; }

  // The reference's literals (N = 100, q = 0.1, 10 shadow rays, t < 1000) as settings.
  mipt_params &settings() { return params_; }
  const std::string &lastAlert() const { return alert_; }
  bool ok() const { return ctx_ != nullptr; }

 private:
  bool upload();
  mipt_ctx *ctx_ = nullptr;
  int device_ = 0;
  mipt_dist *dist_ = nullptr;
  int dist_mode_ = MIPT_SHARD_SAMPLES;
  SceneData scene_;
  bool loaded_ = false;
  std::vector<unsigned char> buffer_;  // w*h*3, index (i + j*w)*3, j = 0 is the bottom row
  std::vector<float> buffer_f32_;
  int buffer_width_ = 0, buffer_height_ = 0;
  mipt_params params_;
  std::string alert_;
};

// Image writers (rows flipped on the way out, like fileio/pngimage.cpp:272-273).
bool write_ppm(const std::string &path, int w, int h, const unsigned char *bottom_up_rgb);
bool write_png(const std::string &path, int w, int h, const unsigned char *bottom_up_rgb);
bool write_image(const std::string &path, int w, int h, const unsigned char *bottom_up_rgb);

}  // namespace host
#endif
