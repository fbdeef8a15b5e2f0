// ray — command-line driver with the reference's calling convention
// (ui/CommandLineUI.cpp:23-107):  ray [options] scene.json output.{png,ppm}
//   -w N     image width (default 512, TraceUI.h:96); height = width / aspectRatio
//   -H N     image height override (the reference can only emit width/aspect)
//   -j FILE  JSON settings: "size", plus this build's "spp", "rr_q", "area_samples",
//            "max_depth", "seed" (the reference hard-codes them)
//   -c DIR   cubemap directory (six P6 files posx/negx/posy/negy/posz/negz.ppm)
//   -r N     accepted and ignored: the path tracer never reads the recursion depth
//   -s N     samples per pixel (reference literal: 100)
//   -d N     HIP device
// Several GPUs, one process per GPU (any launcher; pin each with -d or HIP_VISIBLE_DEVICES):
//   --shard I/N    render only the bands of rows that shard I of N owns (band b of --band-rows rows
//                  belongs to shard b % N, the split mipt_params.shard_* describes); the other rows
//                  of the output stay black, as traceSetup leaves them (RayTracer.cpp:427)
//   --band-rows R  rows per band (default 8)
//   --combine OUT PART...   no rendering: add the shards' .ppm outputs into OUT.  The shards own
//                  disjoint rows, so the sum is byte for byte the frame one GPU renders
// Several GPUs of one node joined over RCCL (libmiptdist.so, include/mipt_dist.h): the split lives inside
// traceImage, as the reference's split over threads does (RayTracer.cpp:482-489)
//   --ranks N      start N ranks of this program, rank r on device -d + r; they split the frame's sample range
//                  (--sharding samples, default: one sum-reduce) or its bands of rows (--sharding tiles: one
//                  gather) and rank 0 writes the frame
//   --rank r --id-file F   (what --ranks passes to the ranks; also for launchers of their own) be rank r;
//                  rank 0 writes the group's id to F, the others read it there
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "RayTracer.h"

static void usage(const char *prog) {
  std::fprintf(stderr, "usage: %s [-w width] [-H height] [-s spp] [-j settings.json] [-c cubemap_dir] [-d device] [-r depth] [--shard I/N] [--band-rows R] [--ranks N [--sharding samples|tiles]] scene.json output.png\n"
                       "       %s --combine output.png part0.ppm part1.ppm ...\n", prog, prog);
}

// --combine: the shards' frames hold disjoint bands of rows and zeros elsewhere
static int combine(int argc, char **argv) {
  if (argc < 4) { usage(argv[0]); return 1; }
  int w = 0, h = 0;
  std::vector<uint8_t> sum;
  try {
    for (int k = 3; k < argc; k++) {
      int pw = 0, ph = 0;
      std::vector<uint8_t> part;
      host::read_ppm(argv[k], &pw, &ph, &part);  // rows bottom-up
      if (k == 3) { w = pw; h = ph; sum.assign(part.size(), 0); }
      if (pw != w || ph != h) { std::fprintf(stderr, "%s: %dx%d, expected %dx%d\n", argv[k], pw, ph, w, h); return 1; }
      for (size_t b = 0; b < sum.size(); b++) {
        if (part[b] && sum[b]) { std::fprintf(stderr, "%s overlaps an earlier part: the shards must own disjoint rows\n", argv[k]); return 1; }
        sum[b] = (uint8_t)(sum[b] + part[b]);
      }
    }
  } catch (const std::exception &e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  if (!host::write_image(argv[2], w, h, sum.data())) { std::fprintf(stderr, "cannot write %s\n", argv[2]); return 1; }
  return 0;
}

// --ranks N without --rank: this process only starts the N ranks (itself again, with --rank r and a file through
// which rank 0 hands the others the group's id) and waits for them.  It has not touched a GPU when it does.
static int launch_ranks(int argc, char **argv, int n_ranks) {
  char tmpl[] = "/tmp/mipt_ranks_XXXXXX";
  if (!mkdtemp(tmpl)) { std::perror("mkdtemp"); return 1; }
  const std::string id_file = std::string(tmpl) + "/id";
  std::vector<pid_t> kids;
  for (int r = 0; r < n_ranks; r++) {
    pid_t pid = fork();
    if (pid < 0) { std::perror("fork"); return 1; }
    if (pid == 0) {
      std::vector<std::string> args(argv, argv + argc);
      const std::string rs = std::to_string(r);
      args.insert(args.begin() + 1, {"--rank", rs, "--id-file", id_file});
      std::vector<char *> av;
      for (auto &a : args) av.push_back(const_cast<char *>(a.c_str()));
      av.push_back(nullptr);
      execv("/proc/self/exe", av.data());
      std::perror("execv");
      _exit(127);
    }
    kids.push_back(pid);
  }
  int worst = 0;
  for (pid_t pid : kids) {
    int st = 0;
    if (waitpid(pid, &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) worst = worst ? worst : (WIFEXITED(st) ? WEXITSTATUS(st) : 1);
  }
  (void)std::remove(id_file.c_str());
  (void)rmdir(tmpl);
  return worst;
}

// The group's id travels through a file: rank 0 writes it (to a temporary name, then renames), the others wait for it.
static bool share_id(int rank, const std::string &path, unsigned char id[MIPT_DIST_ID_BYTES]) {
  if (rank == 0) {
    if (mipt_dist_new_id(id) != MIPT_OK) { std::fprintf(stderr, "%s\n", mipt_dist_last_error()); return false; }
    const std::string tmp = path + ".tmp";
    std::ofstream out(tmp, std::ios::binary);
    out.write(reinterpret_cast<const char *>(id), MIPT_DIST_ID_BYTES);
    out.close();
    return out.good() && std::rename(tmp.c_str(), path.c_str()) == 0;
  }
  for (int tries = 0; tries < 1200; tries++) {  // two minutes
    std::ifstream in(path, std::ios::binary);
    if (in && in.read(reinterpret_cast<char *>(id), MIPT_DIST_ID_BYTES)) return true;
    usleep(100000);
  }
  std::fprintf(stderr, "rank %d: no id in %s after two minutes\n", rank, path.c_str());
  return false;
}

int main(int argc, char **argv) {
  if (argc > 1 && std::string(argv[1]) == "--combine") return combine(argc, argv);
  int width = 512, height = 0, spp = -1, device = 0;
  int shard_index = 0, shard_count = 1, band_rows = 8;
  int n_ranks = 0, rank = -1, sharding = MIPT_SHARD_SAMPLES;
  std::string id_file;
  std::string settings, cubemap;
  int i = 1;
  for (; i < argc && argv[i][0] == '-' && argv[i][1] != '\0'; i++) {
    std::string opt = argv[i];
    auto need = [&](const char *name) -> const char * {
      if (i + 1 >= argc) { std::fprintf(stderr, "option %s needs a value\n", name); std::exit(1); }
      return argv[++i];
    };
    if (opt == "-w") width = std::atoi(need("-w"));
    else if (opt == "-H") height = std::atoi(need("-H"));
    else if (opt == "-s") spp = std::atoi(need("-s"));
    else if (opt == "-j") settings = need("-j");
    else if (opt == "-c") cubemap = need("-c");
    else if (opt == "-d") device = std::atoi(need("-d"));
    else if (opt == "-r") (void)need("-r");
    else if (opt == "--shard") {
      if (std::sscanf(need("--shard"), "%d/%d", &shard_index, &shard_count) != 2 || shard_count < 1 || shard_index < 0 || shard_index >= shard_count) {
        std::fprintf(stderr, "--shard wants I/N with 0 <= I < N\n");
        return 1;
      }
    } else if (opt == "--band-rows") band_rows = std::atoi(need("--band-rows"));
    else if (opt == "--ranks") n_ranks = std::atoi(need("--ranks"));
    else if (opt == "--rank") rank = std::atoi(need("--rank"));
    else if (opt == "--id-file") id_file = need("--id-file");
    else if (opt == "--sharding") {
      const std::string m = need("--sharding");
      if (m == "samples") sharding = MIPT_SHARD_SAMPLES;
      else if (m == "tiles") sharding = MIPT_SHARD_TILES;
      else { std::fprintf(stderr, "--sharding wants samples or tiles\n"); return 1; }
    }
    else if (opt == "-h") { usage(argv[0]); return 0; }
    else { std::fprintf(stderr, "Invalid argument: '%s'.\n", argv[i]); usage(argv[0]); return 1; }
  }
  if (argc - i != 2) { std::fprintf(stderr, "Error: exactly one scene file and one output file are needed\n"); usage(argv[0]); return 1; }
  const char *rayName = argv[i];
  const char *imgName = argv[i + 1];
  if (n_ranks < 0 || (n_ranks > 0 && rank >= n_ranks) || (rank >= 0 && (n_ranks < 1 || id_file.empty()))) {
    std::fprintf(stderr, "--ranks N [--rank r --id-file F]: 0 <= r < N\n");
    return 1;
  }
  if (n_ranks > 0 && rank < 0) return launch_ranks(argc, argv, n_ranks);  // before anything here has touched a GPU
  if (rank >= 0) device += rank;  // one GPU per rank: -d names rank 0's

  host::RayTracer rt(device);
  if (!rt.ok()) { std::fprintf(stderr, "%s\n", rt.lastAlert().c_str()); return 2; }
  if (rank >= 0) {
    unsigned char id[MIPT_DIST_ID_BYTES];
    if (n_ranks > 1 && !share_id(rank, id_file, id)) return 2;
    if (!rt.joinRanks(rank, n_ranks, id, sharding)) { std::fprintf(stderr, "rank %d: %s\n", rank, rt.lastAlert().c_str()); return 2; }
  }
  mipt_params &p = rt.settings();
  if (!settings.empty()) {
    try {
      std::ifstream in(settings);
      std::stringstream ss;
      ss << in.rdbuf();
      host::Json j = host::Json::parse(ss.str());
      if (j.has("size")) width = (int)j.at("size").number();           // TraceUI.cc:44
      if (j.has("spp")) p.spp = (int)j.at("spp").number();
      if (j.has("rr_q")) p.rr_q = j.at("rr_q").number();
      if (j.has("area_samples")) p.area_samples = (int)j.at("area_samples").number();
      if (j.has("max_depth")) p.max_depth = (int)j.at("max_depth").number();
      if (j.has("seed")) p.seed = (uint64_t)j.at("seed").number();
    } catch (const std::exception &e) {
      std::fprintf(stderr, "settings: %s\n", e.what());
      return 1;
    }
  }
  if (spp > 0) p.spp = spp;
  p.sample_begin = 0;
  p.sample_end = p.spp;
  rt.setShard(shard_index, shard_count, band_rows);

  if (!rt.loadScene(rayName)) {
    std::fprintf(stderr, "Unable to load ray file '%s': %s\n", rayName, rt.lastAlert().c_str());
    return 1;
  }
  if (!cubemap.empty() && !rt.loadCubemap(cubemap.c_str())) {
    std::fprintf(stderr, "%s\n", rt.lastAlert().c_str());
    return 1;
  }
  if (height <= 0) height = (int)(width / rt.aspectRatio() + 0.5);  // CommandLineUI.cpp:73-74
  auto t0 = std::chrono::steady_clock::now();
  rt.traceSetup(width, height);
  rt.traceImage(width, height);
  rt.waitRender();
  double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (!rt.lastAlert().empty()) { std::fprintf(stderr, "render failed: %s\n", rt.lastAlert().c_str()); return 2; }
  unsigned char *buf;
  int w, h;
  rt.getBuffer(buf, w, h);
  if (rank > 0) return 0;  // the frame is rank 0's
  if (!host::write_image(imgName, w, h, buf)) { std::fprintf(stderr, "cannot write %s\n", imgName); return 1; }
  std::fprintf(stderr, "%dx%d, %d spp: %.3f s, %.2f Msamples/s\n", w, h, p.spp, sec, (double)w * h * p.spp / sec / 1e6);
  return 0;
}
