/*
 * mipt.h — C ABI of the MI355X path-tracing library (libmipt.so).
 *
 * The library replaces the reference renderer's Monte-Carlo hot path:
 * RayTracer::tracePath and everything it calls (Scene::intersect, the BVH,
 * the primitive tests, Material::shadeBRDF, the light shadow walks).  The host
 * keeps scene loading, the camera and the RayTracer::traceImage / tracePixel
 * entry points and calls through the functions below.
 *
 * All citations are file:line under the reference tree (ray/src/...).
 * Plain C: POD structs, raw pointers and sizes, int status codes, no
 * exceptions cross this boundary.
 */
#ifndef MIPT_H_
#define MIPT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIPT_VERSION 1

/* ---- status codes --------------------------------------------------------- */
#define MIPT_OK 0
#define MIPT_ERR_INVALID (-1)  /* bad argument / bad scene description   */
#define MIPT_ERR_NOSCENE (-2)  /* render call before mipt_upload_scene   */
#define MIPT_ERR_DEVICE (-3)   /* HIP runtime error (message has detail) */
#define MIPT_ERR_NOMEM (-4)    /* host or device allocation failed       */
#define MIPT_ERR_LIMIT (-5)    /* size beyond a documented limit         */

/* ---- scene description (what the host's JSON loader produces) -------------- */

/* Object kinds: SceneObjects/{Sphere,Box,Square,trimesh}.cpp. */
#define MIPT_OBJ_SPHERE 0  /* unit sphere at origin      Sphere.cpp:9-40   */
#define MIPT_OBJ_BOX 1     /* unit cube  +-0.5           Box.cpp:11-74     */
#define MIPT_OBJ_SQUARE 2  /* unit XY square at z=0      Square.cpp:8-43   */
#define MIPT_OBJ_TRIMESH 3 /* triangle mesh              trimesh.cpp:70-200 */

/* Light kinds: scene/light.{h,cpp}. */
#define MIPT_LIGHT_DIRECTIONAL 0 /* light.cpp:10-45   */
#define MIPT_LIGHT_POINT 1       /* light.cpp:47-85   */
#define MIPT_LIGHT_RECT_AREA 2   /* light.cpp:88-147  */

/* Material as read by Material::shadeBRDF (scene/material.cpp:88-170).
 * roughness / metallic / index are the RAW scalars written in the scene file;
 * the luminance re-weighting of MaterialParameter::intensityValue
 * (material.cpp:232-241) is applied inside the library, as the reference does
 * on every access.  Defaults of a default-constructed Material
 * (material.h:152-158): colours 0, roughness 1, metallic 0, index 1. */
typedef struct mipt_material {
  double kd[3]; /* diffuse       */
  double ka[3]; /* ambient       */
  double kt[3]; /* transmissive (only read by the shadow walk, light.cpp:78) */
  double roughness;
  double metallic;
  double index;
  /* Texture-mapped parameters ({"mapped": file}, JsonParser.cpp:86-111;
   * MaterialParameter::value, material.cpp:224-230): 0 = the constant above,
   * k > 0 = mipt_scene_desc.textures[k - 1] looked up at the hit's UV
   * coordinates (TextureMap::getMappedValue, material.cpp:186-222). */
  int32_t kd_texture;
  int32_t ka_texture;
} mipt_material;

/* One Geometry added to the Scene (scene/scene.cpp:103-107), in file order.
 * xform is the local->world matrix in glm memory order (column-major, 16
 * doubles, xform[4*c + r]); MatrixTransform (scene/scene.h:65-97) derives the
 * inverse and the normal matrix from it and so does the library. */
typedef struct mipt_object {
  int32_t type;     /* MIPT_OBJ_*                                  */
  int32_t material; /* index into mipt_scene_desc.materials        */
  int32_t mesh;     /* index into mipt_scene_desc.meshes, or -1    */
  int32_t reserved;
  double xform[16];
} mipt_object;

/* One Trimesh (SceneObjects/trimesh.h:18-97) in its local space.  Faces are
 * already triangulated (quads split as JsonParser.cpp:253-257 does);
 * degenerate faces are dropped by the library exactly as Trimesh::addFace does
 * (trimesh.cpp:32-46, trimesh.h:139-141).  normals is NULL or one vector per
 * vertex (interpolated as trimesh.cpp:154-163); uvs is NULL or one (u, v) per
 * vertex (the "UV" key, JsonParser.cpp:271-279; interpolated as
 * trimesh.cpp:165-172).  Hits on a mesh without uvs, and on spheres, have UV
 * (0, 0) (ray.h:71); squares and boxes set their own (Square.cpp:41,
 * Box.cpp:62-71).  colors is NULL or one RGB per vertex (OBJ files only,
 * JsonParser.cpp:609-612): on a mesh WITHOUT uvs the hit's diffuse colour is
 * the interpolated vertex colour instead of the material's (trimesh.cpp:170-189);
 * with uvs the colours are never read. */
typedef struct mipt_mesh {
  const double *vertices; /* 3 * n_vertices */
  int64_t n_vertices;
  const int32_t *faces; /* 3 * n_faces   */
  int64_t n_faces;
  const double *normals; /* NULL or 3 * n_vertices */
  const double *uvs;     /* NULL or 2 * n_vertices */
  const double *colors;  /* NULL or 3 * n_vertices */
} mipt_mesh;

/* A light, fields as the JSON parser hands them to the constructors
 * (parser/JsonParser.cpp:133-171). */
typedef struct mipt_light {
  int32_t type; /* MIPT_LIGHT_* */
  int32_t reserved;
  double color[3];
  double position[3];  /* point: position; area: corner                      */
  double direction[3]; /* directional: "direction" as written (unnormalised) */
  double u[3];         /* area: uVec (used unnormalised, light.cpp:109)      */
  double v[3];         /* area: vVec                                          */
  double uL, vL;       /* area: uLength, vLength                              */
  float atten[3];      /* constant, linear, quadratic — stored as float in the
                          reference (light.h:98-100,158-160)                  */
  float pad;
} mipt_light;

/* Camera after Camera::update() (scene/camera.cpp:84-88): the host keeps the
 * camera class and passes the four vectors rayThrough (camera.cpp:20-29) uses. */
typedef struct mipt_camera {
  double eye[3];
  double look[3];
  double u[3];
  double v[3];
} mipt_camera;

/* One image (cubemap face or material texture): 8-bit RGB, row 0 = bottom row,
 * as readImage returns it (fileio/pngimage.cpp:210-221).  Cubemap face order is
 * CubeMap::tMap's (scene/cubeMap.cpp:7-64): +x, -x, +y, -y, +z, -z. */
typedef struct mipt_image {
  const uint8_t *rgb; /* 3 * width * height */
  int32_t width;
  int32_t height;
} mipt_image;

typedef struct mipt_scene_desc {
  mipt_camera camera;
  double ambient[3]; /* Scene::ambient() */
  const mipt_material *materials;
  int32_t n_materials;
  const mipt_object *objects;
  int32_t n_objects;
  const mipt_mesh *meshes;
  int32_t n_meshes;
  const mipt_light *lights;
  int32_t n_lights;
  const mipt_image *cubemap; /* NULL, or 6 faces (TraceUI::getCubeMap()) */
  const mipt_image *textures; /* material textures (Scene::getTexture, scene.cpp:140-147) */
  int32_t n_textures;
} mipt_scene_desc;

/* ---- render parameters ----------------------------------------------------- */

/* flags */
#define MIPT_FLAG_FIXED_AREA_PATTERN 1u /* area lights use a fixed stratified
                                           pattern instead of random points  */
#define MIPT_FLAG_COUNTERS 2u           /* collect traversal / shading counters
                                           (slower instrumented kernels)     */
#define MIPT_FLAG_PROFILE 4u            /* bracket every kernel launch with HIP
                                           events on its stream and accumulate
                                           the durations (mipt_get_kernel_times) */

#define MIPT_FLAG_CULL_NEGLIGIBLE 8u    /* off by default: do not trace the shadow
                                           rays of a hit whose whole direct-diffuse
                                           term is bounded below 2^-52 (throughput *
                                           kd / pi * (1 - metallic) * sum of light
                                           colours).  The reference traces them: a
                                           "metallic 1" material reads back as
                                           0.9999999999999999 (material.cpp:232-241)
                                           and keeps a diffuse weight of 1.1e-16.
                                           Changes a sample by < 1e-15            */

/* The literals of the reference are parameters here and default to the
 * reference's values (mipt_default_params):
 *   spp 100 (RayTracer.cpp:61), rr_q 0.1 (RayTracer.cpp:229,277),
 *   area_samples 10 (light.cpp:118,145), t_max 1000 (scene.cpp:129). */
typedef struct mipt_params {
  int32_t width;
  int32_t height;
  int32_t spp;          /* samples per pixel of the whole job                 */
  int32_t area_samples; /* shadow rays per rectangle area light               */
  double rr_q;          /* Russian-roulette kill probability; 0 disables it   */
  double t_max;         /* hits at t >= t_max are misses                      */
  int32_t max_depth;    /* rays at depth >= max_depth are not traced (the
                           reference recursion is unbounded); 1..255, 64 by
                           default                                            */
  uint32_t flags;
  uint64_t seed;
  /* sharding: this call renders rows y with (y / band_rows) % shard_count ==
   * shard_index, and samples [sample_begin, sample_end) of every such pixel. */
  int32_t shard_index;
  int32_t shard_count;
  int32_t band_rows;
  int32_t sample_begin;
  int32_t sample_end;
  int32_t batch_samples; /* path samples per wavefront batch (also the most
                            rays one kernel launch takes), at most 2^28.  The
                            workspace takes about 480 bytes of device memory
                            per sample of a batch plus 8 per shadow ray of a
                            hit.  0 = as large as the job and 60 % of the
                            card's free memory allow (between 2^20 and 2^28):
                            large batches are faster                          */
} mipt_params;

/* Work counters (summed over everything rendered since the last reset). */
typedef struct mipt_counters {
  uint64_t samples;      /* trace(x,y) calls                                  */
  uint64_t path_rays;    /* camera + bounce + mirror rays traced              */
  uint64_t shadow_rays;  /* shadow rays traced                                */
  uint64_t shaded_hits;  /* tracePath nodes that survived Russian roulette    */
  uint64_t node_visits;  /* BVH nodes fetched, closest-hit + shadow kernels   */
  uint64_t prim_tests;   /* primitive intersection tests, both kernels        */
  uint64_t shadow_node_visits; /* the shadow kernel's share of node_visits    */
  uint64_t shadow_prim_tests;  /* the shadow kernel's share of prim_tests     */
  uint64_t ray_stages;   /* ray-state records read+written by kernel stages   */
  uint64_t nan_samples;  /* samples whose radiance was NaN (counted as 0)     */
} mipt_counters;

/* Device time per kernel, summed over launches made with MIPT_FLAG_PROFILE since
 * the last mipt_reset_counters.  Index: MIPT_K_*. */
#define MIPT_K_GENERATE 0
#define MIPT_K_INTERSECT 1
#define MIPT_K_SHADE 2
#define MIPT_K_SHADOW 3
#define MIPT_K_RESOLVE 4
#define MIPT_K_SHADOW_SUM 5 /* per-hit sum of the shadow rays' results */
#define MIPT_K_TAIL 6       /* the launch that finishes the deep, nearly empty levels of a path tree */
#define MIPT_K_COUNT 7
typedef struct mipt_kernel_times {
  double ms[MIPT_K_COUNT];        /* summed HIP-event durations, milliseconds */
  uint64_t launches[MIPT_K_COUNT];
  uint64_t items[MIPT_K_COUNT];   /* rays / hits / pixels the launches processed */
} mipt_kernel_times;

typedef struct mipt_ctx mipt_ctx;

/* ---- entry points ---------------------------------------------------------- */

/* Library/ABI version (MIPT_VERSION). */
int mipt_version(void);

/* Last error message of a context (or of the calling thread when ctx is NULL). */
const char *mipt_last_error(const mipt_ctx *ctx);

/* Fill *p with the reference's literals for a w x h frame, one shard. */
void mipt_default_params(mipt_params *p, int width, int height);

/* Create / destroy a rendering context on HIP device `device`.
 * Replaces constructing a RayTracer (RayTracer.cpp:310-317). */
int mipt_create(mipt_ctx **out, int device);
void mipt_destroy(mipt_ctx *ctx);

/* Choices a caller may make about how the library works; none changes a result.  No counterpart in the
 * reference.  They take effect at the next mipt_upload_scene.  (The library reads no environment variable
 * unless the process sets MIPT_DEBUG=1, which opens the tuning and test knobs of the same names in upper
 * case, MIPT_BUILDER=gpu, MIPT_LDS_STACK, ...)
 *   "builder"      0 = BVH built on the host (binned SAH; default), 1 = on the device (clustering; tens of
 *                  milliseconds for 10 M triangles, ~1.14x the traversal time); scenes below 4096 primitives
 *                  always use the host builder
 *   "leaf_max"     1..8, primitives per leaf the device builder aims for (default 2)
 *   "lds_stack"    1..24, traversal-stack entries per lane kept in LDS (default 24; the rest lives in HBM)
 *   "lean_any"     -1 = automatic, 0 = keep the analytic primitives' tests inside the shadow walk and the closest-hit walk
 *   "vis_bytes"    -1 = automatic, 0 = fp64 shadow-ray records, 1 = one byte wherever that is exact
 *   "tail_rays", "fuse_rays"   -1 = automatic, else the widest tree level (rays) that the depth-first
 *                  finishing launches take (0 = never)
 *   "overlap_rays" widest level whose shadow rays run on the second launch stream (default 2^18)
 * Returns MIPT_ERR_INVALID for an unknown name or a value out of range. */
int mipt_set_option(mipt_ctx *ctx, const char *name, int64_t value);

/* Hand the loaded scene to the device: replaces what RayTracer::loadScene
 * leaves behind (RayTracer.cpp:333-415): Scene::add + ComputeBoundingBox
 * (scene.cpp:54-107), Trimesh::buildTree (trimesh.cpp:62-68) and
 * Scene::buildTree (scene.cpp:149-154).  All arrays are copied. */
int mipt_upload_scene(mipt_ctx *ctx, const mipt_scene_desc *desc);

/* RayTracer::traceImage + waitRender + getBuffer (RayTracer.cpp:462-491,
 * 514-526, 319-324): render the whole frame described by `p` (all shards of
 * this call's shard_index) and return
 *   rgb_f32 : optional host buffer, 3*w*h floats, mean of the per-sample-clamped
 *             radiance (RayTracer.cpp:54,75) before 8-bit truncation;
 *   rgb8    : optional host buffer, 3*w*h bytes, index (i + j*w)*3, j = 0 is the
 *             bottom row, value (int)(255*mean) truncated (RayTracer.cpp:66,97-99).
 * Pixels outside this call's shard are left 0 (traceSetup zero-fills, :427). */
int mipt_render(mipt_ctx *ctx, const mipt_params *p, float *rgb_f32, uint8_t *rgb8);

/* The same work with every buffer resident in HBM, for callers that hold
 * device memory (multi-GPU reduce/gather, benchmarks):
 * adds, for every owned pixel and sample in [sample_begin, sample_end), the
 * clamped sample radiance into d_sum (device, 3*w*h doubles, caller-zeroed).
 * Runs on HIP stream `stream` (a hipStream_t, or NULL) and returns after the
 * work is enqueued and its queue sizes are known (it synchronises `stream`). */
int mipt_render_accumulate(mipt_ctx *ctx, const mipt_params *p, double *d_sum, void *stream);

/* mipt_render_accumulate for callers that cannot allocate device memory (tight serving loops that
 * must not fragment or grow the heap, harnesses that forbid hipMalloc).  No counterpart in the
 * reference.  Like the allocating call it is synchronous: it synchronises `stream` after every tree
 * level to read two counters back and makes small blocking host-to-device copies (light table, row
 * list), so it cannot run under HIP graph capture.  Same arguments, same result in d_sum; the queues, hit records, per-sample sums and
 * shadow-ray results live in a static 48 MiB device array instead of buffers sized to the job
 * (one copy per GPU the library is loaded on, held until the process ends), and the call makes
 * no hipMalloc / hipFree.  Work proceeds in batches of 2^14 path samples whatever batch_samples
 * says, so it is slower than the allocating call on large frames.  Limits, reported as
 * MIPT_ERR_LIMIT: height <= 16384 rows; shadow rays per shaded hit (area_samples per area light
 * + 1 per other light) <= 65536; the overflow stack for path trees that grow (glossy enclosures)
 * holds 2^18 rays and cannot be enlarged (the deepest rays are traced first, so a batch parks at
 * most 2^14 rays per tree level: enough for any tree of depth <= 16, and for deeper ones unless
 * nearly every hit spawns two rays).  MIPT_FLAG_COUNTERS / MIPT_FLAG_PROFILE are refused
 * (MIPT_ERR_INVALID).  Every call on one GPU shares that GPU's array: calls that overlap on
 * one GPU (two contexts, two streams) race; calls on different GPUs do not. */
int mipt_render_accumulate_static(mipt_ctx *ctx, const mipt_params *p, double *d_sum, void *stream);

/* d_sum / spp -> float image and/or RGB8 image, all device pointers
 * (RayTracer.cpp:75,97-99). */
int mipt_resolve(mipt_ctx *ctx, const double *d_sum, int width, int height, int spp,
                 float *d_rgb_f32, uint8_t *d_rgb8, void *stream);

/* RayTracer::tracePixel (RayTracer.cpp:59-101): the mean colour of pixel (i,j)
 * over p->spp samples, identical to the same pixel of mipt_render. */
int mipt_trace_pixel(mipt_ctx *ctx, const mipt_params *p, int i, int j, double out_rgb[3]);

/* Scene::intersect (scene.cpp:114-138) for n world-space rays given as host
 * arrays org[3n], dir[3n] (dir normalised by the caller).  Outputs (host):
 *   t[n]      hit parameter, or t_max on a miss
 *   prim[n]   reference-order primitive index (objects in the top-level BVH's
 *             visiting order, faces in their mesh BVH's), or -1
 *   normal[3n] world-space unit normal as isect::getN() returns it
 * Test hook: the reference has no such call but every parity test needs it. */
int mipt_intersect(mipt_ctx *ctx, int64_t n, const double *org, const double *dir, double t_max,
                   double *t, int32_t *prim, double *normal);

/* What the closest hit of each ray hands to shadeBRDF as surface colours: out
 * uv[2n] = isect::getUVCoordinates() (Square.cpp:41, Box.cpp:62-71,
 * trimesh.cpp:165-172; (0, 0) elsewhere), kd[3n] and ka[3n] = Material::kd(i),
 * ka(i) (material.h:197-212 through MaterialParameter::value,
 * material.cpp:224-230).  Zeros on a miss.  Test hook. */
int mipt_hit_material(mipt_ctx *ctx, int64_t n, const double *org, const double *dir, double t_max,
                      double *uv, double *kd, double *ka);

/* Light::shadowAttenuation * nothing else (light.cpp:15-35,63-85,114-147) for n
 * shading points: fire[3n] is firePos (material.cpp:117); the ray direction is
 * the light's getDirection(firePos) for point/directional lights, area lights
 * ignore it.  key[n] selects the random stream of each point (area lights).
 * out[3n] receives the returned colour. Test hook. */
int mipt_shadow_attenuation(mipt_ctx *ctx, const mipt_params *p, int light, int64_t n,
                            const double *fire, const uint64_t *key, double *out);

/* The shadow rays of n shaded hits, ray by ray, through the kernels a render launches for them (the persistent
 * any-hit kernel of the scene's kind and, where it runs, the pass over the analytic primitives; the walk through
 * translucent occluders in scenes that have any), not the one-ray-per-thread loop of mipt_shadow_attenuation.
 * Per hit k: P[3k] the hit point, N[3k] its normal (firePos = P + 3e-8 N, material.cpp:117), key[k] its random
 * stream, prim[k] the primitive it lies on as mipt_intersect reports it.  out[3 * (k * R + r)], R = rays per hit
 * (area_samples per area light + 1 per other light, lights in scene order): what ray r contributes before the
 * division by the sample count (light.cpp:118-144), i.e. colour * min(1, 1/denom), or 0 where it is blocked.
 * Test hook. */
int mipt_shadow_rays(mipt_ctx *ctx, const mipt_params *p, int64_t n, const double *P, const double *N,
                     const uint64_t *key, const int32_t *prim, double *out);

/* Counters since the last reset (filled only by calls with MIPT_FLAG_COUNTERS). */
int mipt_get_counters(mipt_ctx *ctx, mipt_counters *out);
int mipt_reset_counters(mipt_ctx *ctx);

/* Per-kernel device time of the launches made with MIPT_FLAG_PROFILE (no
 * counterpart in the reference, whose only timer is clock() around traceImage,
 * ui/CommandLineUI.cpp:78-101). */
int mipt_get_kernel_times(mipt_ctx *ctx, mipt_kernel_times *out);

/* Number of primitives (analytic objects + non-degenerate triangles) and BVH
 * nodes of the uploaded scene, and bytes of device memory they occupy. */
int mipt_scene_stats(mipt_ctx *ctx, int64_t *n_prims, int64_t *n_nodes, int64_t *device_bytes);

#ifdef __cplusplus
}
#endif
#endif /* MIPT_H_ */
