/*
 * oracle.h — C interface of the CPU oracle (liboracle.so).
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain fp64 CPU restatement of the
 * reference renderer's path-tracing path, used to check the HIP library and as
 * the timed CPU baseline.  Nothing in the product (libmipt.so, the host shim,
 * the Python package) links, loads or calls it; only tests/, the smoke check and
 * bench.py's cpu_baseline leg do.
 *
 * PARITY UNPINNED per ray: the reference ships no golden vectors, unit tests
 * or known-answer fixtures for this path and cannot be compiled here (glm,
 * FLTK and GLU headers are absent).  What the reference does publish are GUI
 * screenshots of its renders of four demo scenes; the restatement is pinned to
 * those statistically (32x32 block means, tests/test_reference_screenshots.py),
 * and beyond that by the reference's source text and by closed-form known
 * answers (tests/test_oracle_kat.py).
 *
 * Scene and parameter structs are the POD declarations of include/mipt.h.
 */
#ifndef MIPT_ORACLE_H_
#define MIPT_ORACLE_H_

#include "../include/mipt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

/* rng_mode of orc_render */
#define ORC_RNG_COUNTER 0    /* counter-based stream keyed by (pixel, sample, path node):
                                the stream the HIP library uses, results comparable per pixel */
#define ORC_RNG_SEQUENTIAL 1 /* one sequential generator per thread consumed in the reference's
                                depth-first order (RayTracer.cpp:228,244,247,268; light.cpp:107-108) */

/* Build the reference's object graph from a scene description: per-object
 * MatrixTransform + world bounding box (scene.cpp:54-107), per-mesh face list and
 * BVH<TrimeshFace> (trimesh.cpp:62-68), top-level BVH<Geometry> (scene.cpp:149-154). */
orc_scene *orc_create(const mipt_scene_desc *desc);
void orc_destroy(orc_scene *s);
const char *orc_last_error(void);

/* RayTracer::traceImage (RayTracer.cpp:462-491) with `threads` column chunks.
 * mean_rgb: 3*w*h doubles (mean of per-sample-clamped radiance), optional.
 * rgb8: 3*w*h bytes in the reference's buffer layout, optional.
 * counters: optional. Honors p->shard_*, sample_begin/end like the library. */
int orc_render(orc_scene *s, const mipt_params *p, int threads, int rng_mode, double *mean_rgb,
               uint8_t *rgb8, mipt_counters *counters);

/* RayTracer::tracePixel (RayTracer.cpp:59-101), counter RNG. */
int orc_trace_pixel(orc_scene *s, const mipt_params *p, int i, int j, double out_rgb[3]);

/* Scene::intersect (scene.cpp:114-138) per ray. prim = reference-order primitive
 * index (see mipt_intersect), -1 on miss; t = t_max on miss. */
int orc_intersect(orc_scene *s, int64_t n, const double *org, const double *dir, double t_max,
                  double *t, int32_t *prim, double *normal);

/* What the closest hit of each ray hands to shadeBRDF: its UV coordinates (Square.cpp:41,
 * Box.cpp:62-71, trimesh.cpp:165-172; (0, 0) elsewhere) and the diffuse and ambient colours
 * as MaterialParameter::value returns them there (material.cpp:224-230).  Zeros on a miss. */
int orc_hit_material(orc_scene *s, int64_t n, const double *org, const double *dir, double t_max,
                     double *uv, double *kd, double *ka);

/* The same closest hit found by testing every primitive with no BVH at all
 * (the commented-out loops of scene.cpp:115-128 / trimesh.cpp:72-84). */
int orc_intersect_brute(orc_scene *s, int64_t n, const double *org, const double *dir,
                        double t_max, double *t, int32_t *prim, double *normal);

/* Light::shadowAttenuation (light.cpp:15-35,63-85,114-147). */
int orc_shadow_attenuation(orc_scene *s, const mipt_params *p, int light, int64_t n,
                           const double *fire, const uint64_t *key, double *out);

/* The same per shadow ray: out[3 * (k * S + s)] is what sample s contributes before the division by the sample
 * count (one pass of the loop of light.cpp:118-144), S = area_samples for a rectangle area light; S = 1 and the
 * whole value for point and directional lights. */
int orc_shadow_samples(orc_scene *s, const mipt_params *p, int light, int64_t n,
                       const double *fire, const uint64_t *key, double *out);

/* Material::shadeBRDF (material.cpp:88-170) for one hit with the lights of the
 * scene: P (hit point), N, V (= wOut direction), w (= -wIn direction, the bounce
 * direction), Lin (indirect radiance), material index, random key. */
int orc_shade_brdf(orc_scene *s, const mipt_params *p, int material, const double P[3],
                   const double N[3], const double V[3], const double w[3], const double Lin[3],
                   uint64_t key, double out[3]);

/* Visiting order of the top-level BVH: order[k] = index (file order) of the k-th
 * object in BVH<Geometry>::allNodes after construction (bvh.h:40-131). */
int orc_object_order(orc_scene *s, int32_t *order);

/* CubeMap::getColor (cubeMap.cpp:7-64) for n directions. */
int orc_cubemap_color(orc_scene *s, int64_t n, const double *dir, double *out);

/* Uniform random number j of stream `key` (the counter generator both sides use). */
/* Mutants of the restatement, for measuring what the published-render pin can detect
 * (tests/test_reference_screenshots.py): 0 = faithful, 1..orc_mutation_count()-1 each switch
 * one quirk of the reference off or one common mistake on (the list is in oracle.cpp).
 * Process-wide. */
int orc_set_mutation(int m);
int orc_mutation_count(void);

double orc_uniform(uint64_t key, uint64_t j);
uint64_t orc_mix(uint64_t key, uint64_t j);
uint64_t orc_sample_key(uint64_t seed, uint64_t pixel, uint64_t sample);

#ifdef __cplusplus
}
#endif
#endif
