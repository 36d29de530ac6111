/*
 * rt_oracle.h -- C interface of the CPU restatement (see rt_oracle.c).
 * TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg; the shipped library never uses it.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_RNG_ROW = 0, ORC_RNG_STREAM = 1 };
enum { ORC_MATH_LIBM = 0, ORC_MATH_PM = 1 };

typedef struct orc_scene orc_scene;

orc_scene *orc_scene_load(const char *path, char *err, int errlen);
orc_scene *orc_scene_parse_text(const char *name, const char *text, char *err, int errlen);
orc_scene *orc_scene_from_triangles(const float *verts, int32_t n_tri);
void orc_scene_free(orc_scene *sc);

int orc_build_bvh(orc_scene *sc);
int32_t orc_bvh_depth(const orc_scene *sc);

/* out = {objects, nodes, materials, lights, width, height, samples, 0} */
void orc_scene_info(const orc_scene *sc, int32_t out[8]);
/* out = {camera xyz, look_at xyz, up xyz, fov, defocus_angle, exposure, sky rgb, 0} */
void orc_scene_params(const orc_scene *sc, float out[16]);
void orc_scene_set_render(orc_scene *sc, int32_t width, int32_t height, int32_t samples);
void orc_scene_set_camera(orc_scene *sc, const float cam[3], const float look_at[3], const float up[3], float fov,
                          float defocus_angle);
/* out = {origin, upper_left, du, dv, lens_u, lens_v (xyz each), defocus_angle} */
void orc_camera(const orc_scene *sc, float out[19]);

/* Raw arrays in the reference's own layouts: objects 84 B, nodes 32 B,
 * materials 24 B, lights 16 B. */
const void *orc_scene_objects(const orc_scene *sc);
const void *orc_scene_nodes(const orc_scene *sc);
const void *orc_scene_materials(const orc_scene *sc);
const void *orc_scene_lights(const orc_scene *sc);

/* rays: n x {origin xyz, direction xyz}.  counters (optional) =
 * {rays, shadow rays, samples, inner-node visits, triangle tests, sphere tests}. */
int orc_trace_closest(const orc_scene *sc, const float *rays, int64_t n, int32_t *idx, float *t, float *u, float *v,
                      uint64_t counters[6]);
int orc_trace_occluded(const orc_scene *sc, const float *rays, const float *dist, int64_t n, uint8_t *out);
int orc_hit_surface(const orc_scene *sc, const float *rays, int64_t n, const int32_t *idx, const float *t,
                    const float *u, const float *v, float *point, float *normal);

/* bgr: height*width*3 bytes (B,G,R, top row first); accum: height*width*3
 * floats (per-pixel radiance sums before the division by samples).  Either may
 * be NULL. */
int orc_render(orc_scene *sc, int rng_mode, int math_mode, uint32_t seed, int32_t samples_per_stream, int threads,
               uint8_t *bgr, float *accum, uint64_t counters[6]);

uint32_t orc_rng_next(uint32_t *state);
float orc_rng_float(uint32_t *state);
uint32_t orc_stream_seed(uint32_t seed, uint32_t pixel, uint32_t block);

/* out = {bsdf rgb, pdf, l xyz}; mat = {color rgb, roughness (already squared), ior, metallic} */
void orc_sample_bsdf(int math_mode, const float v[3], const float mat[6], uint32_t *rng, float out[7]);
/* out = {bsdf rgb, pdf} */
void orc_eval_bsdf(int math_mode, const float v[3], const float l[3], const float mat[6], float out[4]);
void orc_trace_path(const orc_scene *sc, int math_mode, const float ray[6], uint32_t *rng, float radiance[3]);
float orc_tonemap(int math_mode, float c, float exposure);
/* which: 0 sin, 1 cos, 2 log, 3 pow(x, y) of include/rt_pmath.h */
void orc_pmath(int which, const float *x, const float *y, float *out, int64_t n);

int orc_write_bmp(const char *path, int32_t w, int32_t h, const uint8_t *bgr);

#ifdef __cplusplus
}
#endif
#endif
