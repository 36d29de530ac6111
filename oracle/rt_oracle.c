/*
 * rt_oracle.c -- CPU restatement of the reference renderer's hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker for the HIP path: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load the
 * library built from it.  The shipped library never links or calls it.
 *
 * It restates, in plain C99, the algorithm of melby-md/raytracer (a C++11 CPU
 * path tracer).  Each function cites the reference lines it follows
 * (paths relative to the reference checkout).  Parity of this restatement with
 * the reference is PINNED: in the build container the reference itself is
 * compiled (oracle/Makefile -> oracle/_ref/) and tests/test_oracle_vs_ref.py
 * checks byte-identical BMP output, BVH node arrays, object permutation and
 * per-ray NearestHit/Occluded results; the committed fixtures under
 * tests/golden/ carry the same answers to machines without the reference.
 *
 * Two independent switches extend the reference so that a GPU render can be
 * compared with it:
 *   rng mode   ORC_RNG_ROW   : one xorshift32 stream per image row, seeded
 *                              69420 + row (src/main.cpp:659) - reference-identical.
 *              ORC_RNG_STREAM: one stream per (pixel, sample block); seeds and
 *                              summation order are defined in orc_render().
 *   math mode  ORC_MATH_LIBM : sinf/cosf/logf/powf from the host libm, as the
 *                              reference binary uses.
 *              ORC_MATH_PM   : the portable definitions of include/rt_pmath.h,
 *                              which a GPU can reproduce bit for bit.
 * ROW+LIBM reproduces the reference binary's image.bmp byte for byte.
 *
 * Build with -ffp-contract=off and without -ffast-math: the reference's
 * canonical build has no FMA contraction (build.sh:14-17 on baseline x86-64).
 */
#define _GNU_SOURCE
#include <errno.h>
#include <float.h>
#include <locale.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/rt_pmath.h"
#include "rt_oracle.h"

/* ------------------------------------------------------------------ */
/* vector algebra: src/common.h:52-305                                 */
/* ------------------------------------------------------------------ */

typedef struct { float x, y, z; } v3;
typedef struct { v3 i, j, k; } m3;     /* column major, src/common.h:68-71 */

#define ORC_PI 3.14159265f              /* src/common.h:73 */

/* Ternary min/max: a NaN first argument is dropped, a NaN second argument is
 * returned (src/common.h:75-81). */
static inline float fmin_t(float a, float b) { return a < b ? a : b; }
static inline float fmax_t(float a, float b) { return a > b ? a : b; }

static inline v3 v3_make(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_splat(float s) { v3 r = {s, s, s}; return r; }
static inline v3 v3_add(v3 a, v3 b) { v3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static inline v3 v3_sub(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static inline v3 v3_neg(v3 a) { v3 r = {-a.x, -a.y, -a.z}; return r; }
static inline v3 v3_mul(v3 a, v3 b) { v3 r = {a.x * b.x, a.y * b.y, a.z * b.z}; return r; }
static inline v3 v3_scale(v3 a, float s) { v3 r = {a.x * s, a.y * s, a.z * s}; return r; }
static inline v3 v3_div(v3 a, float s) { v3 r = {a.x / s, a.y / s, a.z / s}; return r; }
static inline v3 v3_rsub(float s, v3 a) { v3 r = {s - a.x, s - a.y, s - a.z}; return r; }
static inline v3 v3_min(v3 a, v3 b) { v3 r = {fmin_t(a.x, b.x), fmin_t(a.y, b.y), fmin_t(a.z, b.z)}; return r; }
static inline v3 v3_max(v3 a, v3 b) { v3 r = {fmax_t(a.x, b.x), fmax_t(a.y, b.y), fmax_t(a.z, b.z)}; return r; }
static inline float v3_get(v3 a, int axis) { return axis == 0 ? a.x : axis == 1 ? a.y : a.z; }

/* src/common.h:238-241: (x*x + y*y) + z*z */
static inline float v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float v3_len2(v3 a) { return v3_dot(a, a); }
static inline float v3_len(v3 a) { return sqrtf(v3_len2(a)); }
/* src/common.h:253-258: three true divisions by the length */
static inline v3 v3_unit(v3 a) { return v3_div(a, v3_len(a)); }
/* src/common.h:260-267 */
static inline v3 v3_cross(v3 a, v3 b)
{
	v3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
	return r;
}
/* src/common.h:233-236 */
static inline v3 v3_lerp(v3 a, v3 b, float t) { return v3_add(a, v3_scale(v3_sub(b, a), t)); }

/* src/common.h:284-287: (v.x*i + v.y*j) + v.z*k */
static inline v3 m3_apply(m3 m, v3 v)
{
	return v3_add(v3_add(v3_scale(m.i, v.x), v3_scale(m.j, v.y)), v3_scale(m.k, v.z));
}

/* src/common.h:289-296 */
static inline m3 m3_basis_from_z(v3 z)
{
	v3 a = (fabsf(z.x) > .9f) ? v3_make(0, 1, 0) : v3_make(1, 0, 0);
	v3 y = v3_unit(v3_cross(z, a));
	v3 x = v3_cross(z, y);
	m3 r = {x, y, z};
	return r;
}

/* src/common.h:298-305 */
static inline m3 m3_transpose(m3 m)
{
	m3 r = {
		{m.i.x, m.j.x, m.k.x},
		{m.i.y, m.j.y, m.k.y},
		{m.i.z, m.j.z, m.k.z},
	};
	return r;
}

/* ------------------------------------------------------------------ */
/* scene model: src/main.h:1-72, src/main.cpp:20-33                    */
/* ------------------------------------------------------------------ */

enum { OBJ_TRIANGLE = 0, OBJ_SPHERE = 1 };      /* src/main.h:1-4 */

typedef struct { v3 v0, v1, v2, n0, n1, n2; } tri_t;   /* src/main.h:11-14 */
typedef struct { v3 center; float radius; } sph_t;     /* src/main.h:16-19 */

/* Same field order and size (84 B) as the reference Object, src/main.h:28-36,
 * so that an object array can be memcmp'd against the reference's. */
typedef struct {
	int32_t type;
	int32_t material_idx;
	int32_t light_idx;
	union { tri_t tri; sph_t sph; } g;
} obj_t;

typedef struct { v3 color; float roughness, ior, metallic; } mat_t;   /* src/main.h:21-26 */
typedef struct { v3 color; int32_t object_idx; } light_t;             /* src/main.h:6-9 */
typedef struct { v3 lo, hi; int32_t index, count; } node_t;           /* src/main.cpp:20-23, 32 B */

typedef struct { v3 o, d; } ray_t;                                     /* src/main.cpp:25-27 */

struct orc_scene {
	v3 camera, up, look_at;
	float fov, defocus_angle, exposure;
	int32_t width, height, samples;
	v3 sky;

	int32_t n_obj, cap_obj;
	obj_t *obj;
	int32_t n_light, cap_light;
	light_t *light;
	int32_t n_mat, cap_mat;
	mat_t *mat;

	int32_t n_node;
	node_t *node;
};

typedef struct {
	uint64_t rays, shadow_rays, samples;
	uint64_t inner_visits, tri_tests, sph_tests;
} counters_t;

_Static_assert(sizeof(obj_t) == 84, "object layout");
_Static_assert(sizeof(node_t) == 32, "node layout");

/* ------------------------------------------------------------------ */
/* scene file grammar: src/parser.cpp:1-417                            */
/* ------------------------------------------------------------------ */

enum { T_STR = 1, T_LBRACE, T_RBRACE, T_LBRACK, T_RBRACK, T_END };

typedef struct {
	const char *file;
	const char *in;
	long pos, prev;
	int failed;
	char *err;
	int errlen;
} lex_t;

typedef struct { int type; const char *s; long n; } tok_t;

/* src/parser.cpp:20-30: message format "ERROR:file[byte+1]: msg" */
static void lex_fail(lex_t *lx, const char *msg)
{
	if (!lx->failed && lx->err)
		snprintf(lx->err, (size_t)lx->errlen, "ERROR:%s[%ld]: %s", lx->file, lx->prev + 1, msg);
	lx->failed = 1;
}

static int is_blank(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n'; }

/* src/parser.cpp:37-45 */
static int is_word(char c)
{
	switch (c) {
	case '\0': case '{': case '}': case '[': case ']': case '#':
		return 0;
	default:
		return !is_blank(c);
	}
}

/* src/parser.cpp:54-101 */
static tok_t lex_next(lex_t *lx)
{
	long start;
	int type = 0;
	if (lx->failed) {
		tok_t t = {T_END, lx->in, 0};
		return t;
	}
	do {
		start = lx->pos;
		char c = lx->in[lx->pos++];
		if (is_blank(c)) {
			while (is_blank(lx->in[lx->pos]))
				lx->pos++;
		} else if (c == '#') {
			/* comment to end of line; an unterminated last comment ends at NUL */
			while (lx->in[lx->pos] != '\n' && lx->in[lx->pos] != '\0')
				lx->pos++;
			if (lx->in[lx->pos] == '\n')
				lx->pos++;
		} else if (c == '\0') {
			type = T_END;
			lx->pos--;            /* stay on the terminator */
		} else if (c == '{') {
			type = T_LBRACE;
		} else if (c == '}') {
			type = T_RBRACE;
		} else if (c == '[') {
			type = T_LBRACK;
		} else if (c == ']') {
			type = T_RBRACK;
		} else {
			while (is_word(lx->in[lx->pos]))
				lx->pos++;
			type = T_STR;
		}
	} while (!type);
	lx->prev = start;
	tok_t t = {type, lx->in + start, lx->pos - start};
	return t;
}

static int tok_is(tok_t t, const char *lit)
{
	size_t n = strlen(lit);
	return (size_t)t.n == n && memcmp(t.s, lit, n) == 0;
}

/* src/parser.cpp:103-122 */
static int lex_cmd(lex_t *lx, tok_t *cmd)
{
	tok_t t = lex_next(lx);
	long start = lx->prev;
	if (lx->failed || t.type == T_END)
		return 0;
	if (t.type == T_STR) {
		if (lex_next(lx).type != T_LBRACE) {
			lex_fail(lx, "Expected '{'");
			return 0;
		}
		*cmd = t;
		lx->prev = start;
		return 1;
	}
	lex_fail(lx, "Expected command");
	return 0;
}

/* src/parser.cpp:124-137 */
static int lex_key(lex_t *lx, tok_t *key)
{
	tok_t t = lex_next(lx);
	if (lx->failed || t.type == T_RBRACE)
		return 0;
	if (t.type == T_STR) {
		*key = t;
		return 1;
	}
	lex_fail(lx, "Expected key");
	return 0;
}

/* src/parser.cpp:139-169: strtof must consume exactly the token */
static float lex_number(lex_t *lx)
{
	tok_t t = lex_next(lx);
	if (lx->failed)
		return 0;
	if (t.type != T_STR) {
		lex_fail(lx, "Expected string");
		return 0;
	}
	char *end = NULL;
	errno = 0;
	float n = strtof(t.s, &end);
	if (end - t.s != t.n || errno) {
		lex_fail(lx, "Invalid number");
		return 0;
	}
	return n;
}

/* src/parser.cpp:171-183.  The reference narrows to int16 after accepting
 * 0..65535; this restatement keeps the accepted value (no wrap), which is
 * identical for every value <= 32767. */
static int32_t lex_dim(lex_t *lx)
{
	tok_t t = lex_next(lx);
	if (lx->failed)
		return 0;
	if (t.type != T_STR) {
		lex_fail(lx, "Expected string");
		return 0;
	}
	char *end = NULL;
	errno = 0;
	long n = strtol(t.s, &end, 10);
	if (end - t.s != t.n || errno) {
		lex_fail(lx, "Invalid integer");
		return 0;
	}
	if (n > (1 << 16) - 1 || n < 0) {
		lex_fail(lx, "Out of bounds integer");
		return 0;
	}
	return (int32_t)n;
}

/* src/parser.cpp:205-222 */
static v3 lex_v3(lex_t *lx)
{
	v3 v = {0, 0, 0};
	if (lex_next(lx).type != T_LBRACK) {
		lex_fail(lx, "Expected array");
		return v;
	}
	v.x = lex_number(lx);
	v.y = lex_number(lx);
	v.z = lex_number(lx);
	if (lx->failed)
		return v;
	if (lex_next(lx).type != T_RBRACK)
		lex_fail(lx, "Expected ']'");
	return v;
}

static obj_t *scene_new_obj(orc_scene *sc)
{
	if (sc->n_obj == sc->cap_obj) {
		sc->cap_obj = sc->cap_obj ? sc->cap_obj * 2 : 1024;
		sc->obj = (obj_t *)realloc(sc->obj, (size_t)sc->cap_obj * sizeof(obj_t));
	}
	obj_t *o = &sc->obj[sc->n_obj++];
	memset(o, 0, sizeof *o);
	return o;
}

/* src/parser.cpp:224-236 (the light's object index is linked after the BVH build) */
static void scene_new_light(orc_scene *sc, int32_t obj_idx, v3 color)
{
	if (sc->n_light == sc->cap_light) {
		sc->cap_light = sc->cap_light ? sc->cap_light * 2 : 64;
		sc->light = (light_t *)realloc(sc->light, (size_t)sc->cap_light * sizeof(light_t));
	}
	int32_t li = sc->n_light++;
	sc->light[li].color = color;
	sc->light[li].object_idx = -1;
	sc->obj[obj_idx].light_idx = li;
}

static mat_t *scene_new_mat(orc_scene *sc)
{
	if (sc->n_mat == sc->cap_mat) {
		sc->cap_mat = sc->cap_mat ? sc->cap_mat * 2 : 64;
		sc->mat = (mat_t *)realloc(sc->mat, (size_t)sc->cap_mat * sizeof(mat_t));
	}
	return &sc->mat[sc->n_mat++];
}

/* Defaults: src/parser.cpp:240-258 */
static orc_scene *scene_alloc_defaults(void)
{
	orc_scene *sc = (orc_scene *)calloc(1, sizeof *sc);
	sc->up = v3_make(0, 0, 1);
	sc->defocus_angle = -1;
	sc->exposure = 1;
	sc->fov = 90;
	sc->width = 512;
	sc->height = 512;
	sc->samples = 20;
	mat_t *m = scene_new_mat(sc);
	m->color = v3_splat(.5f);
	m->roughness = 1;
	m->ior = 1.5f;
	m->metallic = 0;
	return sc;
}

/* src/parser.cpp:238-417 */
static orc_scene *scene_parse(const char *file, const char *text, char *err, int errlen)
{
	orc_scene *sc = scene_alloc_defaults();
	lex_t lx = {file, text, 0, 0, 0, err, errlen};

	int32_t cur_mat = 0;
	int area_light = 0;
	v3 area_color = {0, 0, 0};

	tok_t cmd, key;
	while (lex_cmd(&lx, &cmd)) {
		if (tok_is(cmd, "sphere")) {
			obj_t *o = scene_new_obj(sc);
			o->type = OBJ_SPHERE;
			o->material_idx = cur_mat;
			o->light_idx = -1;
			while (lex_key(&lx, &key)) {
				if (tok_is(key, "radius"))
					o->g.sph.radius = lex_number(&lx);
				else if (tok_is(key, "center"))
					o->g.sph.center = lex_v3(&lx);
				else
					lex_fail(&lx, "Unknown key");
			}
		} else if (tok_is(cmd, "triangle_mesh")) {
			while (lex_key(&lx, &key)) {
				if (!tok_is(key, "vertices")) {
					lex_fail(&lx, "Unknown key");
					continue;
				}
				if (lex_next(&lx).type != T_LBRACK) {
					lex_fail(&lx, "Expected array");
					break;
				}
				for (;;) {
					/* src/parser.cpp:193-203: peek for ']' */
					long save = lx.pos;
					tok_t t = lex_next(&lx);
					if (lx.failed || t.type == T_RBRACK)
						break;
					lx.pos = save;

					obj_t *o = scene_new_obj(sc);
					int32_t oi = sc->n_obj - 1;
					o->type = OBJ_TRIANGLE;
					o->material_idx = cur_mat;
					o->light_idx = -1;
					float f[9];
					for (int q = 0; q < 9; q++)
						f[q] = lex_number(&lx);
					if (lx.failed)
						break;
					tri_t *tr = &o->g.tri;
					tr->v0 = v3_make(f[0], f[1], f[2]);
					tr->v1 = v3_make(f[3], f[4], f[5]);
					tr->v2 = v3_make(f[6], f[7], f[8]);
					/* flat normal, src/parser.cpp:337-343 */
					v3 n = v3_unit(v3_cross(v3_sub(tr->v1, tr->v0), v3_sub(tr->v2, tr->v0)));
					tr->n0 = tr->n1 = tr->n2 = n;
					if (area_light)
						scene_new_light(sc, oi, area_color);
				}
			}
		} else if (tok_is(cmd, "material")) {
			mat_t *m = scene_new_mat(sc);
			cur_mat = sc->n_mat - 1;
			m->color = v3_splat(.7f);
			m->roughness = 1;
			m->ior = 1.5f;
			m->metallic = 0;
			while (lex_key(&lx, &key)) {
				if (tok_is(key, "color")) {
					m->color = lex_v3(&lx);
				} else if (tok_is(key, "roughness")) {
					float r = lex_number(&lx);
					m->roughness = r * r;       /* stored squared, src/parser.cpp:368-369 */
				} else if (tok_is(key, "ior")) {
					m->ior = lex_number(&lx);
				} else if (tok_is(key, "metallic")) {
					m->metallic = lex_number(&lx);
				} else {
					lex_fail(&lx, "Unknown key");
				}
			}
		} else if (tok_is(cmd, "area_light")) {
			area_light = 1;
			while (lex_key(&lx, &key)) {
				if (tok_is(key, "color"))
					area_color = lex_v3(&lx);
				else
					lex_fail(&lx, "Unknown key");
			}
		} else if (tok_is(cmd, "render")) {
			while (lex_key(&lx, &key)) {
				if (tok_is(key, "camera")) sc->camera = lex_v3(&lx);
				else if (tok_is(key, "look_at")) sc->look_at = lex_v3(&lx);
				else if (tok_is(key, "up")) sc->up = lex_v3(&lx);
				else if (tok_is(key, "fov")) sc->fov = lex_number(&lx);
				else if (tok_is(key, "defocus_angle")) sc->defocus_angle = lex_number(&lx);
				else if (tok_is(key, "exposure")) sc->exposure = lex_number(&lx);
				else if (tok_is(key, "width")) sc->width = lex_dim(&lx);
				else if (tok_is(key, "height")) sc->height = lex_dim(&lx);
				else if (tok_is(key, "samples")) sc->samples = lex_dim(&lx);
				else if (tok_is(key, "sky_box_color")) sc->sky = lex_v3(&lx);
				else lex_fail(&lx, "Unknown key");
			}
		} else {
			lex_fail(&lx, "Unknown command");
		}
		if (lx.failed)
			break;
	}
	if (lx.failed) {
		orc_scene_free(sc);
		return NULL;
	}
	return sc;
}

orc_scene *orc_scene_load(const char *path, char *err, int errlen)
{
	setlocale(LC_NUMERIC, "C");        /* src/main.cpp:596 */
	FILE *f = fopen(path, "rb");
	if (!f) {
		if (err)
			snprintf(err, (size_t)errlen, "Couldn't open '%s'", path);  /* src/parser.cpp:261-262 */
		return NULL;
	}
	fseek(f, 0, SEEK_END);
	long len = ftell(f);
	rewind(f);
	char *text = (char *)malloc((size_t)len + 1);
	size_t got = fread(text, 1, (size_t)len, f);
	fclose(f);
	text[got] = '\0';
	orc_scene *sc = scene_parse(path, text, err, errlen);
	free(text);
	return sc;
}

orc_scene *orc_scene_parse_text(const char *name, const char *text, char *err, int errlen)
{
	setlocale(LC_NUMERIC, "C");
	return scene_parse(name, text, err, errlen);
}

/* A scene made of bare triangles (synthetic soups): flat normals as the parser
 * computes them, default material, no lights. */
orc_scene *orc_scene_from_triangles(const float *verts, int32_t n_tri)
{
	orc_scene *sc = scene_alloc_defaults();
	sc->cap_obj = n_tri > 0 ? n_tri : 1;
	sc->obj = (obj_t *)calloc((size_t)sc->cap_obj, sizeof(obj_t));
	sc->n_obj = n_tri;
	for (int32_t i = 0; i < n_tri; i++) {
		const float *f = verts + (size_t)i * 9;
		obj_t *o = &sc->obj[i];
		o->type = OBJ_TRIANGLE;
		o->material_idx = 0;
		o->light_idx = -1;
		tri_t *tr = &o->g.tri;
		tr->v0 = v3_make(f[0], f[1], f[2]);
		tr->v1 = v3_make(f[3], f[4], f[5]);
		tr->v2 = v3_make(f[6], f[7], f[8]);
		v3 n = v3_unit(v3_cross(v3_sub(tr->v1, tr->v0), v3_sub(tr->v2, tr->v0)));
		tr->n0 = tr->n1 = tr->n2 = n;
	}
	return sc;
}

void orc_scene_free(orc_scene *sc)
{
	if (!sc)
		return;
	free(sc->obj);
	free(sc->light);
	free(sc->mat);
	free(sc->node);
	free(sc);
}

/* ------------------------------------------------------------------ */
/* BVH build: src/main.cpp:133-248, 607-622                            */
/* ------------------------------------------------------------------ */

/* src/main.cpp:133-168 */
static void node_fit(orc_scene *sc, int32_t ni)
{
	node_t *nd = &sc->node[ni];
	nd->lo = v3_splat(FLT_MAX);
	nd->hi = v3_splat(-FLT_MAX);
	int32_t last = nd->index + nd->count;
	for (int32_t i = nd->index; i < last; i++) {
		v3 lo = v3_splat(FLT_MAX), hi = v3_splat(-FLT_MAX);
		const obj_t *o = &sc->obj[i];
		if (o->type == OBJ_SPHERE) {
			lo = v3_sub(o->g.sph.center, v3_splat(o->g.sph.radius));
			hi = v3_add(o->g.sph.center, v3_splat(o->g.sph.radius));
		} else if (o->type == OBJ_TRIANGLE) {
			lo = o->g.tri.v0;
			lo = v3_min(lo, o->g.tri.v1);
			lo = v3_min(lo, o->g.tri.v2);
			hi = o->g.tri.v0;
			hi = v3_max(hi, o->g.tri.v1);
			hi = v3_max(hi, o->g.tri.v2);
		}
		nd->lo = v3_min(nd->lo, lo);
		nd->hi = v3_max(nd->hi, hi);
	}
}

/* src/main.cpp:170-248.  Midpoint split on the longest axis, two-pointer swap
 * partition, children at consecutive indices, explicit 64-entry stack whose
 * overflow abandons the current node (src/main.cpp:182-185). */
int orc_build_bvh(orc_scene *sc)
{
	if (sc->n_obj <= 0)
		return -1;
	free(sc->node);
	/* a binary tree over n leaves never needs more than 2n-1 nodes */
	sc->node = (node_t *)calloc((size_t)sc->n_obj * 2, sizeof(node_t));
	int32_t n_node = 1;
	int32_t cur = 0;
	sc->node[0].index = 0;
	sc->node[0].count = sc->n_obj;
	node_fit(sc, 0);

	int32_t stack[64];
	int sp = 0;

	for (;;) {
		node_t *nd = &sc->node[cur];
		if (sp >= 64) {
			cur = stack[--sp];
			continue;
		}
		if (nd->count <= 2) {
			if (sp == 0)
				break;
			cur = stack[--sp];
			continue;
		}

		v3 ext = v3_sub(nd->hi, nd->lo);
		int axis = 0;
		if (ext.y > ext.x)
			axis = 1;
		if (ext.z > v3_get(ext, axis))
			axis = 2;
		float split = v3_get(nd->lo, axis) + v3_get(ext, axis) * .5f;

		int32_t i = nd->index;
		int32_t j = i + nd->count - 1;
		while (i <= j) {
			const obj_t *o = &sc->obj[i];
			v3 c = {0, 0, 0};
			if (o->type == OBJ_SPHERE)
				c = o->g.sph.center;
			else if (o->type == OBJ_TRIANGLE)
				c = v3_scale(v3_add(v3_add(o->g.tri.v0, o->g.tri.v1), o->g.tri.v2), 1.f / 3.f);
			if (v3_get(c, axis) < split) {
				i++;
			} else {
				obj_t tmp = sc->obj[i];
				sc->obj[i] = sc->obj[j];
				sc->obj[j] = tmp;
				j--;
			}
		}

		int32_t left_count = i - nd->index;
		if (left_count == 0 || left_count == nd->count) {
			if (sp == 0)
				break;
			cur = stack[--sp];
			continue;
		}

		int32_t l = n_node++;
		int32_t r = n_node++;
		sc->node[l].index = nd->index;
		sc->node[l].count = left_count;
		sc->node[r].index = i;
		sc->node[r].count = nd->count - left_count;
		nd->index = l;
		nd->count = 0;
		node_fit(sc, l);
		node_fit(sc, r);
		stack[sp++] = r;
		cur = l;
	}
	sc->n_node = n_node;

	/* src/main.cpp:617-622: lights point at their object's new position */
	for (int32_t i = 0; i < sc->n_obj; i++)
		if (sc->obj[i].light_idx >= 0)
			sc->light[sc->obj[i].light_idx].object_idx = i;
	return 0;
}

/* ------------------------------------------------------------------ */
/* intersection: src/main.cpp:250-478                                  */
/* ------------------------------------------------------------------ */

/* src/main.cpp:250-279 */
static inline float hit_triangle(const tri_t *tr, const ray_t *ray, float *pu, float *pv)
{
	v3 e0 = v3_sub(tr->v0, tr->v2);
	v3 e1 = v3_sub(tr->v1, tr->v2);
	v3 pvec = v3_cross(ray->d, e1);
	float det = v3_dot(e0, pvec);
	if (det > -.0001f && det < .0001f)
		return FLT_MAX;
	v3 tvec = v3_sub(ray->o, tr->v2);
	float u = v3_dot(tvec, pvec) / det;
	if (u < 0 || u > 1)
		return FLT_MAX;
	v3 qvec = v3_cross(tvec, e0);
	float v = v3_dot(ray->d, qvec) / det;
	if (v < 0 || u + v > 1)
		return FLT_MAX;
	float t = v3_dot(e1, qvec) / det;
	*pu = u;
	*pv = v;
	if (t > .0001f)
		return t;
	return FLT_MAX;
}

/* src/main.cpp:281-301 */
static inline float hit_sphere(const sph_t *s, const ray_t *ray)
{
	v3 oc = v3_sub(s->center, ray->o);
	float h = v3_dot(ray->d, oc);
	float c = v3_len2(oc) - s->radius * s->radius;
	float delta = h * h - c;
	if (delta < .001f)
		return FLT_MAX;
	float sq = sqrtf(delta);
	float t = h - sq;
	if (t < .001f) {
		t = h + sq;
		if (t < .001f)
			return FLT_MAX;
	}
	return t;
}

/* src/main.cpp:303-330.  bounds[0] = lo, bounds[1] = hi; the sign bit (not a
 * comparison) picks the near slab; six true divisions. */
static inline float hit_box(const ray_t *ray, const node_t *nd, float max_t)
{
	const v3 *b = &nd->lo;        /* lo, hi are adjacent: b[0], b[1] */
	float tmin = 0.f, tmax = FLT_MAX;

	int s = signbit(ray->d.x) ? 1 : 0;
	float t1 = (b[s].x - ray->o.x) / ray->d.x;
	float t2 = (b[!s].x - ray->o.x) / ray->d.x;
	tmin = fmax_t(t1, tmin);
	tmax = fmin_t(t2, tmax);

	s = signbit(ray->d.y) ? 1 : 0;
	t1 = (b[s].y - ray->o.y) / ray->d.y;
	t2 = (b[!s].y - ray->o.y) / ray->d.y;
	tmin = fmax_t(t1, tmin);
	tmax = fmin_t(t2, tmax);

	s = signbit(ray->d.z) ? 1 : 0;
	t1 = (b[s].z - ray->o.z) / ray->d.z;
	t2 = (b[!s].z - ray->o.z) / ray->d.z;
	tmin = fmax_t(t1, tmin);
	tmax = fmin_t(t2, tmax);

	return (tmax >= tmin && tmin < max_t && tmax > 0) ? tmin : FLT_MAX;
}

typedef struct {
	int32_t idx;
	float t, u, v;        /* u, v only meaningful for triangles */
} hit_t;

/* src/main.cpp:332-422.  Returns the winning object index or -1.  The root box
 * is never tested; near child first, ties go to the second child; strict '<'
 * keeps the first-visited primitive on equal t. */
static inline int32_t nearest_hit(const orc_scene *sc, const ray_t *ray, hit_t *out, counters_t *cn)
{
	float best = FLT_MAX;
	int32_t best_idx = -1;
	float bu = 0, bv = 0;
	const node_t *nodes = sc->node;
	const node_t *nd = nodes;
	const node_t *stack[64];
	int sp = 0;

	for (;;) {
		if (nd->count > 0) {
			int32_t last = nd->index + nd->count;
			for (int32_t i = nd->index; i < last; i++) {
				const obj_t *o = &sc->obj[i];
				if (o->type == OBJ_SPHERE) {
					cn->sph_tests++;
					float t = hit_sphere(&o->g.sph, ray);
					if (t < best) {
						best = t;
						best_idx = i;
						bu = bv = 0;      /* barycentrics are a triangle notion */
					}
				} else if (o->type == OBJ_TRIANGLE) {
					cn->tri_tests++;
					float u = 0, v = 0;
					float t = hit_triangle(&o->g.tri, ray, &u, &v);
					if (t < best) {
						best = t;
						best_idx = i;
						bu = u;
						bv = v;
					}
				}
			}
			if (sp == 0)
				break;
			nd = stack[--sp];
		} else {
			cn->inner_visits++;
			const node_t *c1 = &nodes[nd->index];
			const node_t *c2 = &nodes[nd->index + 1];
			float d1 = hit_box(ray, c1, best);
			float d2 = hit_box(ray, c2, best);
			const node_t *nearc, *farc;
			float neard, fard;
			if (d1 < d2) {
				nearc = c1; neard = d1; farc = c2; fard = d2;
			} else {
				nearc = c2; neard = d2; farc = c1; fard = d1;
			}
			if (neard < FLT_MAX) {
				nd = nearc;
				if (fard < FLT_MAX)
					stack[sp++] = farc;
			} else if (fard < FLT_MAX) {
				nd = farc;
			} else {
				if (sp == 0)
					break;
				nd = stack[--sp];
			}
		}
	}
	out->idx = best_idx;
	out->t = best;
	out->u = bu;
	out->v = bv;
	return best_idx;
}

/* Hit point and shading normal of a recorded hit: src/main.cpp:353-354 (sphere),
 * src/main.cpp:365-368 (triangle). */
static inline void hit_surface(const orc_scene *sc, const ray_t *ray, const hit_t *h, v3 *point, v3 *normal)
{
	const obj_t *o = &sc->obj[h->idx];
	*point = v3_add(ray->o, v3_scale(ray->d, h->t));
	if (o->type == OBJ_SPHERE) {
		*normal = v3_div(v3_sub(*point, o->g.sph.center), o->g.sph.radius);
	} else {
		float w = 1 - h->u - h->v;
		const tri_t *tr = &o->g.tri;
		*normal = v3_unit(v3_add(v3_add(v3_scale(tr->n0, h->u), v3_scale(tr->n1, h->v)), v3_scale(tr->n2, w)));
	}
}

/* src/main.cpp:424-478: any-hit, first child first, no ordering */
static inline int occluded(const orc_scene *sc, const ray_t *ray, float dist, counters_t *cn)
{
	const node_t *nodes = sc->node;
	const node_t *nd = nodes;
	const node_t *stack[64];
	int sp = 0;

	for (;;) {
		if (nd->count > 0) {
			int32_t last = nd->index + nd->count;
			for (int32_t i = nd->index; i < last; i++) {
				const obj_t *o = &sc->obj[i];
				if (o->type == OBJ_SPHERE) {
					cn->sph_tests++;
					if (hit_sphere(&o->g.sph, ray) < dist)
						return 1;
				} else if (o->type == OBJ_TRIANGLE) {
					cn->tri_tests++;
					float u = 0, v = 0;
					if (hit_triangle(&o->g.tri, ray, &u, &v) < dist)
						return 1;
				}
			}
			if (sp == 0)
				break;
			nd = stack[--sp];
		} else {
			cn->inner_visits++;
			const node_t *c1 = &nodes[nd->index];
			const node_t *c2 = &nodes[nd->index + 1];
			int h1 = hit_box(ray, c1, dist) < FLT_MAX;
			int h2 = hit_box(ray, c2, dist) < FLT_MAX;
			if (h1) {
				nd = c1;
				if (h2)
					stack[sp++] = c2;
			} else if (h2) {
				nd = c2;
			} else {
				if (sp == 0)
					break;
				nd = stack[--sp];
			}
		}
	}
	return 0;
}

/* ------------------------------------------------------------------ */
/* random numbers: src/main.cpp:83-131                                 */
/* ------------------------------------------------------------------ */

/* src/main.cpp:83-93: xorshift32 (13, 17, 5) */
static inline uint32_t rng_next(uint32_t *st)
{
	uint32_t x = *st;
	x ^= x << 13;
	x ^= x >> 17;
	x ^= x << 5;
	*st = x;
	return x;
}

/* src/main.cpp:95-98: top 24 bits -> [0,1) */
static inline float rng_float(uint32_t *st)
{
	return (float)(rng_next(st) >> 8) / (1 << 24);
}

/* src/main.cpp:100-103 */
static inline uint32_t rng_range(uint32_t *st, uint32_t lo, uint32_t hi)
{
	return rng_next(st) % (hi - lo + 1) + lo;
}

/* src/main.cpp:115-131: uniform barycentrics by folding the unit square */
static inline v3 rng_barycentric(uint32_t *st)
{
	float r1 = rng_float(st);
	float r2 = rng_float(st);
	float u, v;
	if (r1 < r2) {
		u = r1 / 2;
		v = r2 - u;
	} else {
		v = r2 / 2;
		u = r1 - v;
	}
	return v3_make(u, v, 1 - u - v);
}

uint32_t orc_rng_next(uint32_t *state) { return rng_next(state); }
float orc_rng_float(uint32_t *state) { return rng_float(state); }

/* Seed of the stream that serves sample block `block` of pixel `pixel`
 * (ORC_RNG_STREAM).  Two rounds of a 32-bit avalanche mixer; zero is remapped
 * because xorshift32 is stuck at zero. */
static inline uint32_t mix32(uint32_t h)
{
	h ^= h >> 16;
	h *= 0x7feb352dU;
	h ^= h >> 15;
	h *= 0x846ca68bU;
	h ^= h >> 16;
	return h;
}

uint32_t orc_stream_seed(uint32_t seed, uint32_t pixel, uint32_t block)
{
	uint32_t h = mix32(pixel + seed);
	h = mix32(h ^ (block * 0x9e3779b9U + 0x7f4a7c15U));
	return h ? h : 0x9e3779b9U;
}

/* ------------------------------------------------------------------ */
/* camera: src/main.cpp:624-648                                        */
/* ------------------------------------------------------------------ */

typedef struct {
	v3 origin, upper_left, du, dv, lens_u, lens_v;
	float defocus_angle;
} camera_t;

static camera_t camera_setup(const orc_scene *sc)
{
	camera_t c;
	float focus = v3_len(v3_sub(sc->look_at, sc->camera));
	float fov_r = sc->fov * ORC_PI / 180;
	float aspect = (float)sc->width / (float)sc->height;
	float vh = 2 * tanf(fov_r / 2) * focus;
	float vw = vh * aspect;

	v3 fwd = v3_unit(v3_sub(sc->look_at, sc->camera));
	v3 right = v3_unit(v3_cross(fwd, sc->up));
	v3 up = v3_cross(right, fwd);

	v3 vu = v3_scale(right, vw);
	v3 vv = v3_scale(v3_neg(up), vh);

	c.origin = sc->camera;
	c.upper_left = v3_sub(v3_sub(v3_add(sc->camera, v3_scale(fwd, focus)), v3_div(vu, 2)), v3_div(vv, 2));
	c.du = v3_div(vu, (float)sc->width);
	c.dv = v3_div(vv, (float)sc->height);

	float defocus_r = sc->defocus_angle * ORC_PI / 180;
	float lens = focus * tanf(defocus_r / 2);
	c.lens_u = v3_scale(right, lens);
	c.lens_v = v3_scale(up, lens);
	c.defocus_angle = sc->defocus_angle;
	return c;
}

void orc_camera(const orc_scene *sc, float out[19])
{
	camera_t c = camera_setup(sc);
	memcpy(out, &c, sizeof(float) * 19);
}

/* ------------------------------------------------------------------ */
/* transport, instantiated once per math mode                          */
/* ------------------------------------------------------------------ */

#define MFN(name) name##_libm
#define M_SIN(x) sinf(x)
#define M_COS(x) cosf(x)
#define M_LOG(x) logf(x)
#define M_POW(x, y) powf(x, y)
#include "rt_oracle_transport.inc"
#undef MFN
#undef M_SIN
#undef M_COS
#undef M_LOG
#undef M_POW

#define MFN(name) name##_pm
#define M_SIN(x) pm_sinf(x)
#define M_COS(x) pm_cosf(x)
#define M_LOG(x) pm_logf(x)
#define M_POW(x, y) pm_powf(x, y)
#include "rt_oracle_transport.inc"
#undef MFN
#undef M_SIN
#undef M_COS
#undef M_LOG
#undef M_POW

/* ------------------------------------------------------------------ */
/* public entry points                                                 */
/* ------------------------------------------------------------------ */

void orc_scene_info(const orc_scene *sc, int32_t out[8])
{
	out[0] = sc->n_obj;
	out[1] = sc->n_node;
	out[2] = sc->n_mat;
	out[3] = sc->n_light;
	out[4] = sc->width;
	out[5] = sc->height;
	out[6] = sc->samples;
	out[7] = 0;
}

void orc_scene_set_render(orc_scene *sc, int32_t width, int32_t height, int32_t samples)
{
	if (width > 0) sc->width = width;
	if (height > 0) sc->height = height;
	if (samples >= 0) sc->samples = samples;
}

void orc_scene_params(const orc_scene *sc, float out[16])
{
	out[0] = sc->camera.x; out[1] = sc->camera.y; out[2] = sc->camera.z;
	out[3] = sc->look_at.x; out[4] = sc->look_at.y; out[5] = sc->look_at.z;
	out[6] = sc->up.x; out[7] = sc->up.y; out[8] = sc->up.z;
	out[9] = sc->fov; out[10] = sc->defocus_angle; out[11] = sc->exposure;
	out[12] = sc->sky.x; out[13] = sc->sky.y; out[14] = sc->sky.z;
	out[15] = 0;
}

void orc_scene_set_camera(orc_scene *sc, const float cam[3], const float look_at[3], const float up[3], float fov,
                          float defocus_angle)
{
	sc->camera = v3_make(cam[0], cam[1], cam[2]);
	sc->look_at = v3_make(look_at[0], look_at[1], look_at[2]);
	sc->up = v3_make(up[0], up[1], up[2]);
	sc->fov = fov;
	sc->defocus_angle = defocus_angle;
}

const void *orc_scene_objects(const orc_scene *sc) { return sc->obj; }
const void *orc_scene_nodes(const orc_scene *sc) { return sc->node; }
const void *orc_scene_materials(const orc_scene *sc) { return sc->mat; }
const void *orc_scene_lights(const orc_scene *sc) { return sc->light; }

/* Depth of the deepest leaf (root = 0); what a traversal stack must hold. */
int32_t orc_bvh_depth(const orc_scene *sc)
{
	if (!sc->node)
		return -1;
	int32_t best = 0;
	int32_t *st = (int32_t *)malloc(sizeof(int32_t) * 2 * 256);
	int sp = 0;
	st[0] = 0; st[1] = 0; sp = 1;
	while (sp) {
		sp--;
		int32_t n = st[2 * sp], d = st[2 * sp + 1];
		if (d > best) best = d;
		if (sc->node[n].count == 0 && sp + 2 <= 256) {
			st[2 * sp] = sc->node[n].index; st[2 * sp + 1] = d + 1; sp++;
			st[2 * sp] = sc->node[n].index + 1; st[2 * sp + 1] = d + 1; sp++;
		}
	}
	free(st);
	return best;
}

int orc_trace_closest(const orc_scene *sc, const float *rays, int64_t n, int32_t *idx, float *t, float *u, float *v,
                      uint64_t counters[6])
{
	if (!sc->node)
		return -1;
	counters_t total;
	memset(&total, 0, sizeof total);
#pragma omp parallel
	{
		counters_t cn;
		memset(&cn, 0, sizeof cn);
#pragma omp for schedule(dynamic, 4096)
		for (int64_t i = 0; i < n; i++) {
			ray_t r;
			memcpy(&r, rays + i * 6, sizeof r);
			hit_t h;
			cn.rays++;
			nearest_hit(sc, &r, &h, &cn);
			idx[i] = h.idx;
			if (t) t[i] = h.t;
			if (u) u[i] = h.u;
			if (v) v[i] = h.v;
		}
#pragma omp critical
		{
			total.rays += cn.rays;
			total.inner_visits += cn.inner_visits;
			total.tri_tests += cn.tri_tests;
			total.sph_tests += cn.sph_tests;
		}
	}
	if (counters) {
		counters[0] = total.rays; counters[1] = 0; counters[2] = 0;
		counters[3] = total.inner_visits; counters[4] = total.tri_tests; counters[5] = total.sph_tests;
	}
	return 0;
}

int orc_trace_occluded(const orc_scene *sc, const float *rays, const float *dist, int64_t n, uint8_t *out)
{
	if (!sc->node)
		return -1;
#pragma omp parallel
	{
		counters_t cn;
		memset(&cn, 0, sizeof cn);
#pragma omp for schedule(dynamic, 4096)
		for (int64_t i = 0; i < n; i++) {
			ray_t r;
			memcpy(&r, rays + i * 6, sizeof r);
			out[i] = (uint8_t)occluded(sc, &r, dist[i], &cn);
		}
	}
	return 0;
}

/* Hit point and normal for a batch of recorded hits (miss rows are zeroed). */
int orc_hit_surface(const orc_scene *sc, const float *rays, int64_t n, const int32_t *idx, const float *t,
                    const float *u, const float *v, float *point, float *normal)
{
	for (int64_t i = 0; i < n; i++) {
		v3 p = {0, 0, 0}, nn = {0, 0, 0};
		if (idx[i] >= 0) {
			ray_t r;
			memcpy(&r, rays + i * 6, sizeof r);
			hit_t h = {idx[i], t[i], u[i], v[i]};
			hit_surface(sc, &r, &h, &p, &nn);
		}
		memcpy(point + i * 3, &p, 12);
		memcpy(normal + i * 3, &nn, 12);
	}
	return 0;
}

int orc_render(orc_scene *sc, int rng_mode, int math_mode, uint32_t seed, int32_t samples_per_stream, int threads,
               uint8_t *bgr, float *accum, uint64_t counters[6])
{
	if (!sc->node)
		return -1;
	if (math_mode == ORC_MATH_PM)
		return render_pm(sc, rng_mode, seed, samples_per_stream, threads, bgr, accum, counters);
	return render_libm(sc, rng_mode, seed, samples_per_stream, threads, bgr, accum, counters);
}

/* Unit hooks for known-answer tests. */
void orc_sample_bsdf(int math_mode, const float v[3], const float mat[6], uint32_t *rng, float out[7])
{
	mat_t m = {{mat[0], mat[1], mat[2]}, mat[3], mat[4], mat[5]};
	v3 vv = {v[0], v[1], v[2]};
	bsample_t s = math_mode == ORC_MATH_PM ? sample_bsdf_pm(vv, &m, rng) : sample_bsdf_libm(vv, &m, rng);
	out[0] = s.bsdf.x; out[1] = s.bsdf.y; out[2] = s.bsdf.z;
	out[3] = s.pdf;
	out[4] = s.l.x; out[5] = s.l.y; out[6] = s.l.z;
}

void orc_eval_bsdf(int math_mode, const float v[3], const float l[3], const float mat[6], float out[4])
{
	mat_t m = {{mat[0], mat[1], mat[2]}, mat[3], mat[4], mat[5]};
	v3 vv = {v[0], v[1], v[2]}, ll = {l[0], l[1], l[2]};
	v3 f = math_mode == ORC_MATH_PM ? bsdf_eval_pm(vv, ll, &m) : bsdf_eval_libm(vv, ll, &m);
	float p = math_mode == ORC_MATH_PM ? bsdf_pdf_pm(vv, ll, &m) : bsdf_pdf_libm(vv, ll, &m);
	out[0] = f.x; out[1] = f.y; out[2] = f.z; out[3] = p;
}

/* One full path (src/main.cpp:496-592) from a given ray and RNG state. */
void orc_trace_path(const orc_scene *sc, int math_mode, const float ray[6], uint32_t *rng, float radiance[3])
{
	counters_t cn;
	memset(&cn, 0, sizeof cn);
	ray_t r;
	memcpy(&r, ray, sizeof r);
	v3 c = math_mode == ORC_MATH_PM ? trace_path_pm(sc, &r, rng, &cn) : trace_path_libm(sc, &r, rng, &cn);
	radiance[0] = c.x; radiance[1] = c.y; radiance[2] = c.z;
}

float orc_tonemap(int math_mode, float c, float exposure)
{
	return math_mode == ORC_MATH_PM ? tonemap_pm(c, exposure) : tonemap_libm(c, exposure);
}

void orc_pmath(int which, const float *x, const float *y, float *out, int64_t n)
{
	for (int64_t i = 0; i < n; i++) {
		switch (which) {
		case 0: out[i] = pm_sinf(x[i]); break;
		case 1: out[i] = pm_cosf(x[i]); break;
		case 2: out[i] = pm_logf(x[i]); break;
		case 4: { float c; pm_sincosf(x[i], &out[i], &c); break; }     /* the sine of the pair */
		case 5: { float s; pm_sincosf(x[i], &s, &out[i]); break; }     /* the cosine of the pair */
		default: out[i] = pm_powf(x[i], y[i]); break;
		}
	}
}

/* src/main.cpp:48-81: 24-bit bottom-up BMP, rows padded to 4 bytes */
int orc_write_bmp(const char *path, int32_t w, int32_t h, const uint8_t *bgr)
{
	int filesize = 54 + 3 * w * h;
	uint8_t fh[14] = {'B', 'M', 0, 0, 0, 0, 0, 0, 0, 0, 54, 0, 0, 0};
	uint8_t ih[40] = {40, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 24, 0};
	uint8_t pad[3] = {0, 0, 0};
	for (int k = 0; k < 4; k++) {
		fh[2 + k] = (uint8_t)(filesize >> (8 * k));
		ih[4 + k] = (uint8_t)(w >> (8 * k));
		ih[8 + k] = (uint8_t)(h >> (8 * k));
	}
	FILE *f = fopen(path, "wb");
	if (!f)
		return -1;
	fwrite(fh, 1, 14, f);
	fwrite(ih, 1, 40, f);
	for (int i = 0; i < h; i++) {
		fwrite(bgr + (size_t)w * (size_t)(h - i - 1) * 3, 3, (size_t)w, f);
		fwrite(pad, 1, (size_t)((4 - (w * 3) % 4) % 4), f);
	}
	fclose(f);
	return 0;
}
