/*
 * ref_harness.cpp -- C entry points into the REAL reference renderer.
 *
 * TEST INFRASTRUCTURE ONLY.  The reference is a single translation unit whose
 * functions are all `static`; this file pulls that translation unit in from
 * where it lies (REF_MAIN_CPP, set by oracle/Makefile to the reference
 * checkout's src/main.cpp) with its main() renamed, and exports thin wrappers.
 * Nothing of the reference is copied into this repository; the built library
 * goes to oracle/_ref/ (git-ignored).  It exists to validate rt_oracle.c and to
 * generate the fixtures under tests/golden/ (tests/golden/make_golden.py).
 */
#define main ref_main
#include REF_MAIN_CPP
#undef main

static Scene g_scene;
static BVHNode g_nodes[16384];
static int g_node_count;

static int CountNodes(void)
{
	int hi = 0;
	int stack[128];
	int sp = 0;
	stack[sp++] = 0;
	while (sp) {
		int n = stack[--sp];
		if (n > hi)
			hi = n;
		if (g_nodes[n].object_count == 0 && sp + 2 <= 128) {
			stack[sp++] = g_nodes[n].index;
			stack[sp++] = g_nodes[n].index + 1;
		}
	}
	return hi + 1;
}

extern "C" {

int ref_run(const char *scene_path)
{
	char arg0[] = "raytracer";
	char *argv[3] = {arg0, (char *)scene_path, nullptr};
	return ref_main(2, argv);
}

/* src/main.cpp:596-622 minus the render loop */
void ref_load(const char *scene_path)
{
	setlocale(LC_NUMERIC, "C");
	memset(&g_scene, 0, sizeof g_scene);
	memset(g_nodes, 0, sizeof g_nodes);
	LoadScene(&g_scene, scene_path);
	BVHTree tree = {g_nodes, 16384, g_scene.objects};
	BuildBVH(&tree, g_scene.object_count);
	g_scene.bvh = tree;
	for (i32 i = 0; i < g_scene.object_count; i++) {
		Object *obj = &g_scene.objects[i];
		if (obj->light_idx >= 0)
			g_scene.lights[obj->light_idx].object_idx = i;
	}
	g_node_count = CountNodes();
}

void ref_info(int32_t out[8])
{
	out[0] = g_scene.object_count;
	out[1] = g_node_count;
	out[2] = g_scene.material_count;
	out[3] = g_scene.light_count;
	out[4] = g_scene.width;
	out[5] = g_scene.height;
	out[6] = g_scene.samples;
	out[7] = (int32_t)sizeof(Object);
}

const void *ref_objects(void) { return g_scene.objects; }
const void *ref_nodes(void) { return g_nodes; }
const void *ref_materials(void) { return g_scene.materials; }
const void *ref_lights(void) { return g_scene.lights; }

void ref_nearest(const float *rays, int64_t n, int32_t *idx, float *t, float *u, float *v, float *point,
                 float *normal)
{
	for (int64_t i = 0; i < n; i++) {
		Ray ray;
		memcpy(&ray, rays + i * 6, sizeof ray);
		Hit hit;
		hit.point = Vec3{0, 0, 0};
		hit.normal = Vec3{0, 0, 0};
		bool did = NearestHit(&g_scene.bvh, &ray, &hit);
		idx[i] = hit.obj_idx;
		float tt = FLT_MAX, uu = 0, vv = 0;
		if (did) {
			Object *obj = &g_scene.objects[hit.obj_idx];
			if (obj->type == OBJ_TRIANGLE)
				tt = HitTriangle(&obj->triangle, &ray, &uu, &vv);
			else
				tt = HitSphere(&obj->sphere, &ray);
		} else {
			hit.point = Vec3{0, 0, 0};
			hit.normal = Vec3{0, 0, 0};
		}
		t[i] = tt;
		u[i] = uu;
		v[i] = vv;
		memcpy(point + i * 3, &hit.point, 12);
		memcpy(normal + i * 3, &hit.normal, 12);
	}
}

void ref_occluded(const float *rays, const float *dist, int64_t n, uint8_t *out)
{
	for (int64_t i = 0; i < n; i++) {
		Ray ray;
		memcpy(&ray, rays + i * 6, sizeof ray);
		out[i] = Occluded(&g_scene.bvh, &ray, dist[i]) ? 1 : 0;
	}
}

void ref_trace_path(const float *ray6, uint32_t *rng, float radiance[3])
{
	Ray ray;
	memcpy(&ray, ray6, sizeof ray);
	Vec3 c = RayTrace(&g_scene, &ray, rng);
	radiance[0] = c.x;
	radiance[1] = c.y;
	radiance[2] = c.z;
}

void ref_sample_bsdf(const float v[3], const float mat[6], uint32_t *rng, float out[7])
{
	Material m = {{mat[0], mat[1], mat[2]}, mat[3], mat[4], mat[5]};
	BSample s = SampleBSDF(Vec3{v[0], v[1], v[2]}, &m, rng);
	out[0] = s.bsdf.x; out[1] = s.bsdf.y; out[2] = s.bsdf.z;
	out[3] = s.pdf;
	out[4] = s.l.x; out[5] = s.l.y; out[6] = s.l.z;
}

void ref_eval_bsdf(const float v[3], const float l[3], const float mat[6], float out[4])
{
	Material m = {{mat[0], mat[1], mat[2]}, mat[3], mat[4], mat[5]};
	Vec3 f = BSDF(Vec3{v[0], v[1], v[2]}, Vec3{l[0], l[1], l[2]}, &m);
	out[0] = f.x; out[1] = f.y; out[2] = f.z;
	out[3] = BSDFPDF(Vec3{v[0], v[1], v[2]}, Vec3{l[0], l[1], l[2]}, &m);
}

float ref_tonemap(float c, float exposure) { return LinearToGamma(c, exposure); }
uint32_t ref_rng_next(uint32_t *state) { return _rand(state); }
float ref_rng_float(uint32_t *state) { return Rand(state); }

void ref_random_triangle(uint32_t *state, float out[3])
{
	Vec3 b = RandomTriangle(state);
	out[0] = b.x; out[1] = b.y; out[2] = b.z;
}

} /* extern "C" */
