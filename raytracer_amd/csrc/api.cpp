// Host half of include/raytracer_amd.h: scene files, BVH build, camera, BMP.
// Nothing here touches the GPU.

#include <cstdio>
#include <cstring>
#include <string>

#include "host_scene.hpp"

namespace rta {

namespace {
thread_local std::string g_error;
}

void set_global_error(const std::string &text) { g_error = text; }
const char *get_global_error() { return g_error.c_str(); }

namespace {

void report(const SceneError &e, char *err, size_t errlen)
{
	g_error = e.text;
	if (err && errlen)
		snprintf(err, errlen, "%s", e.text.c_str());
}

template <typename Make>
int make_scene(rta_scene **out, char *err, size_t errlen, Make &&make)
{
	if (!out)
		return RTA_E_INVALID;
	*out = nullptr;
	try {
		HostScene *impl = make();
		*out = new rta_scene{impl};
		return RTA_OK;
	} catch (const SceneError &e) {
		report(e, err, errlen);
		return e.code;
	} catch (const std::exception &e) {
		report(SceneError{RTA_E_INVALID, e.what()}, err, errlen);
		return RTA_E_INVALID;
	}
}

}  // namespace
}  // namespace rta

using namespace rta;

extern "C" {

const char *rta_global_error(void) { return get_global_error(); }

int rta_scene_load(const char *path, rta_scene **out, char *err, size_t errlen)
{
	if (!path)
		return RTA_E_INVALID;
	return make_scene(out, err, errlen, [&] { return load_scene(path); });
}

int rta_scene_parse(const char *name, const char *text, rta_scene **out, char *err, size_t errlen)
{
	if (!name || !text)
		return RTA_E_INVALID;
	return make_scene(out, err, errlen, [&] { return parse_scene(name, text); });
}

int rta_scene_from_triangles(const float *verts, int64_t n_tri, rta_scene **out)
{
	if (!verts || n_tri <= 0 || n_tri > 0x7ffffff0)
		return RTA_E_INVALID;
	return make_scene(out, nullptr, 0, [&] {
		HostScene *sc = new HostScene();
		sc->prim_type.reserve((size_t)n_tri);
		sc->geom.reserve((size_t)n_tri * 9);
		sc->normals.reserve((size_t)n_tri * 9);
		for (int64_t k = 0; k < n_tri; k++)
			sc->add_triangle(verts + k * 9, 0);
		return sc;
	});
}

void rta_scene_free(rta_scene *scene)
{
	if (!scene)
		return;
	delete scene->impl;
	delete scene;
}

int rta_scene_build_bvh(rta_scene *scene)
{
	if (!scene)
		return RTA_E_INVALID;
	try {
		scene->impl->build_bvh();
		return RTA_OK;
	} catch (const SceneError &e) {
		report(e, nullptr, 0);
		return e.code;
	} catch (const std::exception &e) {
		report(SceneError{RTA_E_INVALID, e.what()}, nullptr, 0);
		return RTA_E_INVALID;
	}
}

int rta_scene_get_settings(const rta_scene *scene, rta_scene_settings *out)
{
	if (!scene || !out)
		return RTA_E_INVALID;
	*out = scene->impl->settings;
	return RTA_OK;
}

int rta_scene_set_settings(rta_scene *scene, const rta_scene_settings *in)
{
	if (!scene || !in)
		return RTA_E_INVALID;
	scene->impl->settings = *in;
	return RTA_OK;
}

int rta_scene_describe(const rta_scene *scene, rta_scene_desc *out)
{
	if (!scene || !out)
		return RTA_E_INVALID;
	if (scene->impl->node_total() == 0) {
		set_global_error("rta_scene_describe: build the BVH first");
		return RTA_E_INVALID;
	}
	scene->impl->describe(out);
	return RTA_OK;
}

int rta_scene_permutation(const rta_scene *scene, const int32_t **perm, int32_t *n)
{
	if (!scene || !perm || !n)
		return RTA_E_INVALID;
	*perm = scene->impl->perm.data();
	*n = (int32_t)scene->impl->perm.size();
	return RTA_OK;
}

int rta_camera_from_settings(const rta_scene_settings *s, rta_camera *out)
{
	if (!s || !out || s->width <= 0 || s->height <= 0)
		return RTA_E_INVALID;
	camera_from_settings(*s, out);
	return RTA_OK;
}

int rta_write_bmp(const char *path, int32_t width, int32_t height, const uint8_t *bgr)
{
	if (!path || !bgr || width <= 0 || height <= 0)
		return RTA_E_INVALID;
	int rc = write_bmp(path, width, height, bgr);
	if (rc != RTA_OK)
		set_global_error(std::string("cannot write '") + path + "'");
	return rc;
}

}  // extern "C"
