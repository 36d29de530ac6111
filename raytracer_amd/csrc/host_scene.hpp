// Host-side scene: file grammar, BVH construction, flat views for upload.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "raytracer_amd.h"

namespace rta {

struct SceneError {
	int code;
	std::string text;
};

// Structure-of-arrays scene.  Primitive k occupies prim_type[k],
// prim_material[k], prim_light[k], geom[9k..9k+9), normals[9k..9k+9).
struct HostScene {
	rta_scene_settings settings;

	std::vector<int32_t> prim_type, prim_material, prim_light;
	std::vector<float> geom, normals;
	std::vector<float> materials;      // 6 per material
	std::vector<float> light_color;    // 3 per light
	std::vector<int32_t> light_prim;

	std::vector<float> node_bounds;    // 6 per node
	std::vector<int32_t> node_index, node_count;
	std::vector<int32_t> perm;         // BVH-order position -> file-order position

	int32_t prim_count() const { return (int32_t)prim_type.size(); }
	int32_t node_total() const { return (int32_t)node_index.size(); }

	HostScene();
	void add_triangle(const float v[9], int32_t material);
	void add_sphere(const float center[3], float radius, int32_t material);
	void build_bvh();
	// Takes a tree built elsewhere for the primitives in their current (file)
	// order and finishes as build_bvh() does: rows into BVH order, lights re-linked.
	void adopt_tree(std::vector<int32_t> &&order, std::vector<float> &&bounds, std::vector<int32_t> &&index,
	                std::vector<int32_t> &&count);
	void describe(rta_scene_desc *out) const;
};

// Throws SceneError.
HostScene *parse_scene(const std::string &name, const std::string &text);
HostScene *load_scene(const std::string &path);

void camera_from_settings(const rta_scene_settings &s, rta_camera *out);
int write_bmp(const char *path, int32_t w, int32_t h, const uint8_t *bgr);

}  // namespace rta

// The opaque handle of include/raytracer_amd.h.
struct rta_scene {
	rta::HostScene *impl;
};
