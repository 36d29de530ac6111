// Closest hit of a pinhole camera's ray from the triangle list of its cell on the image plane
// (cells.hip), certified as in traverse_wide.hpp.
//
// The list holds every triangle that the reference's HitTriangle (src/main.cpp:250-279) can accept
// for a ray through the cell, so testing the list - with the reference's own arithmetic - does
// what step 1 of the walk's proof asks of the walk: every accepted triangle that can be as near
// as the answer is tested.  Steps 2
// and 3 (no tie; the hit triangle's leaf box passes the reference's box test) are the walk's own
// certify().  A ray without a cell, or one the lists were not made for, is left in doubt like a
// ray the walk gives up on.
#pragma once

#include "traverse_wide.hpp"

namespace rta {

// Must be reached by every lane of the wave (`active`: the lane holds a ray).
template <bool COUNT>
RTA_DEV void closest_from_cell(const DeviceTree &tree, const bool active, const Ray &ray, const int32_t cell,
                               const uint32_t *cell_start, const uint32_t *cell_list, const unsigned long long *cell_keys,
                               HitRec &hit, bool &doubt, WorkCount &work)
{
	WideTraversal<COUNT, false> tr;
	tr.rest(0);
	uint32_t k = 0, end = 0;
	if (active) {
		tr.ray = ray;
		tr.rcp = mk(1.f / ray.d.x, 1.f / ray.d.y, 1.f / ray.d.z);
		// the rays the lists were made for (cells.hip); NaNs fail the comparison
		tr.doubt = !(ray_is_tame(ray) && norm2(ray.d) <= 1.0002f) || cell < 0;
		// the spheres are in no list: each is tested, with the reference's own arithmetic
		if (!tr.doubt)
			for (uint32_t k = 0; k < tree.n_loose; k++) {
				const uint32_t i = ((ConstWords)tree.loose)[k];
				const ConstQuads rec = (ConstQuads)tree.prims + kPrimQuads * (size_t)i;
				const Quad p0 = rec[0], p1 = rec[1];
				const float t = test_sphere(mk(p0.x, p0.y, p0.z), p1.x, ray);
				if (t < tr.best) {
					tr.best = t;
					tr.prim = (int32_t)i;
					tr.tie = false;
				} else if (t == tr.best && t < FLT_MAX) {
					tr.tie = true;
				}
			}
		if (!tr.doubt) {
			k = cell_start[cell];
			end = cell_start[cell + 1];
		}
	}
	// One triangle per lane and trip, for as long as some lane has one left.  The entries come nearest first
	// by the least distance at which each can be hit at all (the low word of its key): once that lies beyond
	// the nearest hit so far, the rest of the list cannot change the answer nor tie with it.
	// (an entry is fetched one trip ahead, together with the triangle of the trip before: one wait per trip, not two)
	uint32_t next_prim = 0;
	float next_bound = 0.f;
	if (k < end) {
		next_bound = __uint_as_float((uint32_t)cell_keys[k]);
		next_prim = cell_list[k];
	}
	while (__ballot(k < end)) {
		if (k < end && next_bound > tr.best)
			k = end;
		if (k < end) {
			const uint32_t i = next_prim;
			if (++k < end) {
				next_bound = __uint_as_float((uint32_t)cell_keys[k]);
				next_prim = cell_list[k];
			}
			if (COUNT) work.tri++;
			const float4 *rec = tree.prims + kPrimQuads * (size_t)i;
			const float4 p0 = rec[0], p1 = rec[1], p2 = rec[2];
			float tu = 0, tv = 0;
			const float t = test_triangle(mk(p0.x, p0.y, p0.z), mk(p1.x, p1.y, p1.z), mk(p2.x, p2.y, p2.z), tr.ray, tu, tv);
			if (t < tr.best) {
				tr.best = t;
				tr.prim = (int32_t)i;
				tr.u = tu;
				tr.v = tv;
				tr.tie = false;
			} else if (t == tr.best && t < FLT_MAX) {
				tr.tie = true;
			}
		}
	}
	if (active)
		tr.certify(tree);
	hit = tr.result();
	doubt = tr.doubt != 0;
}

}  // namespace rta
