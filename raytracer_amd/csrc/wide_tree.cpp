// Host-side builder of the 4-wide tree of the certified traversal.
//
// Topology: a binary tree with one triangle per leaf - up to kWideSahLimit triangles built by
// the surface-area heuristic (the walk's proof does not depend on the tree), beyond that the
// reference's tree (src/main.cpp:170-248) with every leaf of more than one primitive cut further
// (median of the centroids on the longest axis) - collapsed: a wide node starts with the two
// children of a binary node and replaces the inner child with the largest surface by its two
// children until it has four or none is left to open.  Spheres stay outside the tree
// (WideTree::spheres); the reference's tree always supplies the leaf boxes that certify an answer.
//
// Boxes: the exact bounds of the primitives' vertices, widened by a padding P
// that depends on the longest triangle edge E below the box.  The padding makes
// the wide tree CONSERVATIVE with respect to the reference's arithmetic: for a
// ray that the kernels admit (|origin - centre| <= reach, |direction| <= 2,
// components of ordinary magnitude) and a triangle that the reference's
// HitTriangle (src/main.cpp:250-279) accepts at distance t, the approximate slab
// test of traverse_wide.hpp finds every padded box around that triangle entered
// no later than t (1 + 2^-20) and left at a positive distance.  Derivation, with
// u = 2^-24, S >= |origin - any vertex|, D = the computed determinant (the test
// rejects |D| < 1e-4), all roundings to nearest, no contraction:
//   cross products carry an error of at most 4u |a||b|, dot products 3u sum|a_i b_i|,
//   so the three numerators are within 8u |d| E (S + E) (u, v) and 8u E^2 (S + |t d|)
//   (t) of the exact ones and D within 8u |d| E^2 of the exact determinant;
//   hence the exact intersection of the ray's line with the triangle's plane lies
//   within rho = 16.02e4 u |d| E^2 (S + E) + 2uE of the triangle, at a parameter
//   within dt = 16.02e4 u E^2 (S + E) + u t of the computed t.
// A box widened by rho + |d| dt (+ the roundings of the slab test itself and of
// the stored edge vectors, a few u of the coordinates) therefore does what is
// claimed; P is twice that.

#include "wide_tree.hpp"

#include <algorithm>
#include <iterator>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace rta {

namespace {

struct BNode {
	float lo[3], hi[3];
	int32_t left, right;      // children, or -1
	int32_t prim;             // >= 0: a primitive
	float edge;               // longest edge vector of a triangle below
};

inline float as_float(uint32_t u)
{
	float f;
	memcpy(&f, &u, 4);
	return f;
}

struct Builder {
	const rta_scene_desc &d;
	std::vector<BNode> b;
	std::vector<int32_t> ids;
	std::vector<float> cent;      // centroids of the primitives of the leaf being cut

	explicit Builder(const rta_scene_desc &desc) : d(desc) {}

	int32_t leaf(int32_t prim)
	{
		const float *g = d.prim_geom + (size_t)prim * 9;
		BNode n;
		for (int k = 0; k < 3; k++) {
			n.lo[k] = std::min(g[k], std::min(g[3 + k], g[6 + k]));
			n.hi[k] = std::max(g[k], std::max(g[3 + k], g[6 + k]));
		}
		// the edge vectors as the kernels hold them: v0 - v2 and v1 - v2 rounded to binary32
		double e0 = 0, e1 = 0;
		for (int k = 0; k < 3; k++) {
			const float a = g[k] - g[6 + k], c = g[3 + k] - g[6 + k];
			e0 += (double)a * a;
			e1 += (double)c * c;
		}
		n.edge = (float)(std::sqrt(std::max(e0, e1)) * (1 + 1e-6));
		n.left = n.right = -1;
		n.prim = prim;
		b.push_back(n);
		return (int32_t)b.size() - 1;
	}

	int32_t join(int32_t l, int32_t r)
	{
		BNode n;
		for (int k = 0; k < 3; k++) {
			n.lo[k] = std::min(b[l].lo[k], b[r].lo[k]);
			n.hi[k] = std::max(b[l].hi[k], b[r].hi[k]);
		}
		n.edge = std::max(b[l].edge, b[r].edge);
		n.left = l;
		n.right = r;
		n.prim = -1;
		b.push_back(n);
		return (int32_t)b.size() - 1;
	}

	// ids[begin, end): primitives of one reference leaf
	int32_t cut(int32_t begin, int32_t end)
	{
		if (end - begin == 1)
			return leaf(ids[begin]);
		float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
		auto centroid = [&](int32_t p, int k) {
			const float *g = d.prim_geom + (size_t)p * 9;
			return (g[k] + g[3 + k] + g[6 + k]) * (1.f / 3.f);
		};
		for (int32_t i = begin; i < end; i++)
			for (int k = 0; k < 3; k++) {
				const float c = centroid(ids[i], k);
				lo[k] = std::min(lo[k], c);
				hi[k] = std::max(hi[k], c);
			}
		int axis = 0;
		for (int k = 1; k < 3; k++)
			if (hi[k] - lo[k] > hi[axis] - lo[axis])
				axis = k;
		const int32_t mid = begin + (end - begin) / 2;
		std::nth_element(ids.begin() + begin, ids.begin() + mid, ids.begin() + end, [&](int32_t x, int32_t y) {
			const float cx = centroid(x, axis), cy = centroid(y, axis);
			return cx < cy || (cx == cy && x < y);
		});
		const int32_t l = cut(begin, mid), r = cut(mid, end);
		return join(l, r);
	}

	// Surface-area heuristic over ids[begin, end) (16 bins per axis on the centroids): the proof of the
	// walk does not depend on the tree, so the wide tree need not keep the reference's midpoint splits.
	int32_t sah(int32_t begin, int32_t end)
	{
		if (end - begin == 1)
			return leaf(ids[begin]);
		auto centroid = [&](int32_t p, int k) {
			const float *g = d.prim_geom + (size_t)p * 9;
			return ((double)g[k] + g[3 + k] + g[6 + k]) * (1.0 / 3.0);
		};
		auto bounds_of = [&](int32_t p, double *lo, double *hi) {
			const float *g = d.prim_geom + (size_t)p * 9;
			for (int k = 0; k < 3; k++) {
				lo[k] = std::min(lo[k], (double)std::min(g[k], std::min(g[3 + k], g[6 + k])));
				hi[k] = std::max(hi[k], (double)std::max(g[k], std::max(g[3 + k], g[6 + k])));
			}
		};
		auto area = [](const double *lo, const double *hi) {
			const double x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
			return x * y + y * z + z * x;
		};
		double clo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, chi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
		for (int32_t i = begin; i < end; i++)
			for (int k = 0; k < 3; k++) {
				const double c = centroid(ids[i], k);
				clo[k] = std::min(clo[k], c);
				chi[k] = std::max(chi[k], c);
			}
		constexpr int kBins = 16;
		int best_axis = -1, best_bin = 0;
		double best_cost = DBL_MAX;
		for (int axis = 0; axis < 3; axis++) {
			const double extent = chi[axis] - clo[axis];
			if (!(extent > 0))
				continue;
			int count[kBins] = {};
			double blo[kBins][3], bhi[kBins][3];
			for (int b2 = 0; b2 < kBins; b2++)
				for (int k = 0; k < 3; k++) {
					blo[b2][k] = DBL_MAX;
					bhi[b2][k] = -DBL_MAX;
				}
			for (int32_t i = begin; i < end; i++) {
				int b2 = (int)((centroid(ids[i], axis) - clo[axis]) / extent * kBins);
				b2 = b2 < 0 ? 0 : (b2 >= kBins ? kBins - 1 : b2);
				count[b2]++;
				bounds_of(ids[i], blo[b2], bhi[b2]);
			}
			// areas of the prefixes and suffixes of the bins
			double right_area[kBins];
			int right_count[kBins];
			double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
			int n = 0;
			for (int b2 = kBins - 1; b2 > 0; b2--) {
				n += count[b2];
				for (int k = 0; k < 3; k++) {
					lo[k] = std::min(lo[k], blo[b2][k]);
					hi[k] = std::max(hi[k], bhi[b2][k]);
				}
				right_count[b2] = n;
				right_area[b2] = n ? area(lo, hi) : 0;
			}
			for (int k = 0; k < 3; k++) {
				lo[k] = DBL_MAX;
				hi[k] = -DBL_MAX;
			}
			n = 0;
			for (int b2 = 0; b2 + 1 < kBins; b2++) {
				n += count[b2];
				for (int k = 0; k < 3; k++) {
					lo[k] = std::min(lo[k], blo[b2][k]);
					hi[k] = std::max(hi[k], bhi[b2][k]);
				}
				if (n == 0 || right_count[b2 + 1] == 0)
					continue;
				const double cost = area(lo, hi) * n + right_area[b2 + 1] * right_count[b2 + 1];
				if (cost < best_cost) {
					best_cost = cost;
					best_axis = axis;
					best_bin = b2;
				}
			}
		}
		int32_t mid;
		if (best_axis < 0) {
			mid = begin + (end - begin) / 2;          // all centroids in one point
		} else {
			const double extent = chi[best_axis] - clo[best_axis];
			auto left = [&](int32_t p) {
				int b2 = (int)((centroid(p, best_axis) - clo[best_axis]) / extent * kBins);
				b2 = b2 < 0 ? 0 : (b2 >= kBins ? kBins - 1 : b2);
				return b2 <= best_bin;
			};
			mid = (int32_t)(std::partition(ids.begin() + begin, ids.begin() + end, left) - ids.begin());
			if (mid == begin || mid == end)
				mid = begin + (end - begin) / 2;
		}
		const int32_t l = sah(begin, mid), r = sah(mid, end);
		return join(l, r);
	}

	int32_t from_reference(int32_t node)
	{
		const int32_t idx = d.node_index[node], cnt = d.node_count[node];
		if (cnt > 0) {
			ids.resize((size_t)cnt);
			for (int32_t k = 0; k < cnt; k++)
				ids[k] = idx + k;
			return cut(0, cnt);
		}
		const int32_t l = from_reference(idx), r = from_reference(idx + 1);
		return join(l, r);
	}
};

inline double surface(const BNode &n)
{
	const double x = (double)n.hi[0] - n.lo[0], y = (double)n.hi[1] - n.lo[1], z = (double)n.hi[2] - n.lo[2];
	return x * y + y * z + z * x;
}

// binary16 neighbours of x, as bit patterns
inline uint16_t half_bits(_Float16 h)
{
	uint16_t b;
	memcpy(&b, &h, 2);
	return b;
}

inline _Float16 half_of(uint16_t b)
{
	_Float16 h;
	memcpy(&h, &b, 2);
	return h;
}

inline uint16_t half_below(double x)
{
	_Float16 h = (_Float16)x;
	if ((double)h <= x)
		return half_bits(h);
	uint16_t b = half_bits(h);
	if ((b & 0x7fffu) == 0)
		return 0x8001u;                   // below zero: the smallest negative subnormal
	return (b & 0x8000u) ? (uint16_t)(b + 1) : (uint16_t)(b - 1);
}

inline uint16_t half_above(double x)
{
	_Float16 h = (_Float16)x;
	if ((double)h >= x)
		return half_bits(h);
	uint16_t b = half_bits(h);
	if ((b & 0x7fffu) == 0)
		return 0x0001u;
	return (b & 0x8000u) ? (uint16_t)(b - 1) : (uint16_t)(b + 1);
}

inline float below(double x)
{
	float f = (float)x;
	return (double)f <= x ? f : std::nextafterf(f, -INFINITY);
}

}  // namespace

WideTree build_wide_tree(const rta_scene_desc &d, bool tame)
{
	WideTree w;
	auto refuse = [&](const char *why) {
		w.enabled = false;
		w.why_not = why;
		w.nodes.clear();
		return w;
	};
	if (!tame)
		return refuse("box coordinates out of the ordinary range");
	if (d.node_count[0] > 0)
		return refuse("the root is a leaf");
	if (d.n_prims >= (1 << 26))
		return refuse("more primitives than a stack entry can number");

	// every primitive's reference leaf, spheres, and nesting of the reference's boxes: a child's
	// box must lie inside its parent's (the root's own box is never tested, src/main.cpp:338-385)
	w.leaf_boxes.assign((size_t)d.n_prims * 2, make_float4(0, 0, 0, 0));
	std::vector<int32_t> covered;      // the primitives that some leaf holds
	bool too_many_spheres = false;
	{
		std::vector<int32_t> todo = {0};
		while (!todo.empty()) {
			const int32_t node = todo.back();
			todo.pop_back();
			const int32_t idx = d.node_index[node], cnt = d.node_count[node];
			const float *nb = d.node_bounds + (size_t)node * 6;
			if (cnt > 0) {
				for (int32_t p = idx; p < idx + cnt; p++) {
					const float *g = d.prim_geom + (size_t)p * 9;
					w.leaf_boxes[(size_t)p * 2] = make_float4(nb[0], nb[1], nb[2], 1.f);      // (1: the primitive is in a leaf)
					w.leaf_boxes[(size_t)p * 2 + 1] = make_float4(nb[3], nb[4], nb[5], 0);
					if (d.prim_type[p] != RTA_PRIM_TRIANGLE) {
						// a few spheres stay outside the tree and are tested for every ray
						if (w.spheres.size() == kWideMaxSpheres)
							too_many_spheres = true;      // (the boxes are still wanted: a small scene is scanned, spheres and all)
						else
							w.spheres.push_back(p);
						continue;
					}
					for (int k = 0; k < 9; k++)
						if (!std::isfinite(g[k]))
							return refuse("non-finite vertex");
					covered.push_back(p);
				}
				continue;
			}
			for (int c = 0; c < 2; c++) {
				const float *cb = d.node_bounds + (size_t)(idx + c) * 6;
				if (node != 0)
					for (int k = 0; k < 3; k++)
						if (!(cb[k] >= nb[k] && cb[3 + k] <= nb[3 + k]))
							return refuse("a child's box reaches outside its parent's");
				todo.push_back(idx + c);
			}
		}
	}

	// the leaf boxes are complete and nested, the spheres few: answers can be certified (step 3 of the walk's
	// proof) whether or not the scene gets a wide tree
	w.boxes_complete = true;
	if (too_many_spheres)
		return refuse("more spheres than the walk tests one by one");
	w.certifiable = true;
	if (covered.size() < 2)
		return refuse("fewer than two triangles");

	Builder bu(d);
	bu.b.reserve((size_t)d.n_prims * 2);
	// Topology: by surface-area heuristic where the build stays short, else the reference's own.  (Both
	// put every covered primitive in exactly one leaf; the reference's tree always supplies the leaf
	// boxes of step 3 of the proof.)
	int32_t root;
	if (!w.spheres.empty() && covered.size() > kWideSahLimit)
		return refuse("spheres in a scene beyond the size of the heuristic build");      // (the reference's leaves mix both kinds)
	if (covered.size() <= kWideSahLimit && (!getenv("RTA_WIDE_MIDPOINT") || !w.spheres.empty())) {
		bu.ids = covered;
		root = bu.sah(0, (int32_t)covered.size());
	} else {
		root = bu.from_reference(0);
	}
	const std::vector<BNode> &b = bu.b;

	// the scene's sphere
	double c[3], r2 = 0, cmax = 0;
	for (int k = 0; k < 3; k++) {
		c[k] = .5 * ((double)b[root].lo[k] + b[root].hi[k]);
		const double h = .5 * ((double)b[root].hi[k] - b[root].lo[k]);
		r2 += h * h;
		cmax = std::max(cmax, std::fabs(c[k]));
	}
	const double radius = std::sqrt(r2);
	if (!(radius > 0) || !std::isfinite(radius) || !std::isfinite(cmax))
		return refuse("degenerate scene box");
	for (int k = 0; k < 3; k++)
		w.center[k] = (float)c[k];

	// The admitted rays: the padding grows with the distance of the farthest admitted origin, so the
	// wider reach (rays from outside the scene, a camera's say) is taken only where the boxes stay tight.
	std::vector<double> candidates(std::begin(kWideReachRadii), std::end(kWideReachRadii));
	if (getenv("RTA_WIDE_REACH"))
		candidates = {atof(getenv("RTA_WIDE_REACH"))};
	for (size_t trial = 0; trial < candidates.size(); trial++) {
		const double reach_radii = candidates[trial];
		const double s_max = reach_radii * radius;
		w.reach = below((reach_radii - 1.001) * radius - 1e-6 * cmax);
		w.span = (float)(s_max * 1.001);
		if (!(w.reach > 0))
			return refuse("degenerate scene box");
		w.nodes.clear();

		const double u = 0x1p-24;
		auto pad = [&](double e) {
			const double rho = 16.02e4 * u * kWideMaxDir * e * e * (s_max + e) + 2 * u * e;
			const double dt = 16.02e4 * u * e * e * (s_max + e);
			const double slab = 8 * u * (s_max + cmax + radius);
			return 2 * (rho + kWideMaxDir * dt + slab);
		};

		// collapse, depth first; recursion depth = levels of wide nodes
		struct Emit {
			const std::vector<BNode> &b;
			WideTree &w;
			decltype(pad) &pad_of;
			double widened_sum = 0;       // sum over children of (mean widening per side) / (mean extent)
			uint64_t widened_n = 0;
			int depth = 0;
			bool overflow = false;        // a plane beyond binary16's range

			uint32_t node(int32_t at, int level)
			{
				depth = std::max(depth, level);
				const uint32_t me = (uint32_t)(w.nodes.size() / 4);
				w.nodes.resize(w.nodes.size() + 4, make_float4(0, 0, 0, 0));
				int32_t kids[4] = {b[at].left, b[at].right, -1, -1};
				int n = 2;
				while (n < 4) {
					int best = -1;
					double area = -1;
					for (int k = 0; k < n; k++)
						if (b[kids[k]].prim < 0 && surface(b[kids[k]]) > area) {
							area = surface(b[kids[k]]);
							best = k;
						}
					if (best < 0)
						break;
					const int32_t open = kids[best];
					for (int k = n; k > best + 1; k--)
						kids[k] = kids[k - 1];
					kids[best] = b[open].left;
					kids[best + 1] = b[open].right;
					n++;
				}
				// planes relative to the centre the kernels subtract from the ray's origin, as binary16
				// rounded outwards
				uint16_t plane[6][4];
				uint32_t ref[4];
				for (int k = 0; k < 4; k++) {
					if (k >= n) {
						for (int a = 0; a < 3; a++) {
							plane[a][k] = 0x7c00u;          // +inf
							plane[3 + a][k] = 0xfc00u;      // -inf
						}
						ref[k] = kWideNone;
						continue;
					}
					const BNode &kid = b[kids[k]];
					const double p = pad_of(kid.edge);
					double extent = 0, widened = 0;
					for (int a = 0; a < 3; a++) {
						plane[a][k] = half_below((double)kid.lo[a] - p - (double)w.center[a]);
						plane[3 + a][k] = half_above((double)kid.hi[a] + p - (double)w.center[a]);
						const double lo_h = (double)half_of(plane[a][k]), hi_h = (double)half_of(plane[3 + a][k]);
						if (!std::isfinite(lo_h) || !std::isfinite(hi_h))
							overflow = true;
						extent += (hi_h - lo_h) / 3;
						widened += ((hi_h - lo_h) - ((double)kid.hi[a] - kid.lo[a])) / 6;
					}
					if (extent > 0) {
						widened_sum += widened / extent;
						widened_n++;
					}
					ref[k] = kid.prim >= 0 ? (kWideLeaf | (uint32_t)kid.prim) : node(kids[k], level + 1);
				}
				float4 *rec = &w.nodes[(size_t)me * 4];
				uint32_t words[12];
				for (int a = 0; a < 6; a++) {
					words[a * 2] = (uint32_t)plane[a][0] | ((uint32_t)plane[a][1] << 16);
					words[a * 2 + 1] = (uint32_t)plane[a][2] | ((uint32_t)plane[a][3] << 16);
				}
				for (int q = 0; q < 3; q++)
					rec[q] = make_float4(as_float(words[q * 4]), as_float(words[q * 4 + 1]), as_float(words[q * 4 + 2]),
					                     as_float(words[q * 4 + 3]));
				rec[3] = make_float4(as_float(ref[0]), as_float(ref[1]), as_float(ref[2]), as_float(ref[3]));
				return me;
			}
		} emit{b, w, pad};
		emit.node(root, 1);
		if (emit.overflow)
			return refuse("coordinates beyond the range of the wide tree's binary16 planes");
		w.depth = emit.depth;
		w.n_nodes = (uint32_t)(w.nodes.size() / 4);
		w.index_mask = 1;
		while (w.index_mask < (uint32_t)d.n_prims || w.index_mask < w.n_nodes)
			w.index_mask = w.index_mask * 2 + 1;
		w.mean_pad_over_extent = emit.widened_n ? emit.widened_sum / (double)emit.widened_n : 0;
		if (w.mean_pad_over_extent <= kWideMaxPadding)
			break;
	}
	if (w.mean_pad_over_extent > kWideHopelessPadding)
		return refuse("triangles too large for the padding the walk needs");
	w.enabled = true;
	return w;
}

}  // namespace rta
