// Per-cell triangle lists for a pinhole camera's rays.
//
// All camera rays start at one point, so which triangles a camera ray can hit is decided by where
// it crosses the image plane.  Once per view, every triangle that some leaf of the reference's tree
// holds is drawn onto the pixel-sized cells of the image plane (the frame and kSkyBorder cells around
// it) - its projected outline widened by what the reference's arithmetic can add - and every cell
// gets the list of the triangles that touch it.  A camera ray is then answered by testing the
// triangles of its cell with the reference's own HitTriangle arithmetic (src/main.cpp:250-279), no
// tree involved, and certified as in traverse_wide.hpp: the list takes the place of step 1 of the
// proof there.
//
// Why the list is complete.  Let HitTriangle accept triangle p for a camera ray: tame components,
// |d| <= dmax = 1.00011 (the rays are normalised), computed determinant D with |D| >= 1e-4 (the test
// rejects smaller ones).  With u = 2^-24, E = the longer of p's stored edge vectors and
// S >= |o - vertex| for p's vertices, the test's three numerators are within 8u |d| E (S + E) resp.
// 8u E^2 (S + |t d|) of their exact values and D within delta = 8u |d| E^2 of the exact determinant
// D* = -d . (e0 x e1) (the roundings are counted in wide_tree.cpp), hence the exact intersection X
// of the ray's line with p's plane lies within
//     rho = 16.02 u dmax E^2 (S + E) / |D| + 2uE
// of p.  The walk's boxes have to take |D| = 1e-4, the worst any ray can do; here the ray's cell is
// known, and with it a lower bound of |D*|: d . n is q . n / |q| for the point q where the ray crosses
// the image plane, q . n is linear in q and |q| convex, so over a cell (widened by the margin below)
// |D*| >= min over its corners |q . n| / max over its corners |q| if q . n keeps its sign, and
// |D| >= max(1e-4, that (1 - 1e-6) - delta).  A wall seen from in front gets a rho of a thousandth of a
// pixel instead of a third of the room.
// X lies on the ray's line, in front of the camera (every point within rho of p is, checked below),
// so X is seen at the ray's own point on the image plane.  Seen from the pinhole, a point within rho
// of a point P of the triangle lands within (rx, ry) pixels of P's image, where with
// x = h a / z - c (a = (P - o).du / du2, z = (P - o).n, see Projection)
//     rx = rho (|h| / |du| + max |x + c| |n|) / (min |z| - rho |n|)
// and ry likewise: the maxima and minima over the triangle are taken at its vertices.  The ray's
// point on the image plane is where its sample was aimed, (pu + .5 + ju, pv + .5 + jv) as k_advance
// computes it, up to the binary32 roundings of the target point and of the normalised direction:
// orders of magnitude below Projection::margin (see projection_of in context.cpp).  A cell gets the
// triangle unless some edge of the projected triangle, moved outwards by sqrt(rx^2 + ry^2) + margin,
// has the whole cell on its far side (or the cell lies outside the bounding rectangle moved likewise).
//
// Nearest first.  Every entry carries a lower bound of the distance at which the test can accept its
// triangle for a ray through the cell, and a cell's entries are sorted by it, so that the tests can stop at
// the first entry whose bound lies beyond the nearest hit so far.  The ray's parameter at X is |X - o| / |d|,
// and X lies at least (depth - rho) sec(theta) from the pinhole: depth = the least distance of the triangle's
// vertices from the pinhole along the camera's axis, theta = the least angle to that axis of any ray through
// the cell; the computed t lies within dt = 16.02 u E^2 (S + E) / |D| + u t of the parameter at X.
#include "kernels.hpp"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace rta {

namespace {

constexpr int kCellBlock = 256;
constexpr double kCameraMaxDir = 1.00011;      // k_camera_cells admits |d|^2 <= 1.0002

struct Outline {
	double x[3], y[3];      // projected vertices, pixel coordinates
	bool edges;             // the three edge tests apply (a proper triangle on the image plane)
	double mx[3], my[3];    // outward unit normals of the edges k -> k + 1
	double bx0, bx1, by0, by1;     // the projected vertices' bounding rectangle
	double nrm[3];          // e0 x e1
	double rho_num;         // rho = rho_num / |D| + two_ue
	double two_ue;
	double delta;           // the computed determinant lies within it of the exact one
	double dt_num;          // dt = dt_num / |D| (+ u t)
	double kx, ky;          // rx = rho kx / (z_min - rho n_len), ry likewise
	double z_min, n_len;    // z_min = 0: no statement about this triangle (a vertex is not safely in front of the camera): it joins every cell
	double ppx, ppy;        // where the camera's axis meets the image plane, pixel coordinates
	double tan2_x, tan2_y;  // (a pixel's width, height / the image plane's distance)^2
	int i0, i1, j0, j1;     // cells that the outline can touch at its widest, clipped to the map (empty when i1 < i0 or j1 < j0)
};

__device__ Outline outline_of(const TreeRef &tree, const uint32_t prim, const Projection &pr, const int map_w, const int map_h)
{
	Outline t;
	const float4 *rec = tree.prims + kPrimQuads * (size_t)prim;
	const float4 p0 = rec[0], p1 = rec[1], p2 = rec[2];
	// the triangle HitTriangle works on: v2, v2 + (v0 - v2), v2 + (v1 - v2) with the stored edge vectors
	const double v[3][3] = {{(double)p0.x + p1.x, (double)p0.y + p1.y, (double)p0.z + p1.z},
	                        {(double)p0.x + p2.x, (double)p0.y + p2.y, (double)p0.z + p2.z},
	                        {(double)p0.x, (double)p0.y, (double)p0.z}};
	const double e0[3] = {p1.x, p1.y, p1.z}, e1[3] = {p2.x, p2.y, p2.z};
	const double l0 = sqrt(e0[0] * e0[0] + e0[1] * e0[1] + e0[2] * e0[2]), l1 = sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
	const double edge = (l0 > l1 ? l0 : l1) * (1 + 1e-6);
	t.nrm[0] = e0[1] * e1[2] - e0[2] * e1[1];
	t.nrm[1] = e0[2] * e1[0] - e0[0] * e1[2];
	t.nrm[2] = e0[0] * e1[1] - e0[1] * e1[0];
	const double u = 0x1p-24;
	double n_len = 0, du_len = sqrt(pr.du2), dv_len = sqrt(pr.dv2), cx = 0, cy = 0;
	for (int k = 0; k < 3; k++) {
		n_len += pr.n[k] * pr.n[k];
		cx += (pr.ul[k] - pr.o[k]) * pr.du[k];
		cy += (pr.ul[k] - pr.o[k]) * pr.dv[k];
	}
	n_len = sqrt(n_len);
	cx /= pr.du2;
	cy /= pr.dv2;
	const double h_abs = pr.h < 0 ? -pr.h : pr.h;
	bool bad_vertex = false;
	double z_min = 1e300, xc_max = 0, yc_max = 0, far = 0;
	for (int c = 0; c < 3; c++) {
		double w[3], z = 0, w2 = 0;
		for (int k = 0; k < 3; k++) {
			w[k] = v[c][k] - pr.o[k];
			z += w[k] * pr.n[k];
			w2 += w[k] * w[k];
		}
		far = w2 > far ? w2 : far;
		const double scale = pr.h / z;
		if (!(scale > 0) || !(scale < 1e6))
			bad_vertex = true;
		double x = 0, y = 0;
		for (int k = 0; k < 3; k++) {
			const double p = pr.o[k] + w[k] * scale - pr.ul[k];
			x += p * pr.du[k];
			y += p * pr.dv[k];
		}
		x /= pr.du2;
		y /= pr.dv2;
		if (!(x > -1e9 && x < 1e9 && y > -1e9 && y < 1e9))
			bad_vertex = true;
		t.x[c] = x;
		t.y[c] = y;
		const double za = z < 0 ? -z : z, xa = x + cx < 0 ? -(x + cx) : x + cx, ya = y + cy < 0 ? -(y + cy) : y + cy;
		z_min = za < z_min ? za : z_min;
		xc_max = xa > xc_max ? xa : xc_max;
		yc_max = ya > yc_max ? ya : yc_max;
	}
	const double reach = sqrt(far) * (1 + 1e-9) + edge;      // S + E
	t.rho_num = 16.02 * u * kCameraMaxDir * edge * edge * reach * (1 + 1e-9);
	t.two_ue = 2 * u * edge;
	t.delta = 8 * u * kCameraMaxDir * edge * edge * (1 + 1e-9);
	t.dt_num = 16.02 * u * edge * edge * reach * (1 + 1e-9);
	t.kx = h_abs / du_len + xc_max * n_len;
	t.ky = h_abs / dv_len + yc_max * n_len;
	t.z_min = z_min;
	t.n_len = n_len;
	t.ppx = -cx;
	t.ppy = -cy;
	const double plane = h_abs / n_len;
	t.tan2_x = pr.du2 / (plane * plane);
	t.tan2_y = pr.dv2 / (plane * plane);
	t.bx0 = fmin(t.x[0], fmin(t.x[1], t.x[2]));
	t.bx1 = fmax(t.x[0], fmax(t.x[1], t.x[2]));
	t.by0 = fmin(t.y[0], fmin(t.y[1], t.y[2]));
	t.by1 = fmax(t.y[0], fmax(t.y[1], t.y[2]));
	t.edges = false;
	double x0 = -1e9, x1 = 1e9, y0 = -1e9, y1 = 1e9;      // (the whole map)
	if (bad_vertex) {
		t.z_min = 0;
	} else {
		// the widest the outline can get: at the least determinant the test lets pass
		const double rho = t.rho_num * 1e4 + t.two_ue;
		if (rho >= 0 && rho < 1e30 && z_min > 2 * rho * n_len) {
			const double rx = rho * t.kx / (z_min - rho * n_len), ry = rho * t.ky / (z_min - rho * n_len);
			// (and by more than the rounding of these few binary64 operations)
			const double widest = sqrt(rx * rx + ry * ry) * (1 + 1e-9) + pr.margin + 1e-9 * (fabs(t.bx0) + fabs(t.bx1) + fabs(t.by0) + fabs(t.by1) + 1);
			if (widest < 1e9)
				x0 = t.bx0 - widest, x1 = t.bx1 + widest, y0 = t.by0 - widest, y1 = t.by1 + widest;
		}
		const double area2 = (t.x[1] - t.x[0]) * (t.y[2] - t.y[0]) - (t.x[2] - t.x[0]) * (t.y[1] - t.y[0]);
		const double extent = (t.bx1 - t.bx0) + (t.by1 - t.by0);
		// a sliver has no usable edge normals: the rectangle alone serves it
		if (fabs(area2) > 1e-9 * extent * extent && fabs(area2) > 1e-12) {
			t.edges = true;
			const double turn = area2 > 0 ? 1.0 : -1.0;
			for (int k = 0; k < 3; k++) {
				const int k1 = k == 2 ? 0 : k + 1;
				const double ex = t.x[k1] - t.x[k], ey = t.y[k1] - t.y[k];
				const double len = sqrt(ex * ex + ey * ey);
				if (!(len > 0)) {
					t.edges = false;
					break;
				}
				t.mx[k] = turn * ey / len;
				t.my[k] = -turn * ex / len;
			}
		}
	}
	// cells [i, i + 1] x [j, j + 1] of the map, whose cell (kSkyBorder, kSkyBorder) is pixel (0, 0)
	const double lim = 1e8;
	const long long i0 = (long long)floor(x0 < -lim ? -lim : x0) + kSkyBorder, i1 = (long long)floor(x1 > lim ? lim : x1) + kSkyBorder;
	const long long j0 = (long long)floor(y0 < -lim ? -lim : y0) + kSkyBorder, j1 = (long long)floor(y1 > lim ? lim : y1) + kSkyBorder;
	t.i0 = (int)(i0 < 0 ? 0 : i0), t.i1 = (int)(i1 > map_w - 1 ? map_w - 1 : i1);
	t.j0 = (int)(j0 < 0 ? 0 : j0), t.j1 = (int)(j1 > map_h - 1 ? map_h - 1 : j1);
	return t;
}

// May a camera ray through cell (ci, cj) of the map be one that the triangle's test accepts?  If so, `nearest`
// is a lower bound of the t the test can return for such a ray.
__device__ bool touches(const Outline &t, const Projection &pr, const int ci, const int cj, float &nearest)
{
	nearest = 0.f;
	if (!(t.z_min > 0))
		return true;
	// the least |determinant| of a ray through the cell, widened by the margin
	const double cx0 = (double)(ci - kSkyBorder) - pr.margin, cx1 = (double)(ci - kSkyBorder) + 1 + pr.margin;
	const double cy0 = (double)(cj - kSkyBorder) - pr.margin, cy1 = (double)(cj - kSkyBorder) + 1 + pr.margin;
	double g_least = 1e300, q_most = 0;
	int positive = 0, negative = 0;
	for (int c = 0; c < 4; c++) {
		const double x = (c & 1) ? cx1 : cx0, y = (c & 2) ? cy1 : cy0;
		double g = 0, q2 = 0;
		for (int k = 0; k < 3; k++) {
			const double q = (pr.ul[k] - pr.o[k]) + x * pr.du[k] + y * pr.dv[k];
			g += q * t.nrm[k];
			q2 += q * q;
		}
		positive += g > 0;
		negative += g < 0;
		g = g < 0 ? -g : g;
		g_least = g < g_least ? g : g_least;
		q_most = q2 > q_most ? q2 : q_most;
	}
	double det = 1e-4;
	if ((positive == 4 || negative == 4) && q_most > 0) {
		const double seen = g_least / sqrt(q_most) * (1 - 1e-6) - t.delta;
		det = seen > det ? seen : det;
	}
	const double rho = t.rho_num / det + t.two_ue;
	if (!(rho >= 0) || !(t.z_min > 2 * rho * t.n_len))
		return true;
	const double rx = rho * t.kx / (t.z_min - rho * t.n_len), ry = rho * t.ky / (t.z_min - rho * t.n_len);
	const double slack = 1e-9 * (fabs(t.bx0) + fabs(t.bx1) + fabs(t.by0) + fabs(t.by1) + 1);
	const double widen = sqrt(rx * rx + ry * ry) * (1 + 1e-9) + pr.margin + slack;
	if (!(widen < 1e9))
		return true;
	const double x0 = (double)(ci - kSkyBorder), y0 = (double)(cj - kSkyBorder);
	if (x0 + 1 < t.bx0 - widen || x0 > t.bx1 + widen || y0 + 1 < t.by0 - widen || y0 > t.by1 + widen)
		return false;
	if (t.edges)
		for (int k = 0; k < 3; k++) {
			// the cell's nearest point to the edge's line, measured along the outward normal
			const double s = t.mx[k] * (x0 + .5 - t.x[k]) + t.my[k] * (y0 + .5 - t.y[k]) - .5 * (fabs(t.mx[k]) + fabs(t.my[k]));
			if (s > widen)
				return false;
		}
	// the point of the cell (widened by a pixel: far more than the margin) nearest to the camera's axis
	const double ax0 = x0 - 1, ax1 = x0 + 2, ay0 = y0 - 1, ay1 = y0 + 2;
	const double dx = t.ppx < ax0 ? ax0 - t.ppx : (t.ppx > ax1 ? t.ppx - ax1 : 0), dy = t.ppy < ay0 ? ay0 - t.ppy : (t.ppy > ay1 ? t.ppy - ay1 : 0);
	const double sec = sqrt(1 + dx * dx * t.tan2_x + dy * dy * t.tan2_y);
	const double bound = ((t.z_min / t.n_len - rho) * sec / kCameraMaxDir * (1 - 2e-6) - t.dt_num / det) * (1 - 1e-6);
	if (bound > 0) {
		float f = (float)bound;
		if ((double)f > bound)
			f = __uint_as_float(__float_as_uint(f) - 1);      // (positive: the next float down)
		nearest = f;
	}
	return true;
}

// One wave per triangle (gridDim.y waves for big outlines): BY_PRIMS, per primitive record that is a triangle in
// some leaf of the reference's tree (TreeRef::leaf_boxes marks those); otherwise per child reference of a wide
// node, which are the same triangles.  FILL: the second pass, which writes the triangle and its key - cell
// number, then the bits of the bound - into the lists whose lengths the first pass counted.
template <bool FILL, bool BY_PRIMS>
__global__ __launch_bounds__(kCellBlock) void k_cell_lists(const TreeRef tree, const uint32_t n_items, const Projection pr,
                                                           const int map_w, const int map_h, uint32_t *counts,
                                                           const uint32_t *starts, uint32_t *list, unsigned long long *keys)
{
	const uint32_t item = blockIdx.x * (kCellBlock / 64) + threadIdx.x / 64, lane = threadIdx.x & 63;
	if (item >= n_items)
		return;
	uint32_t prim;
	if (BY_PRIMS) {
		prim = item;
		if ((__float_as_uint(tree.prims[kPrimQuads * (size_t)prim].w) & 1u) || tree.leaf_boxes[2 * (size_t)prim].w != 1.f)
			return;           // a sphere (tested for every ray), or in no leaf
	} else {
		const uint32_t ref = reinterpret_cast<const uint32_t *>(tree.wide + 4 * (size_t)(item >> 2) + 3)[item & 3];
		if (ref == kWideNone || !(ref & kWideLeaf))
			return;
		prim = ref & ~kWideLeaf;
	}
	const Outline t = outline_of(tree, prim, pr, map_w, map_h);
	if (t.i1 < t.i0 || t.j1 < t.j0)
		return;
	const uint32_t w = (uint32_t)(t.i1 - t.i0 + 1), n = w * (uint32_t)(t.j1 - t.j0 + 1);
	for (uint32_t k = lane + 64 * blockIdx.y; k < n; k += 64 * gridDim.y) {
		const int ci = t.i0 + (int)(k % w), cj = t.j0 + (int)(k / w);
		float nearest;
		if (!touches(t, pr, ci, cj, nearest))
			continue;
		const size_t cell = (size_t)cj * map_w + ci;
		const uint32_t at = atomicAdd(&counts[cell], 1u);
		if (FILL) {
			list[starts[cell] + at] = prim;
			keys[starts[cell] + at] = ((unsigned long long)cell << 32) | __float_as_uint(nearest);
		}
	}
}

template <bool FILL>
void launch_lists(hipStream_t s, const TreeRef &tree, uint32_t n_wide, uint32_t n_prims, const Projection &pr, int map_w, int map_h,
                  uint32_t *counts, const uint32_t *starts, uint32_t *list, unsigned long long *keys)
{
	const bool by_prims = tree.wide == nullptr;
	const uint32_t items = by_prims ? n_prims : 4 * n_wide;
	// (a scene of a few triangles can have triangles that fill the frame; a scene of many has waves enough)
	const unsigned waves_per_item = items < (1u << 12) ? 64 : (items < (1u << 16) ? 4 : 1);
	const dim3 grid((items + kCellBlock / 64 - 1) / (kCellBlock / 64), waves_per_item);
	if (by_prims)
		hipLaunchKernelGGL((k_cell_lists<FILL, true>), grid, dim3(kCellBlock), 0, s, tree, items, pr, map_w, map_h, counts, starts, list, keys);
	else
		hipLaunchKernelGGL((k_cell_lists<FILL, false>), grid, dim3(kCellBlock), 0, s, tree, items, pr, map_w, map_h, counts, starts, list, keys);
}

}  // namespace

size_t cell_scan_temp_bytes(size_t n)
{
	size_t bytes = 0;
	uint32_t *none = nullptr;
	(void)rocprim::exclusive_scan(nullptr, bytes, none, none, 0u, n, rocprim::plus<uint32_t>(), hipStream_t(nullptr));
	return bytes;
}

hipError_t launch_cell_count(hipStream_t s, const TreeRef &tree, uint32_t n_wide, uint32_t n_prims, const Projection &pr, int map_w,
                             int map_h, uint32_t *counts, uint32_t *starts, void *temp, size_t temp_bytes)
{
	const size_t n = (size_t)map_w * map_h + 1;
	hipError_t e = hipMemsetAsync(counts, 0, n * sizeof(uint32_t), s);
	if (e != hipSuccess)
		return e;
	launch_lists<false>(s, tree, n_wide, n_prims, pr, map_w, map_h, counts, nullptr, nullptr, nullptr);
	// starts[cell] = entries before the cell's; starts[cells] = all entries
	return rocprim::exclusive_scan(temp, temp_bytes, counts, starts, 0u, n, rocprim::plus<uint32_t>(), s);
}

size_t cell_sort_temp_bytes(size_t entries)
{
	size_t bytes = 0;
	unsigned long long *no_keys = nullptr;
	uint32_t *no_values = nullptr;
	(void)rocprim::radix_sort_pairs(nullptr, bytes, no_keys, no_keys, no_values, no_values, entries, 0u, 64u, hipStream_t(nullptr));
	return bytes;
}

hipError_t launch_cell_fill(hipStream_t s, const TreeRef &tree, uint32_t n_wide, uint32_t n_prims, const Projection &pr, int map_w,
                            int map_h, uint32_t *counts, const uint32_t *starts, size_t entries, uint32_t *list_unsorted,
                            unsigned long long *keys_unsorted, uint32_t *list, unsigned long long *keys, void *temp, size_t temp_bytes)
{
	const size_t n = (size_t)map_w * map_h + 1;
	hipError_t e = hipMemsetAsync(counts, 0, n * sizeof(uint32_t), s);
	if (e != hipSuccess)
		return e;
	launch_lists<true>(s, tree, n_wide, n_prims, pr, map_w, map_h, counts, starts, list_unsorted, keys_unsorted);
	if ((e = hipGetLastError()) != hipSuccess || entries == 0)
		return e;
	// by cell, then nearest first (the cells' entries are contiguous already: `starts` stays valid)
	uint32_t cell_bits = 1;
	while (cell_bits < 32 && (n >> cell_bits))
		cell_bits++;
	return rocprim::radix_sort_pairs(temp, temp_bytes, keys_unsorted, keys, list_unsorted, list, entries, 0u, 32u + cell_bits, s);
}

}  // namespace rta
