// Per-cell triangle lists for a pinhole camera's rays.
//
// All camera rays start at one point, so which triangles a camera ray can hit is decided by where
// it crosses the image plane.  Once per view, every triangle of the wide tree is drawn onto the
// pixel-sized cells of the image plane (the frame and kSkyBorder cells around it) - its projected
// outline widened by what the reference's arithmetic can add - and every cell gets the list of the
// triangles that touch it.  A camera ray is then answered by testing the triangles of its cell with
// the reference's own HitTriangle arithmetic (src/main.cpp:250-279), no tree involved, and certified
// as in traverse_wide.hpp: the list takes the place of step 1 of the proof there.
//
// Why the list is complete.  Let the reference's HitTriangle accept triangle p for an admitted ray
// (|origin - any vertex| <= S, |direction| <= kWideMaxDir, components of ordinary magnitude).  By the
// derivation in wide_tree.cpp the exact intersection X of the ray's line with p's plane lies within
//     rho = 16.02e4 u |d| E^2 (S + E) + 2uE,   u = 2^-24, E = p's longest stored edge vector,
// of p.  X lies on the ray's line, in front of the camera (every point within rho of p is, checked
// below), so X is seen at the ray's own point on the image plane.  Seen from the pinhole, a point
// within rho of a point P of the triangle lands within (rx, ry) pixels of P's image, where with
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
// the first entry whose bound lies beyond the nearest hit so far.  A camera ray's direction is normalised
// (|d| within 1 +- 1e-6), so its parameter at X is |X - o| up to that factor, and X lies at least
// (depth - rho) sec(theta) from the pinhole: depth = the least distance of the triangle's vertices from the
// pinhole along the camera's axis, theta = the least angle to that axis of any ray through the cell; the
// computed t lies within dt = 16.02e4 u E^2 (S + E) + u t of the parameter at X (wide_tree.cpp).
#include "kernels.hpp"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace rta {

namespace {

constexpr int kCellBlock = 256;

struct Outline {
	double x[3], y[3];      // projected vertices, pixel coordinates
	double widen;           // pixels
	bool everywhere;        // no statement about this triangle: it joins every cell
	bool edges;             // the three edge tests apply (a proper triangle on the image plane)
	double mx[3], my[3];    // outward unit normals of the edges k -> k + 1
	double depth;           // least distance of a point within rho of the triangle from the pinhole, along the camera's axis
	double dt;              // see above (without the u t part)
	double ppx, ppy;        // where the camera's axis meets the image plane, pixel coordinates
	double tan2_x, tan2_y;  // (a pixel's width, height / the image plane's distance)^2
	int i0, i1, j0, j1;     // cells of the bounding rectangle, clipped to the map (empty when i1 < i0 or j1 < j0)
};

__device__ Outline outline_of(const TreeRef &tree, const uint32_t prim, const Projection &pr, const double span,
                              const int map_w, const int map_h)
{
	Outline t;
	const float4 *rec = tree.prims + kPrimQuads * (size_t)prim;
	const float4 p0 = rec[0], p1 = rec[1], p2 = rec[2];
	// the triangle HitTriangle works on: v2, v2 + (v0 - v2), v2 + (v1 - v2) with the stored edge vectors
	const double v[3][3] = {{(double)p0.x + p1.x, (double)p0.y + p1.y, (double)p0.z + p1.z},
	                        {(double)p0.x + p2.x, (double)p0.y + p2.y, (double)p0.z + p2.z},
	                        {(double)p0.x, (double)p0.y, (double)p0.z}};
	const double e0 = sqrt((double)p1.x * p1.x + (double)p1.y * p1.y + (double)p1.z * p1.z);
	const double e1 = sqrt((double)p2.x * p2.x + (double)p2.y * p2.y + (double)p2.z * p2.z);
	const double edge = (e0 > e1 ? e0 : e1) * (1 + 1e-6);
	const double u = 0x1p-24;
	const double rho = (16.02e4 * u * (double)kWideMaxDir * edge * edge * (span + edge) + 2 * u * edge) * (1 + 1e-9);
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
	t.everywhere = !(rho >= 0 && rho < 1e30);
	double z_min = 1e300, xc_max = 0, yc_max = 0;
	for (int c = 0; c < 3; c++) {
		double w[3], z = 0;
		for (int k = 0; k < 3; k++) {
			w[k] = v[c][k] - pr.o[k];
			z += w[k] * pr.n[k];
		}
		const double scale = pr.h / z;
		if (!(scale > 0) || !(scale < 1e6))
			t.everywhere = true;
		double x = 0, y = 0;
		for (int k = 0; k < 3; k++) {
			const double p = pr.o[k] + w[k] * scale - pr.ul[k];
			x += p * pr.du[k];
			y += p * pr.dv[k];
		}
		x /= pr.du2;
		y /= pr.dv2;
		if (!(x > -1e9 && x < 1e9 && y > -1e9 && y < 1e9))
			t.everywhere = true;
		t.x[c] = x;
		t.y[c] = y;
		const double za = z < 0 ? -z : z, xa = x + cx < 0 ? -(x + cx) : x + cx, ya = y + cy < 0 ? -(y + cy) : y + cy;
		z_min = za < z_min ? za : z_min;
		xc_max = xa > xc_max ? xa : xc_max;
		yc_max = ya > yc_max ? ya : yc_max;
	}
	// every point within rho of the triangle must lie well in front of the camera
	if (!(z_min > 2 * rho * n_len))
		t.everywhere = true;
	double widen = 0;
	if (!t.everywhere) {
		const double rx = rho * (h_abs / du_len + xc_max * n_len) / (z_min - rho * n_len);
		const double ry = rho * (h_abs / dv_len + yc_max * n_len) / (z_min - rho * n_len);
		widen = sqrt(rx * rx + ry * ry) * (1 + 1e-9) + pr.margin;
		if (!(widen < 1e9))
			t.everywhere = true;
	}
	t.widen = widen;
	t.depth = t.everywhere ? 0 : (z_min / n_len - rho) * (1 - 1e-9);
	t.dt = 16.02e4 * u * edge * edge * (span + edge) * (1 + 1e-9);
	t.ppx = -cx;
	t.ppy = -cy;
	const double plane = h_abs / n_len;
	t.tan2_x = pr.du2 / (plane * plane);
	t.tan2_y = pr.dv2 / (plane * plane);
	double x0 = -1e9, x1 = 1e9, y0 = -1e9, y1 = 1e9;
	t.edges = false;
	if (!t.everywhere) {
		x0 = fmin(t.x[0], fmin(t.x[1], t.x[2]));
		x1 = fmax(t.x[0], fmax(t.x[1], t.x[2]));
		y0 = fmin(t.y[0], fmin(t.y[1], t.y[2]));
		y1 = fmax(t.y[0], fmax(t.y[1], t.y[2]));
		// (and by more than the rounding of these few binary64 operations)
		const double slack = 1e-9 * (fabs(x0) + fabs(x1) + fabs(y0) + fabs(y1) + 1);
		t.widen = widen + slack;
		x0 -= t.widen, x1 += t.widen, y0 -= t.widen, y1 += t.widen;
		const double area2 = (t.x[1] - t.x[0]) * (t.y[2] - t.y[0]) - (t.x[2] - t.x[0]) * (t.y[1] - t.y[0]);
		const double extent = (x1 - x0) + (y1 - y0);
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

// may a camera ray through cell (ci, cj) of the map be one that the triangle's test accepts?
__device__ bool touches(const Outline &t, const int ci, const int cj)
{
	if (!t.edges)
		return true;
	const double ccx = (double)(ci - kSkyBorder) + .5, ccy = (double)(cj - kSkyBorder) + .5;
	for (int k = 0; k < 3; k++) {
		// the cell's nearest point to the edge's line, measured along the outward normal
		const double s = t.mx[k] * (ccx - t.x[k]) + t.my[k] * (ccy - t.y[k]) - .5 * (fabs(t.mx[k]) + fabs(t.my[k]));
		if (s > t.widen)
			return false;
	}
	return true;
}

// lower bound of the t that HitTriangle can return for the triangle and a camera ray through the cell
__device__ float nearest_t(const Outline &t, const int ci, const int cj)
{
	if (t.everywhere || !(t.depth > 0))
		return 0.f;
	// the point of the cell (widened by a pixel: far more than the margin) nearest to the camera's axis
	const double x0 = (double)(ci - kSkyBorder) - 1, x1 = x0 + 3, y0 = (double)(cj - kSkyBorder) - 1, y1 = y0 + 3;
	const double dx = t.ppx < x0 ? x0 - t.ppx : (t.ppx > x1 ? t.ppx - x1 : 0), dy = t.ppy < y0 ? y0 - t.ppy : (t.ppy > y1 ? t.ppy - y1 : 0);
	const double sec = sqrt(1 + dx * dx * t.tan2_x + dy * dy * t.tan2_y);
	const double bound = (t.depth * sec * (1 - 2e-6) - t.dt) * (1 - 1e-6);
	if (!(bound > 0))
		return 0.f;
	float f = (float)bound;
	if ((double)f > bound)
		f = __uint_as_float(__float_as_uint(f) - 1);      // (positive: the next float down)
	return f;
}

// One wave per child reference of a wide node (gridDim.y waves for big outlines); FILL: the second pass,
// which writes the triangle and its key - cell number, then the bits of nearest_t - into the lists whose
// lengths the first pass counted.
template <bool FILL>
__global__ __launch_bounds__(kCellBlock) void k_cell_lists(const TreeRef tree, const uint32_t n_wide, const Projection pr,
                                                           const double span, const int map_w, const int map_h,
                                                           uint32_t *counts, const uint32_t *starts, uint32_t *list,
                                                           unsigned long long *keys)
{
	const uint32_t item = blockIdx.x * (kCellBlock / 64) + threadIdx.x / 64, lane = threadIdx.x & 63;
	if (item >= 4 * n_wide)
		return;
	const uint32_t ref = reinterpret_cast<const uint32_t *>(tree.wide + 4 * (size_t)(item >> 2) + 3)[item & 3];
	if (ref == kWideNone || !(ref & kWideLeaf))
		return;
	const uint32_t prim = ref & ~kWideLeaf;
	const Outline t = outline_of(tree, prim, pr, span, map_w, map_h);
	if (t.i1 < t.i0 || t.j1 < t.j0)
		return;
	const uint32_t w = (uint32_t)(t.i1 - t.i0 + 1), n = w * (uint32_t)(t.j1 - t.j0 + 1);
	for (uint32_t k = lane + 64 * blockIdx.y; k < n; k += 64 * gridDim.y) {
		const int ci = t.i0 + (int)(k % w), cj = t.j0 + (int)(k / w);
		if (!touches(t, ci, cj))
			continue;
		const size_t cell = (size_t)cj * map_w + ci;
		const uint32_t at = atomicAdd(&counts[cell], 1u);
		if (FILL) {
			list[starts[cell] + at] = prim;
			keys[starts[cell] + at] = ((unsigned long long)cell << 32) | __float_as_uint(nearest_t(t, ci, cj));
		}
	}
}

}  // namespace

size_t cell_scan_temp_bytes(size_t n)
{
	size_t bytes = 0;
	uint32_t *none = nullptr;
	(void)rocprim::exclusive_scan(nullptr, bytes, none, none, 0u, n, rocprim::plus<uint32_t>(), hipStream_t(nullptr));
	return bytes;
}

static dim3 cell_grid(uint32_t n_wide)
{
	// (a scene of a few triangles can have triangles that fill the frame; a scene of millions has waves enough)
	const unsigned waves_per_leaf = n_wide < (1u << 14) ? 16 : 1;
	return dim3((4 * n_wide + kCellBlock / 64 - 1) / (kCellBlock / 64), waves_per_leaf);
}

hipError_t launch_cell_count(hipStream_t s, const TreeRef &tree, uint32_t n_wide, const Projection &pr, double span, int map_w,
                             int map_h, uint32_t *counts, uint32_t *starts, void *temp, size_t temp_bytes)
{
	const size_t n = (size_t)map_w * map_h + 1;
	hipError_t e = hipMemsetAsync(counts, 0, n * sizeof(uint32_t), s);
	if (e != hipSuccess)
		return e;
	hipLaunchKernelGGL(k_cell_lists<false>, cell_grid(n_wide), dim3(kCellBlock), 0, s, tree, n_wide, pr, span, map_w, map_h, counts,
	                   (const uint32_t *)nullptr, (uint32_t *)nullptr, (unsigned long long *)nullptr);
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

hipError_t launch_cell_fill(hipStream_t s, const TreeRef &tree, uint32_t n_wide, const Projection &pr, double span, int map_w,
                            int map_h, uint32_t *counts, const uint32_t *starts, size_t entries, uint32_t *list_unsorted,
                            unsigned long long *keys_unsorted, uint32_t *list, unsigned long long *keys, void *temp, size_t temp_bytes)
{
	const size_t n = (size_t)map_w * map_h + 1;
	hipError_t e = hipMemsetAsync(counts, 0, n * sizeof(uint32_t), s);
	if (e != hipSuccess)
		return e;
	hipLaunchKernelGGL(k_cell_lists<true>, cell_grid(n_wide), dim3(kCellBlock), 0, s, tree, n_wide, pr, span, map_w, map_h, counts,
	                   starts, list_unsorted, keys_unsorted);
	if ((e = hipGetLastError()) != hipSuccess || entries == 0)
		return e;
	// by cell, then nearest first (the cells' entries are contiguous already: `starts` stays valid)
	uint32_t cell_bits = 1;
	while (cell_bits < 32 && (n >> cell_bits))
		cell_bits++;
	return rocprim::radix_sort_pairs(temp, temp_bytes, keys_unsorted, keys, list_unsorted, list, entries, 0u, 32u + cell_bits, s);
}

}  // namespace rta
