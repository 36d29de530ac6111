// Device BSDF: GGX specular + Lambert diffuse with mixture sampling, the
// reference's material model (src/shading.cpp:1-137).  Elementary functions come
// from include/rt_pmath.h so that a CPU evaluation of the same definitions gives
// the same bits.
#pragma once

#include "device_math.hpp"
#include "rt_pmath.h"

namespace rta {

struct Material {
	V3 color;
	float alpha;      // the file's roughness, squared by the parser
	float ior, metallic;
};

struct BsdfSample {
	V3 f;             // bsdf * cos
	float pdf;
	V3 l;             // sampled direction, local frame
};

// Both samplers below begin alike - two random numbers, the angle 2 pi r1, its sine and cosine - so the part they
// share is done once for a wave whose lanes sample different lobes: the sine and cosine are binary64 polynomials.
struct LobeDraws {
	float r2, sin_phi, cos_phi;
};

RTA_DEV LobeDraws lobe_draws(uint32_t &rng)
{
	LobeDraws d;
	const float r1 = rng_float(rng);
	d.r2 = rng_float(rng);
	const float phi = 2 * kPiF * r1;
	pm_sincosf(phi, &d.sin_phi, &d.cos_phi);
	return d;
}

// src/shading.cpp:1-14
RTA_DEV V3 sample_cosine(const LobeDraws &d)
{
	float s = __builtin_sqrtf(d.r2);
	float x = d.cos_phi * s;
	float y = d.sin_phi * s;
	float z = __builtin_sqrtf(1 - d.r2);
	return mk(x, y, z);
}

// src/shading.cpp:19-37: visible-normal sampling on the spherical cap; the
// fused multiply-add is explicit in the reference.
RTA_DEV V3 sample_vndf(V3 v, float alpha, const LobeDraws &d)
{
	V3 vh = unit(mk(alpha * v.x, alpha * v.y, v.z));
	float z = __builtin_fmaf((1 - d.r2), (1 + vh.z), -vh.z);
	float st = __builtin_sqrtf(lesser(1.f, greater(0.f, 1 - z * z)));
	float x = st * d.cos_phi;
	float y = st * d.sin_phi;
	V3 h = mk(x, y, z) + vh;
	V3 n = unit(mk(alpha * h.x, alpha * h.y, h.z));
	return (n * 2.f) * dot(n, v) - v;
}

// src/shading.cpp:39-50
RTA_DEV float vndf_pdf(V3 v, V3 l, float alpha)
{
	V3 h = unit(v + l);
	float a2 = alpha * alpha;
	float ndf = a2 / (kPiF * pm_powf(h.z * h.z * (a2 - 1) + 1, 2.f));
	float vis = 1 / (__builtin_fabsf(v.z) + __builtin_sqrtf(a2 + (1 - a2) * v.z * v.z));
	return ndf * vis / 2;
}

// src/shading.cpp:56-86
RTA_DEV V3 bsdf_eval(V3 v, V3 l, const Material &m)
{
	if (l.z <= 0)
		return mk(0, 0, 0);
	V3 h = unit(v + l);
	float a2 = m.alpha * m.alpha;
	float ndf = a2 / (kPiF * pm_powf(pm_powf(h.z, 2.f) * (a2 - 1) + 1, 2.f));
	float vis_v = l.z * __builtin_sqrtf(v.z * v.z * (1 - a2) + a2);
	float vis_l = v.z * __builtin_sqrtf(l.z * l.z * (1 - a2) + a2);
	float vis = .5f / (vis_v + vis_l);
	V3 f0 = mix(splat(pm_powf((1 - m.ior) / (1 + m.ior), 2.f)), m.color, m.metallic);
	V3 fr = f0 + one_minus(f0) * pm_powf(1 - dot(h, v), 5.f);
	V3 diffuse = ((one_minus(fr) * m.color) / kPiF) * (1 - m.metallic);
	V3 spec = fr * (vis * ndf);
	return (diffuse + spec) * l.z;
}

// src/shading.cpp:88-100
RTA_DEV void lobe_weights(const Material &m, float &w_cos, float &w_vndf)
{
	float c = 1 - m.metallic;
	float s = 1.f + c;
	w_vndf = 1.f / s;
	w_cos = c / s;
}

// src/shading.cpp:102-113
RTA_DEV float bsdf_pdf(V3 v, V3 l, const Material &m)
{
	float wc, wv;
	lobe_weights(m, wc, wv);
	float pc = greater(l.z, 0.f) / kPiF;
	float pv = vndf_pdf(v, l, m.alpha);
	return pc * wc + pv * wv;
}

// src/shading.cpp:115-137
RTA_DEV BsdfSample sample_bsdf(V3 v, const Material &m, uint32_t &rng)
{
	float wc, wv;
	lobe_weights(m, wc, wv);
	const bool cosine = rng_float(rng) < wc;
	const LobeDraws d = lobe_draws(rng);
	V3 l = cosine ? sample_cosine(d) : sample_vndf(v, m.alpha, d);
	BsdfSample s;
	s.pdf = bsdf_pdf(v, l, m);
	s.f = bsdf_eval(v, l, m);
	s.l = l;
	return s;
}

// src/main.cpp:480-483
RTA_DEV float power_heuristic(float f, float g) { return f * f / (f * f + g * g); }

// src/main.cpp:485-494: solid-angle density of a point on a triangle
RTA_DEV float triangle_pdf(V3 v0, V3 v1, V3 v2, V3 from, V3 on_tri, V3 tri_normal)
{
	float area = norm(cross(v1 - v0, v2 - v0)) / 2;
	V3 dir = unit(from - on_tri);
	float l2 = norm2(from - on_tri);
	return l2 / dot(tri_normal, dir) / area;
}

// src/main.cpp:36-46
RTA_DEV float tonemap(float c, float exposure)
{
	c *= exposure;
	float m = c / (1.f + c);
	if (m <= .0031308f)
		return m * 12.92f;
	return 1.055f * pm_powf(m, 1 / 2.4f) - .055f;
}

}  // namespace rta
