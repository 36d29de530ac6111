// Device-side vector algebra with the reference's operation order
// (src/common.h:52-305).  Every expression keeps the reference's association
// and uses true IEEE division and square root, so that results are the bits an
// x86 build of the reference produces.  Compile with -ffp-contract=off.
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

namespace rta {

#define RTA_DEV __device__ __forceinline__

struct V3 {
	float x, y, z;
};

constexpr float kPiF = 3.14159265f;     // src/common.h:73

// Ternary min/max (src/common.h:75-81): `a < b ? a : b`.  Written as selects so
// that a NaN in `a` yields `b` and a NaN in `b` yields `b`, like the reference.
RTA_DEV float lesser(float a, float b) { return a < b ? a : b; }
RTA_DEV float greater(float a, float b) { return a > b ? a : b; }

RTA_DEV V3 mk(float x, float y, float z) { return V3{x, y, z}; }
RTA_DEV V3 splat(float s) { return V3{s, s, s}; }
RTA_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RTA_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RTA_DEV V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
RTA_DEV V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RTA_DEV V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
RTA_DEV V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
RTA_DEV V3 one_minus(V3 a) { return V3{1.f - a.x, 1.f - a.y, 1.f - a.z}; }

RTA_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RTA_DEV float norm2(V3 a) { return dot(a, a); }
RTA_DEV float norm(V3 a) { return __builtin_sqrtf(norm2(a)); }
RTA_DEV V3 unit(V3 a) { return a / norm(a); }
RTA_DEV V3 cross(V3 a, V3 b)
{
	return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
RTA_DEV V3 mix(V3 a, V3 b, float t) { return a + (b - a) * t; }

// Orthonormal frame around a unit normal (src/common.h:289-296) and the two
// products the integrator needs (src/common.h:284-287, 298-305).
struct Frame {
	V3 x, y, z;
};

RTA_DEV Frame frame_from_normal(V3 n)
{
	V3 a = (__builtin_fabsf(n.x) > .9f) ? mk(0, 1, 0) : mk(1, 0, 0);
	V3 y = unit(cross(n, a));
	V3 x = cross(n, y);
	return Frame{x, y, n};
}

// world = x*v.x + y*v.y + z*v.z
RTA_DEV V3 to_world(const Frame &f, V3 v) { return (f.x * v.x + f.y * v.y) + f.z * v.z; }

// local = transpose(frame) * v, evaluated column by column like the reference:
// v.x*(x.x, y.x, z.x) + v.y*(x.y, y.y, z.y) + v.z*(x.z, y.z, z.z)
RTA_DEV V3 to_local(const Frame &f, V3 v)
{
	V3 c0 = mk(f.x.x, f.y.x, f.z.x), c1 = mk(f.x.y, f.y.y, f.z.y), c2 = mk(f.x.z, f.y.z, f.z.z);
	return (c0 * v.x + c1 * v.y) + c2 * v.z;
}

// xorshift32 (src/main.cpp:83-98)
RTA_DEV uint32_t rng_next(uint32_t &s)
{
	uint32_t x = s;
	x ^= x << 13;
	x ^= x >> 17;
	x ^= x << 5;
	s = x;
	return x;
}

RTA_DEV float rng_float(uint32_t &s) { return (float)(rng_next(s) >> 8) / 16777216.f; }

RTA_DEV uint32_t mix32(uint32_t h)
{
	h ^= h >> 16;
	h *= 0x7feb352dU;
	h ^= h >> 15;
	h *= 0x846ca68bU;
	h ^= h >> 16;
	return h;
}

// Seed of the stream serving block `block` of pixel `pixel`; never zero because
// xorshift32 is stuck there.
RTA_DEV uint32_t stream_seed(uint32_t seed, uint32_t pixel, uint32_t block)
{
	uint32_t h = mix32(pixel + seed);
	h = mix32(h ^ (block * 0x9e3779b9U + 0x7f4a7c15U));
	return h ? h : 0x9e3779b9U;
}

}  // namespace rta
