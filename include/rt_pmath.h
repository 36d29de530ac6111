/*
 * rt_pmath.h -- portable, bit-reproducible single-precision elementary
 * functions: pm_sinf, pm_cosf, pm_logf, pm_powf.
 *
 * Why they exist.  The renderer's transport code calls sinf/cosf/logf/powf
 * (reference: src/main.cpp:45,682-683; src/shading.cpp:9-10,29-30,45,68,77,79).
 * glibc and the ROCm device library round these differently in the last bit,
 * and one differing bit changes a path's length and every later random draw of
 * its stream.  To let a GPU render be compared with a CPU render bit for bit,
 * both sides evaluate THESE definitions instead: every function below is a
 * fixed sequence of IEEE-754 binary64 +,-,*,/ operations (no fused
 * multiply-add, no library call), followed by one rounding to binary32.  Any
 * conforming compiler that does not contract or reassociate (build with
 * -ffp-contract=off, no fast-math) produces identical bits on x86-64 and on
 * gfx950.
 *
 * Accuracy: the binary64 intermediate is good to < 1e-13 relative, so the
 * rounded result equals the correctly rounded value except when the exact
 * value lies within ~1e-6 ulp of a rounding boundary; tests compare against
 * glibc's binary64 functions and allow 1 ulp.
 *
 * Domain notes: pm_sinf/pm_cosf are accurate for |x| < 1e5 (the renderer only
 * passes [0, 2*pi]); non-finite input gives NaN.  pm_powf handles the finite
 * cases the renderer meets (see the function); it is not a full C99 powf.
 */
#ifndef RT_PMATH_H
#define RT_PMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#  define RT_PM_FN __host__ __device__ static inline
#else
#  define RT_PM_FN static inline
#endif

RT_PM_FN double pm__from_bits(uint64_t u)
{
	double d;
	__builtin_memcpy(&d, &u, sizeof d);
	return d;
}

RT_PM_FN uint64_t pm__to_bits(double d)
{
	uint64_t u;
	__builtin_memcpy(&u, &d, sizeof u);
	return u;
}

RT_PM_FN float pm__nanf(void)
{
	uint32_t u = 0x7fc00000u;
	float f;
	__builtin_memcpy(&f, &u, sizeof f);
	return f;
}

RT_PM_FN float pm__inff(void)
{
	uint32_t u = 0x7f800000u;
	float f;
	__builtin_memcpy(&f, &u, sizeof f);
	return f;
}

/* sin and cos of r, |r| <= pi/4 (+ slack), Taylor series in Horner form. */
RT_PM_FN double pm__sin_kernel(double r)
{
	double z = r * r;
	double p = -7.6471637318198164759e-13;          /* -1/15! */
	p = p * z + 1.6059043836821614599e-10;          /*  1/13! */
	p = p * z + -2.5052108385441718775e-08;         /* -1/11! */
	p = p * z + 2.7557319223985890653e-06;          /*  1/9!  */
	p = p * z + -1.9841269841269841270e-04;         /* -1/7!  */
	p = p * z + 8.3333333333333333333e-03;          /*  1/5!  */
	p = p * z + -1.6666666666666666667e-01;         /* -1/3!  */
	return r + r * (z * p);
}

RT_PM_FN double pm__cos_kernel(double r)
{
	double z = r * r;
	double p = 4.7794773323873852974e-14;           /*  1/16! */
	p = p * z + -1.1470745597729724714e-11;         /* -1/14! */
	p = p * z + 2.0876756987868098979e-09;          /*  1/12! */
	p = p * z + -2.7557319223985890653e-07;         /* -1/10! */
	p = p * z + 2.4801587301587301587e-05;          /*  1/8!  */
	p = p * z + -1.3888888888888888889e-03;         /* -1/6!  */
	p = p * z + 4.1666666666666666667e-02;          /*  1/4!  */
	p = p * z + -0.5;
	return 1.0 + z * p;
}

/* Quadrant reduction shared by sin and cos: x = k*(pi/2) + r. */
RT_PM_FN double pm__reduce_pio2(double x, int *quadrant)
{
	const double two_over_pi = 6.36619772367581382433e-01;
	/* pi/2 split so that k * PIO2_HI is exact for |k| < 2^20 */
	const double pio2_hi = 1.57079632673412561417e+00;   /* 33 significant bits */
	const double pio2_lo = 6.07710050650619224932e-11;
	double kd = x * two_over_pi;
	kd = kd + (kd < 0.0 ? -0.5 : 0.5);
	int k = (int)kd;
	double kf = (double)k;
	*quadrant = k & 3;
	return (x - kf * pio2_hi) - kf * pio2_lo;
}

RT_PM_FN float pm_sinf(float xf)
{
	double x = (double)xf;
	if (!(x > -1e5 && x < 1e5))
		return pm__nanf();
	int q;
	double r = pm__reduce_pio2(x, &q);
	double v;
	switch (q) {
	case 0:  v = pm__sin_kernel(r); break;
	case 1:  v = pm__cos_kernel(r); break;
	case 2:  v = -pm__sin_kernel(r); break;
	default: v = -pm__cos_kernel(r); break;
	}
	return (float)v;
}

RT_PM_FN float pm_cosf(float xf)
{
	double x = (double)xf;
	if (!(x > -1e5 && x < 1e5))
		return pm__nanf();
	int q;
	double r = pm__reduce_pio2(x, &q);
	double v;
	switch (q) {
	case 0:  v = pm__cos_kernel(r); break;
	case 1:  v = -pm__sin_kernel(r); break;
	case 2:  v = -pm__cos_kernel(r); break;
	default: v = pm__sin_kernel(r); break;
	}
	return (float)v;
}

/*
 * Both at once, with the bits of pm_sinf(x) and pm_cosf(x): the reduction and the
 * two kernels are evaluated once and the quadrant only picks and negates, which
 * on a SIMD machine avoids running both kernels for each of the two calls when
 * neighbouring lanes fall into different quadrants.
 */
RT_PM_FN void pm_sincosf(float xf, float *s_out, float *c_out)
{
	double x = (double)xf;
	if (!(x > -1e5 && x < 1e5)) {
		*s_out = pm__nanf();
		*c_out = pm__nanf();
		return;
	}
	int q;
	double r = pm__reduce_pio2(x, &q);
	double sk = pm__sin_kernel(r), ck = pm__cos_kernel(r);
	double sv = (q & 1) ? ck : sk;         /* q: 0 sin, 1 cos, 2 -sin, 3 -cos */
	double cv = (q & 1) ? sk : ck;         /* q: 0 cos, 1 -sin, 2 -cos, 3 sin */
	if (q & 2)
		sv = -sv;
	if (q == 1 || q == 2)
		cv = -cv;
	*s_out = (float)sv;
	*c_out = (float)cv;
}

/* Natural log of a positive finite binary64 value (binary32 inputs widened to
 * binary64 are always normal there, so float subnormals need no special path). */
RT_PM_FN double pm__log_pos(double x)
{
	const double ln2 = 6.93147180559945286227e-01;
	uint64_t bits = pm__to_bits(x);
	int e = (int)((bits >> 52) & 0x7ff) - 1023;
	double m = pm__from_bits((bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
	if (m > 1.41421356237309514547) {
		m = m * 0.5;
		e = e + 1;
	}
	/* log(m) = 2*atanh(s), s = (m-1)/(m+1), |s| <= 0.1716 */
	double s = (m - 1.0) / (m + 1.0);
	double z = s * s;
	double p = 5.8823529411764705066e-02;           /* 1/17 */
	p = p * z + 6.6666666666666665741e-02;          /* 1/15 */
	p = p * z + 7.6923076923076927347e-02;          /* 1/13 */
	p = p * z + 9.0909090909090911614e-02;          /* 1/11 */
	p = p * z + 1.1111111111111110494e-01;          /* 1/9  */
	p = p * z + 1.4285714285714284921e-01;          /* 1/7  */
	p = p * z + 2.0000000000000001110e-01;          /* 1/5  */
	p = p * z + 3.3333333333333331483e-01;          /* 1/3  */
	double logm = 2.0 * s + 2.0 * s * (z * p);
	return (double)e * ln2 + logm;
}

RT_PM_FN float pm_logf(float xf)
{
	double x = (double)xf;
	if (x > 0.0 && x < 1e300)
		return (float)pm__log_pos(x);
	if (x == 0.0)
		return -pm__inff();
	if (x > 0.0)
		return xf;                /* +inf */
	return pm__nanf();            /* negative or NaN */
}

/* e^t for a binary64 t, result binary64 (caller rounds). */
RT_PM_FN double pm__exp(double t)
{
	const double inv_ln2 = 1.44269504088896338700e+00;
	const double ln2_hi = 6.93147180369123816490e-01;   /* 33 significant bits */
	const double ln2_lo = 1.90821492927058770002e-10;
	if (t > 709.0)
		t = 709.0;                 /* far beyond binary32 overflow */
	if (t < -740.0)
		return 0.0;
	double kd = t * inv_ln2;
	kd = kd + (kd < 0.0 ? -0.5 : 0.5);
	int k = (int)kd;
	double kf = (double)k;
	double r = (t - kf * ln2_hi) - kf * ln2_lo;          /* |r| <= 0.3466 */
	double p = 1.6059043836821614599e-10;           /* 1/13! */
	p = p * r + 2.0876756987868098979e-09;          /* 1/12! */
	p = p * r + 2.5052108385441718775e-08;          /* 1/11! */
	p = p * r + 2.7557319223985890653e-07;          /* 1/10! */
	p = p * r + 2.7557319223985890653e-06;          /* 1/9!  */
	p = p * r + 2.4801587301587301587e-05;          /* 1/8!  */
	p = p * r + 1.9841269841269841270e-04;          /* 1/7!  */
	p = p * r + 1.3888888888888888889e-03;          /* 1/6!  */
	p = p * r + 8.3333333333333333333e-03;          /* 1/5!  */
	p = p * r + 4.1666666666666666667e-02;          /* 1/4!  */
	p = p * r + 1.6666666666666666667e-01;          /* 1/3!  */
	p = p * r + 0.5;
	p = p * r + 1.0;
	p = p * r + 1.0;
	/* scale by 2^k in two steps so that k down to -1074 stays representable */
	int k1 = k / 2;
	int k2 = k - k1;
	double s1 = pm__from_bits((uint64_t)(k1 + 1023) << 52);
	double s2 = pm__from_bits((uint64_t)(k2 + 1023) << 52);
	return (p * s1) * s2;
}

/*
 * x^y.  Exact-by-construction small integer powers first (the renderer's
 * y == 2 and y == 5 call sites: src/shading.cpp:45,68,77,79), then the general
 * positive-base case through exp(y*log(x)) (src/main.cpp:45).
 */
RT_PM_FN float pm_powf(float x, float y)
{
	if (y == 2.0f)
		return x * x;             /* the correctly rounded square */
	if (y == 5.0f) {
		double d = (double)x;
		double d2 = d * d;
		return (float)(d2 * d2 * d);
	}
	if (y == 0.0f)
		return 1.0f;
	if (x > 0.0f && x < 3.0e38f) {
		double t = (double)y * pm__log_pos((double)x);
		return (float)pm__exp(t);
	}
	if (x == 0.0f)
		return y > 0.0f ? 0.0f : pm__inff();
	if (x > 0.0f)
		return y > 0.0f ? x : 0.0f;   /* +inf */
	return pm__nanf();            /* negative base with non-integer y, or NaN */
}

#endif /* RT_PMATH_H */
