/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the NeRF inference hot path of
 * fnysalehi/Surface-Irradiance-Estimation-from-Neural-Radiance-Fields.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, link, import or call anything in oracle/. The product
 * (libngp_hip.so / pyngp) never does.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or snapshots
 * for this path, cannot be compiled here (CUDA + absent tiny-cuda-nn), and its
 * encoding/MLP arithmetic lives in the un-vendored, un-pinned submodule
 * dependencies/tiny-cuda-nn (.gitmodules:13-15). The grid / MLP / SH maths in
 * this directory restate the public upstream tiny-cuda-nn definitions
 * (SURVEY.md Appendix B); everything else follows the reference file:line
 * cited next to each function.
 *
 * All arithmetic is IEEE fp32 with no FMA contraction (build with
 * -ffp-contract=off) so that results do not depend on the host compiler.
 */
#ifndef ORC_COMMON_H
#define ORC_COMMON_H

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;

static inline v3 v3_make(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_div(v3 a, v3 b) { return v3_make(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static inline v3 v3_divs(v3 a, float s) { return v3_make(a.x / s, a.y / s, a.z / s); }
static inline v3 v3_adds(v3 a, float s) { return v3_make(a.x + s, a.y + s, a.z + s); }
/* dot = (x*x' + y*y') + z*z', left to right, as a plain loop would do. */
static inline float v3_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 v3_cross(v3 a, v3 b) {
	return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float v3_length(v3 a) { return sqrtf(v3_dot(a, a)); }
/* tcnn: normalize(v) = v / length(v) (component-wise division). */
static inline v3 v3_normalize(v3 a) { return v3_divs(a, v3_length(a)); }
static inline float v3_get(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
static inline v3 v3_min(v3 a, v3 b) { return v3_make(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
static inline v3 v3_max(v3 a, v3 b) { return v3_make(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }

/* Column-major 3x3 (m[c*3+r]) times vector: c0*x, then + c1*y, then + c2*z. */
static inline v3 m3_mulv(const float* m, v3 v) {
	v3 r = v3_make(m[0] * v.x, m[1] * v.x, m[2] * v.x);
	r = v3_add(r, v3_make(m[3] * v.y, m[4] * v.y, m[5] * v.y));
	r = v3_add(r, v3_make(m[6] * v.z, m[7] * v.z, m[8] * v.z));
	return r;
}

typedef struct { v3 min, max; } aabb_t;

static inline int aabb_contains(const aabb_t* b, v3 p) { /* bounding_box.cuh:313-318 */
	return p.x >= b->min.x && p.x <= b->max.x && p.y >= b->min.y && p.y <= b->max.y && p.z >= b->min.z && p.z <= b->max.z;
}

/* bounding_box.cuh:172-219 -- slab test, returns FLT_MAX,FLT_MAX on a miss. */
static inline void aabb_ray_intersect(const aabb_t* b, v3 pos, v3 dir, float* out_tmin, float* out_tmax) {
	float tmin = (b->min.x - pos.x) / dir.x;
	float tmax = (b->max.x - pos.x) / dir.x;
	if (tmin > tmax) { float t = tmin; tmin = tmax; tmax = t; }
	float tymin = (b->min.y - pos.y) / dir.y;
	float tymax = (b->max.y - pos.y) / dir.y;
	if (tymin > tymax) { float t = tymin; tymin = tymax; tymax = t; }
	if (tmin > tymax || tymin > tmax) { *out_tmin = FLT_MAX; *out_tmax = FLT_MAX; return; }
	if (tymin > tmin) tmin = tymin;
	if (tymax < tmax) tmax = tymax;
	float tzmin = (b->min.z - pos.z) / dir.z;
	float tzmax = (b->max.z - pos.z) / dir.z;
	if (tzmin > tzmax) { float t = tzmin; tzmin = tzmax; tzmax = t; }
	if (tmin > tzmax || tzmin > tmax) { *out_tmin = FLT_MAX; *out_tmax = FLT_MAX; return; }
	if (tzmin > tmin) tmin = tzmin;
	if (tzmax < tmax) tmax = tzmax;
	*out_tmin = tmin; *out_tmax = tmax;
}

/* ---------------------------------------------------------------- fp16 emulation
 * gcc 11 has no _Float16 on x86, so binary16 is emulated exactly:
 * round-to-nearest-even from double, gradual underflow, overflow to inf. */
static inline float orc_half_to_float(uint16_t h) {
	uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
	uint32_t exp = (h >> 10) & 0x1fu;
	uint32_t man = h & 0x3ffu;
	uint32_t bits;
	if (exp == 0) {
		if (man == 0) {
			bits = sign;
		} else { /* subnormal: value = man * 2^-24 */
			float f = (float)man * 5.9604644775390625e-08f;
			memcpy(&bits, &f, 4);
			bits |= sign;
		}
	} else if (exp == 31) {
		bits = sign | 0x7f800000u | (man << 13);
	} else {
		bits = sign | ((exp + 112u) << 23) | (man << 13);
	}
	float out;
	memcpy(&out, &bits, 4);
	return out;
}

static inline uint16_t orc_double_to_half(double d) {
	uint16_t sign = 0;
	if (d != d) return 0x7e00u;
	if (signbit(d)) { sign = 0x8000u; d = -d; }
	if (d == 0.0) return sign;
	if (d >= 65520.0) return (uint16_t)(sign | 0x7c00u); /* rounds to inf */
	int e;
	(void)frexp(d, &e); /* d = m * 2^e, m in [0.5,1) -> floor(log2 d) = e-1 */
	int exp2 = e - 1;
	double quantum = exp2 < -14 ? 5.9604644775390625e-08 /* 2^-24 */ : ldexp(1.0, exp2 - 10);
	double q = rint(d / quantum); /* default rounding mode: nearest even; scaling is exact */
	double r = q * quantum;
	if (r == 0.0) return sign;
	(void)frexp(r, &e);
	exp2 = e - 1;
	if (exp2 < -14) { /* subnormal */
		return (uint16_t)(sign | (uint16_t)(r / 5.9604644775390625e-08));
	}
	uint32_t man = (uint32_t)(r / ldexp(1.0, exp2 - 10)) - 1024u;
	return (uint16_t)(sign | ((uint32_t)(exp2 + 15) << 10) | man);
}

static inline uint16_t orc_float_to_half(float f) { return orc_double_to_half((double)f); }

/* fp16 a + b with a single rounding (the sum of two halfs is exact in double). */
static inline uint16_t orc_half_add(uint16_t a, uint16_t b) {
	return orc_double_to_half((double)orc_half_to_float(a) + (double)orc_half_to_float(b));
}

/* fp16 fma(a, b, c) with a single rounding (__hfma / v_pk_fma_f16). The product of two halfs is exact in double
 * (22 significant bits); the sum p + c may not be, so it is formed as an exact pair (s, e) with TwoSum and the pair
 * is rounded: s alone decides unless it sits exactly on an fp16 rounding boundary, where the sign of e breaks the
 * tie (RN is monotonic and every fp16 midpoint is a double, so s on one side of a midpoint means p + c is too). */
static inline uint16_t orc_pair_to_half(double s, double e) {
	if (e == 0.0 || s != s) return orc_double_to_half(s);
	uint16_t sign = 0;
	double d = s;
	if (signbit(d)) { sign = 0x8000u; d = -d; e = -e; }
	if (d == 65520.0) return (uint16_t)(sign | (e < 0.0 ? 0x7bffu : 0x7c00u));
	if (d > 65520.0 || d == 0.0) return orc_double_to_half(s);
	int ex;
	(void)frexp(d, &ex);
	int exp2 = ex - 1;
	double quantum = exp2 < -14 ? 5.9604644775390625e-08 : ldexp(1.0, exp2 - 10);
	double scaled = d / quantum; /* exact: quantum is a power of two */
	double fl = floor(scaled);
	if (scaled - fl != 0.5) return orc_double_to_half(s);
	double q = e > 0.0 ? fl + 1.0 : fl;
	return (uint16_t)(sign | orc_double_to_half(q * quantum));
}
static inline uint16_t orc_half_fma(uint16_t a, uint16_t b, uint16_t c) {
	double p = (double)orc_half_to_float(a) * (double)orc_half_to_float(b);
	double cc = (double)orc_half_to_float(c);
	double s = p + cc;
	double bb = s - p;
	double e = (p - (s - bb)) + (cc - bb);
	return orc_pair_to_half(s, e);
}

#ifdef __cplusplus
}
#endif
#endif
