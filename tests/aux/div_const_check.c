/* Exhaustive check of the reciprocal + one-FMA-correction division used by the HIP kernels
 * (csrc/nerf_device.h: div_const) against IEEE division, for the constant divisors on the march path.
 * usage: div_const_check <stride>   -- prints the number of mismatches. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline float div_const(float x, float d, float rc) {
	float q0 = x * rc;
	float r = fmaf(-q0, d, x);
	return fmaf(r, rc, q0);
}
static inline uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
	uint32_t stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 1u;
	const float stepsize = 1.73205080757f / 1024.0f;           /* nerf_device.cuh:31 */
	const float max_cone = stepsize * 128.0f * 1024.0f / 128.0f; /* nerf_device.cuh:35 */
	const float max_step = stepsize * 128.0f;
	const float d[3] = {stepsize, max_cone, max_step - stepsize};
	const uint32_t lo = bits(1e-9f), hi = bits(65536.0f);
	long long bad = 0;
	for (int k = 0; k < 3; ++k) {
		const float dk = d[k], rc = 1.0f / dk;
#pragma omp parallel for reduction(+ : bad) schedule(static)
		for (uint32_t u = lo; u < hi; u += stride) {
			float x = from_bits(u);
			if (bits(div_const(x, dk, rc)) != bits(x / dk)) ++bad;
			if (bits(div_const(-x, dk, rc)) != bits(-x / dk)) ++bad;
		}
	}
	/* zero maps to zero */
	for (int k = 0; k < 3; ++k) if (div_const(0.0f, d[k], 1.0f / d[k]) != 0.0f) ++bad;
	printf("%lld\n", bad);
	return bad != 0;
}
