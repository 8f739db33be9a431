// PCG32 (O'Neill's pcg32 in Wenzel Jakob's single-header form, the `default_rng_t` of the reference through
// tiny-cuda-nn's dependencies/pcg32/pcg32.h -- an un-vendored submodule, absent from the reference mount). This is a
// restatement of the published generator: 64-bit LCG state, XSH-RR output, logarithmic skip-ahead, next_float from
// the top 23 bits. Usable from host and device code.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define NGP_HD __host__ __device__ inline
#else
#define NGP_HD inline
#endif

namespace ngp {

struct Pcg32 {
	uint64_t state, inc;
	static constexpr uint64_t MULT = 0x5851f42d4c957f2dULL;

	NGP_HD void seed(uint64_t initstate, uint64_t initseq = 1u) {
		state = 0u;
		inc = (initseq << 1u) | 1u;
		next_uint();
		state += initstate;
		next_uint();
	}
	NGP_HD uint32_t next_uint() {
		uint64_t oldstate = state;
		state = oldstate * MULT + inc;
		uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
		uint32_t rot = (uint32_t)(oldstate >> 59u);
		return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31u));
	}
	NGP_HD float next_float() {
		union { uint32_t u; float f; } x;
		x.u = (next_uint() >> 9) | 0x3f800000u;
		return x.f - 1.0f;
	}
	// skip `delta` draws ahead (the reference's no-argument advance() uses 2^32)
	NGP_HD void advance(uint64_t delta = (1ull << 32)) {
		uint64_t cur_mult = MULT, cur_plus = inc, acc_mult = 1u, acc_plus = 0u;
		while (delta > 0) {
			if (delta & 1) {
				acc_mult *= cur_mult;
				acc_plus = acc_plus * cur_mult + cur_plus;
			}
			cur_plus = (cur_mult + 1) * cur_plus;
			cur_mult *= cur_mult;
			delta >>= 1;
		}
		state = acc_mult * state + acc_plus;
	}
};

} // namespace ngp
