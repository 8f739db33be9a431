// Device-side building blocks of the NeRF inference path, written for CDNA4 (gfx950, wave64).
//
// Arithmetic contract (DESIGN.md "Numerics"): IEEE fp32, no FMA contraction (-ffp-contract=off), correctly
// rounded division/sqrt; hash-grid features accumulate in fp16 exactly as tcnn's kernel_grid does (one fp16 fma per
// corner and feature, the tvec-era sequence; -DNGP_TCNN_LEGACY_ENCODE: the older one, see accumulate_corner); MLP layers run on
// v_mfma_f32_16x16x32_f16 (fp16 operands, fp32 accumulate) with activations rounded to fp16 between layers.
// Each function cites the reference kernel/device function whose behaviour it reproduces.
#pragma once

#include "ngp_kernels.h"

namespace ngp {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

struct f3 {
	float x, y, z;
};

#define NGP_DEV __device__ __forceinline__

NGP_DEV f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
NGP_DEV f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
NGP_DEV f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
NGP_DEV f3 mul3(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
NGP_DEV f3 div3(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
NGP_DEV f3 scale3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
NGP_DEV f3 adds3(f3 a, float s) { return mk3(a.x + s, a.y + s, a.z + s); }
NGP_DEV float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
NGP_DEV f3 normalize3(f3 a) {
	float len = __builtin_sqrtf(dot3(a, a));
	return mk3(a.x / len, a.y / len, a.z / len);
}
// column-major 3x3 times vector, accumulated column by column
NGP_DEV f3 m3_mulv(const float* m, f3 v) {
	f3 r = mk3(m[0] * v.x, m[1] * v.x, m[2] * v.x);
	r = add3(r, mk3(m[3] * v.y, m[4] * v.y, m[5] * v.y));
	r = add3(r, mk3(m[6] * v.z, m[7] * v.z, m[8] * v.z));
	return r;
}

// ---------------------------------------------------------------------------------------------------------
// Low-discrepancy sequences: random_val.cuh:274-336. Sobol dimension 0 has direction numbers 1<<(31-bit), i.e.
// sobol(index, 0) is a plain bit reversal, which gfx950 does in one v_bfrev_b32.
NGP_DEV uint32_t laine_karras_permutation(uint32_t x, uint32_t seed) {
	x += seed;
	x ^= x * 0x6c50b47cu;
	x ^= x * 0xb82f1e52u;
	x ^= x * 0xc7afe638u;
	x ^= x * 0x8d22f6e6u;
	return x;
}
NGP_DEV uint32_t nested_uniform_scramble_base2(uint32_t x, uint32_t seed) {
	x = __builtin_bitreverse32(x);
	x = laine_karras_permutation(x, seed);
	return __builtin_bitreverse32(x);
}
NGP_DEV uint32_t hash_combine(uint32_t seed, uint32_t v) { return seed ^ (v + (seed << 6) + (seed >> 2)); }
NGP_DEV float ld_random_val_dim0(uint32_t index, uint32_t seed) {
	index = nested_uniform_scramble_base2(index, seed);
	uint32_t x = nested_uniform_scramble_base2(__builtin_bitreverse32(index), hash_combine(seed, 0u));
	return (float)x * 2.3283064365386963e-10f;
}

// ---------------------------------------------------------------------------------------------------------
// Exponential stepping: nerf_device.cuh:378-428
NGP_DEV float stepsize() { return 1.73205080757f / (float)NERF_STEPS; }
NGP_DEV float max_cone_stepsize() { return stepsize() * (float)(1u << (NERF_CASCADES - 1)) * (float)NERF_STEPS / (float)NERF_GRIDSIZE; }

// x / d for a compile-time constant d, bit-identical to the IEEE division: q0 = x*rc, r = fma(-q0, d, x),
// q = fma(r, rc, q0) with rc = RN(1/d) is correctly rounded (Markstein); verified exhaustively for the three
// stepping constants over every float in [1e-9, 65536) and its negatives. 3 instructions instead of ~11.
NGP_DEV float div_const(float x, float d, float rc) {
	float q0 = x * rc;
	float r = __builtin_fmaf(-q0, d, x);
	return __builtin_fmaf(r, rc, q0);
}
NGP_DEV float div_stepsize(float x) { return div_const(x, stepsize(), 1.0f / stepsize()); }
NGP_DEV float div_max_cone_stepsize(float x) { return div_const(x, max_cone_stepsize(), 1.0f / max_cone_stepsize()); }

// The piecewise map's constants depend on the cone angle alone -- five logf / expf and two divisions that the reference's inline
// functions form on every call. A persistent kernel forms them once (make_stepping) and carries them; same expressions, same values.
// The middle branch (exponential steps, t between at and bt) is log(t) / log(1 + c) one way and exp(n log(1 + c)) back. The reference is built with
// --use_fast_math: its logf / expf there are lg2.approx / ex2.approx with a multiply and its division is approximate; v_log_f32 / v_exp_f32 are the
// gfx950 counterparts (1 ulp), so each direction is ONE transcendental and one multiply -- glibc-grade logf / expf and an IEEE division cost ~95
// VALU instructions per empty-space step and ~55 per sample of a scene with aabb_scale > 1, where the march is more than half of the frame
// (profiles/r3_sections_configs.txt). The difference to the oracle's libm is ~1e-5 of a step in n: a ray in ~1e4 lands on the other side of a lattice
// point, as it already did by an ulp of the precise functions (tests: the cascaded scenes' sample counts within 5e-3). -DNGP_PRECISE_STEPPING restores them.
struct Stepping {
	float cone_angle, log1p_c, a, b, at, bt;
	float to_scale, from_scale; // ln 2 / log(1 + c), log(1 + c) / ln 2
};
NGP_DEV Stepping make_stepping(float cone_angle) {
	Stepping s;
	s.cone_angle = cone_angle;
	s.log1p_c = s.a = s.b = s.at = s.bt = s.to_scale = s.from_scale = 0.0f;
	if (cone_angle <= 1e-5f) return s;
	s.log1p_c = logf(1.0f + cone_angle);
	s.a = (logf(stepsize()) - logf(s.log1p_c)) / s.log1p_c;
	s.b = (logf(max_cone_stepsize()) - logf(s.log1p_c)) / s.log1p_c;
	s.at = expf(s.a * s.log1p_c);
	s.bt = expf(s.b * s.log1p_c);
	s.to_scale = 0.6931471805599453f / s.log1p_c;
	s.from_scale = s.log1p_c * 1.4426950408889634f;
	return s;
}
NGP_DEV float to_stepping_space(float t, const Stepping& s) {
	if (s.cone_angle <= 1e-5f) return div_stepsize(t);
	if (t <= s.at) return div_stepsize(t - s.at) + s.a;
#ifdef NGP_PRECISE_STEPPING
	else if (t <= s.bt) return logf(t) / s.log1p_c;
#else
	else if (t <= s.bt) return __builtin_amdgcn_logf(t) * s.to_scale;
#endif
	else return div_max_cone_stepsize(t - s.bt) + s.b;
}
NGP_DEV float from_stepping_space(float n, const Stepping& s) {
	if (s.cone_angle <= 1e-5f) return n * stepsize();
	if (n <= s.a) return (n - s.a) * stepsize() + s.at;
#ifdef NGP_PRECISE_STEPPING
	else if (n <= s.b) return expf(n * s.log1p_c);
#else
	else if (n <= s.b) return __builtin_amdgcn_exp2f(n * s.from_scale);
#endif
	else return (n - s.b) * max_cone_stepsize() + s.bt;
}
NGP_DEV float advance_n_steps(float t, const Stepping& s, float n) { return from_stepping_space(to_stepping_space(t, s) + n, s); }
NGP_DEV float calc_dt(float t, const Stepping& s) { return advance_n_steps(t, s, 1.0f) - t; }
NGP_DEV float to_stepping_space(float t, float cone_angle) { return to_stepping_space(t, make_stepping(cone_angle)); }
NGP_DEV float from_stepping_space(float n, float cone_angle) { return from_stepping_space(n, make_stepping(cone_angle)); }
NGP_DEV float advance_n_steps(float t, float cone_angle, float n) { return advance_n_steps(t, make_stepping(cone_angle), n); }
NGP_DEV float calc_dt(float t, float cone_angle) { return calc_dt(t, make_stepping(cone_angle)); }
NGP_DEV float warp_dt(float dt) { // nerf_device.cuh:306-309
	const float max_stepsize = stepsize() * (float)(1u << (NERF_CASCADES - 1));
	return div_const(dt - stepsize(), max_stepsize - stepsize(), 1.0f / (max_stepsize - stepsize()));
}
NGP_DEV float unwarp_dt(float dt) { // nerf_device.cuh:311-314
	float max_stepsize = stepsize() * (float)(1u << (NERF_CASCADES - 1));
	return dt * (max_stepsize - stepsize()) + stepsize();
}

// ---------------------------------------------------------------------------------------------------------
// Occupancy grid: nerf_device.cuh:316-367,430-447; Morton code as tcnn's morton3D.
NGP_DEV uint32_t expand_bits(uint32_t v) {
	v = (v * 0x00010001u) & 0xFF0000FFu;
	v = (v * 0x00000101u) & 0x0F00F00Fu;
	v = (v * 0x00000011u) & 0xC30C30C3u;
	v = (v * 0x00000005u) & 0x49249249u;
	return v;
}
NGP_DEV uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z) { return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2); }
NGP_DEV uint32_t morton3D_invert(uint32_t x) {
	x = x & 0x49249249u;
	x = (x | (x >> 2)) & 0xc30c30c3u;
	x = (x | (x >> 4)) & 0x0f00f00fu;
	x = (x | (x >> 8)) & 0xff0000ffu;
	x = (x | (x >> 16)) & 0x0000ffffu;
	return x;
}

NGP_DEV bool density_grid_occupied_at(f3 pos, const uint8_t* __restrict__ bitfield, uint32_t mip) {
	float mip_scale = __builtin_ldexpf(1.0f, -(int)mip);
	pos = adds3(scale3(adds3(pos, -0.5f), mip_scale), 0.5f);
	int ix = (int)(pos.x * (float)NERF_GRIDSIZE);
	int iy = (int)(pos.y * (float)NERF_GRIDSIZE);
	int iz = (int)(pos.z * (float)NERF_GRIDSIZE);
	if (ix < 0 || ix >= (int)NERF_GRIDSIZE || iy < 0 || iy >= (int)NERF_GRIDSIZE || iz < 0 || iz >= (int)NERF_GRIDSIZE) return false;
	uint32_t idx = morton3D((uint32_t)ix, (uint32_t)iy, (uint32_t)iz);
	return (bitfield[idx / 8 + (NERF_GRID_N_CELLS / 8) * mip] & (1u << (idx % 8))) != 0;
}

// The same lookup with a 4x4x4-block summary of the bitfield held in LDS (s_coarse: [mip][1024] words). In Morton
// order a 4x4x4 block is 64 consecutive cells = 8 consecutive bytes, so block = idx >> 6. An empty block answers
// "not occupied" without touching global memory; the decision (and therefore every t the march visits) is unchanged.
NGP_DEV bool density_grid_occupied_at_lds(f3 pos, const uint8_t* __restrict__ bitfield, const uint32_t* s_coarse, uint32_t mip) {
	float mip_scale = __builtin_ldexpf(1.0f, -(int)mip);
	pos = adds3(scale3(adds3(pos, -0.5f), mip_scale), 0.5f);
	int ix = (int)(pos.x * (float)NERF_GRIDSIZE);
	int iy = (int)(pos.y * (float)NERF_GRIDSIZE);
	int iz = (int)(pos.z * (float)NERF_GRIDSIZE);
	if (ix < 0 || ix >= (int)NERF_GRIDSIZE || iy < 0 || iy >= (int)NERF_GRIDSIZE || iz < 0 || iz >= (int)NERF_GRIDSIZE) return false;
	uint32_t idx = morton3D((uint32_t)ix, (uint32_t)iy, (uint32_t)iz);
	uint32_t block = idx >> 6;
	if (!((s_coarse[mip * COARSE_WORDS_PER_MIP + (block >> 5)] >> (block & 31u)) & 1u)) return false;
	return (bitfield[idx / 8 + (NERF_GRID_N_CELLS / 8) * mip] & (1u << (idx % 8))) != 0;
}

// Occupancy lookup for the march: 0 = the cell is occupied; otherwise log2 of the side (in cells of this mip) of the
// largest aligned empty block around pos that the LDS summaries can vouch for: 1 cell (-> 1), 4x4x4 (-> 4) or
// 16x16x16 (-> 16). Out-of-range positions count as a single empty cell, like density_grid_occupied_at.
// s_coarse: [mip][1024] words (4^3 blocks = morton >> 6), s_coarse16: [mip][16] words (16^3 blocks = morton >> 12).
// The 64 occupancy bits of a 4x4x4 block are one aligned 8-byte word of the Morton-ordered bitfield; a marching lane
// keeps the last block it read (a ray spends ~18 consecutive samples in one block), so most lookups touch no memory.
struct OccBlockCache {
	uint32_t key; // (mip << 26) | block, 0xffffffff = empty
	uint2 bits;
};

NGP_DEV uint32_t empty_block_size_at(f3 pos, const uint8_t* __restrict__ bitfield, const uint32_t* s_coarse, const uint32_t* s_coarse16, uint32_t mip,
                                     OccBlockCache& cache) {
	float mip_scale = __builtin_ldexpf(1.0f, -(int)mip);
	pos = adds3(scale3(adds3(pos, -0.5f), mip_scale), 0.5f);
	int ix = (int)(pos.x * (float)NERF_GRIDSIZE);
	int iy = (int)(pos.y * (float)NERF_GRIDSIZE);
	int iz = (int)(pos.z * (float)NERF_GRIDSIZE);
	if (ix < 0 || ix >= (int)NERF_GRIDSIZE || iy < 0 || iy >= (int)NERF_GRIDSIZE || iz < 0 || iz >= (int)NERF_GRIDSIZE) return 1u;
	uint32_t idx = morton3D((uint32_t)ix, (uint32_t)iy, (uint32_t)iz);
	uint32_t b16 = idx >> 12;
	if (!((s_coarse16[mip * 16u + (b16 >> 5)] >> (b16 & 31u)) & 1u)) return 16u;
	uint32_t b4 = idx >> 6;
	if (!((s_coarse[mip * COARSE_WORDS_PER_MIP + (b4 >> 5)] >> (b4 & 31u)) & 1u)) return 4u;
	const uint32_t key = (mip << 26) | b4;
	if (cache.key != key) {
		cache.bits = *(const uint2*)(bitfield + (size_t)b4 * 8 + (size_t)(NERF_GRID_N_CELLS / 8) * mip);
		cache.key = key;
	}
	const uint32_t word = (idx & 32u) ? cache.bits.y : cache.bits.x; // bit (idx % 8) of byte idx / 8 == bit (idx & 63) of the block word
	return ((word >> (idx & 31u)) & 1u) ? 0u : 1u;
}

// The same answer for the fused kernel's march, cheapest case first. A ray spends ~18 consecutive samples inside one 4x4x4 block of cells,
// so the lane keeps the block it read last (its 64 occupancy bits) under a key that needs no Morton code: equal key => the answer is
// one bit test, no LDS summary and no memory is touched. Only a lane that has moved to another block pays for the Morton code of the
// block, the two LDS summaries and -- if the block is not empty -- the 8-byte word of the bitfield.
struct OccBlock {
	uint32_t key; // (x >> 2) | (y >> 2) << 5 | (z >> 2) << 10 | mip << 15 of the held block; 0xffffffff = none
	uint2 bits;   // bit (morton & 63) = cell (x & 3, y & 3, z & 3) of the block
};
// lds_mips: the summaries of that many cascades are in LDS (s_coarse), those of the outer ones are read from g_coarse (global, a 4 KB
// table per cascade that the vector L1 keeps): the five-cascade kernel fits a third workgroup per CU that way.
NGP_DEV uint32_t occupancy_state_at(f3 pos, const uint8_t* __restrict__ bitfield, const uint32_t* s_coarse, const uint32_t* s_coarse16, uint32_t mip, OccBlock& cache,
                                    uint32_t lds_mips = NERF_CASCADES, const uint32_t* __restrict__ g_coarse = nullptr) {
	float mip_scale = __builtin_ldexpf(1.0f, -(int)mip);
	pos = adds3(scale3(adds3(pos, -0.5f), mip_scale), 0.5f);
	const int ix = (int)(pos.x * (float)NERF_GRIDSIZE);
	const int iy = (int)(pos.y * (float)NERF_GRIDSIZE);
	const int iz = (int)(pos.z * (float)NERF_GRIDSIZE);
	if (((uint32_t)ix | (uint32_t)iy | (uint32_t)iz) >= NERF_GRIDSIZE) return 1u; // (a negative coordinate has its top bit set)
	const uint32_t x = (uint32_t)ix, y = (uint32_t)iy, z = (uint32_t)iz;
	const uint32_t key = (x >> 2) | ((y >> 2) << 5) | ((z >> 2) << 10) | (mip << 15);
	if (cache.key != key) {
		const uint32_t b4 = morton3D(x >> 2, y >> 2, z >> 2), b16 = b4 >> 6; // == morton3D(x, y, z) >> 6, >> 12
		if (!((s_coarse16[mip * 16u + (b16 >> 5)] >> (b16 & 31u)) & 1u)) return 16u; // (32^3 / 64^3 blocks, tried: hardly ever empty where a 16^3 one is)
		const uint32_t cw = mip < lds_mips ? s_coarse[mip * COARSE_WORDS_PER_MIP + (b4 >> 5)] : g_coarse[mip * COARSE_WORDS_PER_MIP + (b4 >> 5)];
		if (!((cw >> (b4 & 31u)) & 1u)) return 4u;
		cache.bits = *(const uint2*)(bitfield + (size_t)b4 * 8 + (size_t)(NERF_GRID_N_CELLS / 8) * mip);
		cache.key = key;
	}
	// the Morton code of the coordinates' low two bits: bit (idx & 63) of the block word
	const uint32_t bit = (x & 1u) | ((y & 1u) << 1) | ((z & 1u) << 2) | ((x & 2u) << 2) | ((y & 2u) << 3) | ((z & 2u) << 4);
	const uint32_t word = (bit & 32u) ? cache.bits.y : cache.bits.x;
	return ((word >> (bit & 31u)) & 1u) ? 0u : 1u;
}

// The climb to coarser cascades (nerf_device.cuh:488-490) through the block summaries alone: 16 / 4 = the aligned 16^3 / 4^3 block of cells of cascade
// `mip` around pos is empty, 0 = it is not (or pos lies outside that cascade's grid). No bitfield word is read and the lane's block cache is left
// alone: with block jumps on, one empty CELL of the next cascade (two cells of this one) reaches no further than the 4^3 block the march
// already holds, so only whole empty blocks of coarser cascades can lengthen the step.
NGP_DEV uint32_t empty_block_summary_at(f3 pos, const uint32_t* s_coarse, const uint32_t* s_coarse16, uint32_t mip, uint32_t lds_mips = NERF_CASCADES,
                                        const uint32_t* __restrict__ g_coarse = nullptr) {
	float mip_scale = __builtin_ldexpf(1.0f, -(int)mip);
	pos = adds3(scale3(adds3(pos, -0.5f), mip_scale), 0.5f);
	const int ix = (int)(pos.x * (float)(NERF_GRIDSIZE / 4)), iy = (int)(pos.y * (float)(NERF_GRIDSIZE / 4)), iz = (int)(pos.z * (float)(NERF_GRIDSIZE / 4));
	if (((uint32_t)ix | (uint32_t)iy | (uint32_t)iz) >= NERF_GRIDSIZE / 4 || pos.x < 0.0f || pos.y < 0.0f || pos.z < 0.0f) return 0u;
	const uint32_t b4 = morton3D((uint32_t)ix, (uint32_t)iy, (uint32_t)iz), b16 = b4 >> 6;
	if (!((s_coarse16[mip * 16u + (b16 >> 5)] >> (b16 & 31u)) & 1u)) return 16u;
	const uint32_t cw = mip < lds_mips ? s_coarse[mip * COARSE_WORDS_PER_MIP + (b4 >> 5)] : g_coarse[mip * COARSE_WORDS_PER_MIP + (b4 >> 5)];
	return ((cw >> (b4 & 31u)) & 1u) ? 0u : 4u;
}

// res is a power of two, so t / res == t * (1/res) bit for bit; inv_res spares the IEEE division sequence
NGP_DEV float distance_to_next_voxel(f3 pos, f3 dir, f3 idir, float res, float inv_res) {
	f3 p = scale3(adds3(pos, -0.5f), res);
	float tx = (__builtin_floorf(p.x + 0.5f + 0.5f * __builtin_copysignf(1.0f, dir.x)) - p.x) * idir.x;
	float ty = (__builtin_floorf(p.y + 0.5f + 0.5f * __builtin_copysignf(1.0f, dir.y)) - p.y) * idir.y;
	float tz = (__builtin_floorf(p.z + 0.5f + 0.5f * __builtin_copysignf(1.0f, dir.z)) - p.z) * idir.z;
	float t = fminf(fminf(tx, ty), tz);
	return fmaxf(t * inv_res, 0.0f);
}
// block: 1 = the reference's voxel step; 4 / 16 = leave a whole aligned empty block of that many cells per side in
// one go. Every lattice point inside an empty block is in an empty cell, so the chain of per-voxel steps would pass
// through it without emitting a sample and leave it at the same lattice point: the first one at or after the block's
// exit (DESIGN.md "Empty-space blocks"). Only the rounding of the exit distance differs (different start point).
NGP_DEV float advance_to_next_voxel(float t, const Stepping& s, f3 pos, f3 dir, f3 idir, uint32_t mip, uint32_t block = 1u) {
	const int shift = block == 16u ? 4 : (block == 4u ? 2 : 0);
	float res = __builtin_ldexpf((float)NERF_GRIDSIZE, -(int)mip - shift);
	float t_target = t + distance_to_next_voxel(pos, dir, idir, res, __builtin_ldexpf(1.0f / (float)NERF_GRIDSIZE, (int)mip + shift));
	t = to_stepping_space(t, s);
	t_target = to_stepping_space(t_target, s);
	return from_stepping_space(t + __builtin_ceilf(fmaxf(t_target - t, 0.5f)), s);
}
NGP_DEV float advance_to_next_voxel(float t, float cone_angle, f3 pos, f3 dir, f3 idir, uint32_t mip, uint32_t block = 1u) {
	return advance_to_next_voxel(t, make_stepping(cone_angle), pos, dir, idir, mip, block);
}
// A render box larger than the occupancy grid (geometry mode: the inflated scene box, load_scene) puts marching rays
// outside the outermost cascade, where the reference steps one virtual cell at a time through space that cannot
// hold a sample. Distance along the ray from `pos` (outside the cube [0.5 - h, 0.5 + h]^3) to that cube; < 0: the ray
// misses it. Every cell on the way is empty, so one step to the entry lands on the same lattice point.
NGP_DEV float grid_cube_entry(f3 pos, f3 idir, float h) {
	const float lo = 0.5f - h, hi = 0.5f + h;
	float tx0 = (lo - pos.x) * idir.x, tx1 = (hi - pos.x) * idir.x;
	float ty0 = (lo - pos.y) * idir.y, ty1 = (hi - pos.y) * idir.y;
	float tz0 = (lo - pos.z) * idir.z, tz1 = (hi - pos.z) * idir.z;
	float tmin = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fminf(tz0, tz1));
	float tmax = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
	tmin = fmaxf(tmin, 0.0f);
	return tmax >= tmin ? tmin : -1.0f;
}
// the whole-step rounding of advance_to_next_voxel for a given distance to the target
NGP_DEV float advance_by_distance(float t, const Stepping& s, float distance) {
	float t_target = t + distance;
	t = to_stepping_space(t, s);
	t_target = to_stepping_space(t_target, s);
	return from_stepping_space(t + __builtin_ceilf(fmaxf(t_target - t, 0.5f)), s);
}
NGP_DEV float advance_by_distance(float t, float cone_angle, float distance) { return advance_by_distance(t, make_stepping(cone_angle), distance); }
NGP_DEV uint32_t mip_from_pos(f3 pos, uint32_t max_cascade) {
	int exponent;
	float maxval = fmaxf(fmaxf(__builtin_fabsf(pos.x - 0.5f), __builtin_fabsf(pos.y - 0.5f)), __builtin_fabsf(pos.z - 0.5f));
	(void)__builtin_frexpf(maxval, &exponent);
	int v = exponent + 1;
	v = v < 0 ? 0 : v;
	v = v > (int)max_cascade ? (int)max_cascade : v;
	return (uint32_t)v;
}

NGP_DEV bool raabb_contains(const ModelParams& M, f3 p) {
	if (!M.r2l_identity) p = m3_mulv(M.r2l, p); // identity * p == p exactly, so skipping the product changes nothing
	return p.x >= M.raabb_min[0] && p.x <= M.raabb_max[0] && p.y >= M.raabb_min[1] && p.y <= M.raabb_max[1] && p.z >= M.raabb_min[2] && p.z <= M.raabb_max[2];
}

// if_unoccupied_advance_to_next_occupied_voxel, nerf_device.cuh:461-494 (CAPPED: the 200-iteration variant :497-534)
template <bool CAPPED>
NGP_DEV float skip_empty_space(float t, const ModelParams& M, f3 o, f3 d, f3 idir) {
	const float cone_angle = M.cone_angle;
	uint32_t i = 1;
	while (!CAPPED || i < 200) {
		f3 pos = add3(o, scale3(d, t));
		if (t >= MAX_DEPTH || !raabb_contains(M, pos)) return MAX_DEPTH;
		uint32_t mip = mip_from_pos(pos, NERF_CASCADES - 1);
		mip = mip > M.max_cascade ? M.max_cascade : mip; // clamp(mip, min_mip = 0, max_mip)
		if (density_grid_occupied_at(pos, M.bitfield, mip)) return t;
		while (mip < M.max_cascade && !density_grid_occupied_at(pos, M.bitfield, mip + 1)) ++mip;
		t = advance_to_next_voxel(t, cone_angle, pos, d, idir, mip);
		++i;
	}
	return MAX_DEPTH;
}

// BoundingBox::ray_intersect, bounding_box.cuh:172-219 (entry distance only; FLT_MAX on a miss)
NGP_DEV float aabb_ray_entry(const float* bmin, const float* bmax, f3 pos, f3 dir) {
	const float FMAX = 3.402823466e+38f;
	float tmin = (bmin[0] - pos.x) / dir.x;
	float tmax = (bmax[0] - pos.x) / dir.x;
	if (tmin > tmax) { float s = tmin; tmin = tmax; tmax = s; }
	float tymin = (bmin[1] - pos.y) / dir.y;
	float tymax = (bmax[1] - pos.y) / dir.y;
	if (tymin > tymax) { float s = tymin; tymin = tymax; tymax = s; }
	if (tmin > tymax || tymin > tmax) return FMAX;
	if (tymin > tmin) tmin = tymin;
	if (tymax < tmax) tmax = tymax;
	float tzmin = (bmin[2] - pos.z) / dir.z;
	float tzmax = (bmax[2] - pos.z) / dir.z;
	if (tzmin > tzmax) { float s = tzmin; tzmin = tzmax; tzmax = s; }
	if (tmin > tzmax || tzmin > tmax) return FMAX;
	if (tzmin > tmin) tmin = tzmin;
	return tmin;
}

// ---------------------------------------------------------------------------------------------------------
// K1 + K2: init_rays_with_payload_kernel_nerf (testbed_nerf.cu:1428-1544, perspective branch of uv_to_ray,
// common_device.cuh:416-483) followed by advance_pos_nerf (:333-362).
struct RayState {
	f3 o, d;
	float t;
	uint32_t idx; // pixel index x + W*y (seeds the start-of-ray jitter); probe rays: ray id
	uint32_t out; // where the ray's pixel lives in frame_buffer / depth_buffer (== idx unless the layout is tile-packed)
	bool alive;
};

// ---- lenses of uv_to_ray (common_device.cuh:249-338, 375-391, 441-462)
NGP_DEV void opencv_lens_distortion_delta(const float* q, float u, float v, float* du, float* dv) {
	const float k1 = q[0], k2 = q[1], p1 = q[2], p2 = q[3];
	const float u2 = u * u, uv = u * v, v2 = v * v, r2 = u2 + v2;
	const float radial = k1 * r2 + k2 * r2 * r2;
	*du = u * radial + 2.0f * p1 * uv + p2 * (r2 + 2.0f * u2);
	*dv = v * radial + 2.0f * p2 * uv + p1 * (r2 + 2.0f * v2);
}
NGP_DEV void opencv_fisheye_lens_distortion_delta(const float* q, float u, float v, float* du, float* dv) {
	const float k1 = q[0], k2 = q[1], k3 = q[2], k4 = q[3];
	const float r = __builtin_sqrtf(u * u + v * v);
	if (r > (float)2.220446049250313e-16) {
		const float theta = atanf(r);
		const float theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
		const float thetad = theta * (1.0f + k1 * theta2 + k2 * theta4 + k3 * theta6 + k4 * theta8);
		*du = u * thetad / r - u;
		*dv = v * thetad / r - v;
	} else {
		*du = 0.0f;
		*dv = 0.0f;
	}
}
// iterative_lens_undistortion: Newton iteration with central differences, at most 100 steps
template <bool FISHEYE>
NGP_DEV void iterative_lens_undistortion(const float* q, float* u, float* v) {
	auto delta = [&](float a, float b, float* da, float* db) {
		if (FISHEYE) opencv_fisheye_lens_distortion_delta(q, a, b, da, db);
		else opencv_lens_distortion_delta(q, a, b, da, db);
	};
	const float x0 = *u, y0 = *v;
	float x = x0, y = y0;
	for (uint32_t i = 0; i < 100u; ++i) {
		const float step0 = fmaxf(1.1920929e-07f, __builtin_fabsf(1e-6f * x));
		const float step1 = fmaxf(1.1920929e-07f, __builtin_fabsf(1e-6f * y));
		float dx0, dx1, b0, b1, f0, f1, c0, c1, g0, g1;
		delta(x, y, &dx0, &dx1);
		delta(x - step0, y, &b0, &b1);
		delta(x + step0, y, &f0, &f1);
		delta(x, y - step1, &c0, &c1);
		delta(x, y + step1, &g0, &g1);
		// J is column-major in the reference: J[col][row]
		const float j00 = 1.0f + (f0 - b0) / (2.0f * step0), j10 = (g0 - c0) / (2.0f * step1);
		const float j01 = (f1 - b1) / (2.0f * step0), j11 = 1.0f + (g1 - c1) / (2.0f * step1);
		const float rx = x + dx0 - x0, ry = y + dx1 - y0;
		// inverse(J) * r with J = [[j00, j10], [j01, j11]] (row, col)
		const float det = j00 * j11 - j10 * j01;
		const float sx = (j11 * rx - j10 * ry) / det, sy = (-j01 * rx + j00 * ry) / det;
		x -= sx;
		y -= sy;
		if (sx * sx + sy * sy < 1e-10f) break;
	}
	*u = x;
	*v = y;
}
// the camera-space direction of uv under the frame's lens; false: no ray for this pixel
// (out of line: the perspective case must not carry the Newton iteration's registers into the persistent kernel)
__device__ __attribute__((noinline)) void lens_direction_general(int lens_mode, const float* lens_params, float u, float v, float sx, float sy, float* out) {
	const float PI = 3.14159265358979323846f;
	f3 dir;
	if (lens_mode == 3) { // LatLong
		float theta = (v - 0.5f) * PI, phi = (u - 0.5f) * PI * 2.0f;
		dir = mk3(sinf(phi) * cosf(theta), sinf(theta), cosf(phi) * cosf(theta));
	} else if (lens_mode == 2) { // FTheta: f_theta_undistortion(uv - screen_center, params, {0, 0, 0}) (common_device.cuh:361-375); sx, sy carry uv - screen_center here
		float xpix = sx * lens_params[5], ypix = sy * lens_params[6];
		float norm = __builtin_sqrtf(xpix * xpix + ypix * ypix);
		float alpha = lens_params[0] + norm * (lens_params[1] + norm * (lens_params[2] + norm * (lens_params[3] + norm * lens_params[4])));
		float sin_alpha = sinf(alpha), cos_alpha = cosf(alpha);
		if (cos_alpha <= 1.17549435e-38f || norm == 0.f) {
			dir = mk3(0.f, 0.f, 0.f); // Ray::invalid(): the zero direction marks the pixel dead downstream
		} else {
			sin_alpha *= 1.f / norm;
			dir = mk3(sin_alpha * xpix, sin_alpha * ypix, cos_alpha);
		}
	} else if (lens_mode == 5) { // Equirectangular
		float ct = (v - 0.5f) * 2.0f;
		float st = __builtin_sqrtf(fmaxf(1.0f - ct * ct, 0.0f));
		float phi = (u - 0.5f) * PI * 2.0f;
		dir = mk3(sinf(phi) * st, ct, cosf(phi) * st);
	} else {
		dir = mk3(sx, sy, 1.0f);
		if (lens_mode == 1) iterative_lens_undistortion<false>(lens_params, &dir.x, &dir.y);
		else if (lens_mode == 4) iterative_lens_undistortion<true>(lens_params, &dir.x, &dir.y);
	}
	out[0] = dir.x; out[1] = dir.y; out[2] = dir.z;
}
NGP_DEV bool lens_direction(const CameraParams& C, float u, float v, f3& dir) {
	dir = mk3((u - C.screen_center[0]) * (float)C.width / C.focal[0], (v - C.screen_center[1]) * (float)C.height / C.focal[1], 1.0f);
	if (C.lens_mode != 0) {
		float q[7], o[3];
		for (int i = 0; i < 7; ++i) q[i] = C.lens_params[i];
		if (C.lens_mode == 2) lens_direction_general(C.lens_mode, q, u, v, u - C.screen_center[0], v - C.screen_center[1], o);
		else lens_direction_general(C.lens_mode, q, u, v, dir.x, dir.y, o);
		dir = mk3(o[0], o[1], o[2]);
	}
	return true;
}
// depth of field of uv_to_ray (common_device.cuh:471-477): the origin moves on the lens disk (square2disk_shirley of a
// 2-d Sobol point, random_val.cuh:112-128, 311-330), the direction keeps the focus point. Out of line, like the lenses.
__device__ __attribute__((noinline)) void apply_aperture(const float* cam_m, float aperture_size, float focus_z, uint32_t spp, uint32_t px_seed, float* origin3, float* dir3) {
	// ld_random_val_2d(spp, seed): Owen-scrambled Sobol, dimensions 0 (bit reversal) and 1 (v_k = v_{k-1} ^ (v_{k-1} >> 1))
	uint32_t index = nested_uniform_scramble_base2(spp, px_seed);
	uint32_t s0 = __builtin_bitreverse32(index), s1 = 0, v = 0x80000000u;
	for (uint32_t bit = 0; bit < 32u; ++bit) {
		if ((index >> bit) & 1u) s1 ^= v;
		v ^= v >> 1;
	}
	const float sx = (float)nested_uniform_scramble_base2(s0, hash_combine(px_seed, 0u)) * 2.3283064365386963e-10f * 2.0f - 1.0f;
	const float sy = (float)nested_uniform_scramble_base2(s1, hash_combine(px_seed, 1u)) * 2.3283064365386963e-10f * 2.0f - 1.0f;
	const float PI = 3.14159265358979323846f;
	float r, phi;
	if (sx * sx > sy * sy) {
		r = sx;
		phi = (PI / 4.0f) * (sy / sx);
	} else {
		r = sy;
		phi = (PI / 2.0f) - (PI / 4.0f) * (sx / sy);
	}
	const float bx = aperture_size * (r * cosf(phi)), by = aperture_size * (r * sinf(phi));
	f3 origin = mk3(origin3[0], origin3[1], origin3[2]), dir = mk3(dir3[0], dir3[1], dir3[2]);
	const f3 lookat = add3(origin, scale3(dir, focus_z));
	origin = add3(origin, add3(scale3(mk3(cam_m[0], cam_m[1], cam_m[2]), bx), scale3(mk3(cam_m[3], cam_m[4], cam_m[5]), by))); // mat2x3(camera) * blur
	dir = mk3((lookat.x - origin.x) / focus_z, (lookat.y - origin.y) / focus_z, (lookat.z - origin.z) / focus_z);
	origin3[0] = origin.x; origin3[1] = origin.y; origin3[2] = origin.z;
	dir3[0] = dir.x; dir3[1] = dir.y; dir3[2] = dir.z;
}

// get_xform_given_rolling_shutter (common_device.cuh:656-659): the camera of ONE pixel of a frame whose camera moves from
// camera0 to camera1 -- pixel_t = rs.x + rs.y u + rs.z v + rs.w motionblur_time, camera_slerp(camera0, camera1, pixel_t)
// (:651-654: rotation slerp, position mix). tcnn's slerp(mat3, mat3, t) (vec.h, not in the reference mount) is the GLM-derived
// trio quat_cast -> slerp -> normalize -> mat3_cast, restated here. Out of line: only moving frames come here.
__device__ __attribute__((noinline)) void camera_at_pixel(const float* m0, const float* m1, const float* rs, float u, float v, float motionblur_time, float* out12) {
	const float t = rs[0] + rs[1] * u + rs[2] * v + rs[3] * motionblur_time;
	auto quat_cast = [](const float* m, float* q /* w x y z */) { // m[3 c + r]
		const float m00 = m[0], m01 = m[1], m02 = m[2], m10 = m[3], m11 = m[4], m12 = m[5], m20 = m[6], m21 = m[7], m22 = m[8];
		const float fx = m00 - m11 - m22, fy = m11 - m00 - m22, fz = m22 - m00 - m11, fw = m00 + m11 + m22;
		int biggest = 0;
		float fb = fw;
		if (fx > fb) { fb = fx; biggest = 1; }
		if (fy > fb) { fb = fy; biggest = 2; }
		if (fz > fb) { fb = fz; biggest = 3; }
		const float bv = __builtin_sqrtf(fb + 1.0f) * 0.5f, mult = 0.25f / bv;
		if (biggest == 0) { q[0] = bv; q[1] = (m12 - m21) * mult; q[2] = (m20 - m02) * mult; q[3] = (m01 - m10) * mult; }
		else if (biggest == 1) { q[0] = (m12 - m21) * mult; q[1] = bv; q[2] = (m01 + m10) * mult; q[3] = (m20 + m02) * mult; }
		else if (biggest == 2) { q[0] = (m20 - m02) * mult; q[1] = (m01 + m10) * mult; q[2] = bv; q[3] = (m12 + m21) * mult; }
		else { q[0] = (m01 - m10) * mult; q[1] = (m20 + m02) * mult; q[2] = (m12 + m21) * mult; q[3] = bv; }
	};
	float qa[4], qb[4], q[4];
	quat_cast(m0, qa);
	quat_cast(m1, qb);
	float cos_theta = ((qa[0] * qb[0] + qa[1] * qb[1]) + qa[2] * qb[2]) + qa[3] * qb[3];
	if (cos_theta < 0.0f) { // the short way round
		for (int i = 0; i < 4; ++i) qb[i] = -qb[i];
		cos_theta = -cos_theta;
	}
	if (cos_theta > 1.0f - 1.1920929e-07f) {
		for (int i = 0; i < 4; ++i) q[i] = qa[i] + (qb[i] - qa[i]) * t;
	} else {
		const float angle = acosf(cos_theta);
		const float sa = sinf((1.0f - t) * angle), sb = sinf(t * angle), s = sinf(angle);
		for (int i = 0; i < 4; ++i) q[i] = (sa * qa[i] + sb * qb[i]) / s;
	}
	const float len = __builtin_sqrtf(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
	const float w = q[0] / len, x = q[1] / len, y = q[2] / len, z = q[3] / len;
	const float xx = x * x, yy = y * y, zz = z * z, xz = x * z, xy = x * y, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
	out12[0] = 1.0f - 2.0f * (yy + zz); out12[1] = 2.0f * (xy + wz); out12[2] = 2.0f * (xz - wy);
	out12[3] = 2.0f * (xy - wz); out12[4] = 1.0f - 2.0f * (xx + zz); out12[5] = 2.0f * (yz + wx);
	out12[6] = 2.0f * (xz + wy); out12[7] = 2.0f * (yz - wx); out12[8] = 1.0f - 2.0f * (xx + yy);
	for (int i = 0; i < 3; ++i) out12[9 + i] = m0[9 + i] * (1.0f - t) + m1[9 + i] * t; // mix(a, b, t) = a (1 - t) + b t
}

// read_envmap (envmap.cuh:24-50): bilinear lat-long lookup of the environment map behind the NeRF, direction ->
// dir_to_spherical_unorm({d.z, -d.x, d.y}) (random_val.cuh:62-72); x wraps, y clamps. Out of line: acosf / atan2f must not
// carry their registers into the persistent kernel, which only comes here when an environment map is set.
__device__ __attribute__((noinline)) void read_envmap(const float4* __restrict__ envmap, int res_x, int res_y, const float* dir3, float* out4) {
	const float PI = 3.14159265358979323846f;
	const float dx = dir3[2], dy = -dir3[0], dz = dir3[1];
	const float cos_theta = fminf(fmaxf(dz, -1.0f), 1.0f);
	const float theta = acosf(cos_theta);
	const float phi = atan2f(dy, dx);
	const float cyl_x = theta / PI, cyl_y = phi / (2.0f * PI) + 0.5f;
	const float fx = cyl_y * (float)(res_x - 1), fy = cyl_x * (float)(res_y - 1);
	const int tx = (int)fx, ty = (int)fy;
	const float wx = fx - (float)tx, wy = fy - (float)ty;
	auto read_val = [&](int px, int py) {
		if (px < 0) px += res_x;
		else if (px >= res_x) px -= res_x;
		py = py > res_y - 1 ? res_y - 1 : py;
		py = py < 0 ? 0 : py;
		return envmap[px + (size_t)py * res_x];
	};
	const float4 v00 = read_val(tx, ty), v10 = read_val(tx + 1, ty), v01 = read_val(tx, ty + 1), v11 = read_val(tx + 1, ty + 1);
	const float w00 = (1 - wx) * (1 - wy), w10 = wx * (1 - wy), w01 = (1 - wx) * wy, w11 = wx * wy;
	out4[0] = ((w00 * v00.x + w10 * v10.x) + w01 * v01.x) + w11 * v11.x;
	out4[1] = ((w00 * v00.y + w10 * v10.y) + w01 * v01.y) + w11 * v11.y;
	out4[2] = ((w00 * v00.z + w10 * v10.z) + w01 * v01.z) + w11 * v11.z;
	out4[3] = ((w00 * v00.w + w10 * v10.w) + w01 * v01.w) + w11 * v11.w;
}

// PLAIN: a static pinhole camera without depth of field (what a benchmark / screenshot frame is): the instantiation carries none of
// the lens, aperture and moving-camera code, whose out-of-line calls and per-lane arrays cost the persistent kernel registers
template <bool PLAIN = false>
NGP_DEV void init_ray(const ModelParams& M, const CameraParams& C, uint32_t x, uint32_t y, RayState& r) {
	r.idx = x + (uint32_t)C.width * y;
	r.out = r.idx;
	float u = ((float)x + C.pixel_offset[0]) / (float)C.width;
	float v = ((float)y + C.pixel_offset[1]) / (float)C.height;
	f3 dir;
	if (PLAIN) dir = mk3((u - C.screen_center[0]) * (float)C.width / C.focal[0], (v - C.screen_center[1]) * (float)C.height / C.focal[1], 1.0f);
	else lens_direction(C, u, v, dir);
	f3 origin;
	if (PLAIN || !C.moving) { // (the static frame reads the camera from the kernel arguments: no per-lane copy of the matrix)
		dir = m3_mulv(C.m, dir);
		origin = mk3(C.m[9], C.m[10], C.m[11]);
		if (!PLAIN && C.aperture_size != 0.0f) {
			float o3[3] = {origin.x, origin.y, origin.z}, d3[3] = {dir.x, dir.y, dir.z}, cm[6];
			for (int i = 0; i < 6; ++i) cm[i] = C.m[i];
			// px = ivec2(uv * resolution)
			apply_aperture(cm, C.aperture_size, C.focus_z, C.spp, (uint32_t)(int)(u * (float)C.width) * 19349663u + (uint32_t)(int)(v * (float)C.height) * 96925573u, o3, d3);
			origin = mk3(o3[0], o3[1], o3[2]);
			dir = mk3(d3[0], d3[1], d3[2]);
		}
	} else { // the camera of this pixel (rolling shutter / motion blur): init_rays_with_payload_kernel_nerf, src/testbed_nerf.cu:1468
		float cam[12], m0[12], m1[12], rs[4];
		for (int i = 0; i < 12; ++i) { m0[i] = C.m[i]; m1[i] = C.m1[i]; }
		for (int i = 0; i < 4; ++i) rs[i] = C.rolling_shutter[i];
		camera_at_pixel(m0, m1, rs, u, v, ld_random_val_dim0(C.spp, r.idx * 72239731u), cam);
		dir = m3_mulv(cam, dir);
		origin = mk3(cam[9], cam[10], cam[11]);
		if (C.aperture_size != 0.0f) {
			float o3[3] = {origin.x, origin.y, origin.z}, d3[3] = {dir.x, dir.y, dir.z};
			apply_aperture(cam, C.aperture_size, C.focus_z, C.spp, (uint32_t)(int)(u * (float)C.width) * 19349663u + (uint32_t)(int)(v * (float)C.height) * 96925573u, o3, d3);
			origin = mk3(o3[0], o3[1], o3[2]);
			dir = mk3(d3[0], d3[1], d3[2]);
		}
	}
	origin = add3(origin, scale3(dir, C.near_distance));
	r.o = origin;
	r.d = mk3(0.f, 0.f, 0.f);
	r.t = 0.f;
	r.alive = false;
	if (dir.x == 0.0f && dir.y == 0.0f && dir.z == 0.0f) return;
	dir = normalize3(dir);
	r.d = dir; // (a ray that misses the render box still has a direction: the environment map behind it is looked up along it)
	float t = fmaxf(aabb_ray_entry(M.raabb_min, M.raabb_max, m3_mulv(M.r2l, origin), m3_mulv(M.r2l, dir)), 0.0f) + 1e-6f;
	if (!raabb_contains(M, add3(origin, scale3(dir, t)))) return;
	r.t = t;
	r.alive = true;
}

NGP_DEV void advance_pos(const ModelParams& M, const CameraParams& C, RayState& r) {
	if (!r.alive) return;
	f3 idir = mk3(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
	float t = advance_n_steps(r.t, M.cone_angle, ld_random_val_dim0(C.spp, r.idx * 786433u));
	t = skip_empty_space<false>(t, M, r.o, r.d, idir);
	if (t >= MAX_DEPTH) r.alive = false;
	else r.t = t;
}

// ---------------------------------------------------------------------------------------------------------
// K5a: two levels of the multiresolution hash grid for one sample, landing directly in MFMA B-operand order.
// tcnn kernel_grid semantics (pos = fma(scale, x, 0.5); 8 corners, bit d of the corner index selects +1 along
// dimension d; result[f] += (half)(weight * (float)value[f]) accumulated in fp16).
struct CornerSet {
	uint32_t index[8];
};

struct CellPos {
	uint32_t gx, gy, gz;
	float wx, wy, wz;
};
NGP_DEV CellPos level_cell(const LevelInfo& L, float x, float y, float z) {
	float fx = __builtin_fmaf(L.scale, x, 0.5f), fy = __builtin_fmaf(L.scale, y, 0.5f), fz = __builtin_fmaf(L.scale, z, 0.5f);
	float flx = __builtin_floorf(fx), fly = __builtin_floorf(fy), flz = __builtin_floorf(fz);
	CellPos c;
	c.gx = (uint32_t)(int)flx; c.gy = (uint32_t)(int)fly; c.gz = (uint32_t)(int)flz;
	c.wx = fx - flx; c.wy = fy - fly; c.wz = fz - flz;
	return c;
}
NGP_DEV void corner_weights(const CellPos& p, float* weight) {
	float wx0 = 1.0f - p.wx, wy0 = 1.0f - p.wy, wz0 = 1.0f - p.wz;
#pragma unroll
	for (int c = 0; c < 8; ++c) {
		int bx = c & 1, by = (c >> 1) & 1, bz = (c >> 2) & 1;
		weight[c] = ((bx ? p.wx : wx0) * (by ? p.wy : wy0)) * (bz ? p.wz : wz0);
	}
}
NGP_DEV bool level_in_xor_range(const LevelInfo& L, const CellPos& p) {
	uint32_t m = p.gx > p.gy ? p.gx : p.gy;
	m = m > p.gz ? m : p.gz;
	return m <= L.coord_max && !L.xor_disabled;
}

// tcnn grid_index, byte offsets into ModelParams::grid -- any position, any level shape
NGP_DEV void level_corners(const LevelInfo& L, const CellPos& p, CornerSet& cs) {
	uint32_t ix[2], iy[2], iz[2];
	if (L.hashed) {
		ix[0] = p.gx;               ix[1] = p.gx + 1u;
		iy[0] = p.gy * 2654435761u; iy[1] = (p.gy + 1u) * 2654435761u;
		iz[0] = p.gz * 805459861u;  iz[1] = (p.gz + 1u) * 805459861u;
	} else {
		uint32_t r2 = L.res * L.res;
		ix[0] = p.gx;         ix[1] = p.gx + 1u;
		iy[0] = p.gy * L.res; iy[1] = (p.gy + 1u) * L.res;
		iz[0] = p.gz * r2;    iz[1] = (p.gz + 1u) * r2;
	}
#pragma unroll
	for (int c = 0; c < 8; ++c) {
		int bx = c & 1, by = (c >> 1) & 1, bz = (c >> 2) & 1;
		uint32_t idx = L.hashed ? (ix[bx] ^ iy[by] ^ iz[bz]) : (ix[bx] + iy[by] + iz[bz]);
		idx = L.mask ? (idx & L.mask) : (idx % L.size);
		cs.index[c] = (L.offset + idx) * 8u; // byte offset of the 4 x fp16 entry
	}
}

// The same 8 entries through the xor layout (byte offsets into ModelParams::xgrid): 2 multiplies, 3 adds and per
// corner one v_xor3 + one v_and_or, for dense and hashed levels alike. Only valid when level_in_xor_range().
NGP_DEV void level_corners_xor(const LevelInfo& L, const CellPos& p, CornerSet& cs) {
	uint32_t ix[2], iy[2], iz[2];
	ix[0] = p.gx << 3;        ix[1] = ix[0] + 8u;
	iy[0] = p.gy * L.mul_y8;  iy[1] = iy[0] + L.mul_y8;
	iz[0] = p.gz * L.mul_z8;  iz[1] = iz[0] + L.mul_z8;
#pragma unroll
	for (int c = 0; c < 8; ++c) {
		int bx = c & 1, by = (c >> 1) & 1, bz = (c >> 2) & 1;
		cs.index[c] = ((ix[bx] ^ iy[by] ^ iz[bz]) & L.mask8) | L.base8;
	}
}

// Gathers are buffer loads (a 128-bit resource descriptor in SGPRs + a 32-bit per-lane byte offset) rather than flat
// loads from a 64-bit address: no 64-bit add per gather, and an out-of-range offset reads zeros instead of faulting.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t GridRsrc;
NGP_DEV GridRsrc make_grid_rsrc(const void* table, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(table), (short)0, (int)bytes, 0x00020000); }
NGP_DEV uint2 gather8(GridRsrc r, uint32_t offset) {
	const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)offset, 0, 0);
	return make_uint2(v.x, v.y);
}

// The corner sum of tcnn's kernel_grid. tiny-cuda-nn is an un-pinned, un-vendored submodule of the reference
// (.gitmodules:13-15) and has published two sequences:
//   tvec era (the tcnn that has tcnn::vec3 / mat4x3, which the reference's sources use throughout):
//       result = fma((T)weight, grid_val(pos_grid_local), result);
//     the weight is rounded to fp16 once, then one fp16 fma (single rounding) per feature. This is what ships:
//     v_cvt_f16_f32 + two v_pk_fma_f16 per corner, bit for bit the oracle's ORC_GRID_ACC_FMA.
//   before the tvec refactor (-DNGP_TCNN_LEGACY_ENCODE, libngp_hip_legacy.so, oracle ORC_GRID_ACC_LEGACY):
//       result[f] += (T)(weight * (float)val[f]);
//     the fp32 product is rounded to fp32, THEN to fp16, then added in fp16. hipcc would fuse the first two steps into
//     v_fma_mix*_f16, which rounds the exact product once -- a different number in ~2e-5 of the cases
//     (tools/micro/mix_probe.hip) -- so the product is formed as an fp32 value of its own (v_fma_mix_f32) and the
//     features are accumulated as packed pairs (v_cvt_pk_f16_f32 + v_pk_add_f16): 8 instructions per corner.
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
struct FeatureAcc {
	half2_t f01, f23;
};
// fp32 product of an fp32 weight and one half of a packed fp16 pair, rounded to fp32: v_fma_mix_f32 reads the fp16
// operand directly (no v_cvt_f32_f16) and, with an fp32 destination, rounds exactly like v_mul_f32
NGP_DEV float product_lo(float w, uint32_t packed) {
	float p;
	asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[0,1,0]" : "=v"(p) : "v"(w), "v"(packed));
	return p;
}
NGP_DEV float product_hi(float w, uint32_t packed) {
	float p;
	asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(p) : "v"(w), "v"(packed));
	return p;
}
NGP_DEV void accumulate_corner(uint2 v, float w, FeatureAcc& r) {
#ifndef NGP_TCNN_LEGACY_ENCODE
	const half2_t wh = {(half_t)w, (half_t)w};
	const uint32_t wv = __builtin_bit_cast(uint32_t, wh);
	uint32_t a01 = __builtin_bit_cast(uint32_t, r.f01), a23 = __builtin_bit_cast(uint32_t, r.f23);
	asm("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(a01) : "v"(wv), "v"(v.x));
	asm("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(a23) : "v"(wv), "v"(v.y));
	r.f01 = __builtin_bit_cast(half2_t, a01);
	r.f23 = __builtin_bit_cast(half2_t, a23);
#else
	half2_t a, b;
	a[0] = (half_t)product_lo(w, v.x);
	a[1] = (half_t)product_hi(w, v.x);
	b[0] = (half_t)product_lo(w, v.y);
	b[1] = (half_t)product_hi(w, v.y);
	r.f01 = r.f01 + a;
	r.f23 = r.f23 + b;
#endif
}
NGP_DEV void store_features(const FeatureAcc& lo, const FeatureAcc& hi, half8& out) {
	out[0] = lo.f01[0]; out[1] = lo.f01[1]; out[2] = lo.f23[0]; out[3] = lo.f23[1];
	out[4] = hi.f01[0]; out[5] = hi.f01[1]; out[6] = hi.f23[0]; out[7] = hi.f23[1];
}

// lane (h, c) of a 16-sample pass encodes levels h and h+4: B-fragment element j<4 is feature j of level h,
// element j>=4 is feature j-4 of level h+4 (the K permutation n(s,h,j) = 32s + 16(j>>2) + 4h + (j&3) that the
// host applied to every weight matrix, see ngp_api.cpp build_weight_fragments).
// The encode is split in two so that a wave can keep the gathers of two 16-sample passes (32 loads per lane) in
// flight before it consumes either: issue computes the 16 addresses and starts the loads, finish forms the 16
// trilinear weights (from the 6 cell fractions kept meanwhile) and accumulates.
struct EncodeInFlight {
	uint2 v[16];
	float wx[2], wy[2], wz[2];
};
NGP_DEV void encode_issue(GridRsrc grid, GridRsrc xgrid, const LevelInfo* lv, int h, float x, float y, float z, EncodeInFlight& e) {
	const LevelInfo& L0 = lv[h];
	const LevelInfo& L1 = lv[h + 4];
	const CellPos p0 = level_cell(L0, x, y, z), p1 = level_cell(L1, x, y, z);
	CornerSet c0, c1;
	// Every render sample lies in the xor layout's range; positions outside [0, 1] (possible through ngp_grid_encode / a
	// render box larger than the training box) take the tcnn-order table for the whole wave. 32-bit byte offsets: both
	// tables are below 2 GiB (checked by the host).
	if (__all((int)level_in_xor_range(L0, p0) & (int)level_in_xor_range(L1, p1))) {
		level_corners_xor(L0, p0, c0);
		level_corners_xor(L1, p1, c1);
#pragma unroll
		for (int c = 0; c < 8; ++c) e.v[c] = gather8(xgrid, c0.index[c]);
#pragma unroll
		for (int c = 0; c < 8; ++c) e.v[8 + c] = gather8(xgrid, c1.index[c]);
#if defined(NGP_EXPERIMENT_EXTRA_GATHERS) // timing experiment only (profiles/r3_extra_gathers.txt): N more lane-loads per lane and pass, of lines the loads above have just touched; results unchanged
#pragma unroll
		for (int c = 0; c < NGP_EXPERIMENT_EXTRA_GATHERS; ++c) {
			const uint2 d = gather8(xgrid, c & 1 ? c1.index[c & 7] : c0.index[c & 7]);
			asm volatile("" ::"v"(d.x), "v"(d.y));
		}
#endif
	} else {
		level_corners(L0, p0, c0);
		level_corners(L1, p1, c1);
#pragma unroll
		for (int c = 0; c < 8; ++c) e.v[c] = gather8(grid, c0.index[c]);
#pragma unroll
		for (int c = 0; c < 8; ++c) e.v[8 + c] = gather8(grid, c1.index[c]);
	}
	e.wx[0] = p0.wx; e.wy[0] = p0.wy; e.wz[0] = p0.wz;
	e.wx[1] = p1.wx; e.wy[1] = p1.wy; e.wz[1] = p1.wz;
}
NGP_DEV half8 encode_finish(const EncodeInFlight& e) {
	FeatureAcc acc[2] = {{{0, 0}, {0, 0}}, {{0, 0}, {0, 0}}};
#pragma unroll
	for (int l = 0; l < 2; ++l) {
		CellPos p;
		p.gx = p.gy = p.gz = 0;
		p.wx = e.wx[l]; p.wy = e.wy[l]; p.wz = e.wz[l];
		float w[8];
		corner_weights(p, w);
#pragma unroll
		for (int c = 0; c < 8; ++c) accumulate_corner(e.v[8 * l + c], w[c], acc[l]);
	}
	half8 out;
	store_features(acc[0], acc[1], out);
	return out;
}
NGP_DEV half8 encode_level_pair(GridRsrc grid, GridRsrc xgrid, const LevelInfo* lv, int h, float x, float y, float z) {
	EncodeInFlight e;
	encode_issue(grid, xgrid, lv, h, x, y, z, e);
	return encode_finish(e);
}

// ---------------------------------------------------------------------------------------------------------
// K5c: tcnn SphericalHarmonics degree 4 -- the four coefficients 4h..4h+3 that lane group h feeds to the rgb head.
NGP_DEV void sh4_quad(int h, float dx01, float dy01, float dz01, float* out4) {
	float x = dx01 * 2.0f - 1.0f, y = dy01 * 2.0f - 1.0f, z = dz01 * 2.0f - 1.0f;
	float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
	float o[16];
	o[0] = 0.28209479177387814f;
	o[1] = -0.48860251190291987f * y;
	o[2] = 0.48860251190291987f * z;
	o[3] = -0.48860251190291987f * x;
	o[4] = 1.0925484305920792f * xy;
	o[5] = -1.0925484305920792f * yz;
	o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
	o[7] = -1.0925484305920792f * xz;
	o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
	o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
	o[10] = 2.8906114426405538f * xy * z;
	o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
	o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
	o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
	o[14] = 1.4453057213202769f * z * (x2 - y2);
	o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		float a = h == 0 ? o[j] : o[4 + j];
		float b = h == 2 ? o[8 + j] : o[12 + j];
		out4[j] = h < 2 ? a : b;
	}
}

// ---------------------------------------------------------------------------------------------------------
// K5b/K5d: the two fully fused MLPs for 16 samples on one wave. Samples sit on the MFMA N axis (lane & 15),
// neurons on the M axis, so every layer's fp32 accumulator tile D[neuron = 4h + r][sample = c] is, after ReLU
// and a cvt to fp16, already the next layer's B operand -- no LDS transpose, no cross-lane traffic
// (cdna_hip_programming.md "An accumulator tile as the next MFMA's operand").
NGP_DEV half8 ld_frag(const uint4* s_w, int f, int lane) {
	union { uint4 u; half8 h; } cv;
	cv.u = s_w[f * 64 + lane];
	return cv.h;
}
NGP_DEV floatx4 mfma16(half8 a, half8 b, floatx4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
// ReLU + round to fp16. max(round(x), 0) == round(max(x, 0)) (rounding keeps the sign), so the ReLU runs on packed
// halves: 4 cvt_pk + 4 pk_max per 8 activations.
NGP_DEV half8 relu_pack(floatx4 lo, floatx4 hi) {
	half8 r;
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		r[j] = (half_t)lo[j];
		r[4 + j] = (half_t)hi[j];
	}
	const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
	return __builtin_elementwise_max(r, zero);
}

struct MlpOut {
	half_t rgb[3];  // valid in lanes with h == 0
	half_t sigma;   // density logit, valid in lanes with h == 0
};

// All 16 SH coefficients of one direction as fp16 (what the owning lane stores once per ray).
// fp32 -> fp16 of a value that is first rounded to fp32 (no fusing of the producing multiply into v_fma_mix*_f16,
// which would round the exact product once; see accumulate_corner)
NGP_DEV half_t to_half_rn(float v) {
	asm("" : "+v"(v));
	return (half_t)v;
}
NGP_DEV void sh4_all(float dx01, float dy01, float dz01, half_t* out16) {
	float x = dx01 * 2.0f - 1.0f, y = dy01 * 2.0f - 1.0f, z = dz01 * 2.0f - 1.0f;
	float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
	out16[0] = (half_t)0.28209479177387814f;
	out16[1] = to_half_rn(-0.48860251190291987f * y);
	out16[2] = to_half_rn(0.48860251190291987f * z);
	out16[3] = to_half_rn(-0.48860251190291987f * x);
	out16[4] = to_half_rn(1.0925484305920792f * xy);
	out16[5] = to_half_rn(-1.0925484305920792f * yz);
	out16[6] = to_half_rn(0.94617469575755997f * z2 - 0.31539156525251999f);
	out16[7] = to_half_rn(-1.0925484305920792f * xz);
	out16[8] = to_half_rn(0.54627421529603959f * x2 - 0.54627421529603959f * y2);
	out16[9] = to_half_rn(0.59004358992664352f * y * (-3.0f * x2 + y2));
	out16[10] = to_half_rn(2.8906114426405538f * xy * z);
	out16[11] = to_half_rn(0.45704579946446572f * y * (1.0f - 5.0f * z2));
	out16[12] = to_half_rn(0.3731763325901154f * z * (5.0f * z2 - 3.0f));
	out16[13] = to_half_rn(0.45704579946446572f * x * (1.0f - 5.0f * z2));
	out16[14] = to_half_rn(1.4453057213202769f * z * (x2 - y2));
	out16[15] = to_half_rn(0.59004358992664352f * x * (-x2 + 3.0f * y2));
}

struct Sh4 { // SH coefficients 4h..4h+3 of the sample's direction, fp16
	half_t v[4];
};
NGP_DEV Sh4 sh4_from_dir(int h, float dx01, float dy01, float dz01) {
	float sh[4];
	sh4_quad(h, dx01, dy01, dz01, sh);
	Sh4 r;
#pragma unroll
	for (int j = 0; j < 4; ++j) r.v[j] = to_half_rn(sh[j]);
	return r;
}

// density head alone (NerfNetwork::density, nerf_network.h): the logit of sample c in lanes 0..15
template <bool DLIN = false>
NGP_DEV half_t density_pass(const uint4* s_w, int lane, half8 enc) {
	const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
	if (DLIN) return (half_t)mfma16(ld_frag(s_w, FRAG_D0, lane), enc, zero)[0]; // configs/nerf/linear.json: the 16 x 32 output layer alone
	floatx4 d0 = mfma16(ld_frag(s_w, FRAG_D0 + 0, lane), enc, zero);
	floatx4 d1 = mfma16(ld_frag(s_w, FRAG_D0 + 1, lane), enc, zero);
	floatx4 d2 = mfma16(ld_frag(s_w, FRAG_D0 + 2, lane), enc, zero);
	floatx4 d3 = mfma16(ld_frag(s_w, FRAG_D0 + 3, lane), enc, zero);
	half8 b0 = relu_pack(d0, d1), b1 = relu_pack(d2, d3);
	floatx4 dens = mfma16(ld_frag(s_w, FRAG_D1 + 0, lane), b0, zero);
	dens = mfma16(ld_frag(s_w, FRAG_D1 + 1, lane), b1, dens);
	return (half_t)dens[0];
}
// RGB_MID: the number of 64x64 layers of the rgb head = its n_hidden_layers - 1 (configs/nerf/base.json: 1;
// base_1layer.json 0, base_3layer.json 2). Their fragments follow FRAG_R1 eight at a time, the output layer's come last.
// RGB_MID -1 (base_0layer.json): the rgb head is its output layer alone (a CutlassMLP: 8 padded rows, zero rows above them in the
// fragment); RGB_MID -2 (linear.json): so is the density head. One MFMA each.
template <int RGB_MID = 1>
NGP_DEV MlpOut mlp_pass(const uint4* s_w, int lane, half8 enc, Sh4 shq) {
	const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
	floatx4 d0, d1, d2, d3, dens;
	half8 b0, b1;
	if (RGB_MID == -2) {
		dens = mfma16(ld_frag(s_w, FRAG_D0, lane), enc, zero);
	} else {
		// density head: 32 -> 64 (ReLU) -> 16
		d0 = mfma16(ld_frag(s_w, FRAG_D0 + 0, lane), enc, zero);
		d1 = mfma16(ld_frag(s_w, FRAG_D0 + 1, lane), enc, zero);
		d2 = mfma16(ld_frag(s_w, FRAG_D0 + 2, lane), enc, zero);
		d3 = mfma16(ld_frag(s_w, FRAG_D0 + 3, lane), enc, zero);
		b0 = relu_pack(d0, d1);
		b1 = relu_pack(d2, d3);
		dens = mfma16(ld_frag(s_w, FRAG_D1 + 0, lane), b0, zero);
		dens = mfma16(ld_frag(s_w, FRAG_D1 + 1, lane), b1, dens);
	}
	// rgb head input: [density out 4h..4h+3 | SH 4h..4h+3]
	half8 rin;
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		rin[j] = (half_t)dens[j];
		rin[4 + j] = shq.v[j];
	}
	MlpOut out;
	out.sigma = rin[0];
	if (RGB_MID < 0) {
		const floatx4 lin = mfma16(ld_frag(s_w, FRAG_R0, lane), rin, zero);
		out.rgb[0] = (half_t)lin[0];
		out.rgb[1] = (half_t)lin[1];
		out.rgb[2] = (half_t)lin[2];
		return out;
	}
	// rgb head: 32 -> 64 (ReLU) [-> 64 (ReLU)] x RGB_MID -> 16
	d0 = mfma16(ld_frag(s_w, FRAG_R0 + 0, lane), rin, zero);
	d1 = mfma16(ld_frag(s_w, FRAG_R0 + 1, lane), rin, zero);
	d2 = mfma16(ld_frag(s_w, FRAG_R0 + 2, lane), rin, zero);
	d3 = mfma16(ld_frag(s_w, FRAG_R0 + 3, lane), rin, zero);
	b0 = relu_pack(d0, d1);
	b1 = relu_pack(d2, d3);
#pragma unroll
	for (int k = 0; k < RGB_MID; ++k) {
		const int f = FRAG_R1 + 8 * k;
		d0 = mfma16(ld_frag(s_w, f + 0, lane), b0, zero);
		d0 = mfma16(ld_frag(s_w, f + 1, lane), b1, d0);
		d1 = mfma16(ld_frag(s_w, f + 2, lane), b0, zero);
		d1 = mfma16(ld_frag(s_w, f + 3, lane), b1, d1);
		d2 = mfma16(ld_frag(s_w, f + 4, lane), b0, zero);
		d2 = mfma16(ld_frag(s_w, f + 5, lane), b1, d2);
		d3 = mfma16(ld_frag(s_w, f + 6, lane), b0, zero);
		d3 = mfma16(ld_frag(s_w, f + 7, lane), b1, d3);
		b0 = relu_pack(d0, d1);
		b1 = relu_pack(d2, d3);
	}
	constexpr int f_out = FRAG_R1 + 8 * (RGB_MID < 0 ? 0 : RGB_MID);
	floatx4 rgb = mfma16(ld_frag(s_w, f_out + 0, lane), b0, zero);
	rgb = mfma16(ld_frag(s_w, f_out + 1, lane), b1, rgb);
	out.rgb[0] = (half_t)rgb[0];
	out.rgb[1] = (half_t)rgb[1];
	out.rgb[2] = (half_t)rgb[2];
	return out;
}

// ---------------------------------------------------------------------------------------------------------
// ERenderMode::Normals: d density logit / d position -- tcnn's input_gradient(dim 3) (src/testbed_nerf.cu:2106-2107) for 16 samples
// on one wave: the density MLP's backward pass on MFMA and GridEncoding's input path on the corner values the forward encode
// already holds (no second gather).
// The four A fragments of W1^T (32 encoding rows x 64 neurons, in the hidden layer's K order) follow the forward fragments in the
// weight buffer (build_normals_fragments_kernel, nerf_kernels.hip): fragment 2t + s = rows 16t.., neurons of K block s.
struct DensityGrad {
	float g[3];     // d (128 logit) / d warped position, summed over this lane's two levels only (the caller adds the four lanes of a sample)
	half_t sigma;   // the density logit (lanes with h == 0)
};
// dy_dx of kernel_grid for linear interpolation (scale * sum over the 4 corner pairs of w_other * (val_right - val_left), fp32)
// times dL_dy, for the two levels of this lane; e.v[8 l + c]: corner c (bit d = +1 along dimension d) of level l
NGP_DEV void encode_gradient(const EncodeInFlight& e, float scale0, float scale1, const float* dLdy, float* g3) {
	g3[0] = g3[1] = g3[2] = 0.0f;
#pragma unroll
	for (int l = 0; l < 2; ++l) {
		const float pos[3] = {e.wx[l], e.wy[l], e.wz[l]};
		const float scale = l ? scale1 : scale0;
#pragma unroll
		for (int gd = 0; gd < 3; ++gd) {
			float grads[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
			for (int idx = 0; idx < 4; ++idx) {
				float weight = scale;
				int corner = 0;
#pragma unroll
				for (int nd = 0; nd < 2; ++nd) {
					const int dim = nd >= gd ? nd + 1 : nd;
					if ((idx & (1 << nd)) == 0) {
						weight *= 1.0f - pos[dim];
					} else {
						weight *= pos[dim];
						corner |= 1 << dim;
					}
				}
				union { uint2 u; half_t h[4]; } left, right;
				left.u = e.v[8 * l + corner];
				right.u = e.v[8 * l + (corner | (1 << gd))];
#pragma unroll
				for (int f = 0; f < 4; ++f) grads[f] += weight * ((float)right.h[f] - (float)left.h[f]);
			}
#pragma unroll
			for (int f = 0; f < 4; ++f) g3[gd] += dLdy[4 * l + f] * grads[f];
		}
	}
}
// linear (wave-uniform): the density head has no hidden layer (configs/nerf/linear.json) -- dL_dy is 128 * W[0][:] itself, unmasked
NGP_DEV DensityGrad density_gradient_pass(const uint4* s_w, const uint4* __restrict__ g_wfrags, int lane, const EncodeInFlight& e, half8 enc, float scale0, float scale1, bool linear = false) {
	const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
	if (linear) {
		// row 0 of the only fragment sits in the lanes with (lane & 15) == 0, in the encoding's K order: element j of lane (h, .) is
		// W[0][16 (j >> 2) + 4 h + (j & 3)] -- feature j & 3 of level h (j < 4) or h + 4
		const half8 w = ld_frag(s_w, FRAG_D0, lane & 48);
		float dLdy[8];
#pragma unroll
		for (int j = 0; j < 8; ++j) dLdy[j] = (float)(half_t)((half_t)128.0f * w[j]);
		DensityGrad out;
		encode_gradient(e, scale0, scale1, dLdy, out.g);
		out.sigma = (half_t)mfma16(ld_frag(s_w, FRAG_D0, lane), enc, zero)[0];
		return out;
	}
	// forward: 32 -> 64 (ReLU) -> 16
	floatx4 d0 = mfma16(ld_frag(s_w, FRAG_D0 + 0, lane), enc, zero);
	floatx4 d1 = mfma16(ld_frag(s_w, FRAG_D0 + 1, lane), enc, zero);
	floatx4 d2 = mfma16(ld_frag(s_w, FRAG_D0 + 2, lane), enc, zero);
	floatx4 d3 = mfma16(ld_frag(s_w, FRAG_D0 + 3, lane), enc, zero);
	const half8 b0 = relu_pack(d0, d1), b1 = relu_pack(d2, d3);
	floatx4 dens = mfma16(ld_frag(s_w, FRAG_D1 + 0, lane), b0, zero);
	dens = mfma16(ld_frag(s_w, FRAG_D1 + 1, lane), b1, dens);
	// backward: one-hot 128 at the logit -> 128 * W2[0][:] masked by the forward ReLU (row 0 of the output layer's fragments sits in
	// the lanes with (lane & 15) == 0, in the hidden layer's K order: exactly the order of b0 / b1)
	const half8 w0 = ld_frag(s_w, FRAG_D1 + 0, lane & 48), w1 = ld_frag(s_w, FRAG_D1 + 1, lane & 48);
	half8 g0, g1;
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		g0[j] = b0[j] > (half_t)0 ? (half_t)((half_t)128.0f * w0[j]) : (half_t)0;
		g1[j] = b1[j] > (half_t)0 ? (half_t)((half_t)128.0f * w1[j]) : (half_t)0;
	}
	union { uint4 u; half8 h; } a;
	floatx4 lo = zero, hi = zero; // dL_dy of level h (rows 4h..4h+3 of tile 0) and of level h + 4 (tile 1)
	a.u = g_wfrags[(FRAG_NORMALS + 0) * 64 + lane]; lo = mfma16(a.h, g0, lo);
	a.u = g_wfrags[(FRAG_NORMALS + 1) * 64 + lane]; lo = mfma16(a.h, g1, lo);
	a.u = g_wfrags[(FRAG_NORMALS + 2) * 64 + lane]; hi = mfma16(a.h, g0, hi);
	a.u = g_wfrags[(FRAG_NORMALS + 3) * 64 + lane]; hi = mfma16(a.h, g1, hi);
	float dLdy[8];
#pragma unroll
	for (int r = 0; r < 4; ++r) {
		dLdy[r] = (float)(half_t)lo[r];
		dLdy[4 + r] = (float)(half_t)hi[r];
	}
	DensityGrad out;
	encode_gradient(e, scale0, scale1, dLdy, out.g);
	out.sigma = (half_t)dens[0];
	return out;
}

// ---------------------------------------------------------------------------------------------------------
// activations, nerf_device.cuh:203-263
// The reference is built with --use_fast_math: its expf is __expf (ex2.approx of x * log2 e) and its divisions are
// approximate. v_exp_f32 / v_rcp_f32 are the gfx950 counterparts (1 ulp); the oracle uses libm, the difference is
// ~1e-7 relative per sample, far inside the image tolerance.
NGP_DEV float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
NGP_DEV float logistic(float x) { return __builtin_amdgcn_rcpf(1.0f + fast_exp(-x)); }
NGP_DEV float network_to_rgb(float v, uint32_t act) {
	switch (act) {
		case 1: return v > 0.0f ? v : 0.0f;
		case 2: return logistic(v);
		case 3: return fast_exp(fminf(fmaxf(v, -10.0f), 10.0f));
		default: return v;
	}
}
NGP_DEV float network_to_density(float v, uint32_t act) {
	switch (act) {
		case 1: return v > 0.0f ? v : 0.0f;
		case 2: return logistic(v);
		case 3: return fast_exp(v);
		default: return v;
	}
}
NGP_DEV float network_to_density_derivative(float v, uint32_t act) { // nerf_device.cuh:245-254
	switch (act) {
		case 1: return v > 0.0f ? 1.0f : 0.0f;
		case 2: { float d = logistic(v); return d * (1.0f - d); }
		case 3: return fast_exp(fminf(fmaxf(v, -15.0f), 15.0f));
		default: return 1.0f;
	}
}
// powf under the reference's --use_fast_math: exp2(y * log2 x), x > 0
NGP_DEV float fast_pow(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }
NGP_DEV float srgb_to_linear(float s) { // common_device.cuh:34-40
	return s <= 0.04045f ? s / 12.92f : fast_pow((s + 0.055f) / 1.055f, 2.4f);
}
NGP_DEV float linear_to_srgb(float l) { // common_device.cuh:58-64
	return l < 0.0031308f ? 12.92f * l : 1.055f * fast_pow(l, 0.41666f) - 0.055f;
}
NGP_DEV uint32_t lanes_below(unsigned long long mask) {
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

} // namespace ngp
