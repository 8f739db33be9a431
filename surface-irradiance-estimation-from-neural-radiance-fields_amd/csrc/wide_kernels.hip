// The NeRF inference path for the "wide" architecture on MI355X (gfx950): configs/nerf/frequency.json -- the original NeRF's
// network inside the reference's renderer: Frequency encodings of position (16 frequencies -> 96 inputs) and direction (4 -> 24),
// a density MLP of 7 hidden layers x 256 neurons and an rgb MLP of 1 x 256 (tcnn CutlassMLP; the reference's loop is the same
// src/testbed_nerf.cu:2056-2138 chain, with every layer a cutlass GEMM launch).
//
// One persistent kernel per sample-per-pixel, like the base.json kernel (nerf_kernels.hip), but organised around the GEMMs that
// now dominate (0.87 MFLOP per sample, 42x base.json's). A workgroup of 4 waves owns 256 ray slots, one per thread: every round
// each live slot marches to its next sample, the workgroup runs the network on 128 of the waiting samples (those that waited
// before go first) with the activations resident in LDS, and each slot composites its own sample: twice as many slots as rows
// keep the rows full. TWO workgroups share a CU (80 KB of LDS each, 2 waves per SIMD): while one
// marches, evaluates sines or exchanges activations, the other one's MFMAs run -- the phases of one workgroup are serial, and
// with one workgroup per CU the matrix pipe idled through all but the GEMM phase. Rays never leave registers; the only HBM
// traffic is occupancy bits, the weight fragments (868 KB, L2-resident) and one frame-buffer write per pixel.
//
//   activations  X[128 samples][264] fp16 in LDS (66 KB, rows padded by 8 halves so that the 16-byte operand reads of 16
//                consecutive samples cover all 64 banks once); a layer is computed in place: all waves read, barrier, all write
//   GEMM         v_mfma_f32_32x32x16_f16, weights on the A side (M = neurons), samples on the B side (N). A wave owns
//                width / 4 neurons (2 M-tiles for 256) and all four 32-sample tiles: 8 accumulator tiles = 128 fp32 registers,
//                every B read feeds two MFMAs (LDS at half its rate when the MFMA pipe is full)
//   A operand    MFMA fragments prepared by the host (ngp_kernels.h WideModel), 16 B per lane, coalesced from L2, streamed through
//                a ring of 3 K-blocks that runs 2 blocks (512 MFMA cycles) ahead of its use and on into the next layer
//   B operand    ds_read_b128 of X[sample][16 kb + 8 h ..], read one K-block ahead into a second register set
//   output tile  lane (n, h) holds neurons 8q + 4h + r of sample n: four 8-byte LDS writes per tile put them back in row n
//   encodings    the sines of a sample are split over two threads (tid and tid + 128): all four waves work through them
//
// The march is the loop of nerf_device.cuh:461-494 on the occupancy bitfield in global memory (L2), one 8-byte word = one 4^3 block
// of cells per load, kept by the lane; like the base.json kernel it leaves empty 4^3 / 16^3 blocks in one step unless
// FrameParams::tune[6] (ngp_set_schedule's block_jumps) is 0, which gives the reference's one-voxel steps and its exact sample sets.
#include "render_common.h"

#include <cstdio>
#include <cstdlib>

namespace ngp {

constexpr int WBLOCK = 256;    // threads of a workgroup = ray slots
constexpr int ROWS = 128;      // samples per round: the first 128 slots (waiting ones first) that hold a sample
constexpr int XS = 264;        // halves per activation row: 256 + 8 of padding, which also carries the row's position / outputs (row_meta)
constexpr int DIR_STRIDE = 24; // halves per direction-encoding row (4 frequencies x 3 x 2, or 16 SH coefficients)
typedef float floatx16 __attribute__((ext_vector_type(16)));

struct WideShared {
	half_t x[ROWS * XS];
	half_t dir[WBLOCK * DIR_STRIDE];       // per ray slot: the encoded direction, constant along the ray
	uint16_t owner[ROWS];                  // the slot whose sample a row carries this round
	uint32_t cnt[8];                       // per wave: waiting / new samples (the round's selection)
	uint32_t coarse16[NERF_CASCADES * 16]; // per cascade: which 16^3-cell blocks of the occupancy grid hold anything (ModelParams::coarse, tail)
	unsigned long long prof[16];           // diagnostic (NGP_PROFILE_SECTIONS=1): the workgroup's section sums, kept here rather than in registers
};
// the 16 bytes behind a row's 256 activations: before the network the sample's position (x, y, z, 1; w = 0: the row is empty), after
// it the sample's rgb outputs
NGP_DEV float4* row_meta(WideShared& S, int row) { return (float4*)(S.x + row * XS + 256); }
static_assert(2 * sizeof(WideShared) <= 160 * 1024, "two workgroups per CU");

// The march's occupancy lookup (cf. empty_block_size_at, nerf_device.h): 0 = the cell is occupied, else the side (in cells of this
// cascade) of the largest aligned empty block around pos that can be vouched for: 16 (summary bits in LDS), 4 (the 64 occupancy
// bits of a 4x4x4 block are one aligned 8-byte word of the Morton-ordered bitfield: all zero = the block is empty) or 1. The
// lane keeps the last word it read: a ray spends many consecutive lookups in one block, and every miss is an L2 round trip that
// nothing hides at one wave per SIMD.
NGP_DEV uint32_t empty_block_size_global(f3 pos, const uint8_t* __restrict__ bitfield, const uint32_t* s_coarse16, uint32_t mip, OccBlockCache& cache) {
	float mip_scale = __builtin_ldexpf(1.0f, -(int)mip);
	pos = adds3(scale3(adds3(pos, -0.5f), mip_scale), 0.5f);
	int ix = (int)(pos.x * (float)NERF_GRIDSIZE);
	int iy = (int)(pos.y * (float)NERF_GRIDSIZE);
	int iz = (int)(pos.z * (float)NERF_GRIDSIZE);
	if (ix < 0 || ix >= (int)NERF_GRIDSIZE || iy < 0 || iy >= (int)NERF_GRIDSIZE || iz < 0 || iz >= (int)NERF_GRIDSIZE) return 1u;
	const uint32_t idx = morton3D((uint32_t)ix, (uint32_t)iy, (uint32_t)iz);
	const uint32_t b16 = idx >> 12;
	if (!((s_coarse16[mip * 16u + (b16 >> 5)] >> (b16 & 31u)) & 1u)) return 16u;
	const uint32_t b4 = idx >> 6, key = (mip << 26) | b4;
	if (cache.key != key) {
		cache.bits = *(const uint2*)(bitfield + (size_t)b4 * 8 + (size_t)(NERF_GRID_N_CELLS / 8) * mip);
		cache.key = key;
	}
	if ((cache.bits.x | cache.bits.y) == 0u) return 4u;
	const uint32_t word = (idx & 32u) ? cache.bits.y : cache.bits.x;
	return ((word >> (idx & 31u)) & 1u) ? 0u : 1u;
}

NGP_DEV floatx16 mfma32(half8 a, half8 b, floatx16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// sin(t) for the encoder's arguments (|t| up to 2^15 pi for positions in the unit cube): Cody-Waite reduction by pi in three fp32
// pieces (fma: the products are exact), then the odd Taylor polynomial of degree 11 on [-pi/2, pi/2]. Absolute error < 2e-7 (measured
// 1.7e-7 over 6.4e6 arguments), i.e. the fp16 rounding of the feature differs from the exactly rounded one in < 1e-4 of the cases;
// 17 instructions where libm's general sinf takes ~55. Arguments beyond 2^20 go to sinf.
NGP_DEV float encoder_sin_reduced(float t) { // |t| < 2^20
	const float k = __builtin_rintf(t * 0.318309886183790672f);
	float r = __builtin_fmaf(-k, 3.14159274101257324f, t);
	r = __builtin_fmaf(-k, -8.74227765734758577e-08f, r);
	r = __builtin_fmaf(-k, -3.55271367880050093e-15f, r);
	const float r2 = r * r;
	float p = -2.50521083854417188e-08f;
	p = __builtin_fmaf(p, r2, 2.75573192239858907e-06f);
	p = __builtin_fmaf(p, r2, -1.98412698412698413e-04f);
	p = __builtin_fmaf(p, r2, 8.33333333333333333e-03f);
	p = __builtin_fmaf(p, r2, -1.66666666666666667e-01f);
	const float s = __builtin_fmaf(r * r2, p, r);
	const uint32_t flip = (uint32_t)(int)k << 31; // sin(r + k pi) = (-1)^k sin(r)
	return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, s) ^ flip);
}
NGP_DEV float encoder_sin(float t) {
	if (!(__builtin_fabsf(t) < 1048576.0f)) return sinf(t);
	return encoder_sin_reduced(t);
}
// cos(t) the same way (the encoding's derivative, ERenderMode::Normals): even Taylor polynomial of degree 12 on [-pi/2, pi/2], |error| < 1e-7
NGP_DEV float encoder_cos(float t) {
	if (!(__builtin_fabsf(t) < 1048576.0f)) return cosf(t);
	const float k = __builtin_rintf(t * 0.318309886183790672f);
	float r = __builtin_fmaf(-k, 3.14159274101257324f, t);
	r = __builtin_fmaf(-k, -8.74227765734758577e-08f, r);
	r = __builtin_fmaf(-k, -3.55271367880050093e-15f, r);
	const float r2 = r * r;
	float p = 2.08767569878680990e-09f;
	p = __builtin_fmaf(p, r2, -2.75573192239858907e-07f);
	p = __builtin_fmaf(p, r2, 2.48015873015873016e-05f);
	p = __builtin_fmaf(p, r2, -1.38888888888888889e-03f);
	p = __builtin_fmaf(p, r2, 4.16666666666666667e-02f);
	p = __builtin_fmaf(p, r2, -0.5f);
	const float c = __builtin_fmaf(p, r2, 1.0f);
	const uint32_t flip = (uint32_t)(int)k << 31; // cos(r + k pi) = (-1)^k cos(r)
	return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, c) ^ flip);
}

// tcnn FrequencyEncoding (encodings/frequency.h): feature j of input x is
//   sin(fma(scalbn(x[j / (2 F)], (j / 2) % F), pi, (j % 2) pi / 2))
// rounded to fp16; inputs beyond 3 * 2 F up to `padded` are ones. This call writes the features of frequencies [f_begin, f_end) of
// the three inputs (4-byte aligned pairs); with `tail` also the ones up to `padded` and zeros from there to k_end (the MFMA K the row
// is read in: the weights' columns there are zeros, which does not make 0 x stale-NaN a zero; 16-byte aligned).
// The usual case -- up to 16 frequencies of inputs around the unit cube: every argument stays below 2^20 -- runs without the per-sine
// range check, frequency by frequency with the six sines of a frequency (three inputs, sine and cosine) as independent chains.
NGP_DEV void frequency_encode(uint32_t n_freq, uint32_t padded, float x, float y, float z, half_t* out, uint32_t f_begin, uint32_t f_end, bool tail, uint32_t k_end = 0) {
	const float PI = 3.14159265358979323846f;
	const float in[3] = {x, y, z};
	if (n_freq <= 16u && __builtin_fabsf(x) <= 8.0f && __builtin_fabsf(y) <= 8.0f && __builtin_fabsf(z) <= 8.0f) { // |2^15 * 8 * pi| < 2^20
		for (uint32_t f = f_begin; f < f_end; ++f) {
#pragma unroll
			for (int d = 0; d < 3; ++d) {
				const float v = __builtin_ldexpf(in[d], (int)f);
				half2_t sc;
				sc[0] = (half_t)encoder_sin_reduced(__builtin_fmaf(v, PI, 0.0f));
				sc[1] = (half_t)encoder_sin_reduced(__builtin_fmaf(v, PI, PI / 2.0f));
				*(half2_t*)(out + (uint32_t)d * 2u * n_freq + 2u * f) = sc;
			}
		}
	} else {
#pragma unroll
		for (int d = 0; d < 3; ++d) {
			for (uint32_t f = f_begin; f < f_end; ++f) {
				const float v = __builtin_ldexpf(in[d], (int)f);
				half2_t sc;
				sc[0] = (half_t)encoder_sin(__builtin_fmaf(v, PI, 0.0f));
				sc[1] = (half_t)encoder_sin(__builtin_fmaf(v, PI, PI / 2.0f));
				*(half2_t*)(out + (uint32_t)d * 2u * n_freq + 2u * f) = sc;
			}
		}
	}
	if (tail) {
		const half2_t ones = {(half_t)1.0f, (half_t)1.0f};
		for (uint32_t j = 6u * n_freq; j < padded; j += 2u) *(half2_t*)(out + j) = ones;
		const half8 zeros = {0, 0, 0, 0, 0, 0, 0, 0};
		for (uint32_t j = padded; j < k_end; j += 8u) *(half8*)(out + j) = zeros; // (padded is a multiple of 8)
	}
}

// tcnn Identity encoding (encodings/identity.h; configs/nerf/none.json): out[j] = in[j] * scale + offset (1, 0 here) rounded to fp16, ones up to
// `padded`; zeros from there to k_end as above
NGP_DEV void identity_encode(uint32_t padded, float x, float y, float z, half_t* out, uint32_t k_end = 0) {
	half8 v = {(half_t)x, (half_t)y, (half_t)z, (half_t)1.0f, (half_t)1.0f, (half_t)1.0f, (half_t)1.0f, (half_t)1.0f};
	*(half8*)out = v; // (padded is a multiple of 8: next_multiple(3, alignment))
	const half8 ones = {(half_t)1.0f, (half_t)1.0f, (half_t)1.0f, (half_t)1.0f, (half_t)1.0f, (half_t)1.0f, (half_t)1.0f, (half_t)1.0f};
	for (uint32_t j = 8u; j < padded; j += 8u) *(half8*)(out + j) = ones;
	const half8 zeros = {0, 0, 0, 0, 0, 0, 0, 0};
	for (uint32_t j = padded; j < k_end; j += 8u) *(half8*)(out + j) = zeros;
}

NGP_DEV uint2 pack4(float a, float b, float c, float d, bool relu) {
	half2_t lo = {(half_t)a, (half_t)b}, hi = {(half_t)c, (half_t)d}; // v_cvt_pk_f16_f32
	if (relu) { // max(round(x), 0) == round(max(x, 0)): v_pk_max_f16
		const half2_t zero = {(half_t)0.0f, (half_t)0.0f};
		lo = __builtin_elementwise_max(lo, zero);
		hi = __builtin_elementwise_max(hi, zero);
	}
	return make_uint2(__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi));
}

// The weights stream: a wave's MFMA A fragments of K-block kb sit in ring stage kb % RING, requested RING - 1 K-blocks (16 MFMAs) before
// their use; the ring runs on into the next layer (its first blocks are requested while this layer's outputs are packed and
// exchanged). The barriers inside the network wait for LDS traffic only (s_waitcnt lgkmcnt(0) + s_barrier): a __syncthreads()
// would also drain the outstanding global loads, i.e. the stream.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#ifndef WIDE_RING
#define WIDE_RING (WIDE_MFMA16 ? 2 : 3) // (a K block of the 16x16x32 form is 32 wide: one block ahead is the same 512 MFMA cycles as two of the 32x32x16 form)
#endif
constexpr int RING = WIDE_RING, AHEAD = RING - 1; // stages of the weight ring, K-blocks it runs ahead
NGP_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
NGP_DEV half8 as_half8(u32x4 u) { return __builtin_bit_cast(half8, u); }

struct LayerFrags { // where this wave's fragments of a layer start (lane included), their K-blocks and M-tiles (0 tiles: no layer)
	const uint4* base;
	int nkb, mt;
};
template <int MT>
NGP_DEV void ring_preload(u32x4 (&ar)[RING][MT], LayerFrags L) {
#pragma unroll
	for (int st = 0; st < AHEAD; ++st)
#pragma unroll
		for (int m = 0; m < MT; ++m)
			if (m < L.mt) ar[st][m] = *(const u32x4*)(L.base + ((size_t)m * L.nkb + st) * 64);
}

#if WIDE_MFMA16
// ---- the GEMMs on v_mfma_f32_16x16x32_f16. MT counts 16-neuron tiles of a wave (4 for 256 neurons), NKB 32-wide K blocks.
// A fragment: lane (r = lane & 15, h = lane >> 4) holds W[16 m + r][32 kb + 8 h ..]; B: lane (c, h) reads X[16 t + c][32 kb + 8 h ..]
// (one ds_read_b128); D: lane (c, h) holds neurons 16 m + 4 h .. + 3 of sample 16 t + c -- four packed halves, one 8-byte LDS write.
// A wave owns 16 MT neurons and all eight 16-sample tiles: MT x 8 accumulator tiles of 4 registers (128 for MT = 4, as before). The
// B operands of half a K block (four sample tiles) are read while the MFMAs of the other half run.
typedef float floatx4w __attribute__((ext_vector_type(4)));
NGP_DEV floatx4w mfma16w(half8 a, half8 b, floatx4w c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
template <int MT, int NKB>
NGP_DEV void wide_hidden_layer(half_t* X, u32x4 (&ar)[RING][MT], const uint4* __restrict__ wf, int wave, int lane, LayerFrags next) {
	const int c = lane & 15, h = lane >> 4;
	const floatx4w zero = {0.f, 0.f, 0.f, 0.f};
	floatx4w acc[MT][8];
	const half_t* col = X + c * XS + 8 * h;
	// B operands: two sample tiles at a time, read while the MFMAs of the previous pair run (MT x 2 MFMAs = 128 cycles for MT = 4: a
	// ds_read_b128's latency); a quarter of a K block in flight instead of a whole one keeps the loop inside 256 registers
	half8 b[2][2];
	b[0][0] = *(const half8*)col;
	b[0][1] = *(const half8*)(col + 16 * XS);
#pragma unroll
	for (int kb = 0; kb < NKB; ++kb) {
		if (kb + AHEAD < NKB) {
#pragma unroll
			for (int m = 0; m < MT; ++m) ar[(kb + AHEAD) % RING][m] = *(const u32x4*)(wf + ((size_t)m * NKB + kb + AHEAD) * 64);
		}
#pragma unroll
		for (int q = 0; q < 4; ++q) { // sample tiles 2q, 2q + 1
			const int nq = (q + 1) & 3, nkb = q == 3 ? kb + 1 : kb;
			if (nkb < NKB) {
				b[(q + 1) & 1][0] = *(const half8*)(col + 16 * (2 * nq) * XS + 32 * nkb);
				b[(q + 1) & 1][1] = *(const half8*)(col + 16 * (2 * nq + 1) * XS + 32 * nkb);
			}
			__builtin_amdgcn_sched_barrier(0); // (the scheduler would sink the reads next to their use)
			const bool last = kb == NKB - 1 && q == 3;
			if (last) lds_barrier(); // every wave has read the layer's input: the in-place writes may start under the last MFMAs (see the 32x32x16 form)
#pragma unroll
			for (int t = 0; t < 2; ++t)
#pragma unroll
				for (int m = 0; m < MT; ++m) acc[m][2 * q + t] = mfma16w(as_half8(ar[kb % RING][m]), b[q & 1][t], kb == 0 ? zero : acc[m][2 * q + t]);
			if (!last) __builtin_amdgcn_sched_barrier(0);
		}
	}
	ring_preload<MT>(ar, next);
#pragma unroll
	for (int t = 0; t < 8; ++t) {
		half_t* row = X + (16 * t + c) * XS + 16 * (wave * MT) + 4 * h;
#pragma unroll
		for (int m = 0; m < MT; ++m) *(uint2*)(row + 16 * m) = pack4(acc[m][t][0], acc[m][t][1], acc[m][t][2], acc[m][t][3], true);
	}
	lds_barrier();
}

// An output layer (at most 16 neurons, no activation): wave w computes sample tiles 2w and 2w + 1; fragments in ar[.][0].
// Returned: acc[t'] = outputs 4h .. 4h + 3 of sample 16 (2 wave + t') + c
struct OutTiles {
	floatx4w t[2];
};
template <int MT, int NKB>
NGP_DEV OutTiles wide_out_layer(const half_t* X, u32x4 (&ar)[RING][MT], const uint4* __restrict__ wf, int wave, int lane, LayerFrags next) {
	const int c = lane & 15, h = lane >> 4;
	const floatx4w zero = {0.f, 0.f, 0.f, 0.f};
	OutTiles acc;
	acc.t[0] = acc.t[1] = zero;
	const half_t* col = X + (32 * wave + c) * XS + 8 * h;
	half8 b[2][2];
	b[0][0] = *(const half8*)col;
	b[0][1] = *(const half8*)(col + 16 * XS);
#pragma unroll
	for (int kb = 0; kb < NKB; ++kb) {
		if (kb + AHEAD < NKB) ar[(kb + AHEAD) % RING][0] = *(const u32x4*)(wf + (size_t)(kb + AHEAD) * 64);
		if (kb + 1 < NKB) {
			b[(kb + 1) & 1][0] = *(const half8*)(col + 32 * (kb + 1));
			b[(kb + 1) & 1][1] = *(const half8*)(col + 16 * XS + 32 * (kb + 1));
		}
		__builtin_amdgcn_sched_barrier(0);
		acc.t[0] = mfma16w(as_half8(ar[kb % RING][0]), b[kb & 1][0], acc.t[0]);
		acc.t[1] = mfma16w(as_half8(ar[kb % RING][0]), b[kb & 1][1], acc.t[1]);
		__builtin_amdgcn_sched_barrier(0);
	}
	ring_preload<MT>(ar, next);
	return acc;
}
#else
// One hidden layer, in place: X[:, 0 .. 128 MT) <- ReLU(W X[:, 0 .. 16 NKB)); ring stages 0..2 hold (or await) K-blocks 0..2.
// MODE (ERenderMode::Normals, the density network's backward pass on the same GEMM): 0 = forward; 1 = forward, and bit (row, neuron) of `mask`
// (ROWS x 32 bytes) records which outputs are positive; 2 = a TRANSPOSED layer of the backward pass: no ReLU, outputs whose mask bit is
// clear become zero (the ReLU's derivative at the forward activation); 3 = transposed, no mask (the first layer: its inputs are the encoding)
template <int MT, int NKB, int MODE = 0>
NGP_DEV void wide_hidden_layer(half_t* X, u32x4 (&ar)[RING][MT], const uint4* __restrict__ wf, int wave, int lane, LayerFrags next, unsigned long long* pr = nullptr, uint8_t* mask = nullptr) {
	const int n = lane & 31, h = lane >> 5;
	unsigned long long ts0 = 0;
	if (pr) ts0 = stamp();
	const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
	floatx16 acc[MT][4];
	const half_t* col = X + n * XS + 8 * h;
	half8 b[2][4]; // the B operands of K block kb + 1 are read while the 4 MT MFMAs of block kb run
#pragma unroll
	for (int t = 0; t < 4; ++t) b[0][t] = *(const half8*)(col + 32 * t * XS);
#pragma unroll
	for (int kb = 0; kb < NKB; ++kb) {
		if (kb + AHEAD < NKB) {
#pragma unroll
			for (int m = 0; m < MT; ++m) ar[(kb + AHEAD) % RING][m] = *(const u32x4*)(wf + ((size_t)m * NKB + kb + AHEAD) * 64);
		}
		if (kb + 1 < NKB) {
#pragma unroll
			for (int t = 0; t < 4; ++t) b[(kb + 1) & 1][t] = *(const half8*)(col + 32 * t * XS + 16 * (kb + 1));
		}
		__builtin_amdgcn_sched_barrier(0); // (the scheduler would sink the reads next to their use, one block late)
		// The last block's operands are in registers: every wave is done reading the layer's input once it gets here, so the barrier that
		// guards the in-place write sits in front of the last 4 MT MFMAs instead of behind the packing -- the tiles are then packed and
		// written one by one as their accumulators complete, under the MFMAs of the tiles behind them.
		if (kb == NKB - 1) lds_barrier();
#pragma unroll
		for (int t = 0; t < 4; ++t)
#pragma unroll
			for (int m = 0; m < MT; ++m) acc[m][t] = mfma32(as_half8(ar[kb % RING][m]), b[kb & 1][t], kb == 0 ? zero : acc[m][t]);
		if (kb < NKB - 1) __builtin_amdgcn_sched_barrier(0);
	}
	ring_preload<MT>(ar, next);
	// Lane (n, h) holds neurons 8 q + 4 h .. + 3 of its sample: 8 bytes per q. Written like that, the 32 lanes of a half wave hit every second
	// bank pair twice (row stride 132 dwords: lanes n and n + 16 share banks -- the stride is the one the 16-byte operand READS need). The two
	// halves of the wave trade one q each instead (v_permlane32_swap: the low half ends up with neurons 16 j .. + 7, the high half with
	// 16 j + 8 .. + 15) and write 16 bytes per lane: half as many LDS instructions, conflict-free like the reads.
#pragma unroll
	for (int t = 0; t < 4; ++t) {
		half_t* row = X + (32 * t + n) * XS + 32 * (wave * MT) + 8 * h;
#pragma unroll
		for (int m = 0; m < MT; ++m)
#pragma unroll
			for (int j = 0; j < 2; ++j) {
				const uint2 p0 = pack4(acc[m][t][8 * j], acc[m][t][8 * j + 1], acc[m][t][8 * j + 2], acc[m][t][8 * j + 3], MODE < 2);
				const uint2 p1 = pack4(acc[m][t][8 * j + 4], acc[m][t][8 * j + 5], acc[m][t][8 * j + 6], acc[m][t][8 * j + 7], MODE < 2);
				const auto sx = __builtin_amdgcn_permlane32_swap(p0.x, p1.x, false, false);
				const auto sy = __builtin_amdgcn_permlane32_swap(p0.y, p1.y, false, false);
				uint4 v = make_uint4(sx[0], sy[0], sx[1], sy[1]); // neurons c0 .. c0 + 7 of row 32 t + n, c0 = 32 (wave MT + m) + 16 j + 8 h
				if (MODE == 1 || MODE == 2) {
					uint8_t* mb = mask + (32 * t + n) * 32 + 4 * (wave * MT + m) + 2 * j + h;
					const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
					if (MODE == 1) { // (after the ReLU a value is positive iff it is not a zero of either sign)
						uint32_t bits = 0;
#pragma unroll
						for (int k = 0; k < 4; ++k) bits |= ((w4[k] & 0x7fffu) ? 1u : 0u) << (2 * k) | ((w4[k] & 0x7fff0000u) ? 1u : 0u) << (2 * k + 1);
						*mb = (uint8_t)bits;
					} else {
						const uint32_t bits = *mb;
						uint32_t o4[4];
#pragma unroll
						for (int k = 0; k < 4; ++k) o4[k] = w4[k] & (((bits >> (2 * k)) & 1u ? 0xffffu : 0u) | ((bits >> (2 * k + 1)) & 1u ? 0xffff0000u : 0u));
						v = make_uint4(o4[0], o4[1], o4[2], o4[3]);
					}
				}
				*(uint4*)(row + 32 * m + 16 * j) = v;
			}
	}
	lds_barrier();
	if (pr) pr[8] += stamp() - ts0; // (K loop and epilogue together: a stamp between them would keep the packing out of the last MFMAs' shadow)
}

// An output layer (at most 32 neurons, no activation): wave w computes its own sample tile w; fragments in ar[.][0].
// One MFMA per K block: a ring that runs AHEAD blocks ahead would be 64 MFMA cycles ahead of an L2 round trip, i.e. every block would wait
// for one (measured: the two output layers cost 10 k cycles of a 96 k round). The accumulator tiles are free here, so ALL of the layer's
// fragments and operands are requested at once: one round trip for the layer.
template <int MT, int NKB>
NGP_DEV floatx16 wide_out_layer(const half_t* X, u32x4 (&ar)[RING][MT], const uint4* __restrict__ wf, int wave, int lane, LayerFrags next) {
	const int n = lane & 31, h = lane >> 5;
	const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
	floatx16 acc = zero;
	const half_t* col = X + (32 * wave + n) * XS + 8 * h;
	u32x4 a[NKB];
	half8 b[NKB];
#pragma unroll
	for (int kb = 0; kb < NKB; ++kb) a[kb] = kb < AHEAD ? ar[kb][0] : *(const u32x4*)(wf + (size_t)kb * 64);
#pragma unroll
	for (int kb = 0; kb < NKB; ++kb) b[kb] = *(const half8*)(col + 16 * kb);
	ring_preload<MT>(ar, next); // (the ring's stages have been copied out: the next layer's first blocks travel behind this layer's)
	__builtin_amdgcn_sched_barrier(0);
#pragma unroll
	for (int kb = 0; kb < NKB; ++kb) acc = mfma32(as_half8(a[kb]), b[kb], acc);
	return acc;
}

#endif

struct WideOut {
	half_t r, g, b, sigma;
	float gx, gy, gz; // ERenderMode::Normals only: d logit / d warped position
};
constexpr int K256 = 256 / WIDE_TILE_K, K128 = 128 / WIDE_TILE_K; // K blocks of the two layer shapes the kernels are instantiated for

template <int MT>
NGP_DEV LayerFrags layer_frags(const WideModel& W, uint32_t l, int wave, int lane) {
	LayerFrags L;
	const uint32_t n_layers = W.n_hidden_density + W.n_hidden_rgb + 2u;
	if (l >= n_layers) {
		L.base = nullptr; L.nkb = 0; L.mt = 0;
		return L;
	}
	const bool is_out = l == W.n_hidden_density || l + 1u == n_layers;
	L.nkb = (int)W.layers[l].n_kblocks;
	L.mt = is_out ? 1 : MT;
	L.base = W.frags + W.layers[l].frag_offset + (is_out ? 0 : (size_t)(wave * MT) * L.nkb * 64) + lane;
	return L;
}
template <int MT>
NGP_DEV void wide_network_prefetch(const WideModel& W, u32x4 (&ar)[RING][MT], int tid) {
	ring_preload<MT>(ar, layer_frags<MT>(W, 0, tid >> 6, tid & 63));
}

// NerfNetwork::inference_mixed_precision_impl (nerf_network.h:105-139) for the workgroup's 128 sample rows. On entry row r of
// S.x holds the position encoding of slot r's sample (zeros beyond it up to the first layer's K) and S.dir its direction
// encoding, both visible (the caller has passed a barrier), and the first layer's fragments have been requested
// (wide_network_prefetch); on exit the thread that owns row `my_row` (-1: none) has its outputs and X may be overwritten.
template <int MT>
NGP_DEV WideOut wide_network(const WideModel& W, WideShared& S, int tid, u32x4 (&ar)[RING][MT], int my_row, unsigned long long* pr = nullptr) {
	const int wave = tid >> 6, lane = tid & 63;
	const int n = lane & 31, h = lane >> 5;
	(void)n; (void)h;
	const int row_id = tid & (ROWS - 1), part = tid >> 7;
	const uint32_t n_layers = W.n_hidden_density + W.n_hidden_rgb + 2u;
	WideOut o;
	o.r = o.g = o.b = o.sigma = (half_t)0.0f;
	for (uint32_t l = 0; l < n_layers; ++l) {
		const bool density_out = l == W.n_hidden_density, rgb_out = l + 1u == n_layers;
		const LayerFrags cur = layer_frags<MT>(W, l, wave, lane), next = layer_frags<MT>(W, l + 1u, wave, lane);
		if (!density_out && !rgb_out) {
#if WIDE_MFMA16
			if (cur.nkb == K256) wide_hidden_layer<MT, K256>(S.x, ar, cur.base, wave, lane, next);
			else wide_hidden_layer<MT, K128>(S.x, ar, cur.base, wave, lane, next);
#else
			if (cur.nkb == K256) wide_hidden_layer<MT, K256>(S.x, ar, cur.base, wave, lane, next, pr);
			else wide_hidden_layer<MT, K128>(S.x, ar, cur.base, wave, lane, next, pr);
#endif
			continue;
		}
#if WIDE_MFMA16
		const OutTiles acc = cur.nkb == K256 ? wide_out_layer<MT, K256>(S.x, ar, cur.base, wave, lane, next) : wide_out_layer<MT, K128>(S.x, ar, cur.base, wave, lane, next);
#else
		unsigned long long ts0 = 0;
		if (pr) ts0 = stamp();
		const floatx16 acc = cur.nkb == K256 ? wide_out_layer<MT, K256>(S.x, ar, cur.base, wave, lane, next) : wide_out_layer<MT, K128>(S.x, ar, cur.base, wave, lane, next);
		if (pr) { asm volatile("" :: "v"(acc[0])); pr[15] += stamp() - ts0; }
#endif
		if (density_out) {
			// the 16 density outputs become columns 0..15 of the rgb network's input (rows of this wave's own tile: no other wave reads them now)
#if WIDE_MFMA16
#pragma unroll
			for (int q = 0; q < 2; ++q) *(uint2*)(S.x + (32 * wave + 16 * q + (lane & 15)) * XS + 4 * (lane >> 4)) = pack4(acc.t[q][0], acc.t[q][1], acc.t[q][2], acc.t[q][3], false);
#else
			half_t* orow = S.x + (32 * wave + n) * XS + 4 * h;
#pragma unroll
			for (int q = 0; q < 2; ++q) *(uint2*)(orow + 8 * q) = pack4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3], false);
#endif
			lds_barrier();
			// [density out | direction encoding | ones up to the network's input alignment | zeros up to the next layer's K]; two threads per row
			half_t* row = S.x + row_id * XS;
			if (my_row >= 0) o.sigma = S.x[my_row * XS];
			if (part == 0) {
				const half_t* d = S.dir + (int)S.owner[row_id] * DIR_STRIDE;
				for (uint32_t c = 0; c < W.dir_dims; c += 8u) *(uint4*)(row + 16u + c) = *(const uint4*)(d + c);
			} else {
				const uint32_t k_end = (uint32_t)WIDE_TILE_K * W.layers[l + 1u].n_kblocks;
				for (uint32_t c = 16u + W.dir_dims; c < k_end; c += 8u) {
					const half_t v = c < W.rgb_in ? (half_t)1.0f : (half_t)0.0f;
					const half8 fill = {v, v, v, v, v, v, v, v};
					*(half8*)(row + c) = fill;
				}
			}
			lds_barrier();
		} else {
#if WIDE_MFMA16
			if ((lane >> 4) == 0) {
#pragma unroll
				for (int q = 0; q < 2; ++q) *(uint2*)row_meta(S, 32 * wave + 16 * q + (lane & 15)) = pack4(acc.t[q][0], acc.t[q][1], acc.t[q][2], 0.f, false);
			}
#else
			if (h == 0) *(uint2*)row_meta(S, 32 * wave + n) = pack4(acc[0], acc[1], acc[2], 0.f, false);
#endif
			lds_barrier();
			if (my_row >= 0) {
				union { uint2 u; half_t hh[4]; } r;
				r.u = *(const uint2*)row_meta(S, my_row);
				o.r = r.hh[0]; o.g = r.hh[1]; o.b = r.hh[2];
			}
		}
	}
	return o;
}

#if !WIDE_MFMA16
template <int MT>
NGP_DEV LayerFrags layer_frags_t(const WideModel& W, int l, int wave, int lane) {
	LayerFrags L;
	if (l < 0) {
		L.base = nullptr; L.nkb = 0; L.mt = 0;
		return L;
	}
	L.nkb = (int)W.layers_t[l].n_kblocks;
	L.mt = MT;
	L.base = W.frags + W.layers_t[l].frag_offset + (size_t)(wave * MT) * L.nkb * 64 + lane;
	return L;
}
// ERenderMode::Normals for this architecture: what tcnn's DifferentiableObject::input_gradient(stream, 3, ...) computes for the workgroup's 128 rows
// (src/testbed_nerf.cu:2106-2107; the CPU checker restates it for the tests). NerfNetwork::backward_impl routes the one-hot loss gradient
// (backprop_scale 128 at output 3) to the density network's output 0 only -- the rgb network does not see it --, so: the density network forward
// with every hidden layer's ReLU mask kept as bits (`mask`: n_hidden_density x ROWS x 32 bytes), g = 128 W_out[0][:] under the last mask, the hidden
// layers TRANSPOSED on the same GEMM kernel (fp16 gradients between layers, each masked by the layer below), and the encoding's own derivative:
// Frequency: dL_dx[d] = sum_k (float)dL_dy[d 2F + k] * (2^f pi cos(input_k)) in fp32 (tcnn encodings/frequency.h); Identity: dL_dy[d].
// Entry and exit as wide_network; the returned sigma is the density logit, (gx, gy, gz) the gradient / 128.
template <int MT>
NGP_DEV WideOut wide_density_gradient(const WideModel& W, WideShared& S, uint8_t* mask, int tid, u32x4 (&ar)[RING][MT], int my_row) {
	const int wave = tid >> 6, lane = tid & 63;
	const int n = lane & 31, h = lane >> 5;
	const int row_id = tid & (ROWS - 1), part = tid >> 7;
	const int NH = (int)W.n_hidden_density;
	WideOut o;
	o.r = o.g = o.b = o.sigma = (half_t)0.0f;
	o.gx = o.gy = o.gz = 0.0f;
	for (int l = 0; l < NH; ++l) {
		const LayerFrags cur = layer_frags<MT>(W, (uint32_t)l, wave, lane), next = layer_frags<MT>(W, (uint32_t)l + 1u, wave, lane);
		if (cur.nkb == K256) wide_hidden_layer<MT, K256, 1>(S.x, ar, cur.base, wave, lane, next, nullptr, mask + (size_t)l * ROWS * 32);
		else wide_hidden_layer<MT, K128, 1>(S.x, ar, cur.base, wave, lane, next, nullptr, mask + (size_t)l * ROWS * 32);
	}
	{ // the density output layer: only the logit is needed (and kept: column 0 of the wave's own rows)
		const LayerFrags cur = layer_frags<MT>(W, (uint32_t)NH, wave, lane), next = layer_frags_t<MT>(W, NH - 1, wave, lane);
		const floatx16 acc = cur.nkb == K256 ? wide_out_layer<MT, K256>(S.x, ar, cur.base, wave, lane, next) : wide_out_layer<MT, K128>(S.x, ar, cur.base, wave, lane, next);
		if (h == 0) *(uint2*)(S.x + (32 * wave + n) * XS) = pack4(acc[0], acc[1], acc[2], acc[3], false);
	}
	lds_barrier();
	if (my_row >= 0) o.sigma = S.x[my_row * XS];
	lds_barrier(); // (the logits have been read: the rows may be overwritten)
	{ // the loss gradient at the last hidden layer: 128 * W_out[0][:] where that layer's activation is positive; two threads per row
		const half_t* w0 = (const half_t*)(W.frags + W.out_row0_offset);
		const uint8_t* mrow = mask + (size_t)(NH - 1) * ROWS * 32 + row_id * 32;
		half_t* xrow = S.x + row_id * XS;
		const uint32_t half_w = W.width / 2u;
		for (uint32_t c = (uint32_t)part * half_w; c < (uint32_t)(part + 1) * half_w; c += 8u) {
			const half8 wv = *(const half8*)(w0 + c);
			const uint32_t bits = mrow[c >> 3];
			half8 g;
#pragma unroll
			for (int i = 0; i < 8; ++i) g[i] = (bits >> i) & 1u ? (half_t)((half_t)128.0f * wv[i]) : (half_t)0.0f;
			*(half8*)(xrow + c) = g;
		}
	}
	lds_barrier();
	for (int l = NH - 1; l >= 1; --l) {
		const LayerFrags cur = layer_frags_t<MT>(W, l, wave, lane), next = layer_frags_t<MT>(W, l - 1, wave, lane);
		if (cur.nkb == K256) wide_hidden_layer<MT, K256, 2>(S.x, ar, cur.base, wave, lane, next, nullptr, mask + (size_t)(l - 1) * ROWS * 32);
		else wide_hidden_layer<MT, K128, 2>(S.x, ar, cur.base, wave, lane, next, nullptr, mask + (size_t)(l - 1) * ROWS * 32);
	}
	{
		const LayerFrags cur = layer_frags_t<MT>(W, 0, wave, lane), next = layer_frags_t<MT>(W, -1, wave, lane);
		if (cur.nkb == K256) wide_hidden_layer<MT, K256, 3>(S.x, ar, cur.base, wave, lane, next);
		else wide_hidden_layer<MT, K128, 3>(S.x, ar, cur.base, wave, lane, next);
	}
	// columns [0, enc_dims) of a row now hold dL/d(encoding) in fp16; the encoding's derivative by the row's two threads (the masks are dead:
	// their memory carries the second thread's partial sums)
	const float4 p = *row_meta(S, row_id);
	const half_t* g = S.x + row_id * XS;
	float gs[3] = {0.f, 0.f, 0.f};
	if (W.pos_identity) {
		if (part == 0) { gs[0] = (float)g[0]; gs[1] = (float)g[1]; gs[2] = (float)g[2]; }
	} else {
		const float PI = 3.14159265358979323846f;
		const uint32_t F = W.pos_freqs, split = (F + 1u) / 2u;
		const float in[3] = {p.x, p.y, p.z};
		for (uint32_t f = part ? split : 0u; f < (part ? F : split); ++f) {
			const float scale = __builtin_ldexpf(PI, (int)f);
#pragma unroll
			for (int d = 0; d < 3; ++d) {
				const float v = __builtin_ldexpf(in[d], (int)f);
				const half2_t gg = *(const half2_t*)(g + (uint32_t)d * 2u * F + 2u * f);
				gs[d] += (float)gg[0] * (scale * encoder_cos(__builtin_fmaf(v, PI, 0.0f)));
				gs[d] += (float)gg[1] * (scale * encoder_cos(__builtin_fmaf(v, PI, PI / 2.0f)));
			}
		}
	}
	float* partial = (float*)mask;
	if (part == 1) { partial[row_id * 4 + 0] = gs[0]; partial[row_id * 4 + 1] = gs[1]; partial[row_id * 4 + 2] = gs[2]; }
	lds_barrier();
	if (part == 0) *row_meta(S, row_id) = make_float4((gs[0] + partial[row_id * 4 + 0]) * (1.0f / 128.0f), (gs[1] + partial[row_id * 4 + 1]) * (1.0f / 128.0f), (gs[2] + partial[row_id * 4 + 2]) * (1.0f / 128.0f), 0.0f);
	lds_barrier();
	if (my_row >= 0) {
		const float4 gq = *row_meta(S, my_row);
		o.gx = gq.x; o.gy = gq.y; o.gz = gq.z;
	}
	return o;
}
#endif

NGP_DEV void encode_direction(const WideModel& W, WideShared& S, int slot, f3 d) {
	const float dx = (d.x + 1.0f) * 0.5f, dy = (d.y + 1.0f) * 0.5f, dz = (d.z + 1.0f) * 0.5f;
	half_t* out = S.dir + slot * DIR_STRIDE;
	if (W.dir_identity) identity_encode(W.dir_dims, dx, dy, dz, out);
	else if (W.dir_freqs) frequency_encode(W.dir_freqs, W.dir_dims, dx, dy, dz, out, 0u, W.dir_freqs, true);
	else sh4_all(dx, dy, dz, out);
}

// the position encodings of the round's samples: row r by threads r (low frequencies) and r + 128 (high frequencies, padding)
NGP_DEV void encode_positions(const WideModel& W, WideShared& S, int tid) {
	const int row = tid & (ROWS - 1), part = tid >> 7;
	const float4 p = *row_meta(S, row);
	if (p.w == 0.0f) return;
	if (W.pos_identity) {
		if (part == 0) identity_encode(W.enc_dims, p.x, p.y, p.z, S.x + row * XS, (uint32_t)WIDE_TILE_K * W.layers[0].n_kblocks);
		return;
	}
	const uint32_t split = (W.pos_freqs + 1u) / 2u;
	frequency_encode(W.pos_freqs, W.enc_dims, p.x, p.y, p.z, S.x + row * XS, part ? split : 0u, part ? W.pos_freqs : split, part != 0, (uint32_t)WIDE_TILE_K * W.layers[0].n_kblocks);
}

template <bool PROBE, int MT, bool NORMALS = false>
NGP_DEV void wide_body(const ModelParams& M, const CameraParams& C, const FrameParams& F, const ProbeParams& P) {
	__shared__ WideShared S;
	__shared__ __attribute__((aligned(16))) uint8_t s_mask[NORMALS ? WIDE_MAX_NORMALS_LAYERS * ROWS * 32 : 16]; // Normals: the hidden layers' ReLU masks (one workgroup per CU then)
	const WideModel& W = M.wide;
	const int tid = threadIdx.x, lane = tid & 63;
	const uint32_t max_cascade = M.max_cascade;
	const float cone_angle = M.cone_angle;
	if (tid == 0) atomicMax(&F.results[4], ~realtime());
	if (tid < (int)NERF_CASCADES * 16) S.coarse16[tid] = M.coarse[NERF_CASCADES * COARSE_WORDS_PER_MIP + tid];
	__syncthreads();
	OccBlockCache occ_cache;
	occ_cache.key = 0xffffffffu;
	occ_cache.bits = make_uint2(0u, 0u);
	const float* cam_last = C.moving ? C.m1 : C.m;
	const f3 cam_fwd = mk3(cam_last[6], cam_last[7], cam_last[8]);
	const f3 cam_pos = mk3(cam_last[9], cam_last[10], cam_last[11]);
	const f3 bg_linear = (PROBE || !F.direct) ? mk3(0.f, 0.f, 0.f)
	                     : F.color_space == 1 ? mk3(F.background[0], F.background[1], F.background[2])
	                                          : mk3(srgb_to_linear(F.background[0]), srgb_to_linear(F.background[1]), srgb_to_linear(F.background[2]));
	const float4 empty_pixel = (!PROBE && F.direct) ? tonemap_pixel(F, bg_linear, 0.f, 0.f, 0.f, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
	const f3 amin = mk3(M.aabb_min[0], M.aabb_min[1], M.aabb_min[2]);
	const f3 adiag = mk3(M.aabb_diag[0], M.aabb_diag[1], M.aabb_diag[2]);

	RayState ray;
	ray.alive = false;
	ray.o = ray.d = mk3(0.f, 0.f, 0.f);
	ray.t = 0.f;
	ray.idx = ray.out = 0;
	f3 idir = mk3(0.f, 0.f, 0.f);
	Accum acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
	uint32_t step = 1, skip_i = 1;
	bool ready = false, counted = false, finished = false, exhausted = false;
	bool held = false; // ready, but the round's 128 rows went to others: first in line next round
	float wx = 0.f, wy = 0.f, wz = 0.f, wdt = 0.f;
	uint32_t n_alive_init = 0, n_hit = 0, n_samples = 0;
	int stall = 0;
	// diagnostic (NGP_PROFILE_SECTIONS=1: FrameParams::prof): cycle sums per section, taken by one lane per workgroup
	const bool prof = F.prof != nullptr && tid == 0;
	unsigned long long t0 = 0;
	if (tid < 16) S.prof[tid] = 0ull;
	__syncthreads();
	unsigned long long t2 = 0;
	auto lap2 = [&](int k) { // finer: [10] march loop, [11] decision, [12] rows + prefetch, [13] encode, [14] composite; [8] K loops, [9] epilogues (wide_hidden_layer)
		const unsigned long long t1 = stamp();
		S.prof[k] += t1 - t2;
		t2 = t1;
	};
	auto lap = [&](int section) { // [0] refill, [1] march + decision, [2] network, [3] encode + composite
		const unsigned long long t1 = stamp();
		S.prof[section] += t1 - t0;
		t0 = t1;
	};

	for (;;) {
		if (prof) { t0 = stamp(); S.prof[4] += 1ull; }
		// ---- retire finished rays (K7) and refill free slots from the strip queue (K1 + the jitter of K2), per wave
		const unsigned long long dead_mask = __ballot(!ray.alive);
		const int n_dead = __popcll(dead_mask);
		if (__any(finished)) {
			bool hit = false;
			if (finished) {
				hit = shade_ray<PROBE, false, NORMALS>(F, P, bg_linear, ray.out, acc, step - 1u, ray.d);
				finished = false;
			}
			n_hit += (uint32_t)__popcll(__ballot(hit));
		}
		if (!exhausted && n_dead >= 16) {
			const uint32_t want = (uint32_t)n_dead >> 4, n_strips = F.n_local_tiles * 4u;
			uint32_t first = 0;
			if (lane == 0) first = atomicAdd(F.queue, want);
			first = __builtin_amdgcn_readfirstlane(first);
			if (first >= n_strips) {
				exhausted = true;
			} else {
				const uint32_t got = n_strips - first < want ? n_strips - first : want;
				const uint32_t r = lanes_below(dead_mask);
				const uint32_t strip = first + (r >> 4);
				const bool take = !ray.alive && r < got * 16u;
				const uint32_t tile_local = strip >> 2, s4 = strip & 3u, i16 = r & 15u; // a strip = a 4 x 4 quarter of the tile
				const uint32_t slot = ((s4 >> 1) * 4u + (i16 >> 2)) * 8u + (s4 & 1u) * 4u + (i16 & 3u);
				const uint32_t tile = F.shard_index + F.shard_count * tile_local;
				bool fresh = false;
				if (PROBE) {
					const uint32_t q = tile * 64u + slot;
					if (take && q < P.n_rays) {
						init_probe_ray(P, q, ray);
						fresh = true;
					}
				} else if (take) {
					const uint32_t x = (tile % F.tiles_x) * 8u + (slot & 7u);
					const uint32_t y = (tile / F.tiles_x) * 8u + (slot >> 3);
					if (x < (uint32_t)C.width && y < (uint32_t)C.height) {
						init_ray<false>(M, C, x, y, ray);
						if (F.packed) ray.out = tile_local * 64u + slot;
						if (!F.envmap) {
							if (F.direct) {
								F.frame_buffer[ray.out] = empty_pixel;
								F.depth_buffer[ray.out] = MAX_DEPTH;
							} else if (F.depth_buffer[ray.out] < 0.01f) {
								F.depth_buffer[ray.out] = MAX_DEPTH;
							}
						} else {
							const bool valid = ray.d.x != 0.0f || ray.d.y != 0.0f || ray.d.z != 0.0f;
							float env[4] = {0.f, 0.f, 0.f, 0.f};
							if (valid) {
								float d3[3] = {ray.d.x, ray.d.y, ray.d.z};
								read_envmap(F.envmap, F.env_w, F.env_h, d3, env);
							}
							if (F.direct) {
								F.frame_buffer[ray.out] = valid ? tonemap_pixel(F, bg_linear, env[0], env[1], env[2], env[3]) : empty_pixel;
								F.depth_buffer[ray.out] = MAX_DEPTH;
							} else {
								if (valid) F.frame_buffer[ray.out] = make_float4(env[0], env[1], env[2], env[3]);
								if (F.depth_buffer[ray.out] < 0.01f) F.depth_buffer[ray.out] = MAX_DEPTH;
							}
						}
						if (ray.alive) {
							ray.t = advance_n_steps(ray.t, cone_angle, ld_random_val_dim0(C.spp, ray.idx * 786433u));
							fresh = true;
						}
					}
				}
				if (fresh) {
					encode_direction(W, S, tid, ray.d); // (the slot's own row: nobody else touches it)
					idir = mk3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
					acc = Accum{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
					step = 1;
					skip_i = 1;
					ready = false;
					counted = PROBE;
				}
				if (PROBE) n_alive_init += (uint32_t)__popcll(__ballot(fresh));
			}
		}

		if (prof) { lap(0); t2 = t0; }
		// ---- K4 / K2: if_unoccupied_advance_to_next_occupied_voxel (nerf_device.cuh:461-494): every marching slot walks to its next
		// sample (or out of the box), at most 64 voxels per round so that one long empty stretch does not hold up the workgroup
		bool newly_counted = false;
		if (ray.alive && !ready) {
			for (int k = 0; k < F.tune[1]; ++k) {
				const f3 pos = add3(ray.o, scale3(ray.d, ray.t));
				bool out = ray.t >= MAX_DEPTH || !raabb_contains(M, pos);
				if (PROBE && skip_i >= 200) out = true;
				if (out) {
					ray.alive = false;
					finished = true;
					break;
				}
				uint32_t mip = mip_from_pos(pos, NERF_CASCADES - 1);
				mip = mip > max_cascade ? max_cascade : mip;
				uint32_t empty = empty_block_size_global(pos, M.bitfield, S.coarse16, mip, occ_cache);
				if (empty == 0u) {
					const float dt = calc_dt(ray.t, cone_angle);
					f3 w = sub3(pos, amin);
					if (M.diag_pow2) w = mul3(w, mk3(M.aabb_inv_diag[0], M.aabb_inv_diag[1], M.aabb_inv_diag[2]));
					else w = div3(w, adiag);
					wx = w.x; wy = w.y; wz = w.z;
					wdt = warp_dt(dt);
					ray.t = ray.t + dt;
					ready = true;
					skip_i = 1;
					newly_counted = !counted;
					counted = true;
					break;
				}
				// climb to the largest empty cascade cell around pos (nerf_device.cuh:488-490)
				while (mip < max_cascade) {
					const uint32_t e = empty_block_size_global(pos, M.bitfield, S.coarse16, mip + 1, occ_cache);
					if (e == 0u) break;
					++mip;
					empty = e;
				}
				const float grid_half = 0.5f * (float)(1u << max_cascade);
				const bool outside = !PROBE && empty == 1u && mip == max_cascade &&
				                     fmaxf(fmaxf(__builtin_fabsf(pos.x - 0.5f), __builtin_fabsf(pos.y - 0.5f)), __builtin_fabsf(pos.z - 0.5f)) > grid_half;
				const float to_grid = outside ? grid_cube_entry(pos, idir, grid_half) : 0.0f;
				if (outside && to_grid < 0.0f) {
					ray.alive = false;
					finished = true;
					break;
				} else if (outside && to_grid > 0.0f) {
					ray.t = advance_by_distance(ray.t, cone_angle, to_grid);
				} else {
					// block jumps (FrameParams::tune[6], on by default like the base.json kernel): leave an empty 4^3 / 16^3 block in one step
					ray.t = advance_to_next_voxel(ray.t, cone_angle, pos, ray.d, idir, mip, (PROBE || !F.tune[6]) ? 1u : empty);
				}
				++skip_i;
			}
		}
		n_alive_init += (uint32_t)__popcll(__ballot(newly_counted));
		if (prof) lap2(10);

		// ---- workgroup decision: run the network once (nearly) a round's worth of samples waits or nothing else can add to them
		const int wave_id = tid >> 6;
		const unsigned long long held_mask = __ballot(ready && held), new_mask = __ballot(ready && !held);
		if (lane == 0) {
			S.cnt[wave_id] = (uint32_t)__popcll(held_mask);
			S.cnt[4 + wave_id] = (uint32_t)__popcll(new_mask);
		}
		const int n_progress = __syncthreads_count((ray.alive && !ready) || finished || (!ray.alive && !exhausted));
		uint32_t held_before = 0, new_before = 0, held_total = 0, new_total = 0;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const uint32_t hc = S.cnt[k], nc = S.cnt[4 + k];
			if (k < wave_id) { held_before += hc; new_before += nc; }
			held_total += hc;
			new_total += nc;
		}
		const int n_ready = (int)(held_total + new_total);
		if (prof) { lap(1); lap2(11); }
		if (n_ready == 0) {
			if (n_progress == 0) break;
			__syncthreads(); // (S.cnt is rewritten next round)
			continue;
		}
		if (n_ready < F.tune[2] && n_progress > 0 && stall < F.tune[3]) {
			++stall;
			__syncthreads();
			continue;
		}
		stall = 0;

		// ---- the round's rows: samples that waited go first, then new ones in slot order
		const uint32_t rank = (ready && held) ? held_before + lanes_below(held_mask) : held_total + new_before + lanes_below(new_mask);
		const bool run = ready && rank < (uint32_t)ROWS;
		const int my_row = run ? (int)rank : -1;
		held = ready && !run;
		// ---- K5: positions to the rows, encodings by thread pairs, then the network on the whole block
		if (run) {
			*row_meta(S, my_row) = make_float4(wx, wy, wz, 1.0f);
			S.owner[my_row] = (uint16_t)tid;
		}
		if (tid < ROWS && tid >= n_ready) {
			*row_meta(S, tid) = make_float4(0.f, 0.f, 0.f, 0.f);
			S.owner[tid] = 0;
		}
		u32x4 ar[RING][MT];
		wide_network_prefetch<MT>(W, ar, tid); // the first layer's weights travel while the sines are computed
		lds_barrier();
		if (prof) lap2(12);
		encode_positions(W, S, tid);
		lds_barrier();
		if (prof) lap2(13);
		if (prof) { lap(3); S.prof[5] += 1ull; S.prof[7] += (unsigned long long)(n_ready < ROWS ? n_ready : ROWS); }
#if WIDE_MFMA16
		const WideOut o = wide_network<MT>(W, S, tid, ar, my_row, prof ? S.prof : nullptr);
#else
		const WideOut o = NORMALS ? wide_density_gradient<MT>(W, S, s_mask, tid, ar, my_row) : wide_network<MT>(W, S, tid, ar, my_row, prof ? S.prof : nullptr);
#endif
		if (prof) { lap(2); t2 = t0; }

		// ---- K6: composite_kernel_nerf (:569-726)
		if (run) {
			ready = false;
			const f3 pos = add3(amin, mul3(mk3(wx, wy, wz), adiag));
			const float sdepth = dot3(cam_fwd, sub3(pos, cam_pos));
			const float T = 1.0f - acc.a;
			const float dt = unwarp_dt(wdt);
			const float alpha = 1.0f - fast_exp(-network_to_density((float)o.sigma, M.density_act) * dt);
			const float weight = alpha * T;
			float cr = network_to_rgb((float)o.r, M.rgb_act), cg = network_to_rgb((float)o.g, M.rgb_act), cb = network_to_rgb((float)o.b, M.rgb_act);
			if (!PROBE && F.render_mode > 1) {
				if (NORMALS) { // src/testbed_nerf.cu:688-693: opposite to the density gradient
					const float dd = network_to_density_derivative((float)o.sigma, M.density_act);
					const f3 nrm = normalize3(mk3(-dd * o.gx, -dd * o.gy, -dd * o.gz));
					cr = nrm.x; cg = nrm.y; cb = nrm.z;
				} else if (F.render_mode == 2) {
					cr = cg = cb = alpha;
				} else if (F.render_mode == 3) {
					cr = (pos.x - 0.5f) / 2.0f + 0.5f; cg = (pos.y - 0.5f) / 2.0f + 0.5f; cb = (pos.z - 0.5f) / 2.0f + 0.5f;
				} else if (F.render_mode == 4) {
					cr = cg = cb = dot3(cam_fwd, sub3(pos, ray.o)) * F.depth_scale;
				}
			}
			acc.r += cr * weight;
			acc.g += cg * weight;
			acc.b += cb * weight;
			acc.a += weight;
			if (weight > acc.max_weight) {
				acc.max_weight = weight;
				acc.depth = sdepth;
			}
			++step;
			if (acc.a > (1.0f - F.min_transmittance)) {
				acc.r /= acc.a; acc.g /= acc.a; acc.b /= acc.a; acc.a /= acc.a;
				ray.alive = false;
				finished = true;
			} else if (step >= MARCH_ITER) {
				ray.alive = false;
			}
		}
		n_samples += (uint32_t)__popcll(__ballot(run));
		if (prof) { lap(3); lap2(14); }
	}
	if (prof) { // [refill, march + decision, network, encode + composite] cycles, rounds, network rounds (twice: the host divides by both), samples
		for (int k = 0; k < 4; ++k) atomicAdd(&F.prof[k], S.prof[k]);
		atomicAdd(&F.prof[4], S.prof[4]);
		atomicAdd(&F.prof[5], S.prof[5]);
		atomicAdd(&F.prof[6], S.prof[5]);
		atomicAdd(&F.prof[7], S.prof[7]);
		for (int k = 8; k < 16; ++k) atomicAdd(&F.prof[56 + k], S.prof[k]); // [64..71]
	}
	finish_launch(F, lane, n_alive_init, n_hit, n_samples);
}

#define NGP_WIDE_KERNEL __global__ __launch_bounds__(WBLOCK) __attribute__((amdgpu_waves_per_eu(2, 2)))
#define NGP_WIDE_KERNEL_1 __global__ __launch_bounds__(WBLOCK) // (the Normals kernels: 113 KB of LDS, one workgroup per CU, one wave per SIMD)

NGP_WIDE_KERNEL void render_nerf_wide256(const ModelParams M, const CameraParams C, const FrameParams F) {
	ProbeParams P{};
	wide_body<false, 64 / WIDE_TILE_M>(M, C, F, P);
}
NGP_WIDE_KERNEL void render_nerf_wide128(const ModelParams M, const CameraParams C, const FrameParams F) {
	ProbeParams P{};
	wide_body<false, 32 / WIDE_TILE_M>(M, C, F, P);
}
NGP_WIDE_KERNEL void trace_probe_wide256(const ModelParams M, const FrameParams F, const ProbeParams P) {
	CameraParams C{};
	wide_body<true, 64 / WIDE_TILE_M>(M, C, F, P);
}
NGP_WIDE_KERNEL void trace_probe_wide128(const ModelParams M, const FrameParams F, const ProbeParams P) {
	CameraParams C{};
	wide_body<true, 32 / WIDE_TILE_M>(M, C, F, P);
}

#if !WIDE_MFMA16
// ERenderMode::Normals: the density network's backward pass per round; 32 KB of masks beside the activations = one workgroup per CU
NGP_WIDE_KERNEL_1 void render_nerf_wide256_normals(const ModelParams M, const CameraParams C, const FrameParams F) {
	ProbeParams P{};
	wide_body<false, 64 / WIDE_TILE_M, true>(M, C, F, P);
}
NGP_WIDE_KERNEL_1 void render_nerf_wide128_normals(const ModelParams M, const CameraParams C, const FrameParams F) {
	ProbeParams P{};
	wide_body<false, 32 / WIDE_TILE_M, true>(M, C, F, P);
}
#endif

// NerfNetwork::inference on explicit inputs (ngp_network_inference): 128 samples per workgroup round; GRADIENT: the density logit's input
// gradient instead (ngp_density_gradient: 3 floats per sample into `out`)
template <int MT, bool GRADIENT = false>
NGP_DEV void wide_inference_body(const ModelParams& M, uint32_t n, const float* __restrict__ pos01, const float* __restrict__ dir01, uint16_t* __restrict__ out) {
	__shared__ WideShared S;
	__shared__ __attribute__((aligned(16))) uint8_t s_mask[GRADIENT ? WIDE_MAX_NORMALS_LAYERS * ROWS * 32 : 16];
	const WideModel& W = M.wide;
	const int tid = threadIdx.x, row = tid & (ROWS - 1), part = tid >> 7;
	for (uint32_t base = blockIdx.x * ROWS; base < n; base += gridDim.x * ROWS) { // (workgroup-uniform trip count)
		const uint32_t i = base + (uint32_t)row;
		const bool run = i < n;
		u32x4 ar[RING][MT];
		wide_network_prefetch<MT>(W, ar, tid);
		if (part == 0) { // row r carries sample base + r; its direction sits in slot r
			*row_meta(S, row) = run ? make_float4(pos01[3 * (size_t)i], pos01[3 * (size_t)i + 1], pos01[3 * (size_t)i + 2], 1.0f) : make_float4(0.f, 0.f, 0.f, 0.f);
			S.owner[row] = (uint16_t)row;
		} else if (run && !GRADIENT) {
			half_t* d = S.dir + row * DIR_STRIDE;
			if (W.dir_identity) identity_encode(W.dir_dims, dir01[3 * (size_t)i], dir01[3 * (size_t)i + 1], dir01[3 * (size_t)i + 2], d);
			else if (W.dir_freqs) frequency_encode(W.dir_freqs, W.dir_dims, dir01[3 * (size_t)i], dir01[3 * (size_t)i + 1], dir01[3 * (size_t)i + 2], d, 0u, W.dir_freqs, true);
			else sh4_all(dir01[3 * (size_t)i], dir01[3 * (size_t)i + 1], dir01[3 * (size_t)i + 2], d);
		}
		lds_barrier();
		encode_positions(W, S, tid);
		lds_barrier();
#if WIDE_MFMA16
		const WideOut o = wide_network<MT>(W, S, tid, ar, (run && part == 0) ? row : -1);
#else
		const WideOut o = GRADIENT ? wide_density_gradient<MT>(W, S, s_mask, tid, ar, (run && part == 0) ? row : -1) : wide_network<MT>(W, S, tid, ar, (run && part == 0) ? row : -1);
#endif
		if (run && part == 0) {
			if (GRADIENT) {
				float* og = (float*)out + 3 * (size_t)i;
				og[0] = o.gx; og[1] = o.gy; og[2] = o.gz;
			} else {
				union { half_t h[4]; uint2 u; } p;
				p.h[0] = o.r; p.h[1] = o.g; p.h[2] = o.b; p.h[3] = o.sigma;
				*(uint2*)(out + 4 * (size_t)i) = p.u;
			}
		}
		lds_barrier(); // (the next round's positions overwrite what this round's threads may still read)
	}
}
NGP_WIDE_KERNEL void network_inference_wide256(const ModelParams M, uint32_t n, const float* __restrict__ pos01, const float* __restrict__ dir01, uint16_t* __restrict__ out) {
	wide_inference_body<64 / WIDE_TILE_M>(M, n, pos01, dir01, out);
}
NGP_WIDE_KERNEL void network_inference_wide128(const ModelParams M, uint32_t n, const float* __restrict__ pos01, const float* __restrict__ dir01, uint16_t* __restrict__ out) {
	wide_inference_body<32 / WIDE_TILE_M>(M, n, pos01, dir01, out);
}
#if !WIDE_MFMA16
NGP_WIDE_KERNEL_1 void density_gradient_wide256(const ModelParams M, uint32_t n, const float* __restrict__ pos01, float* __restrict__ out) {
	wide_inference_body<64 / WIDE_TILE_M, true>(M, n, pos01, pos01, (uint16_t*)out);
}
NGP_WIDE_KERNEL_1 void density_gradient_wide128(const ModelParams M, uint32_t n, const float* __restrict__ pos01, float* __restrict__ out) {
	wide_inference_body<32 / WIDE_TILE_M, true>(M, n, pos01, pos01, (uint16_t*)out);
}
#endif
// the position encoding alone (ngp_grid_encode's counterpart for this architecture): n x enc_dims halves
__global__ void frequency_encode_kernel(const ModelParams M, uint32_t n, const float* __restrict__ pos01, uint16_t* __restrict__ out) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	half_t* o = (half_t*)out + (size_t)M.wide.enc_dims * i;
	if (M.wide.pos_identity) identity_encode(M.wide.enc_dims, pos01[3 * (size_t)i], pos01[3 * (size_t)i + 1], pos01[3 * (size_t)i + 2], o);
	else frequency_encode(M.wide.pos_freqs, M.wide.enc_dims, pos01[3 * (size_t)i], pos01[3 * (size_t)i + 1], pos01[3 * (size_t)i + 2], o, 0u, M.wide.pos_freqs, true);
}

// ---------------------------------------------------------------------------------------------------------
// launchers: persistent grids of what is resident at once -- two workgroups per CU (80 KB of LDS, 256 registers per lane each)
template <typename K>
static int wide_blocks_per_cu(K kernel) {
	int n = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, WBLOCK, 0) != hipSuccess || n < 1) n = 1;
	return n;
}
// The round's schedule (FrameParams::tune as this kernel reads it; of ngp_set_schedule's knobs only block_jumps applies here):
// [1] voxel steps a marching slot may take per round, [2] run the network once this many samples wait (of the 128 a round takes) ...
// [3] ... or after this many extra march rounds. WIDE_TUNE="steps,go,stall" overrides the defaults (experiments).
static void wide_schedule(FrameParams& G) {
	static const struct T { int v[3]; } t = []() {
		T r = {{4, 128, 2}};
		if (const char* e = getenv("WIDE_TUNE")) (void)sscanf(e, "%d,%d,%d", &r.v[0], &r.v[1], &r.v[2]);
		if (r.v[0] < 1 || r.v[0] > 1024) r.v[0] = 4;
		if (r.v[1] < 1 || r.v[1] > ROWS) r.v[1] = 128;
		if (r.v[2] < 0 || r.v[2] > 64) r.v[2] = 2;
		return r;
	}();
	G.tune[1] = t.v[0];
	G.tune[2] = t.v[1];
	G.tune[3] = t.v[2];
}
static int wide_blocks(const FrameParams& F, int n_cus, int per_cu) {
	if (const char* e = getenv("NGP_BLOCKS_PER_CU")) { int v = atoi(e); if (v > 0 && v < per_cu) per_cu = v; } // experiments only
	int n_blocks = n_cus * per_cu;
	const int needed = (int)((F.n_local_tiles + 3) / 4); // 256 ray slots = four 8x8 tiles per workgroup
	if (n_blocks > needed) n_blocks = needed > 0 ? needed : 1;
	return n_blocks;
}
void launch_render_nerf_wide(const ModelParams& M, const CameraParams& C, const FrameParams& F, int n_cus, hipStream_t stream) {
#if !WIDE_MFMA16
	if (F.render_mode == 7) { // ERenderMode::Normals
		static const int per_cu256n = wide_blocks_per_cu(render_nerf_wide256_normals), per_cu128n = wide_blocks_per_cu(render_nerf_wide128_normals);
		const int nb = wide_blocks(F, n_cus, M.wide.width == 256 ? per_cu256n : per_cu128n);
		FrameParams G = F;
		G.n_waves = (uint32_t)nb * (WBLOCK / 64);
		wide_schedule(G);
		if (M.wide.width == 256) hipLaunchKernelGGL(render_nerf_wide256_normals, dim3(nb), dim3(WBLOCK), 0, stream, M, C, G);
		else hipLaunchKernelGGL(render_nerf_wide128_normals, dim3(nb), dim3(WBLOCK), 0, stream, M, C, G);
		return;
	}
#endif
	static const int per_cu256 = wide_blocks_per_cu(render_nerf_wide256), per_cu128 = wide_blocks_per_cu(render_nerf_wide128);
	const int n_blocks = wide_blocks(F, n_cus, M.wide.width == 256 ? per_cu256 : per_cu128);
	FrameParams G = F;
	G.n_waves = (uint32_t)n_blocks * (WBLOCK / 64);
	wide_schedule(G);
	if (M.wide.width == 256) hipLaunchKernelGGL(render_nerf_wide256, dim3(n_blocks), dim3(WBLOCK), 0, stream, M, C, G);
	else hipLaunchKernelGGL(render_nerf_wide128, dim3(n_blocks), dim3(WBLOCK), 0, stream, M, C, G);
}
void launch_trace_probe_wide(const ModelParams& M, const FrameParams& F, const ProbeParams& P, int n_cus, hipStream_t stream) {
	static const int per_cu256 = wide_blocks_per_cu(trace_probe_wide256), per_cu128 = wide_blocks_per_cu(trace_probe_wide128);
	const int n_blocks = wide_blocks(F, n_cus, M.wide.width == 256 ? per_cu256 : per_cu128);
	FrameParams G = F;
	G.n_waves = (uint32_t)n_blocks * (WBLOCK / 64);
	wide_schedule(G);
	if (M.wide.width == 256) hipLaunchKernelGGL(trace_probe_wide256, dim3(n_blocks), dim3(WBLOCK), 0, stream, M, G, P);
	else hipLaunchKernelGGL(trace_probe_wide128, dim3(n_blocks), dim3(WBLOCK), 0, stream, M, G, P);
}
void launch_network_inference_wide(const ModelParams& M, uint32_t n, const float* pos01, const float* dir01, uint16_t* out, int n_cus, hipStream_t stream) {
	if (n == 0) return;
	int n_blocks = (int)((n + ROWS - 1) / ROWS);
	if (n_blocks > 2 * n_cus) n_blocks = 2 * n_cus;
	if (M.wide.width == 256) hipLaunchKernelGGL(network_inference_wide256, dim3(n_blocks), dim3(WBLOCK), 0, stream, M, n, pos01, dir01, out);
	else hipLaunchKernelGGL(network_inference_wide128, dim3(n_blocks), dim3(WBLOCK), 0, stream, M, n, pos01, dir01, out);
}
void launch_density_gradient_wide(const ModelParams& M, uint32_t n, const float* pos01, float* out, int n_cus, hipStream_t stream) {
	if (n == 0) return;
#if !WIDE_MFMA16
	int n_blocks = (int)((n + ROWS - 1) / ROWS);
	if (n_blocks > n_cus) n_blocks = n_cus;
	if (M.wide.width == 256) hipLaunchKernelGGL(density_gradient_wide256, dim3(n_blocks), dim3(WBLOCK), 0, stream, M, n, pos01, out);
	else hipLaunchKernelGGL(density_gradient_wide128, dim3(n_blocks), dim3(WBLOCK), 0, stream, M, n, pos01, out);
#else
	fprintf(stderr, "[ngp] the density gradient of the wide architecture is not built in the WIDE_MFMA16 experiment\n");
#endif
}
void launch_frequency_encode(const ModelParams& M, uint32_t n, const float* pos01, uint16_t* out, hipStream_t stream) {
	if (n == 0) return;
	hipLaunchKernelGGL(frequency_encode_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, M, n, pos01, out);
}

} // namespace ngp
