/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_common.h). PARITY UNPINNED.
 *
 * NeRF inference path: ray generation, occupancy-grid marching, hash-grid + SH
 * encoding, fully fused MLPs, alpha compositing, shading and the post pass.
 * Every function names the reference file:line it restates; tcnn semantics are
 * restated from the public upstream definition because
 * dependencies/tiny-cuda-nn is absent from the reference mount.
 */
#include "oracle.h"
#include "orc_common.h"

#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ constants
 * nerf_device.cuh:24-42 */
#define NERF_GRIDSIZE 128u
#define NERF_GRID_N_CELLS (128u * 128u * 128u)
#define NERF_STEPS 1024u
#define NERF_CASCADES 8u
static const float SQRT3 = 1.73205080757f;
static inline float STEPSIZE(void) { return SQRT3 / (float)NERF_STEPS; }
static inline float MIN_CONE_STEPSIZE(void) { return STEPSIZE(); }
static inline float MAX_CONE_STEPSIZE(void) { return STEPSIZE() * (float)(1u << (NERF_CASCADES - 1)) * (float)NERF_STEPS / (float)NERF_GRIDSIZE; }
static const float NERF_MIN_OPTICAL_THICKNESS = 0.01f;
static const float MAX_DEPTH = ORC_MAX_DEPTH;

/* ------------------------------------------------------------------ Morton (tcnn common_device.h) */
static inline uint32_t expand_bits(uint32_t v) {
	v = (v * 0x00010001u) & 0xFF0000FFu;
	v = (v * 0x00000101u) & 0x0F00F00Fu;
	v = (v * 0x00000011u) & 0xC30C30C3u;
	v = (v * 0x00000005u) & 0x49249249u;
	return v;
}
static inline uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z) {
	return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
static inline uint32_t morton3D_invert(uint32_t x) {
	x = x & 0x49249249u;
	x = (x | (x >> 2)) & 0xc30c30c3u;
	x = (x | (x >> 4)) & 0x0f00f00fu;
	x = (x | (x >> 8)) & 0xff0000ffu;
	x = (x | (x >> 16)) & 0x0000ffffu;
	return x;
}

/* ------------------------------------------------------------------ colour (common_device.cuh:34-64) */
float orc_srgb_to_linear(float srgb) {
	if (srgb <= 0.04045f) return srgb / 12.92f;
	return powf((srgb + 0.055f) / 1.055f, 2.4f);
}
float orc_linear_to_srgb(float linear) {
	if (linear < 0.0031308f) return 12.92f * linear;
	return 1.055f * powf(linear, 0.41666f) - 0.055f;
}

/* ------------------------------------------------------------------ sampling sequences
 * random_val.cuh:207-370 (Burley 2019 Owen-scrambled Sobol; direction numbers are data). */
static const uint32_t SOBOL_DIRECTIONS[5][32] = {
	{0x80000000, 0x40000000, 0x20000000, 0x10000000, 0x08000000, 0x04000000, 0x02000000, 0x01000000,
	 0x00800000, 0x00400000, 0x00200000, 0x00100000, 0x00080000, 0x00040000, 0x00020000, 0x00010000,
	 0x00008000, 0x00004000, 0x00002000, 0x00001000, 0x00000800, 0x00000400, 0x00000200, 0x00000100,
	 0x00000080, 0x00000040, 0x00000020, 0x00000010, 0x00000008, 0x00000004, 0x00000002, 0x00000001},
	{0x80000000, 0xc0000000, 0xa0000000, 0xf0000000, 0x88000000, 0xcc000000, 0xaa000000, 0xff000000,
	 0x80800000, 0xc0c00000, 0xa0a00000, 0xf0f00000, 0x88880000, 0xcccc0000, 0xaaaa0000, 0xffff0000,
	 0x80008000, 0xc000c000, 0xa000a000, 0xf000f000, 0x88008800, 0xcc00cc00, 0xaa00aa00, 0xff00ff00,
	 0x80808080, 0xc0c0c0c0, 0xa0a0a0a0, 0xf0f0f0f0, 0x88888888, 0xcccccccc, 0xaaaaaaaa, 0xffffffff},
	{0x80000000, 0xc0000000, 0x60000000, 0x90000000, 0xe8000000, 0x5c000000, 0x8e000000, 0xc5000000,
	 0x68800000, 0x9cc00000, 0xee600000, 0x55900000, 0x80680000, 0xc09c0000, 0x60ee0000, 0x90550000,
	 0xe8808000, 0x5cc0c000, 0x8e606000, 0xc5909000, 0x6868e800, 0x9c9c5c00, 0xeeee8e00, 0x5555c500,
	 0x8000e880, 0xc0005cc0, 0x60008e60, 0x9000c590, 0xe8006868, 0x5c009c9c, 0x8e00eeee, 0xc5005555},
	{0x80000000, 0xc0000000, 0x20000000, 0x50000000, 0xf8000000, 0x74000000, 0xa2000000, 0x93000000,
	 0xd8800000, 0x25400000, 0x59e00000, 0xe6d00000, 0x78080000, 0xb40c0000, 0x82020000, 0xc3050000,
	 0x208f8000, 0x51474000, 0xfbea2000, 0x75d93000, 0xa0858800, 0x914e5400, 0xdbe79e00, 0x25db6d00,
	 0x58800080, 0xe54000c0, 0x79e00020, 0xb6d00050, 0x800800f8, 0xc00c0074, 0x200200a2, 0x50050093},
	{0x80000000, 0x40000000, 0x20000000, 0xb0000000, 0xf8000000, 0xdc000000, 0x7a000000, 0x9d000000,
	 0x5a800000, 0x2fc00000, 0xa1600000, 0xf0b00000, 0xda880000, 0x6fc40000, 0x81620000, 0x40bb0000,
	 0x22878000, 0xb3c9c000, 0xfb65a000, 0xddb2d000, 0x78022800, 0x9c0b3c00, 0x5a0fb600, 0x2d0ddb00,
	 0xa2878080, 0xf3c9c040, 0xdb65a020, 0x6db2d0b0, 0x800228f8, 0x400b3cdc, 0x200fb67a, 0xb00ddb9d},
};

static uint32_t sobol(uint32_t index, uint32_t dim) { /* random_val.cuh:207-264 */
	uint32_t X = 0;
	for (uint32_t bit = 0; bit < 32; ++bit) {
		if ((index >> bit) & 1u) X ^= SOBOL_DIRECTIONS[dim][bit];
	}
	return X;
}
static uint32_t hash_combine(uint32_t seed, uint32_t v) { return seed ^ (v + (seed << 6) + (seed >> 2)); }
static uint32_t reverse_bits(uint32_t x) {
	x = (((x & 0xaaaaaaaau) >> 1) | ((x & 0x55555555u) << 1));
	x = (((x & 0xccccccccu) >> 2) | ((x & 0x33333333u) << 2));
	x = (((x & 0xf0f0f0f0u) >> 4) | ((x & 0x0f0f0f0fu) << 4));
	x = (((x & 0xff00ff00u) >> 8) | ((x & 0x00ff00ffu) << 8));
	return (x >> 16) | (x << 16);
}
static uint32_t laine_karras_permutation(uint32_t x, uint32_t seed) {
	x += seed;
	x ^= x * 0x6c50b47cu;
	x ^= x * 0xb82f1e52u;
	x ^= x * 0xc7afe638u;
	x ^= x * 0x8d22f6e6u;
	return x;
}
static uint32_t nested_uniform_scramble_base2(uint32_t x, uint32_t seed) {
	x = reverse_bits(x);
	x = laine_karras_permutation(x, seed);
	x = reverse_bits(x);
	return x;
}
static const float LD_SCALE = 2.3283064365386963e-10f; /* float(1.0/(1ull<<32)) */

float orc_ld_random_val(uint32_t index, uint32_t seed, uint32_t dim) { /* random_val.cuh:332-336 */
	index = nested_uniform_scramble_base2(index, seed);
	return (float)nested_uniform_scramble_base2(sobol(index, dim), hash_combine(seed, dim)) * LD_SCALE;
}
static void ld_random_val_2d(uint32_t index, uint32_t seed, float* out) { /* random_val.cuh:311-330 */
	index = nested_uniform_scramble_base2(index, seed);
	for (uint32_t i = 0; i < 2; ++i) {
		uint32_t x = nested_uniform_scramble_base2(sobol(index, i), hash_combine(seed, i));
		out[i] = (float)x * LD_SCALE;
	}
}
static float fractf_(float x) { return x - floorf(x); }
void orc_ld_random_pixel_offset(uint32_t spp, float* out2) { /* random_val.cuh:365-370 */
	float a[2], b[2];
	ld_random_val_2d(0, 0xdeadbeefu, a);
	ld_random_val_2d(spp, 0xdeadbeefu, b);
	out2[0] = fractf_((0.5f - a[0]) + b[0]);
	out2[1] = fractf_((0.5f - a[1]) + b[1]);
}

/* ------------------------------------------------------------------ prepared model */
typedef struct {
	uint32_t offsets[ORC_MAX_LEVELS + 1];
	uint32_t resolutions[ORC_MAX_LEVELS];
	float scales[ORC_MAX_LEVELS];
	uint32_t enc_dims;
	/* float copies of the fp16 weights, row-major out x in per layer */
	float* density_w;
	float* rgb_w;
	uint64_t n_density_w, n_rgb_w;
	const uint16_t* grid;
	aabb_t aabb, render_aabb;
	uint32_t dir_dims;   /* width of the direction encoding (16 for SH degree 4, 6 n_freq for Frequency), padded to the alignment */
	uint32_t rgb_in;     /* next_multiple(density out + dir_dims, alignment), nerf_network.h:97 */
	uint32_t rgb_out;    /* next_multiple(3, alignment): 16 FullyFusedMLP, 8 CutlassMLP */
} prepared_t;

static uint32_t next_multiple_u32(uint32_t v, uint32_t d) { return ((v + d - 1) / d) * d; }

/* tcnn GridEncoding constructor + grid_scale/grid_resolution. */
int orc_grid_layout(const orc_nerf_model* m, uint32_t* offsets, uint32_t* resolutions, float* scales) {
	if (m->n_levels == 0 || m->n_levels > ORC_MAX_LEVELS) return -1;
	float log2_pls = log2f(m->per_level_scale);
	uint32_t offset = 0;
	for (uint32_t l = 0; l < m->n_levels; ++l) {
		float scale = exp2f((float)l * log2_pls) * (float)m->base_resolution - 1.0f;
		uint32_t res = (uint32_t)ceilf(scale) + 1u;
		uint32_t max_params = 0xFFFFFFFFu / 2u;
		uint32_t params_in_level = powf((float)res, 3.0f) > (float)max_params ? max_params : res * res * res;
		params_in_level = next_multiple_u32(params_in_level, 8u);
		uint32_t cap = 1u << m->log2_hashmap_size;
		if (params_in_level > cap) params_in_level = cap;
		offsets[l] = offset;
		resolutions[l] = res;
		scales[l] = scale;
		offset += params_in_level;
	}
	offsets[m->n_levels] = offset;
	return 0;
}

static uint64_t mlp_n_params(uint32_t in, uint32_t width, uint32_t n_hidden, uint32_t out_padded) {
	if (n_hidden == 0) return (uint64_t)out_padded * in; /* tcnn CutlassMLP with no hidden layer: one (padded output) x (input) matrix */
	return (uint64_t)width * in + (uint64_t)(n_hidden - 1) * width * width + (uint64_t)out_padded * width;
}

static uint32_t align_up(uint32_t v, uint32_t a) { return (v + a - 1) / a * a; }
static void network_shapes(const orc_nerf_model* m, uint32_t* enc_dims, uint32_t* dir_dims, uint32_t* rgb_in, uint32_t* rgb_out) {
	const uint32_t al = m->mlp_alignment ? m->mlp_alignment : 16u;
	*enc_dims = m->pos_encoding == 1 ? align_up(3u * 2u * m->pos_n_frequencies, al) : m->pos_encoding == 2 ? align_up(3u, al) : m->n_levels * m->n_features_per_level;
	*dir_dims = m->dir_encoding == 1 ? align_up(3u * 2u * m->dir_n_frequencies, al) : m->dir_encoding == 2 ? align_up(3u, al) : 16u;
	*rgb_in = align_up(m->density_out_dims + *dir_dims, al);
	*rgb_out = align_up(3u, al);
}

uint64_t orc_n_params(const orc_nerf_model* m) {
	uint32_t offsets[ORC_MAX_LEVELS + 1], res[ORC_MAX_LEVELS];
	float scales[ORC_MAX_LEVELS];
	uint32_t enc, dir, rgb_in, rgb_out;
	network_shapes(m, &enc, &dir, &rgb_in, &rgb_out);
	uint64_t ng = 0;
	if (m->pos_encoding == 0) {
		if (orc_grid_layout(m, offsets, res, scales)) return 0;
		ng = (uint64_t)offsets[m->n_levels] * m->n_features_per_level;
	}
	uint64_t nd = mlp_n_params(enc, m->n_neurons, m->n_hidden_density, m->density_out_dims);
	uint64_t nr = mlp_n_params(rgb_in, m->n_neurons, m->n_hidden_rgb, rgb_out);
	return nd + nr + ng;
}

int orc_nerf_prepare(orc_nerf_model* m) {
	if (m->n_hidden_density > 8 || m->n_hidden_rgb > 8) return -2;
	if (m->n_neurons > 256) return -4;
	prepared_t* p = (prepared_t*)calloc(1, sizeof(prepared_t));
	if (!p) return -1;
	if (m->pos_encoding == 0 && orc_grid_layout(m, p->offsets, p->resolutions, p->scales)) { free(p); return -1; }
	network_shapes(m, &p->enc_dims, &p->dir_dims, &p->rgb_in, &p->rgb_out);
	p->n_density_w = mlp_n_params(p->enc_dims, m->n_neurons, m->n_hidden_density, m->density_out_dims);
	p->n_rgb_w = mlp_n_params(p->rgb_in, m->n_neurons, m->n_hidden_rgb, p->rgb_out);
	uint64_t need = p->n_density_w + p->n_rgb_w + (m->pos_encoding != 0 ? 0 : (uint64_t)p->offsets[m->n_levels] * m->n_features_per_level);
	if (m->n_params != need) { free(p); return -3; }
	p->density_w = (float*)malloc(sizeof(float) * p->n_density_w);
	p->rgb_w = (float*)malloc(sizeof(float) * p->n_rgb_w);
	for (uint64_t i = 0; i < p->n_density_w; ++i) p->density_w[i] = orc_half_to_float(m->params[i]);
	for (uint64_t i = 0; i < p->n_rgb_w; ++i) p->rgb_w[i] = orc_half_to_float(m->params[p->n_density_w + i]);
	p->grid = m->params + p->n_density_w + p->n_rgb_w;
	p->aabb.min = v3_make(m->aabb_min[0], m->aabb_min[1], m->aabb_min[2]);
	p->aabb.max = v3_make(m->aabb_max[0], m->aabb_max[1], m->aabb_max[2]);
	p->render_aabb.min = v3_make(m->render_aabb_min[0], m->render_aabb_min[1], m->render_aabb_min[2]);
	p->render_aabb.max = v3_make(m->render_aabb_max[0], m->render_aabb_max[1], m->render_aabb_max[2]);
	m->prepared = p;
	return 0;
}

void orc_nerf_release(orc_nerf_model* m) {
	prepared_t* p = (prepared_t*)m->prepared;
	if (!p) return;
	free(p->density_w);
	free(p->rgb_w);
	free(p);
	m->prepared = NULL;
}

/* ------------------------------------------------------------------ K5a hash grid
 * tcnn kernel_grid / pos_fract / grid_index / coherent prime hash (Appendix B.1):
 *   pos = fma(scale, x, 0.5); cell = floor(pos); w = pos - cell
 *   index = dense (x + y*R + z*R^2, strides stop once they exceed the level size) or
 *           (x*1 ^ y*2654435761 ^ z*805459861), then % level size
 *   fp16 accumulation over the corners in the order 0..7 (bit d of the corner index selects the +1 neighbour along
 *   dimension d), by one of the two published sequences (orc_nerf_model::grid_accumulate, oracle.h):
 *     result = fma((half)weight, value, result)        or        result[f] += (half)(weight * (float)value[f]) */
static inline uint32_t grid_index(uint32_t hashmap_size, uint32_t res, const uint32_t* pg) {
	uint32_t stride = 1, index = 0;
	for (uint32_t dim = 0; dim < 3 && stride <= hashmap_size; ++dim) {
		index += pg[dim] * stride;
		stride *= res;
	}
	if (hashmap_size < stride) {
		index = (pg[0] * 1u) ^ (pg[1] * 2654435761u) ^ (pg[2] * 805459861u);
	}
	return index % hashmap_size;
}

static void grid_encode_one(const orc_nerf_model* m, const prepared_t* p, const float* x, uint16_t* out) {
	const uint32_t F = m->n_features_per_level;
	for (uint32_t l = 0; l < m->n_levels; ++l) {
		const uint32_t size = p->offsets[l + 1] - p->offsets[l];
		const uint16_t* level = p->grid + (uint64_t)p->offsets[l] * F;
		const float scale = p->scales[l];
		const uint32_t res = p->resolutions[l];
		float pos[3];
		uint32_t pg[3];
		for (int d = 0; d < 3; ++d) {
			float v = fmaf(scale, x[d], 0.5f);
			float fl = floorf(v);
			pg[d] = (uint32_t)(int)fl;
			pos[d] = v - fl;
		}
		uint16_t result[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		for (uint32_t idx = 0; idx < 8; ++idx) {
			float weight = 1.0f;
			uint32_t pgl[3];
			for (uint32_t d = 0; d < 3; ++d) {
				if ((idx & (1u << d)) == 0) {
					weight *= 1.0f - pos[d];
					pgl[d] = pg[d];
				} else {
					weight *= pos[d];
					pgl[d] = pg[d] + 1u;
				}
			}
			const uint16_t* val = level + (uint64_t)grid_index(size, res, pgl) * F;
			if (m->grid_accumulate == ORC_GRID_ACC_FMA) { /* result = fma((T)weight, val, result): one fp16 rounding per term */
				const uint16_t wh = orc_float_to_half(weight);
				for (uint32_t f = 0; f < F; ++f) result[f] = orc_half_fma(wh, val[f], result[f]);
			} else { /* result[f] += (T)(weight * (float)val[f]) */
				for (uint32_t f = 0; f < F; ++f) {
					float prod = weight * orc_half_to_float(val[f]);
					result[f] = orc_half_add(result[f], orc_float_to_half(prod));
				}
			}
		}
		for (uint32_t f = 0; f < F; ++f) out[l * F + f] = result[f];
	}
}

void orc_grid_encode(const orc_nerf_model* m, uint32_t n, const float* pos01, uint16_t* out) {
	const prepared_t* p = (const prepared_t*)m->prepared;
	for (uint32_t i = 0; i < n; ++i) grid_encode_one(m, p, pos01 + 3 * (size_t)i, out + (size_t)p->enc_dims * i);
}

/* ------------------------------------------------------------------ K5c spherical harmonics, degree 4
 * tcnn SphericalHarmonicsEncoding (Appendix B.3): input mapped 2x-1, 16 real SH values, cast to half. */
static void sh4_one(const float* d01, uint16_t* out) {
	float x = d01[0] * 2.0f - 1.0f, y = d01[1] * 2.0f - 1.0f, z = d01[2] * 2.0f - 1.0f;
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
	for (int i = 0; i < 16; ++i) out[i] = orc_float_to_half(o[i]);
}

void orc_sh4_encode(uint32_t n, const float* dir01, uint16_t* out) {
	for (uint32_t i = 0; i < n; ++i) sh4_one(dir01 + 3 * (size_t)i, out + 16 * (size_t)i);
}

/* ------------------------------------------------------------------ K5b/K5d fully fused MLP
 * tcnn FullyFusedMLP (Appendix B.2): bias-free, row-major (out x in) fp16 weights, ReLU on hidden
 * layers, activations stored as fp16 between layers, fp16 output. The reference accumulates inside
 * tensor-core fragments; here each dot product is accumulated exactly (double), rounded to fp32
 * (the MFMA accumulator type of the MI355X build) and then to fp16. */
static void mlp_layer(const float* w, uint32_t n_out, uint32_t n_in, const float* in, int relu, float* out_f, uint16_t* out_h) {
	for (uint32_t o = 0; o < n_out; ++o) {
		const float* row = w + (size_t)o * n_in;
		double acc = 0.0;
		for (uint32_t i = 0; i < n_in; ++i) acc += (double)row[i] * (double)in[i];
		float a = (float)acc;
		if (relu && !(a > 0.0f)) a = 0.0f;
		uint16_t h = orc_float_to_half(a);
		if (out_h) out_h[o] = h;
		out_f[o] = orc_half_to_float(h);
	}
}
/* ORC_MLP_ACC_FP16_K16 -- the bracket on the reference's own arithmetic. tcnn's FullyFusedMLP (kernel_mlp_fused and its
 * threadblock_*layer* helpers; called at nerf_network.h:120,130) keeps every layer's result in
 * wmma::fragment<wmma::accumulator, 16, 16, 16, __half> and issues one wmma::mma_sync per 16-wide block of the inner dimension:
 * the tensor core forms the block's 16 products exactly, adds them and the incoming fp16 accumulator in a wider internal format
 * and hands back an fp16 accumulator. Restated as: after every 16-wide K block the running sum is rounded to fp16 once
 * (products and the in-block sum exact). The hardware's internal width and truncation are not published, so this is a model of
 * the reference's rounding points, not a bit-level claim; it exists to measure how far fp16 accumulation moves an image away
 * from the exact-sum result the MI355X build (fp32 MFMA accumulators) reproduces. ReLU acts on the fp16 fragment. */
static void mlp_layer_fp16_k16(const float* w, uint32_t n_out, uint32_t n_in, const float* in, int relu, float* out_f, uint16_t* out_h) {
	for (uint32_t o = 0; o < n_out; ++o) {
		const float* row = w + (size_t)o * n_in;
		uint16_t acc = 0;
		for (uint32_t k0 = 0; k0 < n_in; k0 += 16u) {
			double block = (double)orc_half_to_float(acc);
			const uint32_t k1 = k0 + 16u < n_in ? k0 + 16u : n_in;
			for (uint32_t i = k0; i < k1; ++i) block += (double)row[i] * (double)in[i];
			acc = orc_double_to_half(block);
		}
		float a = orc_half_to_float(acc);
		if (relu && !(a > 0.0f)) { a = 0.0f; acc = 0; }
		if (out_h) out_h[o] = acc;
		out_f[o] = a;
	}
}
/* ORC_MLP_ACC_IDEAL: no rounding anywhere between the stored parameters and the logits (float64 throughout) */
static void mlp_layer_ideal(const float* w, uint32_t n_out, uint32_t n_in, const double* in, int relu, double* out) {
	for (uint32_t o = 0; o < n_out; ++o) {
		const float* row = w + (size_t)o * n_in;
		double acc = 0.0;
		for (uint32_t i = 0; i < n_in; ++i) acc += (double)row[i] * in[i];
		out[o] = relu && !(acc > 0.0) ? 0.0 : acc;
	}
}
static void mlp_forward_ideal(const float* w, uint32_t n_in, uint32_t width, uint32_t n_hidden, uint32_t n_out, const double* in, double* out) {
	double a[256], b[256];
	if (n_hidden == 0) { /* configs/nerf/linear.json: the output layer alone */
		mlp_layer_ideal(w, n_out, n_in, in, 0, out);
		return;
	}
	mlp_layer_ideal(w, width, n_in, in, 1, a);
	w += (size_t)width * n_in;
	double* cur = a;
	double* nxt = b;
	for (uint32_t l = 1; l < n_hidden; ++l) {
		mlp_layer_ideal(w, width, width, cur, 1, nxt);
		w += (size_t)width * width;
		double* t = cur; cur = nxt; nxt = t;
	}
	mlp_layer_ideal(w, n_out, width, cur, 0, out);
}

/* returns pointer past the consumed weights */
static void mlp_forward(uint32_t mode, const float* w, uint32_t n_in, uint32_t width, uint32_t n_hidden, uint32_t n_out, const float* in, float* out_f, uint16_t* out_h) {
	void (*layer)(const float*, uint32_t, uint32_t, const float*, int, float*, uint16_t*) = mode == ORC_MLP_ACC_FP16_K16 ? mlp_layer_fp16_k16 : mlp_layer;
	float a[256], b[256];
	if (n_hidden == 0) { /* configs/nerf/linear.json: the output layer alone (output_activation None) */
		layer(w, n_out, n_in, in, 0, out_f, out_h);
		return;
	}
	layer(w, width, n_in, in, 1, a, NULL);
	w += (size_t)width * n_in;
	float* cur = a;
	float* nxt = b;
	for (uint32_t l = 1; l < n_hidden; ++l) {
		layer(w, width, width, cur, 1, nxt, NULL);
		w += (size_t)width * width;
		float* t = cur; cur = nxt; nxt = t;
	}
	layer(w, n_out, width, cur, 0, out_f, out_h);
}

/* tcnn FrequencyEncoding (encodings/frequency.h): out[j] = sin(scalbn(x[j / (2 n_freq)], (j / 2) % n_freq) * pi
 * + (j % 2) * pi / 2), cast to half; padded outputs are 1 (tcnn pads encodings with ones). The argument is one fma (what nvcc makes
 * of the expression); the sine is libm's -- tcnn calls the hardware approximation __sinf, whose error at these arguments (up to
 * 2^15 pi) is not specified closely enough to restate: PARITY UNPINNED like the rest of tcnn's arithmetic. */
/* tcnn IdentityEncoding (encodings/identity.h; configs/nerf/none.json): out[j] = in[j] * scale + offset (1 and 0: the defaults) cast to half,
 * padded outputs are 1. PARITY UNPINNED like the rest of tcnn's arithmetic. */
static void identity_encode_one(uint32_t n_dims, uint32_t padded, const float* x, uint16_t* out) {
	for (uint32_t j = 0; j < padded; ++j) out[j] = orc_float_to_half(j < n_dims ? x[j] * 1.0f + 0.0f : 1.0f);
}
static void frequency_encode_one(uint32_t n_dims, uint32_t n_freq, uint32_t padded, const float* x, uint16_t* out) {
	const float PI = 3.14159265358979323846f;
	const uint32_t n = n_dims * 2u * n_freq;
	for (uint32_t j = 0; j < n; ++j) {
		const uint32_t log2_frequency = (j / 2u) % n_freq, feature = j / (n_freq * 2u);
		const float phase_shift = (float)(j % 2u) * (PI / 2.0f);
		const float v = scalbnf(x[feature], (int)log2_frequency);
		out[j] = orc_float_to_half(sinf(fmaf(v, PI, phase_shift))); /* x * PI + phase_shift as nvcc contracts it (-fmad is its default) */
	}
	for (uint32_t j = n; j < padded; ++j) out[j] = orc_float_to_half(1.0f);
}
void orc_frequency_encode(uint32_t n, uint32_t n_dims, uint32_t n_frequencies, const float* x, uint16_t* out) {
	const uint32_t w = n_dims * 2u * n_frequencies;
	for (uint32_t i = 0; i < n; ++i) frequency_encode_one(n_dims, n_frequencies, w, x + (size_t)n_dims * i, out + (size_t)w * i);
}

/* the float64 network (ORC_MLP_ACC_IDEAL): stored fp16 parameters, everything else exact to double rounding -- trilinear
 * interpolation, spherical harmonics / sines, both MLPs; logits out4 = rgb, density */
static void nerf_network_one_ideal(const orc_nerf_model* m, const prepared_t* p, const float* pos01, const float* dir01, double* out4) {
	double enc[ORC_MAX_LEVELS * 8 > 256 ? ORC_MAX_LEVELS * 8 : 256];
	const double PI = 3.14159265358979323846;
	if (m->pos_encoding == 1) {
		const uint32_t nf = m->pos_n_frequencies, n = 3u * 2u * nf;
		for (uint32_t j = 0; j < n; ++j) enc[j] = sin(ldexp((double)pos01[j / (nf * 2u)], (int)((j / 2u) % nf)) * PI + (double)(j % 2u) * (PI / 2.0));
		for (uint32_t j = n; j < p->enc_dims; ++j) enc[j] = 1.0;
	} else if (m->pos_encoding == 2) {
		for (uint32_t j = 0; j < p->enc_dims; ++j) enc[j] = j < 3u ? (double)pos01[j] : 1.0;
	} else {
		const uint32_t F = m->n_features_per_level;
		for (uint32_t l = 0; l < m->n_levels; ++l) {
			const uint32_t size = p->offsets[l + 1] - p->offsets[l];
			const uint16_t* level = p->grid + (uint64_t)p->offsets[l] * F;
			double pos[3];
			uint32_t pg[3];
			for (int d = 0; d < 3; ++d) {
				const double v = (double)p->scales[l] * (double)pos01[d] + 0.5, fl = floor(v);
				pg[d] = (uint32_t)(int)fl;
				pos[d] = v - fl;
			}
			double result[8] = {0, 0, 0, 0, 0, 0, 0, 0};
			for (uint32_t idx = 0; idx < 8; ++idx) {
				double weight = 1.0;
				uint32_t pgl[3];
				for (uint32_t d = 0; d < 3; ++d) {
					if ((idx & (1u << d)) == 0) { weight *= 1.0 - pos[d]; pgl[d] = pg[d]; }
					else { weight *= pos[d]; pgl[d] = pg[d] + 1u; }
				}
				const uint16_t* val = level + (uint64_t)grid_index(size, p->resolutions[l], pgl) * F;
				for (uint32_t f = 0; f < F; ++f) result[f] += weight * (double)orc_half_to_float(val[f]);
			}
			for (uint32_t f = 0; f < F; ++f) enc[l * F + f] = result[f];
		}
	}
	double rgb_in[128], dens[32];
	mlp_forward_ideal(p->density_w, p->enc_dims, m->n_neurons, m->n_hidden_density, m->density_out_dims, enc, dens);
	for (uint32_t i = 0; i < m->density_out_dims; ++i) rgb_in[i] = dens[i];
	double* dir = rgb_in + m->density_out_dims;
	if (m->dir_encoding == 1) {
		const uint32_t nf = m->dir_n_frequencies, n = 3u * 2u * nf;
		for (uint32_t j = 0; j < n; ++j) dir[j] = sin(ldexp((double)dir01[j / (nf * 2u)], (int)((j / 2u) % nf)) * PI + (double)(j % 2u) * (PI / 2.0));
		for (uint32_t j = n; j < p->dir_dims; ++j) dir[j] = 1.0;
	} else if (m->dir_encoding == 2) {
		for (uint32_t j = 0; j < p->dir_dims; ++j) dir[j] = j < 3u ? (double)dir01[j] : 1.0;
	} else {
		const double x = (double)dir01[0] * 2.0 - 1.0, y = (double)dir01[1] * 2.0 - 1.0, z = (double)dir01[2] * 2.0 - 1.0;
		const double xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
		dir[0] = 0.28209479177387814;
		dir[1] = -0.48860251190291987 * y;
		dir[2] = 0.48860251190291987 * z;
		dir[3] = -0.48860251190291987 * x;
		dir[4] = 1.0925484305920792 * xy;
		dir[5] = -1.0925484305920792 * yz;
		dir[6] = 0.94617469575755997 * z2 - 0.31539156525251999;
		dir[7] = -1.0925484305920792 * xz;
		dir[8] = 0.54627421529603959 * x2 - 0.54627421529603959 * y2;
		dir[9] = 0.59004358992664352 * y * (-3.0 * x2 + y2);
		dir[10] = 2.8906114426405538 * xy * z;
		dir[11] = 0.45704579946446572 * y * (1.0 - 5.0 * z2);
		dir[12] = 0.3731763325901154 * z * (5.0 * z2 - 3.0);
		dir[13] = 0.45704579946446572 * x * (1.0 - 5.0 * z2);
		dir[14] = 1.4453057213202769 * z * (x2 - y2);
		dir[15] = 0.59004358992664352 * x * (-x2 + 3.0 * y2);
	}
	for (uint32_t i = m->density_out_dims + p->dir_dims; i < p->rgb_in; ++i) rgb_in[i] = 1.0;
	double rgb_out[16];
	mlp_forward_ideal(p->rgb_w, p->rgb_in, m->n_neurons, m->n_hidden_rgb, p->rgb_out, rgb_in, rgb_out);
	out4[0] = rgb_out[0];
	out4[1] = rgb_out[1];
	out4[2] = rgb_out[2];
	out4[3] = dens[0];
}

/* nerf_network.h:105-139: pos enc -> density MLP -> [density out | dir enc] -> rgb MLP; row 3 <- density logit. */
static void nerf_network_one(const orc_nerf_model* m, const prepared_t* p, const float* pos01, const float* dir01, uint16_t* out4) {
	if (m->mlp_accumulate == ORC_MLP_ACC_IDEAL) { /* (the fp16 interface rounds the float64 logits once) */
		double o[4];
		nerf_network_one_ideal(m, p, pos01, dir01, o);
		for (int k = 0; k < 4; ++k) out4[k] = orc_double_to_half(o[k]);
		return;
	}
	uint16_t enc_h[ORC_MAX_LEVELS * 8 > 256 ? ORC_MAX_LEVELS * 8 : 256];
	float enc[ORC_MAX_LEVELS * 8 > 256 ? ORC_MAX_LEVELS * 8 : 256];
	if (m->pos_encoding == 1) frequency_encode_one(3, m->pos_n_frequencies, p->enc_dims, pos01, enc_h);
	else if (m->pos_encoding == 2) identity_encode_one(3, p->enc_dims, pos01, enc_h);
	else grid_encode_one(m, p, pos01, enc_h);
	for (uint32_t i = 0; i < p->enc_dims; ++i) enc[i] = orc_half_to_float(enc_h[i]);
	float rgb_in[128];
	uint16_t dens_h[32];
	mlp_forward(m->mlp_accumulate, p->density_w, p->enc_dims, m->n_neurons, m->n_hidden_density, m->density_out_dims, enc, rgb_in, dens_h);
	uint16_t dir_h[64];
	if (m->dir_encoding == 1) frequency_encode_one(3, m->dir_n_frequencies, p->dir_dims, dir01, dir_h);
	else if (m->dir_encoding == 2) identity_encode_one(3, p->dir_dims, dir01, dir_h);
	else sh4_one(dir01, dir_h);
	for (uint32_t i = 0; i < p->dir_dims; ++i) rgb_in[m->density_out_dims + i] = orc_half_to_float(dir_h[i]);
	for (uint32_t i = m->density_out_dims + p->dir_dims; i < p->rgb_in; ++i) rgb_in[i] = 1.0f; /* alignment padding of the rgb network's input */
	float rgb_out[16];
	uint16_t rgb_h[16];
	mlp_forward(m->mlp_accumulate, p->rgb_w, p->rgb_in, m->n_neurons, m->n_hidden_rgb, p->rgb_out, rgb_in, rgb_out, rgb_h);
	out4[0] = rgb_h[0];
	out4[1] = rgb_h[1];
	out4[2] = rgb_h[2];
	out4[3] = dens_h[0];
}
/* the logits as the compositor reads them: the fp16 network outputs, or -- ORC_MLP_ACC_IDEAL -- the float64 logits rounded to fp32 */
static void nerf_network_one_f(const orc_nerf_model* m, const prepared_t* p, const float* pos01, const float* dir01, float* out4) {
	if (m->mlp_accumulate == ORC_MLP_ACC_IDEAL) {
		double o[4];
		nerf_network_one_ideal(m, p, pos01, dir01, o);
		for (int k = 0; k < 4; ++k) out4[k] = (float)o[k];
		return;
	}
	uint16_t h[4];
	nerf_network_one(m, p, pos01, dir01, h);
	for (int k = 0; k < 4; ++k) out4[k] = orc_half_to_float(h[k]);
}

/* ------------------------------------------------------------------ ERenderMode::Normals: d density logit / d position
 * The reference calls tcnn's DifferentiableObject::input_gradient(stream, 3, input, input) (src/testbed_nerf.cu:2106-2107): a one-hot
 * "loss gradient" of backprop_scale = 128 at output 3 -- the density logit, which NerfNetwork::backward_impl routes to output 0 of the
 * density MLP (nerf_network.h) --, the MLP's backward pass (fp16 gradients between layers, ReLU masks from the forward activations),
 * GridEncoding::backward's input path (kernel_grid_backward_input: dL_dx[d] = sum_k (float)dL_dy[k] * dy_dx[d][k] in fp32, with the
 * dy_dx that kernel_grid forms for linear interpolation: scale * sum over the 4 corner pairs of w_other * (val_right - val_left)),
 * and a division by the scale. tcnn is not in the mount (PARITY UNPINNED): its fused backward's rounding points are restated as
 * "exact sum, rounded to fp32, then to fp16" like the forward pass.
 * Frequency encodings (configs/nerf/frequency.json): tcnn's forward pass stores dy_dx[j] = scalbnf(1, log2_frequency) * PI * __cosf(input) in
 * fp32 beside each feature, and frequency_encoding_backward sums dL_dx[d] = sum_k (float)dL_dy[d 2F + k] * dy_dx[d 2F + k] in fp32, k ascending
 * (encodings/frequency.h). Identity encodings: dL_dx[d] = dL_dy[d] * scale (1). */
static void density_gradient_one(const orc_nerf_model* m, const prepared_t* p, const float* x, float* grad3, uint16_t* logit_out) {
	const uint32_t F = m->n_features_per_level, W = m->n_neurons, E = p->enc_dims, NH = m->n_hidden_density;
	uint16_t enc_h[ORC_MAX_LEVELS * 8 > 256 ? ORC_MAX_LEVELS * 8 : 256];
	if (m->pos_encoding == 1) frequency_encode_one(3, m->pos_n_frequencies, p->enc_dims, x, enc_h);
	else if (m->pos_encoding == 2) identity_encode_one(3, p->enc_dims, x, enc_h);
	else grid_encode_one(m, p, x, enc_h);
	float act[9][256] = {{0.0f}}; /* act[0] = encoding, act[k] = hidden layer k (as floats of the fp16 values) */
	for (uint32_t i = 0; i < E; ++i) act[0][i] = orc_half_to_float(enc_h[i]);
	const float* w = p->density_w;
	const float* layer_w[9];
	uint32_t n_in = E;
	for (uint32_t l = 0; l < NH; ++l) {
		layer_w[l] = w;
		mlp_layer(w, W, n_in, act[l], 1, act[l + 1], NULL);
		w += (size_t)W * n_in;
		n_in = W;
	}
	uint16_t out_h[32];
	float out_f[32];
	mlp_layer(w, m->density_out_dims, n_in, act[NH], 0, out_f, out_h);
	if (logit_out) *logit_out = out_h[0];
	/* backward: d(128 * logit) / d hidden NH = 128 * W_out[0][:], masked by the forward ReLU (no mask on the encoding itself when
	 * there is no hidden layer) */
	float g[256], g_prev[256];
	for (uint32_t i = 0; i < n_in; ++i) g[i] = NH == 0 || act[NH][i] > 0.0f ? orc_half_to_float(orc_float_to_half(128.0f * w[i])) : 0.0f;
	for (uint32_t l = NH; l-- > 0;) {
		const uint32_t n_prev = l == 0 ? E : W;
		for (uint32_t i = 0; i < n_prev; ++i) {
			double acc = 0.0;
			for (uint32_t o = 0; o < W; ++o) acc += (double)layer_w[l][(size_t)o * n_prev + i] * (double)g[o];
			float v = (float)acc;
			if (l > 0 && !(act[l][i] > 0.0f)) v = 0.0f;
			g_prev[i] = orc_half_to_float(orc_float_to_half(v));
		}
		memcpy(g, g_prev, sizeof(float) * n_prev);
	}
	/* g = dL_dy over the encoding (fp16 values). dy_dx and the sum over features, level by level */
	float result[3] = {0.0f, 0.0f, 0.0f};
	if (m->pos_encoding == 1) {
		const float PI = 3.14159265358979323846f;
		const uint32_t nf = m->pos_n_frequencies;
		for (uint32_t d = 0; d < 3; ++d) {
			for (uint32_t k = 0; k < 2u * nf; ++k) {
				const uint32_t log2_frequency = k / 2u;
				const float input = fmaf(scalbnf(x[d], (int)log2_frequency), PI, (float)(k % 2u) * (PI / 2.0f));
				const float dy_dx = scalbnf(1.0f, (int)log2_frequency) * PI * cosf(input);
				result[d] += g[d * 2u * nf + k] * dy_dx;
			}
		}
		for (int d = 0; d < 3; ++d) grad3[d] = result[d] * (1.0f / 128.0f);
		return;
	}
	if (m->pos_encoding == 2) {
		for (int d = 0; d < 3; ++d) grad3[d] = g[d] * (1.0f / 128.0f);
		return;
	}
	for (uint32_t l = 0; l < m->n_levels; ++l) {
		const uint32_t size = p->offsets[l + 1] - p->offsets[l];
		const uint16_t* level = p->grid + (uint64_t)p->offsets[l] * F;
		const float scale = p->scales[l];
		const uint32_t res = p->resolutions[l];
		float pos[3];
		uint32_t pg[3];
		for (int d = 0; d < 3; ++d) {
			float v = fmaf(scale, x[d], 0.5f);
			float fl = floorf(v);
			pg[d] = (uint32_t)(int)fl;
			pos[d] = v - fl;
		}
		for (uint32_t gd = 0; gd < 3; ++gd) {
			float grads[8] = {0, 0, 0, 0, 0, 0, 0, 0};
			for (uint32_t idx = 0; idx < 4; ++idx) {
				float weight = scale;
				uint32_t pgl[3];
				for (uint32_t nd = 0; nd < 2; ++nd) {
					const uint32_t dim = nd >= gd ? nd + 1 : nd;
					if ((idx & (1u << nd)) == 0) {
						weight *= 1.0f - pos[dim];
						pgl[dim] = pg[dim];
					} else {
						weight *= pos[dim];
						pgl[dim] = pg[dim] + 1u;
					}
				}
				pgl[gd] = pg[gd];
				const uint16_t* left = level + (uint64_t)grid_index(size, res, pgl) * F;
				pgl[gd] = pg[gd] + 1u;
				const uint16_t* right = level + (uint64_t)grid_index(size, res, pgl) * F;
				for (uint32_t f = 0; f < F; ++f) grads[f] += weight * (orc_half_to_float(right[f]) - orc_half_to_float(left[f]));
			}
			for (uint32_t f = 0; f < F; ++f) result[gd] += g[l * F + f] * grads[f];
		}
	}
	for (int d = 0; d < 3; ++d) grad3[d] = result[d] * (1.0f / 128.0f);
}

void orc_density_gradient(const orc_nerf_model* m, uint32_t n, const float* pos01, float* grad) {
	const prepared_t* p = (const prepared_t*)m->prepared;
#pragma omp parallel for schedule(dynamic, 64)
	for (int64_t i = 0; i < (int64_t)n; ++i) density_gradient_one(m, p, pos01 + 3 * i, grad + 3 * i, NULL);
}

void orc_nerf_network(const orc_nerf_model* m, uint32_t n, const float* pos01, const float* dir01, uint16_t* out) {
	const prepared_t* p = (const prepared_t*)m->prepared;
#pragma omp parallel for schedule(dynamic, 64)
	for (int64_t i = 0; i < (int64_t)n; ++i) nerf_network_one(m, p, pos01 + 3 * i, dir01 + 3 * i, out + 4 * i);
}

/* ------------------------------------------------------------------ K8/K9 occupancy bitfield
 * testbed_nerf.cu:284-331, 2863-2877 */
void orc_density_grid_to_bitfield(const float* grid, uint32_t max_cascade, uint8_t* bitfield, float* out_mean) {
	const uint32_t n_elements = NERF_GRID_N_CELLS;
	/* reduce_sum of fmaxf(val,0)/n_elements over level 0. The GPU reduction order is unspecified;
	 * the sum is taken in double so that any order rounds to the same float. */
	double sum = 0.0;
	for (uint32_t i = 0; i < n_elements; ++i) sum += (double)(fmaxf(grid[i], 0.0f) / (float)n_elements);
	float mean = (float)sum;
	if (out_mean) *out_mean = mean;
	float thresh = NERF_MIN_OPTICAL_THICKNESS < mean ? NERF_MIN_OPTICAL_THICKNESS : mean;
	const uint32_t n_bytes_total = n_elements / 8 * NERF_CASCADES;
	const uint32_t n_nonzero = n_elements / 8 * (max_cascade + 1);
	for (uint32_t i = 0; i < n_bytes_total; ++i) {
		if (i >= n_nonzero) { bitfield[i] = 0; continue; }
		uint8_t bits = 0;
		for (uint32_t j = 0; j < 8; ++j) bits |= grid[(size_t)i * 8 + j] > thresh ? (uint8_t)(1u << j) : 0;
		bitfield[i] = bits;
	}
	for (uint32_t level = 1; level < NERF_CASCADES; ++level) {
		const uint8_t* prev = bitfield + (size_t)(level - 1) * (n_elements / 8);
		uint8_t* next = bitfield + (size_t)level * (n_elements / 8);
		for (uint32_t i = 0; i < n_elements / 64; ++i) {
			uint8_t bits = 0;
			for (uint32_t j = 0; j < 8; ++j) bits |= prev[(size_t)i * 8 + j] > 0 ? (uint8_t)(1u << j) : 0;
			uint32_t x = morton3D_invert(i >> 0) + NERF_GRIDSIZE / 8;
			uint32_t y = morton3D_invert(i >> 1) + NERF_GRIDSIZE / 8;
			uint32_t z = morton3D_invert(i >> 2) + NERF_GRIDSIZE / 8;
			next[morton3D(x, y, z)] |= bits;
		}
	}
}

/* ------------------------------------------------------------------ stepping (nerf_device.cuh:378-428) */
static float to_stepping_space(float t, float cone_angle) {
	if (cone_angle <= 1e-5f) return t / MIN_CONE_STEPSIZE();
	float log1p_c = logf(1.0f + cone_angle);
	float a = (logf(MIN_CONE_STEPSIZE()) - logf(log1p_c)) / log1p_c;
	float b = (logf(MAX_CONE_STEPSIZE()) - logf(log1p_c)) / log1p_c;
	float at = expf(a * log1p_c);
	float bt = expf(b * log1p_c);
	if (t <= at) return (t - at) / MIN_CONE_STEPSIZE() + a;
	else if (t <= bt) return logf(t) / log1p_c;
	else return (t - bt) / MAX_CONE_STEPSIZE() + b;
}
static float from_stepping_space(float n, float cone_angle) {
	if (cone_angle <= 1e-5f) return n * MIN_CONE_STEPSIZE();
	float log1p_c = logf(1.0f + cone_angle);
	float a = (logf(MIN_CONE_STEPSIZE()) - logf(log1p_c)) / log1p_c;
	float b = (logf(MAX_CONE_STEPSIZE()) - logf(log1p_c)) / log1p_c;
	float at = expf(a * log1p_c);
	float bt = expf(b * log1p_c);
	if (n <= a) return (n - a) * MIN_CONE_STEPSIZE() + at;
	else if (n <= b) return expf(n * log1p_c);
	else return (n - b) * MAX_CONE_STEPSIZE() + bt;
}
static float advance_n_steps(float t, float cone_angle, float n) { return from_stepping_space(to_stepping_space(t, cone_angle) + n, cone_angle); }
static float calc_dt(float t, float cone_angle) { return advance_n_steps(t, cone_angle, 1.0f) - t; }

/* nerf_device.cuh:306-314 */
static float warp_dt(float dt) {
	float max_stepsize = MIN_CONE_STEPSIZE() * (float)(1u << (NERF_CASCADES - 1));
	return (dt - MIN_CONE_STEPSIZE()) / (max_stepsize - MIN_CONE_STEPSIZE());
}
static float unwarp_dt(float dt) {
	float max_stepsize = MIN_CONE_STEPSIZE() * (float)(1u << (NERF_CASCADES - 1));
	return dt * (max_stepsize - MIN_CONE_STEPSIZE()) + MIN_CONE_STEPSIZE();
}

/* ------------------------------------------------------------------ occupancy lookups (nerf_device.cuh:316-367,430-447) */
static uint32_t cascaded_grid_idx_at(v3 pos, uint32_t mip) {
	float mip_scale = scalbnf(1.0f, -(int)mip);
	pos = v3_adds(pos, -0.5f);
	pos = v3_scale(pos, mip_scale);
	pos = v3_adds(pos, 0.5f);
	int ix = (int)(pos.x * (float)NERF_GRIDSIZE);
	int iy = (int)(pos.y * (float)NERF_GRIDSIZE);
	int iz = (int)(pos.z * (float)NERF_GRIDSIZE);
	if (ix < 0 || ix >= (int)NERF_GRIDSIZE || iy < 0 || iy >= (int)NERF_GRIDSIZE || iz < 0 || iz >= (int)NERF_GRIDSIZE) return 0xFFFFFFFFu;
	return morton3D((uint32_t)ix, (uint32_t)iy, (uint32_t)iz);
}
static int density_grid_occupied_at(v3 pos, const uint8_t* bitfield, uint32_t mip) {
	uint32_t idx = cascaded_grid_idx_at(pos, mip);
	if (idx == 0xFFFFFFFFu) return 0;
	return (bitfield[idx / 8 + (NERF_GRID_N_CELLS * mip) / 8] & (1u << (idx % 8))) != 0;
}
static float signf_(float x) { return copysignf(1.0f, x); }
static float distance_to_next_voxel(v3 pos, v3 dir, v3 idir, float res) {
	v3 p = v3_scale(v3_adds(pos, -0.5f), res);
	float tx = (floorf(p.x + 0.5f + 0.5f * signf_(dir.x)) - p.x) * idir.x;
	float ty = (floorf(p.y + 0.5f + 0.5f * signf_(dir.y)) - p.y) * idir.y;
	float tz = (floorf(p.z + 0.5f + 0.5f * signf_(dir.z)) - p.z) * idir.z;
	float t = fminf(fminf(tx, ty), tz);
	return fmaxf(t / res, 0.0f);
}
static float advance_to_next_voxel(float t, float cone_angle, v3 pos, v3 dir, v3 idir, uint32_t mip) {
	float res = scalbnf((float)NERF_GRIDSIZE, -(int)mip);
	float t_target = t + distance_to_next_voxel(pos, dir, idir, res);
	t = to_stepping_space(t, cone_angle);
	t_target = to_stepping_space(t_target, cone_angle);
	return from_stepping_space(t + ceilf(fmaxf(t_target - t, 0.5f)), cone_angle);
}
static uint32_t mip_from_pos(v3 pos, uint32_t max_cascade) {
	int exponent;
	float maxval = fmaxf(fmaxf(fabsf(pos.x - 0.5f), fabsf(pos.y - 0.5f)), fabsf(pos.z - 0.5f));
	(void)frexpf(maxval, &exponent);
	int v = exponent + 1;
	if (v < 0) v = 0;
	if (v > (int)max_cascade) v = (int)max_cascade;
	return (uint32_t)v;
}
static uint32_t clamp_u32(uint32_t v, uint32_t lo, uint32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* nerf_device.cuh:461-494 (and the 200-iteration variant :497-534 when capped != 0) */
static float if_unoccupied_advance_to_next_occupied_voxel(float t, float cone_angle, v3 o, v3 d, v3 idir, const uint8_t* grid,
                                                         uint32_t min_mip, uint32_t max_mip, const aabb_t* aabb, const float* to_local, int capped) {
	uint32_t i = 1;
	while (!capped || i < 200) {
		v3 pos = v3_add(o, v3_scale(d, t));
		if (t >= MAX_DEPTH || !aabb_contains(aabb, m3_mulv(to_local, pos))) return MAX_DEPTH;
		uint32_t mip = clamp_u32(mip_from_pos(pos, NERF_CASCADES - 1), min_mip, max_mip);
		if (!grid || density_grid_occupied_at(pos, grid, mip)) return t;
		while (mip < max_mip && !density_grid_occupied_at(pos, grid, mip + 1)) ++mip;
		t = advance_to_next_voxel(t, cone_angle, pos, d, idir, mip);
		++i;
	}
	return MAX_DEPTH;
}

/* ------------------------------------------------------------------ K1 ray generation
 * testbed_nerf.cu:1428-1544 with the perspective branch of uv_to_ray (common_device.cuh:416-483).
 * Restated for: no foveation, no hidden-area mask, Perspective lens, no distortion map, zero parallax
 * shift, zero aperture, plane_z >= 0, render mode Shade, no env map, static camera (camera0 == camera1,
 * so the per-pixel camera_slerp of get_xform_given_rolling_shutter is the identity up to rounding). */
/* lenses of uv_to_ray, common_device.cuh:249-338 (OpenCV / fisheye distortion, Newton undistortion), :375-391 (lat-long,
 * equirectangular), :441-462 */
static void opencv_delta(const float* q, float u, float v, float* du, float* dv) {
	const float k1 = q[0], k2 = q[1], p1 = q[2], p2 = q[3];
	const float u2 = u * u, uv = u * v, v2 = v * v, r2 = u2 + v2;
	const float radial = k1 * r2 + k2 * r2 * r2;
	*du = u * radial + 2.0f * p1 * uv + p2 * (r2 + 2.0f * u2);
	*dv = v * radial + 2.0f * p2 * uv + p1 * (r2 + 2.0f * v2);
}
static void fisheye_delta(const float* q, float u, float v, float* du, float* dv) {
	const float k1 = q[0], k2 = q[1], k3 = q[2], k4 = q[3];
	const float r = sqrtf(u * u + v * v);
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
static v3 lens_direction(const orc_camera* cam, float u, float v);
static void lens_undistort(int fisheye, const float* q, float* u, float* v) {
	void (*delta)(const float*, float, float, float*, float*) = fisheye ? fisheye_delta : opencv_delta;
	const float x0 = *u, y0 = *v;
	float x = x0, y = y0;
	for (uint32_t i = 0; i < 100u; ++i) {
		const float step0 = fmaxf(1.1920929e-07f, fabsf(1e-6f * x));
		const float step1 = fmaxf(1.1920929e-07f, fabsf(1e-6f * y));
		float dx0, dx1, b0, b1, f0, f1, c0, c1, g0, g1;
		delta(q, x, y, &dx0, &dx1);
		delta(q, x - step0, y, &b0, &b1);
		delta(q, x + step0, y, &f0, &f1);
		delta(q, x, y - step1, &c0, &c1);
		delta(q, x, y + step1, &g0, &g1);
		const float j00 = 1.0f + (f0 - b0) / (2.0f * step0), j10 = (g0 - c0) / (2.0f * step1);
		const float j01 = (f1 - b1) / (2.0f * step0), j11 = 1.0f + (g1 - c1) / (2.0f * step1);
		const float rx = x + dx0 - x0, ry = y + dx1 - y0;
		const float det = j00 * j11 - j10 * j01;
		const float sx = (j11 * rx - j10 * ry) / det, sy = (-j01 * rx + j00 * ry) / det;
		x -= sx;
		y -= sy;
		if (sx * sx + sy * sy < 1e-10f) break;
	}
	*u = x;
	*v = y;
}
void orc_lens_direction(const orc_camera* cam, float u, float v, float* dir3) {
	v3 d = lens_direction(cam, u, v);
	dir3[0] = d.x; dir3[1] = d.y; dir3[2] = d.z;
}
static v3 lens_direction(const orc_camera* cam, float u, float v) {
	const float PI = 3.14159265358979323846f;
	if (cam->lens_mode == 3) {
		float theta = (v - 0.5f) * PI, phi = (u - 0.5f) * PI * 2.0f;
		return v3_make(sinf(phi) * cosf(theta), sinf(theta), cosf(phi) * cosf(theta));
	}
	if (cam->lens_mode == 5) {
		float ct = (v - 0.5f) * 2.0f;
		float st = sqrtf(fmaxf(1.0f - ct * ct, 0.0f));
		float phi = (u - 0.5f) * PI * 2.0f;
		return v3_make(sinf(phi) * st, ct, cosf(phi) * st);
	}
	if (cam->lens_mode == 2) { /* FTheta: f_theta_undistortion(uv - screen_center, params, error_direction = 0), common_device.cuh:361-375, 441-446.
	                            * params: polynomial r0..r4 of the angle in the pixel radius, then the resolution the intrinsics were given at */
		const float* q = cam->lens_params;
		float xpix = (u - cam->screen_center[0]) * q[5], ypix = (v - cam->screen_center[1]) * q[6];
		float norm = sqrtf(xpix * xpix + ypix * ypix);
		float alpha = q[0] + norm * (q[1] + norm * (q[2] + norm * (q[3] + norm * q[4])));
		float sin_alpha = sinf(alpha), cos_alpha = cosf(alpha);
		if (cos_alpha <= FLT_MIN || norm == 0.f) return v3_make(0.f, 0.f, 0.f); /* Ray::invalid() */
		sin_alpha *= 1.f / norm;
		return v3_make(sin_alpha * xpix, sin_alpha * ypix, cos_alpha);
	}
	v3 dir = v3_make((u - cam->screen_center[0]) * (float)cam->width / cam->focal_length[0],
	                 (v - cam->screen_center[1]) * (float)cam->height / cam->focal_length[1], 1.0f);
	if (cam->lens_mode == 1) lens_undistort(0, cam->lens_params, &dir.x, &dir.y);
	else if (cam->lens_mode == 4) lens_undistort(1, cam->lens_params, &dir.x, &dir.y);
	return dir;
}

/* depth of field, common_device.cuh:471-477: square2disk_shirley (random_val.cuh:112-128) of ld_random_val_2d(spp, px hash) */
void orc_apply_aperture(const orc_camera* cam, float u, float v, float* origin3, float* dir3) {
	if (cam->aperture_size == 0.0f || cam->focus_z < 0.0f) return;
	int px = (int)(u * (float)cam->width), py = (int)(v * (float)cam->height);
	float sq[2];
	ld_random_val_2d(cam->spp_index, (uint32_t)px * 19349663u + (uint32_t)py * 96925573u, sq);
	float a = sq[0] * 2.0f - 1.0f, b = sq[1] * 2.0f - 1.0f;
	const float PI = 3.14159265358979323846f;
	float r, phi;
	if (a * a > b * b) {
		r = a;
		phi = (PI / 4.0f) * (b / a);
	} else {
		r = b;
		phi = (PI / 2.0f) - (PI / 4.0f) * (a / b);
	}
	float bx = cam->aperture_size * (r * cosf(phi)), by = cam->aperture_size * (r * sinf(phi));
	v3 origin = v3_make(origin3[0], origin3[1], origin3[2]), dir = v3_make(dir3[0], dir3[1], dir3[2]);
	v3 lookat = v3_add(origin, v3_scale(dir, cam->focus_z));
	origin = v3_add(origin, v3_add(v3_scale(v3_make(cam->matrix[0], cam->matrix[1], cam->matrix[2]), bx), v3_scale(v3_make(cam->matrix[3], cam->matrix[4], cam->matrix[5]), by)));
	dir = v3_make((lookat.x - origin.x) / cam->focus_z, (lookat.y - origin.y) / cam->focus_z, (lookat.z - origin.z) / cam->focus_z);
	origin3[0] = origin.x; origin3[1] = origin.y; origin3[2] = origin.z;
	dir3[0] = dir.x; dir3[1] = dir.y; dir3[2] = dir.z;
}

static int camera_moves(const orc_camera* cam) { return cam->has_matrix1 && memcmp(cam->matrix, cam->matrix1, sizeof(cam->matrix)) != 0; }
static void quat_cast(const float* m, float* q /* w x y z */) { /* m[3 c + r] */
	const float m00 = m[0], m01 = m[1], m02 = m[2], m10 = m[3], m11 = m[4], m12 = m[5], m20 = m[6], m21 = m[7], m22 = m[8];
	const float fx = m00 - m11 - m22, fy = m11 - m00 - m22, fz = m22 - m00 - m11, fw = m00 + m11 + m22;
	int biggest = 0;
	float fb = fw;
	if (fx > fb) { fb = fx; biggest = 1; }
	if (fy > fb) { fb = fy; biggest = 2; }
	if (fz > fb) { fb = fz; biggest = 3; }
	const float bv = sqrtf(fb + 1.0f) * 0.5f, mult = 0.25f / bv;
	if (biggest == 0) { q[0] = bv; q[1] = (m12 - m21) * mult; q[2] = (m20 - m02) * mult; q[3] = (m01 - m10) * mult; }
	else if (biggest == 1) { q[0] = (m12 - m21) * mult; q[1] = bv; q[2] = (m01 + m10) * mult; q[3] = (m20 + m02) * mult; }
	else if (biggest == 2) { q[0] = (m20 - m02) * mult; q[1] = (m01 + m10) * mult; q[2] = bv; q[3] = (m12 + m21) * mult; }
	else { q[0] = (m01 - m10) * mult; q[1] = (m20 + m02) * mult; q[2] = (m12 + m21) * mult; q[3] = bv; }
}
void orc_camera_at_pixel(const orc_camera* cam, float u, float v, uint32_t idx, float* out12) {
	if (!camera_moves(cam)) { memcpy(out12, cam->matrix, sizeof(cam->matrix)); return; }
	const float* rs = cam->rolling_shutter;
	const float t = rs[0] + rs[1] * u + rs[2] * v + rs[3] * orc_ld_random_val(cam->spp_index, idx * 72239731u, 0);
	float qa[4], qb[4], q[4];
	quat_cast(cam->matrix, qa);
	quat_cast(cam->matrix1, qb);
	float cos_theta = ((qa[0] * qb[0] + qa[1] * qb[1]) + qa[2] * qb[2]) + qa[3] * qb[3];
	if (cos_theta < 0.0f) {
		for (int i = 0; i < 4; ++i) qb[i] = -qb[i];
		cos_theta = -cos_theta;
	}
	if (cos_theta > 1.0f - 1.1920929e-07f) {
		for (int i = 0; i < 4; ++i) q[i] = qa[i] + (qb[i] - qa[i]) * t;
	} else {
		const float angle = acosf(cos_theta);
		const float sa = sinf((1.0f - t) * angle), sb = sinf(t * angle), sn = sinf(angle);
		for (int i = 0; i < 4; ++i) q[i] = (sa * qa[i] + sb * qb[i]) / sn;
	}
	const float len = sqrtf(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
	const float w = q[0] / len, x = q[1] / len, y = q[2] / len, z = q[3] / len;
	const float xx = x * x, yy = y * y, zz = z * z, xz = x * z, xy = x * y, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
	out12[0] = 1.0f - 2.0f * (yy + zz); out12[1] = 2.0f * (xy + wz); out12[2] = 2.0f * (xz - wy);
	out12[3] = 2.0f * (xy - wz); out12[4] = 1.0f - 2.0f * (xx + zz); out12[5] = 2.0f * (yz + wx);
	out12[6] = 2.0f * (xz + wy); out12[7] = 2.0f * (yz - wx); out12[8] = 1.0f - 2.0f * (xx + yy);
	for (int i = 0; i < 3; ++i) out12[9 + i] = cam->matrix[9 + i] * (1.0f - t) + cam->matrix1[9 + i] * t;
}
static void init_ray_dir(const orc_nerf_model* m, const orc_camera* cam, uint32_t x, uint32_t y, orc_payload* payload, float* unit_dir3);
void orc_init_ray(const orc_nerf_model* m, const orc_camera* cam, uint32_t x, uint32_t y, orc_payload* payload) {
	float d3[3];
	init_ray_dir(m, cam, x, y, payload, d3);
}
/* unit_dir3: the normalised ray direction of the pixel ((0,0,0) for a pixel without a ray) -- the reference reads the
 * environment map along it before the render-box test (src/testbed_nerf.cu:1524-1528), the payload keeps it only for rays that enter the box */
static void init_ray_dir(const orc_nerf_model* m, const orc_camera* cam, uint32_t x, uint32_t y, orc_payload* payload, float* unit_dir3) {
	const prepared_t* p = (const prepared_t*)m->prepared;
	unit_dir3[0] = unit_dir3[1] = unit_dir3[2] = 0.0f;
	uint32_t idx = x + (uint32_t)cam->width * y;
	float off[2];
	orc_ld_random_pixel_offset(cam->snap_to_pixel_centers ? 0u : cam->spp_index, off);
	float u = ((float)x + off[0]) / (float)cam->width;
	float v = ((float)y + off[1]) / (float)cam->height;
	v3 dir = lens_direction(cam, u, v);
	orc_camera pc = *cam; /* the camera of this pixel (src/testbed_nerf.cu:1468) */
	orc_camera_at_pixel(cam, u, v, idx, pc.matrix);
	dir = m3_mulv(pc.matrix, dir);
	v3 origin = v3_make(pc.matrix[9], pc.matrix[10], pc.matrix[11]);
	{
		float o3[3] = {origin.x, origin.y, origin.z}, d3[3] = {dir.x, dir.y, dir.z};
		orc_apply_aperture(&pc, u, v, o3, d3);
		origin = v3_make(o3[0], o3[1], o3[2]);
		dir = v3_make(d3[0], d3[1], d3[2]);
	}
	origin = v3_add(origin, v3_scale(dir, cam->near_distance));

	memset(payload, 0, sizeof(*payload));
	payload->max_weight = 0.0f;
	if (dir.x == 0.0f && dir.y == 0.0f && dir.z == 0.0f) { /* !ray.is_valid() */
		payload->origin[0] = origin.x; payload->origin[1] = origin.y; payload->origin[2] = origin.z;
		payload->alive = 0;
		return;
	}
	dir = v3_normalize(dir);
	unit_dir3[0] = dir.x; unit_dir3[1] = dir.y; unit_dir3[2] = dir.z;
	float tmin, tmax;
	aabb_ray_intersect(&p->render_aabb, m3_mulv(m->render_aabb_to_local, origin), m3_mulv(m->render_aabb_to_local, dir), &tmin, &tmax);
	float t = fmaxf(tmin, 0.0f) + 1e-6f;
	payload->origin[0] = origin.x; payload->origin[1] = origin.y; payload->origin[2] = origin.z;
	if (!aabb_contains(&p->render_aabb, m3_mulv(m->render_aabb_to_local, v3_add(origin, v3_scale(dir, t))))) {
		payload->alive = 0;
		return;
	}
	payload->dir[0] = dir.x; payload->dir[1] = dir.y; payload->dir[2] = dir.z;
	payload->t = t;
	payload->idx = idx;
	payload->n_steps = 0;
	payload->alive = 1;
}

/* K2: testbed_nerf.cu:333-362 */
void orc_advance_pos(const orc_nerf_model* m, const orc_camera* cam, orc_payload* payload) {
	const prepared_t* p = (const prepared_t*)m->prepared;
	if (!payload->alive) return;
	v3 origin = v3_make(payload->origin[0], payload->origin[1], payload->origin[2]);
	v3 dir = v3_make(payload->dir[0], payload->dir[1], payload->dir[2]);
	v3 idir = v3_make(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
	float cone_angle = m->cone_angle_constant; /* calc_cone_angle returns the constant, nerf_device.cuh:369-376 */
	float t = advance_n_steps(payload->t, cone_angle, orc_ld_random_val(cam->spp_index, payload->idx * 786433u, 0));
	t = if_unoccupied_advance_to_next_occupied_voxel(t, cone_angle, origin, dir, idir, m->density_grid_bitfield, 0, m->max_cascade, &p->render_aabb, m->render_aabb_to_local, 0);
	if (t >= MAX_DEPTH) payload->alive = 0;
	else payload->t = t;
}

/* ------------------------------------------------------------------ activations (nerf_device.cuh:203-263) */
static float logistic_(float x) { return 1.0f / (1.0f + expf(-x)); }
static float clampf_(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
static float network_to_rgb(float val, uint32_t act) {
	switch (act) {
		case ORC_ACT_NONE: return val;
		case ORC_ACT_RELU: return val > 0.0f ? val : 0.0f;
		case ORC_ACT_LOGISTIC: return logistic_(val);
		case ORC_ACT_EXPONENTIAL: return expf(clampf_(val, -10.0f, 10.0f));
	}
	return 0.0f;
}
static float network_to_density(float val, uint32_t act) {
	switch (act) {
		case ORC_ACT_NONE: return val;
		case ORC_ACT_RELU: return val > 0.0f ? val : 0.0f;
		case ORC_ACT_LOGISTIC: return logistic_(val);
		case ORC_ACT_EXPONENTIAL: return expf(val);
	}
	return 0.0f;
}

/* ------------------------------------------------------------------ K3-K6 for one ray
 * generate_next_nerf_network_inputs (testbed_nerf.cu:430-477) + network + composite_kernel_nerf (:528-735,
 * render mode Shade, no glow, show_accel < 0) chained sample by sample. The reference batches 1..8 samples
 * between compactions (:2080-2081); a ray's result does not depend on that batching because compositing
 * re-checks termination after every sample and samples past the terminating one are discarded.
 * The loop bound restates MARCH_ITER (:46,2056): a ray that is still alive after that many steps is never
 * compacted into the hit buffer and therefore never shaded. Returns the number of samples evaluated. */
uint32_t orc_trace_ray(const orc_nerf_model* m, const float* cam_matrix, const orc_render_opts* o, orc_payload* payload, float* rgba, float* depth) {
	const prepared_t* p = (const prepared_t*)m->prepared;
	if (!payload->alive) return 0;
	v3 origin = v3_make(payload->origin[0], payload->origin[1], payload->origin[2]);
	v3 dir = v3_make(payload->dir[0], payload->dir[1], payload->dir[2]);
	v3 idir = v3_make(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
	v3 cam_fwd = v3_make(cam_matrix[6], cam_matrix[7], cam_matrix[8]);
	v3 cam_pos = v3_make(cam_matrix[9], cam_matrix[10], cam_matrix[11]);
	float cone_angle = m->cone_angle_constant;
	float t = payload->t;
	v3 diag = v3_sub(p->aabb.max, p->aabb.min);
	float lr = rgba[0], lg = rgba[1], lb = rgba[2], la = rgba[3];
	float local_depth = *depth;
	uint32_t n = 0;
	uint32_t step = 1;
	for (; step < ORC_MARCH_ITER; ++step) {
		t = if_unoccupied_advance_to_next_occupied_voxel(t, cone_angle, origin, dir, idir, m->density_grid_bitfield, 0, m->max_cascade,
		                                                &p->render_aabb, m->render_aabb_to_local, o->capped_skip);
		if (t >= MAX_DEPTH) { payload->alive = 0; break; }
		float dt = calc_dt(t, cone_angle);
		/* NerfCoordinate written by the generator ... */
		v3 warped = v3_div(v3_sub(v3_add(origin, v3_scale(dir, t)), p->aabb.min), diag); /* warp_position, nerf_device.cuh:265 */
		float wdir[3] = {(dir.x + 1.0f) * 0.5f, (dir.y + 1.0f) * 0.5f, (dir.z + 1.0f) * 0.5f}; /* warp_direction :290 */
		float wdt = warp_dt(dt);
		t += dt;
		/* ... network ... */
		float wpos[3] = {warped.x, warped.y, warped.z};
		float out4[4];
		nerf_network_one_f(m, p, wpos, wdir, out4);
		++n;
		/* ... and read back by the compositor */
		v3 pos = v3_add(p->aabb.min, v3_mul(warped, diag)); /* unwarp_position */
		float T = 1.0f - la;
		float dtu = unwarp_dt(wdt);
		float alpha = 1.0f - expf(-network_to_density(out4[3], m->density_activation) * dtu);
		float weight = alpha * T;
		float r = network_to_rgb(out4[0], m->rgb_activation);
		float g = network_to_rgb(out4[1], m->rgb_activation);
		float b = network_to_rgb(out4[2], m->rgb_activation);
		if (o->render_mode == 7) { /* ERenderMode::Normals, testbed_nerf.cu:688-693: opposite to the density gradient */
			float g3[3];
			density_gradient_one(m, p, wpos, g3, NULL);
			float dd;
			float sv = out4[3];
			switch (m->density_activation) { /* network_to_density_derivative, nerf_device.cuh:245-254 */
				case 1: dd = sv > 0.0f ? 1.0f : 0.0f; break;
				case 2: { float dn = 1.0f / (1.0f + expf(-sv)); dd = dn * (1.0f - dn); } break;
				case 3: dd = expf(fminf(fmaxf(sv, -15.0f), 15.0f)); break;
				default: dd = 1.0f;
			}
			v3 nrm = v3_make(-dd * g3[0], -dd * g3[1], -dd * g3[2]);
			float len = sqrtf(v3_dot(nrm, nrm));
			r = nrm.x / len; g = nrm.y / len; b = nrm.z / len;
		} else if (o->render_mode == 2) { /* ERenderMode::AO, testbed_nerf.cu:700-702 */
			r = g = b = alpha;
		} else if (o->render_mode == 3) { /* Positions :694-695 */
			r = (pos.x - 0.5f) / 2.0f + 0.5f; g = (pos.y - 0.5f) / 2.0f + 0.5f; b = (pos.z - 0.5f) / 2.0f + 0.5f;
		} else if (o->render_mode == 4) { /* Depth :698-699 */
			r = g = b = v3_dot(cam_fwd, v3_sub(pos, origin)) * o->depth_scale;
		}
		lr += r * weight; lg += g * weight; lb += b * weight; la += weight;
		if (weight > payload->max_weight) {
			payload->max_weight = weight;
			local_depth = v3_dot(cam_fwd, v3_sub(pos, cam_pos));
		}
		if (la > (1.0f - o->min_transmittance)) {
			lr /= la; lg /= la; lb /= la; la /= la;
			payload->alive = 0;
			break;
		}
	}
	payload->t = t;
	payload->n_steps = (uint16_t)(n > 65535u ? 65535u : n);
	rgba[0] = lr; rgba[1] = lg; rgba[2] = lb; rgba[3] = la;
	*depth = local_depth;
	return n;
}

/* K7: shade_kernel_nerf (testbed_nerf.cu:1361-1401), render mode Shade; with o->depth_test the
 * geometry variant (testbed_geometry_training.cu:1826-1871). */
static void shade_one(const orc_render_opts* o, const float* rgba, float depth, uint32_t idx, uint32_t n_steps, float* frame_buffer, float* depth_buffer) {
	if (o->depth_test && depth > depth_buffer[idx]) return;
	float tmp[4] = {rgba[0], rgba[1], rgba[2], rgba[3]};
	if (o->render_mode == 7) { /* ERenderMode::Normals, :1379-1381 */
		float len = sqrtf(tmp[0] * tmp[0] + tmp[1] * tmp[1] + tmp[2] * tmp[2]);
		for (int c = 0; c < 3; ++c) tmp[c] = (0.5f * (tmp[c] / len) + 0.5f) * tmp[3];
	}
	if (o->render_mode == 5) { /* ERenderMode::Cost, :1382-1384 (n_steps: the ray's total, see include/ngp_hip.h NGP_RENDER_COST) */
		tmp[0] = tmp[1] = tmp[2] = (float)n_steps / 128.0f;
		tmp[3] = 1.0f;
	}
	if (!o->train_in_linear_colors && o->render_mode == 0) { /* only ERenderMode::Shade converts (:1393): 1 = ShadeEnvMap / ShadeGridEnvMap keep the network's sRGB values */
		tmp[0] = orc_srgb_to_linear(tmp[0]);
		tmp[1] = orc_srgb_to_linear(tmp[1]);
		tmp[2] = orc_srgb_to_linear(tmp[2]);
	}
	float* fb = frame_buffer + 4 * (size_t)idx;
	float one_minus_a = 1.0f - tmp[3];
	for (int c = 0; c < 4; ++c) fb[c] = tmp[c] + fb[c] * one_minus_a;
	if (tmp[3] > 0.2f) depth_buffer[idx] = depth;
}

/* read_envmap, envmap.cuh:24-50, with dir_to_spherical_unorm (random_val.cuh:62-72): bilinear, x wraps, y clamps */
void orc_read_envmap(const float* envmap, int32_t res_x, int32_t res_y, const float* dir3, float* out4) {
	const float PI = 3.14159265358979323846f;
	const float dx = dir3[2], dy = -dir3[0], dz = dir3[1];
	const float cos_theta = fminf(fmaxf(dz, -1.0f), 1.0f);
	const float theta = acosf(cos_theta);
	const float phi = atan2f(dy, dx);
	const float cyl_x = theta / PI, cyl_y = phi / (2.0f * PI) + 0.5f;
	const float fx = cyl_y * (float)(res_x - 1), fy = cyl_x * (float)(res_y - 1);
	const int tx = (int)fx, ty = (int)fy;
	const float wx = fx - (float)tx, wy = fy - (float)ty;
	const float* v[4];
	for (int k = 0; k < 4; ++k) {
		int px = tx + (k & 1), py = ty + (k >> 1);
		if (px < 0) px += res_x;
		else if (px >= res_x) px -= res_x;
		py = py > res_y - 1 ? res_y - 1 : py;
		py = py < 0 ? 0 : py;
		v[k] = envmap + 4 * ((size_t)px + (size_t)py * res_x);
	}
	const float w00 = (1 - wx) * (1 - wy), w10 = wx * (1 - wy), w01 = (1 - wx) * wy, w11 = wx * wy;
	for (int c = 0; c < 4; ++c) out4[c] = ((w00 * v[0][c] + w10 * v[1][c]) + w01 * v[2][c]) + w11 * v[3][c];
}

void orc_render_nerf(const orc_nerf_model* m, const orc_camera* cam, const orc_render_opts* o, float* frame_buffer, float* depth_buffer, orc_render_stats* stats) {
	const int64_t n_pixels = (int64_t)cam->width * cam->height;
	uint64_t n_alive = 0, n_hit = 0, n_samples = 0;
#ifdef _OPENMP
	if (o->n_threads > 0) omp_set_num_threads(o->n_threads);
#endif
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : n_alive, n_hit, n_samples)
	for (int64_t i = 0; i < n_pixels; ++i) {
		uint32_t x = (uint32_t)(i % cam->width), y = (uint32_t)(i / cam->width);
		orc_payload payload;
		float d3[3];
		init_ray_dir(m, cam, x, y, &payload, d3);
		/* testbed_nerf.cu:1490-1493 */
		if (depth_buffer[i] < 0.01f) depth_buffer[i] = MAX_DEPTH;
		if (o->envmap && (d3[0] != 0.0f || d3[1] != 0.0f || d3[2] != 0.0f)) orc_read_envmap(o->envmap, o->env_w, o->env_h, d3, frame_buffer + 4 * i); /* :1526-1528 */
		orc_advance_pos(m, cam, &payload);
		if (!payload.alive) continue;
		++n_alive;
		float rgba[4] = {0, 0, 0, 0};
		float depth = 0.0f;
		n_samples += orc_trace_ray(m, camera_moves(cam) ? cam->matrix1 : cam->matrix, o, &payload, rgba, &depth); /* depth along camera_matrix1 (:2412) */
		/* compact_kernel_nerf (:1403-1426): dead rays with alpha > 0.001 reach shading; rays that are
		 * still alive when the march loop ends never do. */
		if (!payload.alive && rgba[3] > 0.001f) {
			++n_hit;
			shade_one(o, rgba, depth, payload.idx, payload.n_steps, frame_buffer, depth_buffer);
		}
	}
	if (stats) {
		stats->n_rays = (uint64_t)n_pixels;
		stats->n_rays_alive_after_init = n_alive;
		stats->n_rays_hit = n_hit;
		stats->n_samples = n_samples;
	}
}

void orc_trace_payloads(const orc_nerf_model* m, const float* cam_matrix, const orc_render_opts* o, uint32_t n, orc_payload* payloads, float* rgba, float* depth, orc_render_stats* stats) {
	uint64_t n_alive = 0, n_hit = 0, n_samples = 0;
#ifdef _OPENMP
	if (o->n_threads > 0) omp_set_num_threads(o->n_threads);
#endif
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : n_alive, n_hit, n_samples)
	for (int64_t i = 0; i < (int64_t)n; ++i) {
		if (!payloads[i].alive) continue;
		++n_alive;
		n_samples += orc_trace_ray(m, cam_matrix, o, &payloads[i], rgba + 4 * i, depth + i);
		if (!payloads[i].alive && rgba[4 * i + 3] > 0.001f) ++n_hit;
	}
	if (stats) {
		stats->n_rays = n;
		stats->n_rays_alive_after_init = n_alive;
		stats->n_rays_hit = n_hit;
		stats->n_samples = n_samples;
	}
}

/* ------------------------------------------------------------------ P1 accumulate + tonemap
 * render_buffer.cu:228-262 (EColorSpace::Linear) and :529-561 (Identity tonemap curve, no DLSS). */
void orc_accumulate(uint32_t n_pixels, const float* frame_buffer, float* accumulate_buffer, float sample_count) {
	for (size_t i = 0; i < (size_t)n_pixels * 4; ++i) {
		accumulate_buffer[i] = (accumulate_buffer[i] * sample_count + frame_buffer[i]) / (sample_count + 1.0f);
	}
}

/* accumulate_kernel / tonemap_kernel with EColorSpace::SRGB (src/render_buffer.cu:241-248, 537-541, 324-340): samples are
 * averaged as sRGB values, the background stays sRGB, the result is linearised before exposure. color_space 0 = Linear. */
void orc_accumulate_cs(uint32_t n_pixels, const float* frame_buffer, float* accumulate_buffer, float sample_count, int32_t color_space) {
	for (size_t i = 0; i < (size_t)n_pixels; ++i) {
		for (int k = 0; k < 4; ++k) {
			float c = frame_buffer[4 * i + k];
			if (color_space == 1 && k < 3) c = orc_linear_to_srgb(c);
			accumulate_buffer[4 * i + k] = (accumulate_buffer[4 * i + k] * sample_count + c) / (sample_count + 1.0f);
		}
	}
}
void orc_tonemap_cs(uint32_t n_pixels, const float* accumulate_buffer, const float* background_rgba, float exposure, int32_t to_srgb, int32_t color_space,
                    float* rgba_out) {
	float bg[4] = {background_rgba[0], background_rgba[1], background_rgba[2], background_rgba[3]};
	if (color_space != 1) for (int k = 0; k < 3; ++k) bg[k] = orc_srgb_to_linear(bg[k]);
	float scale = powf(2.0f, exposure);
	for (size_t i = 0; i < (size_t)n_pixels; ++i) {
		float c[4] = {accumulate_buffer[4 * i], accumulate_buffer[4 * i + 1], accumulate_buffer[4 * i + 2], accumulate_buffer[4 * i + 3]};
		float weight = (1.0f - c[3]) * bg[3];
		for (int k = 0; k < 3; ++k) c[k] += bg[k] * weight;
		c[3] += weight;
		for (int k = 0; k < 3; ++k) {
			if (color_space == 1) c[k] = orc_srgb_to_linear(c[k]);
			c[k] *= scale;
			if (to_srgb) c[k] = orc_linear_to_srgb(c[k]);
		}
		for (int k = 0; k < 4; ++k) rgba_out[4 * i + k] = c[k];
	}
}

void orc_tonemap(uint32_t n_pixels, const float* accumulate_buffer, const float* background_rgba, float exposure, int32_t to_srgb, float* rgba_out) {
	/* colour space is Linear, so the sRGB background colour is linearised first */
	float bg[4] = {orc_srgb_to_linear(background_rgba[0]), orc_srgb_to_linear(background_rgba[1]), orc_srgb_to_linear(background_rgba[2]), background_rgba[3]};
	float scale = powf(2.0f, exposure);
	for (size_t i = 0; i < (size_t)n_pixels; ++i) {
		float c[4] = {accumulate_buffer[4 * i], accumulate_buffer[4 * i + 1], accumulate_buffer[4 * i + 2], accumulate_buffer[4 * i + 3]};
		float weight = (1.0f - c[3]) * bg[3];
		for (int k = 0; k < 3; ++k) c[k] += bg[k] * weight;
		c[3] += weight;
		for (int k = 0; k < 3; ++k) {
			c[k] *= scale;
			if (to_srgb) c[k] = orc_linear_to_srgb(c[k]);
		}
		for (int k = 0; k < 4; ++k) rgba_out[4 * i + k] = c[k];
	}
}


/* ------------------------------------------------------------------ density-grid refresh
 * Testbed::update_density_grid_nerf, src/testbed_nerf.cu:2772-2861; kernels :185-232 (sample generation, splat)
 * and :253-276 (decayed maximum). default_rng_t is pcg32 from tiny-cuda-nn's dependencies/pcg32/pcg32.h (an
 * un-vendored submodule, absent from the mount): restated here from the published generator -- 64-bit LCG
 * (multiplier 0x5851f42d4c957f2d), XSH-RR output, skip-ahead in log time, next_float from the top 23 bits. */
static uint32_t pcg32_next_uint(orc_pcg32* r) {
	uint64_t oldstate = r->state;
	r->state = oldstate * 0x5851f42d4c957f2dULL + r->inc;
	uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
	uint32_t rot = (uint32_t)(oldstate >> 59u);
	return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31u));
}
void orc_pcg32_seed(orc_pcg32* r, uint64_t initstate, uint64_t initseq) {
	r->state = 0u;
	r->inc = (initseq << 1u) | 1u;
	pcg32_next_uint(r);
	r->state += initstate;
	pcg32_next_uint(r);
}
uint32_t orc_pcg32_next_uint(orc_pcg32* r) { return pcg32_next_uint(r); }
static float pcg32_next_float(orc_pcg32* r) {
	union { uint32_t u; float f; } x;
	x.u = (pcg32_next_uint(r) >> 9) | 0x3f800000u;
	return x.f - 1.0f;
}
void orc_pcg32_advance(orc_pcg32* r, uint64_t delta) {
	uint64_t cur_mult = 0x5851f42d4c957f2dULL, cur_plus = r->inc, acc_mult = 1u, acc_plus = 0u;
	while (delta > 0) {
		if (delta & 1) {
			acc_mult *= cur_mult;
			acc_plus = acc_plus * cur_mult + cur_plus;
		}
		cur_plus = (cur_mult + 1) * cur_plus;
		cur_mult *= cur_mult;
		delta >>= 1;
	}
	r->state = acc_mult * r->state + acc_plus;
}

/* generate_grid_samples_nerf_nonuniform (:185-213) + NerfNetwork::density + splat_..._max_nearest_neighbor (:215-232) */
static void grid_samples_splat(const orc_nerf_model* m, const prepared_t* p, uint32_t n_samples, orc_pcg32 rng0, uint32_t step, uint32_t n_cascades, float thresh,
                               const float* grid_in, float* grid_tmp) {
#pragma omp parallel for schedule(dynamic, 4096)
	for (int64_t ii = 0; ii < (int64_t)n_samples; ++ii) {
		const uint32_t i = (uint32_t)ii;
		orc_pcg32 rng = rng0;
		orc_pcg32_advance(&rng, (uint64_t)i * 4u);
		uint32_t level = (uint32_t)(pcg32_next_float(&rng) * (float)n_cascades) % n_cascades;
		uint32_t idx = 0;
		for (uint32_t j = 0; j < 10; ++j) {
			idx = ((i + step * n_samples) * 56924617u + j * 19349663u + 96925573u) % NERF_GRID_N_CELLS;
			idx += level * NERF_GRID_N_CELLS;
			if (grid_in[idx] > thresh) break;
		}
		uint32_t pos_idx = idx % NERF_GRID_N_CELLS;
		float x = (float)morton3D_invert(pos_idx >> 0), y = (float)morton3D_invert(pos_idx >> 1), z = (float)morton3D_invert(pos_idx >> 2);
		float rx = pcg32_next_float(&rng), ry = pcg32_next_float(&rng), rz = pcg32_next_float(&rng);
		float scale = scalbnf(1.0f, (int)level);
		float pos[3] = {((x + rx) / (float)NERF_GRIDSIZE - 0.5f) * scale + 0.5f, ((y + ry) / (float)NERF_GRIDSIZE - 0.5f) * scale + 0.5f,
		                ((z + rz) / (float)NERF_GRIDSIZE - 0.5f) * scale + 0.5f};
		float pos01[3];
		for (int d = 0; d < 3; ++d) { /* warp_position */
			float lo = d == 0 ? p->aabb.min.x : (d == 1 ? p->aabb.min.y : p->aabb.min.z);
			float hi = d == 0 ? p->aabb.max.x : (d == 1 ? p->aabb.max.y : p->aabb.max.z);
			pos01[d] = (pos[d] - lo) / (hi - lo);
		}
		uint16_t enc_h[ORC_MAX_LEVELS * 8];
		float enc[ORC_MAX_LEVELS * 8], dens_f[64];
		uint16_t dens_h[32];
		if (m->pos_encoding == 2) identity_encode_one(3, p->enc_dims, pos01, enc_h);
		else if (m->pos_encoding == 1) frequency_encode_one(3, m->pos_n_frequencies, p->enc_dims, pos01, enc_h); /* any NerfNetwork: NerfNetwork::density, nerf_network.h */
		else grid_encode_one(m, p, pos01, enc_h);
		for (uint32_t k = 0; k < p->enc_dims; ++k) enc[k] = orc_half_to_float(enc_h[k]);
		mlp_forward(m->mlp_accumulate == ORC_MLP_ACC_FP16_K16 ? ORC_MLP_ACC_FP16_K16 : ORC_MLP_ACC_EXACT, p->density_w, p->enc_dims, m->n_neurons, m->n_hidden_density, m->density_out_dims, enc, dens_f, dens_h);
		float thickness = network_to_density(orc_half_to_float(dens_h[0]), m->density_activation) * MIN_CONE_STEPSIZE();
		/* atomicMax on the bit pattern: positive floats order like unsigned ints */
		union { float f; uint32_t u; } nv, ov;
		nv.f = thickness;
#pragma omp critical(orc_grid_splat)
		{
			ov.f = grid_tmp[idx];
			if (nv.u > ov.u) grid_tmp[idx] = nv.f;
		}
	}
}

void orc_update_density_grid(const orc_nerf_model* m, float* grid, uint32_t max_cascade, orc_pcg32* rng, uint32_t* ema_step, float decay, uint32_t n_uniform,
                             uint32_t n_nonuniform) {
	const prepared_t* p = (const prepared_t*)m->prepared;
	const uint32_t n_cascades = max_cascade + 1, n_elements = NERF_GRID_N_CELLS * n_cascades;
	float* tmp = (float*)calloc(n_elements, sizeof(float));
	grid_samples_splat(m, p, n_uniform, *rng, *ema_step, n_cascades, -0.01f, grid, tmp);
	orc_pcg32_advance(rng, 1ull << 32);
	grid_samples_splat(m, p, n_nonuniform, *rng, *ema_step, n_cascades, NERF_MIN_OPTICAL_THICKNESS, grid, tmp);
	orc_pcg32_advance(rng, 1ull << 32);
	for (uint32_t i = 0; i < n_elements; ++i) { /* ema_grid_samples_nerf */
		float prev = grid[i];
		grid[i] = prev < 0.f ? prev : fmaxf(prev * decay, tmp[i]);
	}
	++*ema_step;
	free(tmp);
}

/* ------------------------------------------------------------------ training step
 * generate_training_samples_nerf (src/testbed_nerf.cu:737-890) and compute_loss_kernel_train_nerf (:893-1213),
 * restated for the default path: no envmap, no error-map CDFs, no explicit rays, no random max level, static
 * cameras, no exposure / depth supervision / sharpness. Rays are processed in index order, so the sample offsets
 * (`base`) are deterministic here where the reference's atomics are not; tests compare ray by ray. */
static uint32_t mip_from_dt(float dt, v3 pos, uint32_t max_cascade) { /* nerf_device.cuh:449-458 */
	uint32_t mip = mip_from_pos(pos, max_cascade);
	dt *= 2.0f * (float)NERF_GRIDSIZE;
	if (dt < 1.0f) return mip;
	int exponent;
	(void)frexpf(dt, &exponent);
	int v = (int)mip < exponent ? exponent : (int)mip;
	if (v > (int)max_cascade) v = (int)max_cascade;
	return (uint32_t)v;
}
static void train_read_pixel(const orc_train_image* im, float u, float v, float* rgba) { /* read_rgba, common_device.cuh:797-829 */
	int px = (int)(u * (float)im->res[0]), py = (int)(v * (float)im->res[1]);
	px = px < 0 ? 0 : (px > im->res[0] - 1 ? im->res[0] - 1 : px);
	py = py < 0 ? 0 : (py > im->res[1] - 1 ? im->res[1] - 1 : py);
	size_t idx = (size_t)px + (size_t)py * (size_t)im->res[0];
	if (im->type == 1) {
		uint32_t raw = ((const uint32_t*)im->pixels)[idx];
		if (raw == 0x00FF00FFu) { rgba[0] = rgba[1] = rgba[2] = rgba[3] = -1.0f; return; }
		float a = (float)(raw >> 24) * (1.0f / 255.0f);
		rgba[0] = orc_srgb_to_linear((float)(raw & 255u) * (1.0f / 255.0f)) * a;
		rgba[1] = orc_srgb_to_linear((float)((raw >> 8) & 255u) * (1.0f / 255.0f)) * a;
		rgba[2] = orc_srgb_to_linear((float)((raw >> 16) & 255u) * (1.0f / 255.0f)) * a;
		rgba[3] = a;
	} else if (im->type == 3) {
		for (int k = 0; k < 4; ++k) rgba[k] = ((const float*)im->pixels)[idx * 4 + k];
	} else {
		rgba[0] = 5.0f; rgba[1] = 0.0f; rgba[2] = 0.0f; rgba[3] = 1.0f;
	}
}
static void train_uv(orc_pcg32* rng, const orc_train_image* im, int snap, float* u, float* v) { /* nerf_device.cuh:592-615 */
	*u = pcg32_next_float(rng);
	*v = pcg32_next_float(rng);
	if (snap) {
		int px = (int)(*u * (float)im->res[0]), py = (int)(*v * (float)im->res[1]);
		px = px < 0 ? 0 : (px > im->res[0] - 1 ? im->res[0] - 1 : px);
		py = py < 0 ? 0 : (py > im->res[1] - 1 ? im->res[1] - 1 : py);
		*u = ((float)px + 0.5f) / (float)im->res[0];
		*v = ((float)py + 0.5f) / (float)im->res[1];
	}
}
static uint32_t train_image_idx(uint32_t base_idx, uint32_t n_rays, uint32_t n_images) { return ((base_idx * n_images) / n_rays) % n_images; } /* :617-638 */

uint32_t orc_train_generate_samples(const orc_nerf_model* m, const orc_train_image* images, const orc_train_opts* o, uint32_t max_samples, uint32_t* numsteps,
                                    uint32_t* base_out, float* rays6, float* coords) {
	const aabb_t aabb = {v3_make(m->aabb_min[0], m->aabb_min[1], m->aabb_min[2]), v3_make(m->aabb_max[0], m->aabb_max[1], m->aabb_max[2])};
	const v3 diag = v3_sub(aabb.max, aabb.min);
	uint32_t counter = 0;
	for (uint32_t i = 0; i < o->n_rays; ++i) {
		numsteps[i] = 0;
		base_out[i] = 0;
		const orc_train_image* im = &images[train_image_idx(i, o->n_rays, o->n_images)];
		orc_pcg32 rng = o->rng;
		orc_pcg32_advance(&rng, (uint64_t)(uint32_t)(i * 16u)); /* N_MAX_RANDOM_SAMPLES_PER_RAY */
		float u, v, px[4];
		train_uv(&rng, im, o->snap_to_pixel_centers, &u, &v);
		train_read_pixel(im, u, v, px);
		if (px[0] < 0.0f) continue;
		(void)pcg32_next_float(&rng); /* motionblur_time */
		orc_camera cam;
		memset(&cam, 0, sizeof(cam));
		cam.width = im->res[0]; cam.height = im->res[1];
		cam.focal_length[0] = im->focal[0]; cam.focal_length[1] = im->focal[1];
		cam.screen_center[0] = im->principal[0]; cam.screen_center[1] = im->principal[1];
		cam.lens_mode = im->lens_mode;
		memcpy(cam.lens_params, im->lens_params, sizeof(cam.lens_params));
		const v3 d_un = m3_mulv(im->xform, lens_direction(&cam, u, v));
		const v3 org = v3_make(im->xform[9], im->xform[10], im->xform[11]);
		const v3 d = v3_normalize(d_un);
		float tmin, tmax;
		aabb_ray_intersect(&aabb, org, d, &tmin, &tmax);
		tmin = fmaxf(tmin, 0.0f);
		const float cone_angle = m->cone_angle_constant;
		const float startt = advance_n_steps(tmin, cone_angle, pcg32_next_float(&rng));
		const v3 idir = v3_make(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
		uint32_t j = 0;
		float t = startt;
		v3 pos;
		while (aabb_contains(&aabb, pos = v3_add(org, v3_scale(d, t))) && j < NERF_STEPS) {
			float dt = calc_dt(t, cone_angle);
			uint32_t mip = mip_from_dt(dt, pos, m->max_cascade);
			if (density_grid_occupied_at(pos, m->density_grid_bitfield, mip)) { ++j; t += dt; }
			else t = advance_to_next_voxel(t, cone_angle, pos, d, idir, mip);
		}
		if (j == 0) continue;
		const uint32_t n = j, base = counter;
		counter += n;
		if (base + n > max_samples) continue;
		numsteps[i] = n;
		base_out[i] = base;
		rays6[i * 6 + 0] = org.x; rays6[i * 6 + 1] = org.y; rays6[i * 6 + 2] = org.z;
		rays6[i * 6 + 3] = d_un.x; rays6[i * 6 + 4] = d_un.y; rays6[i * 6 + 5] = d_un.z;
		float* c = coords + (size_t)base * 7;
		t = startt;
		j = 0;
		while (aabb_contains(&aabb, pos = v3_add(org, v3_scale(d, t))) && j < n) {
			float dt = calc_dt(t, cone_angle);
			uint32_t mip = mip_from_dt(dt, pos, m->max_cascade);
			if (density_grid_occupied_at(pos, m->density_grid_bitfield, mip)) {
				v3 w = v3_div(v3_sub(pos, aabb.min), diag); /* warp_position */
				c[0] = w.x; c[1] = w.y; c[2] = w.z; c[3] = warp_dt(dt);
				c[4] = (d.x + 1.0f) * 0.5f; c[5] = (d.y + 1.0f) * 0.5f; c[6] = (d.z + 1.0f) * 0.5f;
				c += 7;
				++j;
				t += dt;
			} else {
				t = advance_to_next_voxel(t, cone_angle, pos, d, idir, mip);
			}
		}
	}
	return counter;
}

static void train_loss_and_gradient(float target, float prediction, int type, float* loss, float* grad) { /* nerf_device.cuh:61-142, 640-658 */
	const float diff = prediction - target, sign = copysignf(1.0f, diff);
	switch (type) {
		case 1: *loss = fabsf(diff); *grad = sign; break;
		case 2: { float den = fabsf(prediction) + 1e-2f; *loss = fabsf(diff) / den; *grad = sign / den; break; }
		case 3: { float den = 0.5f * (fabsf(prediction) + fabsf(target)) + 1e-2f; *loss = fabsf(diff) / den; *grad = sign / den; break; }
		case 4: {
			const float alpha = 0.1f, ad = fabsf(diff);
			*loss = (ad > alpha ? (ad - 0.5f * alpha) : (0.5f / alpha * diff * diff)) / 5.0f;
			*grad = (ad > alpha ? (diff > 0.0f ? 1.0f : -1.0f) : (diff / alpha)) / 5.0f;
			break;
		}
		case 5: { float div = fabsf(diff) + 1.0f; *loss = logf(div); *grad = sign / div; break; }
		case 6: { float den = prediction * prediction + 1e-2f; *loss = diff * diff / den; *grad = 2.0f * diff / den; break; }
		default: *loss = diff * diff; *grad = 2.0f * diff; break;
	}
}
static float network_to_rgb_derivative(float val, uint32_t act) { /* :214-223 */
	switch (act) {
		case ORC_ACT_RELU: return val > 0.0f ? 1.0f : 0.0f;
		case ORC_ACT_LOGISTIC: { float s = logistic_(val); return s * (1.0f - s); }
		case ORC_ACT_EXPONENTIAL: return expf(clampf_(val, -10.0f, 10.0f));
		default: return 1.0f;
	}
}
static float network_to_density_derivative(float val, uint32_t act) { /* :245-254 */
	switch (act) {
		case ORC_ACT_RELU: return val > 0.0f ? 1.0f : 0.0f;
		case ORC_ACT_LOGISTIC: { float s = logistic_(val); return s * (1.0f - s); }
		case ORC_ACT_EXPONENTIAL: return expf(clampf_(val, -15.0f, 15.0f));
		default: return 1.0f;
	}
}

/* network_output: fp16 x 4 per marched sample (rgb, density logit). Outputs per ray: compacted_numsteps (the prefix
 * of the ray's samples that receives a gradient), loss; per marched sample: dloss fp16 x 4 at the same index as its
 * coordinate (rows behind a ray's prefix stay zero). Every ray fits (no target-batch clamp). */
void orc_train_loss(const orc_nerf_model* m, const orc_train_image* images, const orc_train_opts* o, const uint32_t* numsteps, const uint32_t* base_in,
                    const float* rays6, const float* coords, const uint16_t* network_output, uint32_t* compacted_numsteps, float* loss_out, uint16_t* dloss) {
	const v3 amin = v3_make(m->aabb_min[0], m->aabb_min[1], m->aabb_min[2]);
	const v3 diag = v3_sub(v3_make(m->aabb_max[0], m->aabb_max[1], m->aabb_max[2]), amin);
	for (uint32_t i = 0; i < o->n_rays; ++i) {
		compacted_numsteps[i] = 0;
		loss_out[i] = 0.0f;
		const uint32_t n = numsteps[i], base = base_in[i];
		if (n == 0) continue;
		const float* cin = coords + (size_t)base * 7;
		const uint16_t* net = network_output + (size_t)base * 4;
		float T = 1.0f;
		const float EPSILON = 1e-4f;
		v3 rgb_ray = v3_make(0.f, 0.f, 0.f);
		uint32_t cn = 0;
		for (; cn < n; ++cn) {
			if (T < EPSILON) break;
			const uint16_t* q = net + (size_t)cn * 4;
			v3 rgb = v3_make(network_to_rgb(orc_half_to_float(q[0]), m->rgb_activation), network_to_rgb(orc_half_to_float(q[1]), m->rgb_activation),
			                 network_to_rgb(orc_half_to_float(q[2]), m->rgb_activation));
			float dt = unwarp_dt(cin[(size_t)cn * 7 + 3]);
			float density = network_to_density(orc_half_to_float(q[3]), m->density_activation);
			float alpha = 1.0f - expf(-density * dt);
			float weight = alpha * T;
			rgb_ray = v3_add(rgb_ray, v3_scale(rgb, weight));
			T *= (1.0f - alpha);
		}
		orc_pcg32 rng = o->rng;
		orc_pcg32_advance(&rng, (uint64_t)(uint32_t)(i * 16u));
		const orc_train_image* im = &images[train_image_idx(i, o->n_rays, o->n_images)];
		float u, v;
		train_uv(&rng, im, o->snap_to_pixel_centers, &u, &v);
		orc_pcg32_advance(&rng, 1); /* motionblur_time */
		v3 bg = v3_make(o->background[0], o->background[1], o->background[2]);
		if (o->random_bg_color) {
			bg.x = pcg32_next_float(&rng);
			bg.y = pcg32_next_float(&rng);
			bg.z = pcg32_next_float(&rng);
		}
		bg = v3_make(orc_srgb_to_linear(bg.x), orc_srgb_to_linear(bg.y), orc_srgb_to_linear(bg.z));
		float tex[4];
		train_read_pixel(im, u, v, tex);
		v3 target;
		if (o->linear_colors || o->color_space == 0) {
			target = v3_make(tex[0] + (1.0f - tex[3]) * bg.x, tex[1] + (1.0f - tex[3]) * bg.y, tex[2] + (1.0f - tex[3]) * bg.z);
			if (!o->linear_colors) {
				target = v3_make(orc_linear_to_srgb(target.x), orc_linear_to_srgb(target.y), orc_linear_to_srgb(target.z));
				bg = v3_make(orc_linear_to_srgb(bg.x), orc_linear_to_srgb(bg.y), orc_linear_to_srgb(bg.z));
			}
		} else {
			bg = v3_make(orc_linear_to_srgb(bg.x), orc_linear_to_srgb(bg.y), orc_linear_to_srgb(bg.z));
			if (tex[3] > 0.0f) {
				target = v3_make(orc_linear_to_srgb(tex[0] / tex[3]) * tex[3] + (1.0f - tex[3]) * bg.x, orc_linear_to_srgb(tex[1] / tex[3]) * tex[3] + (1.0f - tex[3]) * bg.y,
				                 orc_linear_to_srgb(tex[2] / tex[3]) * tex[3] + (1.0f - tex[3]) * bg.z);
			} else {
				target = bg;
			}
		}
		if (cn == n) rgb_ray = v3_add(rgb_ray, v3_scale(bg, T));
		compacted_numsteps[i] = cn;
		if (cn == 0) continue;
		float lx, ly, lz, gx, gy, gz;
		train_loss_and_gradient(target.x, rgb_ray.x, o->loss_type, &lx, &gx);
		train_loss_and_gradient(target.y, rgb_ray.y, o->loss_type, &ly, &gy);
		train_loss_and_gradient(target.z, rgb_ray.z, o->loss_type, &lz, &gz);
		const v3 lgrad = v3_make(gx, gy, gz);
		loss_out[i] = ((lx + ly + lz) / 3.0f) / (float)o->n_rays;
		const float loss_scale = o->loss_scale / (float)o->n_rays;
		const float output_l2_reg = m->rgb_activation == ORC_ACT_EXPONENTIAL ? 1e-4f : 0.0f;
		const float output_l1_reg_density = o->density_grid_mean < 0.01f ? 1e-4f : 0.0f;
		const v3 ray_o = v3_make(rays6[i * 6 + 0], rays6[i * 6 + 1], rays6[i * 6 + 2]);
		v3 rgb_ray2 = v3_make(0.f, 0.f, 0.f);
		T = 1.0f;
		for (uint32_t j = 0; j < cn; ++j) {
			const float* c = cin + (size_t)j * 7;
			const v3 pos = v3_add(v3_mul(v3_make(c[0], c[1], c[2]), diag), amin);
			const float depth = v3_length(v3_sub(pos, ray_o));
			const float dt = unwarp_dt(c[3]);
			const uint16_t* q = net + (size_t)j * 4;
			const float o0 = orc_half_to_float(q[0]), o1 = orc_half_to_float(q[1]), o2 = orc_half_to_float(q[2]), o3 = orc_half_to_float(q[3]);
			const v3 rgb = v3_make(network_to_rgb(o0, m->rgb_activation), network_to_rgb(o1, m->rgb_activation), network_to_rgb(o2, m->rgb_activation));
			const float density = network_to_density(o3, m->density_activation);
			const float alpha = 1.0f - expf(-density * dt);
			const float weight = alpha * T;
			rgb_ray2 = v3_add(rgb_ray2, v3_scale(rgb, weight));
			T *= (1.0f - alpha);
			const v3 suffix = v3_sub(rgb_ray, rgb_ray2);
			const v3 dl = v3_scale(lgrad, weight);
			uint16_t* out = dloss + (size_t)(base + j) * 4;
			out[0] = orc_float_to_half(loss_scale * (dl.x * network_to_rgb_derivative(o0, m->rgb_activation) + fmaxf(0.0f, output_l2_reg * o0)));
			out[1] = orc_float_to_half(loss_scale * (dl.y * network_to_rgb_derivative(o1, m->rgb_activation) + fmaxf(0.0f, output_l2_reg * o1)));
			out[2] = orc_float_to_half(loss_scale * (dl.z * network_to_rgb_derivative(o2, m->rgb_activation) + fmaxf(0.0f, output_l2_reg * o2)));
			const float dd = network_to_density_derivative(o3, m->density_activation);
			const float by_mlp = dd * (dt * v3_dot(lgrad, v3_sub(v3_scale(rgb, T), suffix)));
			out[3] = orc_float_to_half(loss_scale * by_mlp + (o3 < 0.0f ? -output_l1_reg_density : 0.0f) + (o3 > -10.0f && depth < o->near_distance ? 1e-4f : 0.0f));
		}
	}
}
