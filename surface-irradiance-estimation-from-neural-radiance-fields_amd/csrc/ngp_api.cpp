// C ABI of libngp_hip (include/ngp_hip.h): model upload, snapshot / transforms.json I/O, frame rendering.
// Host logic only; every device computation lives in nerf_kernels.hip. There is no CPU fallback: each entry point
// that computes needs a HIP device and fails with an error otherwise.
#include "ngp_host.h"

#include <map>
#include <mutex>

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <dirent.h>
#include <fstream>
#include <sstream>
#include <sys/stat.h>

using namespace ngp;

namespace {

uint16_t float_to_half(float f) { // round to nearest even, for "params_type": "float" snapshots
	uint32_t x;
	memcpy(&x, &f, 4);
	uint32_t sign = (x >> 16) & 0x8000u;
	int32_t exp = (int32_t)((x >> 23) & 0xff) - 127 + 15;
	uint32_t man = x & 0x7fffffu;
	if (((x >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (man ? 0x200u : 0));
	if (exp >= 31) return (uint16_t)(sign | 0x7c00u);
	if (exp <= 0) {
		if (exp < -10) return (uint16_t)sign;
		man |= 0x800000u;
		uint32_t shift = (uint32_t)(14 - exp);
		uint32_t half_man = man >> shift;
		uint32_t rem = man & ((1u << shift) - 1u);
		uint32_t halfway = 1u << (shift - 1);
		if (rem > halfway || (rem == halfway && (half_man & 1u))) ++half_man;
		return (uint16_t)(sign | half_man);
	}
	uint32_t half = (uint32_t)(exp << 10) | (man >> 13);
	uint32_t rem = man & 0x1fffu;
	if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) ++half;
	return (uint16_t)(sign | half);
}

std::string inflate_all(const void* data, size_t n) { // zlib or gzip container (zstr, src/testbed.cu:262-266)
	z_stream zs;
	memset(&zs, 0, sizeof(zs));
	if (inflateInit2(&zs, 15 + 32) != Z_OK) throw std::runtime_error("inflateInit2 failed");
	zs.next_in = (Bytef*)data;
	zs.avail_in = (uInt)n;
	std::string out;
	std::vector<char> buf(1 << 20);
	int rc;
	do {
		zs.next_out = (Bytef*)buf.data();
		zs.avail_out = (uInt)buf.size();
		rc = inflate(&zs, Z_NO_FLUSH);
		if (rc != Z_OK && rc != Z_STREAM_END) {
			inflateEnd(&zs);
			throw std::runtime_error("inflate failed: corrupt .ingp stream");
		}
		out.append(buf.data(), buf.size() - zs.avail_out);
	} while (rc != Z_STREAM_END);
	inflateEnd(&zs);
	return out;
}

std::string deflate_gzip(const std::string& in, int level) {
	z_stream zs;
	memset(&zs, 0, sizeof(zs));
	if (deflateInit2(&zs, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("deflateInit2 failed");
	zs.next_in = (Bytef*)in.data();
	zs.avail_in = (uInt)in.size();
	std::string out;
	std::vector<char> buf(1 << 20);
	int rc;
	do {
		zs.next_out = (Bytef*)buf.data();
		zs.avail_out = (uInt)buf.size();
		rc = deflate(&zs, Z_FINISH);
		out.append(buf.data(), buf.size() - zs.avail_out);
	} while (rc != Z_STREAM_END);
	deflateEnd(&zs);
	return out;
}

// ------------------------------------------------------------------------------------------------ sampling (host)
// ld_random_pixel_offset (random_val.cuh:365-370) is a per-frame constant, so it is evaluated once on the host.
// Sobol dimensions 0 and 1 need no table: dim 0 is a bit reversal, dim 1's direction numbers obey v[i] = v[i-1] ^ (v[i-1] >> 1).
uint32_t reverse_bits32(uint32_t x) {
	x = ((x & 0xaaaaaaaau) >> 1) | ((x & 0x55555555u) << 1);
	x = ((x & 0xccccccccu) >> 2) | ((x & 0x33333333u) << 2);
	x = ((x & 0xf0f0f0f0u) >> 4) | ((x & 0x0f0f0f0fu) << 4);
	x = ((x & 0xff00ff00u) >> 8) | ((x & 0x00ff00ffu) << 8);
	return (x >> 16) | (x << 16);
}
uint32_t lk_perm(uint32_t x, uint32_t seed) {
	x += seed;
	x ^= x * 0x6c50b47cu;
	x ^= x * 0xb82f1e52u;
	x ^= x * 0xc7afe638u;
	x ^= x * 0x8d22f6e6u;
	return x;
}
uint32_t nus2(uint32_t x, uint32_t seed) { return reverse_bits32(lk_perm(reverse_bits32(x), seed)); }
uint32_t hash_combine(uint32_t seed, uint32_t v) { return seed ^ (v + (seed << 6) + (seed >> 2)); }
uint32_t sobol_dim(uint32_t index, int dim) {
	if (dim == 0) return reverse_bits32(index);
	uint32_t v = 0x80000000u, X = 0;
	for (int bit = 0; bit < 32; ++bit) {
		if ((index >> bit) & 1u) X ^= v;
		v ^= v >> 1;
	}
	return X;
}
void ld_random_val_2d(uint32_t index, uint32_t seed, float* out) {
	index = nus2(index, seed);
	for (int i = 0; i < 2; ++i) out[i] = (float)nus2(sobol_dim(index, i), hash_combine(seed, (uint32_t)i)) * 2.3283064365386963e-10f;
}
void ld_random_pixel_offset(uint32_t spp, float* out) {
	float a[2], b[2];
	ld_random_val_2d(0, 0xdeadbeefu, a);
	ld_random_val_2d(spp, 0xdeadbeefu, b);
	for (int i = 0; i < 2; ++i) {
		float v = (0.5f - a[i]) + b[i];
		out[i] = v - floorf(v);
	}
}

// ------------------------------------------------------------------------------------------------ model
uint32_t next_multiple(uint32_t v, uint32_t d) { return ((v + d - 1) / d) * d; }

// tcnn GridEncoding level table
void build_levels(const ngp_model_desc& d, LevelInfo* lv, uint32_t* total_entries) {
	float log2_pls = log2f(d.per_level_scale);
	uint32_t offset = 0;
	for (uint32_t l = 0; l < d.n_levels; ++l) {
		float scale = exp2f((float)l * log2_pls) * (float)d.base_resolution - 1.0f;
		if (!(scale >= 0.0f && scale < 1073741824.0f)) throw std::runtime_error("invalid hash grid configuration (a level's resolution is out of range)");
		uint32_t res = (uint32_t)ceilf(scale) + 1u;
		uint32_t max_params = 0xFFFFFFFFu / 2u;
		uint32_t n = powf((float)res, 3.0f) > (float)max_params ? max_params : res * res * res;
		n = next_multiple(n, 8u);
		n = std::min(n, 1u << d.log2_hashmap_size);
		// grid_index: strides accumulate while stride <= size; hashed iff the final stride exceeds the level size
		uint32_t stride = 1;
		for (int dim = 0; dim < 3 && stride <= n; ++dim) stride *= res;
		lv[l].scale = scale;
		lv[l].res = res;
		lv[l].size = n;
		lv[l].offset = offset;
		lv[l].hashed = n < stride ? 1u : 0u;
		lv[l].mask = (n & (n - 1)) == 0 ? n - 1 : 0u;
		lv[l].xor_disabled = 0;
		if ((uint64_t)offset + n > 0x1FFFFFFFull) throw std::runtime_error("grid encoding too large for 32-bit gather offsets (more than 2^29 entries)");
		offset += n;
	}
	*total_entries = offset;
}

// Xor layout of the hash-grid table for the render kernels. tcnn's grid_index has two shapes -- a dense
// x + y*res + z*res^2 (wrapped modulo the level size when a corner coordinate reaches res) and a prime-multiplier
// xor hash -- and the two halves of a wave work on levels of different shape. Dense levels are therefore re-laid
// out at load time with power-of-two strides, entry (x, y, z) at x | y << b | z << 2b for x, y, z in [0, res],
// 2^b > res, each holding the entry tcnn's formula (including its wrap) would have fetched; then
//   dense:  8x ^ y * (8 << b) ^ z * (8 << 2b)        (disjoint bit fields: xor == add)
//   hashed: 8x ^ y * (8 * 2654435761) ^ z * (8 * 805459861), masked with 8 * (size - 1)
// is ONE formula with per-level multipliers, and aligning every level to its power-of-two footprint turns
// "+ offset" into an OR. Same table entries, so the features are bit-identical; the cost is HBM nobody misses
// (Lego-shaped model: 23 MB -> 38 MB).
uint64_t pow2_ceil(uint64_t v) {
	uint64_t p = 1;
	while (p < v) p <<= 1;
	return p;
}
void build_xor_layout(LevelInfo* lv, uint32_t n_levels, const uint16_t* grid /* tcnn order, 4 halves per entry */, std::vector<uint64_t>& table) {
	uint64_t cursor = 0;
	std::vector<uint32_t> bits(n_levels, 0), wrapped(n_levels, 0);
	for (uint32_t l = 0; l < n_levels; ++l) {
		LevelInfo& L = lv[l];
		uint64_t bytes;
		if (L.hashed) {
			if ((L.size & (L.size - 1)) != 0) throw std::runtime_error("hashed grid level whose size is not a power of two");
			bytes = (uint64_t)L.size * 8u;
		} else if ((uint64_t)L.res * L.res * L.res > (uint64_t)L.size) {
			// tcnn's grid_index forms its strides in uint32: at res = 65536 (level 6 of the upstream aabb_scale-128 configuration,
			// per_level_scale 4) res^2 wraps to 0, the loop's guard `stride <= hashmap_size` keeps going and the level is indexed
			// as (x + y * 65536 + z * 0) % size -- dense by the code's own test, with z dropped. Same entries here: x < res and a
			// power-of-two res keep the fields disjoint (add == xor), so the hashed form serves it with multipliers (res, 0);
			// the corner x + 1 == res would carry into y's field, hence coord_max = res - 2 (beyond it the wave takes
			// level_corners on the tcnn-order table, which wraps exactly like tcnn).
			const bool pow2 = (L.res & (L.res - 1)) == 0 && (L.size & (L.size - 1)) == 0 && L.res * L.res == 0u;
			wrapped[l] = pow2 ? 1 : 2; // 2: no xor form -- every wave takes the tcnn-order table for this level
			bytes = pow2 ? (uint64_t)L.size * 8u : 8u;
		} else {
			uint32_t b = 0;
			while ((1u << b) <= L.res) ++b; // 2^b > res: coordinates 0..res fit
			bits[l] = b;
			bytes = pow2_ceil(((uint64_t)(L.res + 1u) << (2 * b)) * 8u);
		}
		cursor = (cursor + bytes - 1) / bytes * bytes;
		if (cursor + bytes > 0xFFFFFFFFull) throw std::runtime_error("hash grid too large for 32-bit gather offsets");
		L.base8 = (uint32_t)cursor;
		if (L.hashed) {
			L.coord_max = 0xFFFFFFFFu;
			L.mul_y8 = 2654435761u * 8u;
			L.mul_z8 = 805459861u * 8u;
			L.mask8 = (L.size - 1u) * 8u;
		} else if (wrapped[l] == 1) {
			L.coord_max = L.res - 2u;
			L.mul_y8 = L.res * 8u;
			L.mul_z8 = 0u;
			L.mask8 = (L.size - 1u) * 8u;
		} else if (wrapped[l] == 2) {
			L.xor_disabled = 1u;
			L.coord_max = 0u;
			L.mul_y8 = L.mul_z8 = L.mask8 = 0u;
		} else {
			L.coord_max = L.res - 1u;
			L.mul_y8 = 8u << bits[l];
			L.mul_z8 = 8u << (2 * bits[l]);
			L.mask8 = 0xFFFFFFFFu;
		}
		cursor += bytes;
	}
	table.assign(cursor / 8u, 0ull);
	const uint64_t* src = (const uint64_t*)grid;
	for (uint32_t l = 0; l < n_levels; ++l) {
		const LevelInfo& L = lv[l];
		uint64_t* dst = table.data() + L.base8 / 8u;
		const uint64_t* level = src + L.offset;
		if (L.hashed || wrapped[l] == 1) {
			std::copy(level, level + L.size, dst);
			continue;
		}
		if (wrapped[l] == 2) continue;
		const uint32_t b = bits[l];
		for (uint32_t z = 0; z <= L.res; ++z)
			for (uint32_t y = 0; y <= L.res; ++y)
				for (uint32_t x = 0; x <= L.res; ++x) dst[x | (y << b) | (z << (2 * b))] = level[(x + y * L.res + z * L.res * L.res) % L.size];
	}
}

uint64_t mlp_n_params(uint32_t n_in, uint32_t width, uint32_t n_hidden, uint32_t n_out) {
	if (n_hidden == 0) return (uint64_t)n_out * n_in; // tcnn CutlassMLP without a hidden layer: one (padded output) x (input) matrix
	return (uint64_t)width * n_in + (uint64_t)(n_hidden - 1) * width * width + (uint64_t)n_out * width;
}

// MFMA A-operand fragments for v_mfma_f32_16x16x32_f16: fragment (tile m, k-step s) holds, in lane l = (h = l>>4,
// row = l&15), element j: W[16m + row][n(s,h,j)], n(s,h,j) = 32s + 16(j>>2) + 4h + (j&3). The K permutation n() is
// the order in which the previous layer's accumulator tiles (and the encoder's level pairs) already sit in the
// B operand's registers, so no activation ever moves between lanes (nerf_device.h mlp_pass).
// n_out rows are stored (a CutlassMLP's output layer: 8); tiles are filled up with zero rows.
void emit_fragments(std::vector<uint16_t>& frags, int first_frag, const uint16_t* W, int n_out, int n_in) {
	int f = first_frag;
	for (int m = 0; m < (n_out + 15) / 16; ++m) {
		for (int s = 0; s < n_in / 32; ++s, ++f) {
			for (int l = 0; l < 64; ++l) {
				int h = l >> 4, row = l & 15;
				for (int j = 0; j < 8; ++j) {
					int k = 32 * s + 16 * (j >> 2) + 4 * h + (j & 3);
					frags[((size_t)f * 64 + l) * 8 + j] = 16 * m + row < n_out ? W[(size_t)(16 * m + row) * n_in + k] : (uint16_t)0;
				}
			}
		}
	}
}

// MFMA A fragments of one layer of the wide architecture (ngp_kernels.h WideModel): [m tile][k block][lane] x 8 fp16, zeros beyond the matrix
WideLayer emit_wide_fragments(std::vector<uint16_t>& frags, const uint16_t* W, uint32_t n_out, uint32_t n_in) {
	constexpr uint32_t TM = (uint32_t)ngp::WIDE_TILE_M, TK = (uint32_t)ngp::WIDE_TILE_K;
	WideLayer L{};
	L.frag_offset = (uint32_t)(frags.size() / 8);
	L.n_kblocks = (uint16_t)((n_in <= 128 ? 128u : 256u) / TK); // the kernels are instantiated for K = 128 and 256 (zero columns beyond the matrix)
	L.n_mtiles = (uint16_t)((n_out + TM - 1) / TM);
	frags.resize(frags.size() + (size_t)L.n_mtiles * L.n_kblocks * 64 * 8, 0);
	uint16_t* out = frags.data() + (size_t)L.frag_offset * 8;
	for (uint32_t m = 0; m < L.n_mtiles; ++m)
		for (uint32_t kb = 0; kb < L.n_kblocks; ++kb)
			for (uint32_t l = 0; l < 64; ++l)
				for (uint32_t j = 0; j < 8; ++j) {
					const uint32_t row = TM * m + (l % TM), col = TK * kb + 8 * (l / TM) + j;
					if (row < n_out && col < n_in) out[(((size_t)m * L.n_kblocks + kb) * 64 + l) * 8 + j] = W[(size_t)row * n_in + col];
				}
	return L;
}

// widths of the wide architecture as NerfNetwork derives them (nerf_network.h:81-100)
struct WideShapes {
	uint32_t alignment, enc_dims, dir_dims, rgb_in, rgb_out;
};
WideShapes wide_shapes(const ngp_model_desc& d) {
	WideShapes w{};
	w.alignment = d.mlp_alignment ? d.mlp_alignment : 16u;
	auto up = [&](uint32_t v) { return (v + w.alignment - 1) / w.alignment * w.alignment; };
	w.enc_dims = d.pos_encoding == 2 ? up(3u) : up(6u * d.pos_n_frequencies);
	w.dir_dims = d.dir_encoding == 1 ? up(6u * d.dir_n_frequencies) : d.dir_encoding == 2 ? up(3u) : 16u;
	w.rgb_in = up(d.density_out_dims + w.dir_dims);
	w.rgb_out = up(3u);
	return w;
}

void free_model(ngp_ctx* ctx) {
	ngp::free_training(ctx);
	if (ctx->d_params) (void)hipFree(ctx->d_params);
	if (ctx->d_xgrid) (void)hipFree(ctx->d_xgrid);
	ctx->d_xgrid = nullptr;
	if (ctx->d_wfrags) (void)hipFree(ctx->d_wfrags);
	if (ctx->d_bitfield) (void)hipFree(ctx->d_bitfield);
	if (ctx->d_coarse) (void)hipFree(ctx->d_coarse);
	ctx->d_coarse = nullptr;
	if (ctx->d_density_f16) (void)hipFree(ctx->d_density_f16);
	if (ctx->d_density_f32) (void)hipFree(ctx->d_density_f32);
	if (ctx->d_partial) (void)hipFree(ctx->d_partial);
	if (ctx->d_density_tmp) (void)hipFree(ctx->d_density_tmp);
	ctx->d_density_tmp = nullptr;
	ctx->d_params = nullptr;
	ctx->d_wfrags = nullptr;
	ctx->d_bitfield = nullptr;
	ctx->d_density_f16 = nullptr;
	ctx->d_density_f32 = nullptr;
	ctx->d_partial = nullptr;
	ctx->model_loaded = false;
}

void set_model_impl(ngp_ctx* ctx, const ngp_model_desc& d) {
	const bool wide = d.pos_encoding >= 1; // Frequency (1) or Identity (2) position encoding: no grid, the wide-MLP kernels
	if (d.pos_encoding > 2 || d.dir_encoding > 2 || (d.mlp_alignment != 0 && d.mlp_alignment != 8 && d.mlp_alignment != 16)) throw std::runtime_error("invalid model descriptor (encoding kinds / mlp_alignment)");
	if (!wide && d.dir_encoding != 0) throw std::runtime_error("unsupported network architecture: a Frequency / Identity direction encoding is implemented together with a Frequency / Identity position encoding (configs/nerf/frequency.json, none.json)");
	if (wide) {
		if ((d.n_neurons != 128 && d.n_neurons != 256) || d.n_hidden_density < 1 || d.n_hidden_rgb < 1 || d.n_hidden_density + d.n_hidden_rgb + 2 > (uint32_t)WIDE_MAX_LAYERS ||
		    d.density_out_dims != 16 || (d.pos_encoding == 1 && (d.pos_n_frequencies < 1 || d.pos_n_frequencies > 40)) || (d.dir_encoding == 1 && (d.dir_n_frequencies < 1 || d.dir_n_frequencies > 4))) {
			throw std::runtime_error("unsupported network architecture: with a Frequency position encoding (configs/nerf/frequency.json) the HIP path implements MLPs of 128 or 256 "
			                         "neurons with 1 or more hidden layers, a 16-wide density output, up to 40 position and 4 direction frequencies");
		}
	} else
	if (d.n_levels != N_LEVELS || d.n_features_per_level != N_FEATURES || d.n_neurons != MLP_WIDTH || d.n_hidden_density > 1 ||
	    d.n_hidden_rgb > 1 + (uint32_t)MAX_RGB_MID || d.density_out_dims != 16 || (d.n_hidden_density == 0 && d.n_hidden_rgb != 0)) {
		throw std::runtime_error("unsupported network architecture: the HIP path is specialised for configs/nerf/base.json and its variants "
		                         "(HashGrid 8 levels x 4 features; density MLP 64x1 hidden -> 16 with an rgb MLP 64 wide of 0 to 3 hidden layers -- "
		                         "base, base_0layer .. base_3layer --, or both heads without a hidden layer -- linear.json)");
	}
	if (!wide && ((d.log2_hashmap_size > 28 && d.log2_hashmap_size != 31) || d.base_resolution == 0 || !(d.per_level_scale > 0.f))) throw std::runtime_error("invalid hash grid configuration");
	if (d.aabb_scale == 0 || (d.aabb_scale & (d.aabb_scale - 1)) != 0) throw std::runtime_error("NeRF dataset's `aabb_scale` must be a power of two"); // testbed_nerf.cu:2707
	if (d.aabb_scale > (1u << (NERF_CASCADES - 1))) throw std::runtime_error("NeRF dataset must have `aabb_scale <= 128`"); // :2711-2718

	ModelParams M{};
	uint32_t total_entries = 0;
	if (!wide) build_levels(d, M.levels, &total_entries);
	const WideShapes ws = wide_shapes(d);
	const uint32_t enc_dims = wide ? ws.enc_dims : d.n_levels * d.n_features_per_level;
	const uint64_t nd = mlp_n_params(enc_dims, d.n_neurons, d.n_hidden_density, d.density_out_dims);
	// (grid models: the rgb input is 16 + 16 wide under either alignment; the output is padded to the rgb network's -- 8 rows for a CutlassMLP)
	const uint64_t nr = wide ? mlp_n_params(ws.rgb_in, d.n_neurons, d.n_hidden_rgb, ws.rgb_out) : mlp_n_params(d.density_out_dims + 16u, d.n_neurons, d.n_hidden_rgb, ws.rgb_out);
	const uint64_t ng = wide ? 0 : (uint64_t)total_entries * d.n_features_per_level;
	if (d.n_params != nd + nr + ng || !d.params_fp16) {
		throw std::runtime_error("parameter count mismatch: snapshot has " + std::to_string(d.n_params) + ", network needs " + std::to_string(nd + nr + ng));
	}
	uint32_t max_cascade = 0;
	while ((1u << max_cascade) < d.aabb_scale) ++max_cascade; // testbed_nerf.cu:2729-2732
	const uint64_t n_grid_expected = (uint64_t)NERF_GRID_N_CELLS * (max_cascade + 1);
	if (d.n_density_grid != 0 && d.n_density_grid != n_grid_expected) throw std::runtime_error("Incompatible number of grid cascades."); // testbed.cu:5350

	if (ctx->device >= 0) NGP_HIP_CHECK(hipDeviceSynchronize()); // frames in flight read the tables about to be freed
	free_model(ctx);
	ctx->params.assign(d.params_fp16, d.params_fp16 + d.n_params);
	ctx->density_grid.assign(d.density_grid_fp16, d.density_grid_fp16 + d.n_density_grid);
	ctx->desc = d;
	ctx->desc.params_fp16 = nullptr;
	ctx->desc.density_grid_fp16 = nullptr;
	ctx->max_cascade = max_cascade;
	ctx->have_desc = true;
	{ // reset_network: m_rng = default_rng_t{m_seed}; density_grid_rng = default_rng_t{m_rng.next_uint()} (testbed.cu:3848-3861, m_seed = 1337)
		Pcg32 rng;
		rng.seed(1337u);
		Pcg32 grid_rng;
		grid_rng.seed(rng.next_uint());
		ctx->grid_rng_state = grid_rng.state;
		ctx->grid_rng_inc = grid_rng.inc;
		ctx->grid_ema_step = 0;
		ctx->grid_updates = 0;
	}
	if (ctx->device < 0) return; // host-only context: the model is parsed and validated, nothing can be rendered

	if (wide) {
		// every layer's weights as MFMA A fragments (wide_kernels.hip)
		std::vector<uint16_t> frags;
		WideModel& WM = M.wide;
		WM.width = d.n_neurons;
		WM.pos_freqs = d.pos_encoding == 1 ? d.pos_n_frequencies : 0u;
		WM.dir_freqs = d.dir_encoding == 1 ? d.dir_n_frequencies : 0u;
		WM.pos_identity = d.pos_encoding == 2 ? 1u : 0u;
		WM.dir_identity = d.dir_encoding == 2 ? 1u : 0u;
		WM.enc_dims = ws.enc_dims;
		WM.dir_dims = ws.dir_dims;
		WM.rgb_in = ws.rgb_in;
		WM.n_hidden_density = d.n_hidden_density;
		WM.n_hidden_rgb = d.n_hidden_rgb;
		const uint16_t* W = ctx->params.data();
		uint32_t l = 0;
		auto emit_mlp = [&](uint32_t n_in, uint32_t n_hidden, uint32_t n_out) {
			WM.layers[l++] = emit_wide_fragments(frags, W, d.n_neurons, n_in);
			W += (size_t)d.n_neurons * n_in;
			for (uint32_t k = 1; k < n_hidden; ++k) {
				WM.layers[l++] = emit_wide_fragments(frags, W, d.n_neurons, d.n_neurons);
				W += (size_t)d.n_neurons * d.n_neurons;
			}
			WM.layers[l++] = emit_wide_fragments(frags, W, n_out, d.n_neurons);
			W += (size_t)n_out * d.n_neurons;
		};
		emit_mlp(ws.enc_dims, d.n_hidden_density, d.density_out_dims);
		emit_mlp(ws.rgb_in, d.n_hidden_rgb, ws.rgb_out);
		if (d.n_hidden_density <= (uint32_t)WIDE_MAX_NORMALS_LAYERS) {
			// ERenderMode::Normals: the density network's hidden layers transposed (the backward pass of tcnn's input_gradient runs the same GEMM
			// kernels on them), zero rows beyond the encoding's width in layer 0, and row 0 of the output layer (the one-hot loss gradient's only row)
			const uint16_t* D = ctx->params.data();
			std::vector<uint16_t> wt((size_t)d.n_neurons * d.n_neurons);
			uint32_t n_in = ws.enc_dims;
			for (uint32_t k = 0; k < d.n_hidden_density; ++k) {
				std::fill(wt.begin(), wt.end(), (uint16_t)0);
				for (uint32_t o = 0; o < d.n_neurons; ++o)
					for (uint32_t i = 0; i < n_in && i < d.n_neurons; ++i) wt[(size_t)i * d.n_neurons + o] = D[(size_t)o * n_in + i];
				WM.layers_t[k] = emit_wide_fragments(frags, wt.data(), d.n_neurons, d.n_neurons);
				D += (size_t)d.n_neurons * n_in;
				n_in = d.n_neurons;
			}
			WM.out_row0_offset = (uint32_t)(frags.size() / 8);
			frags.insert(frags.end(), D, D + d.n_neurons); // (D now points at the output layer: row 0 = the density logit's weights)
		}
		NGP_HIP_CHECK(hipMalloc((void**)&ctx->d_wfrags, frags.size() * sizeof(uint16_t)));
		NGP_HIP_CHECK(hipMemcpy(ctx->d_wfrags, frags.data(), frags.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
		WM.frags = ctx->d_wfrags;
	} else {
	// grid table
	NGP_HIP_CHECK(hipMalloc(&ctx->d_params, ng * sizeof(uint16_t)));
	NGP_HIP_CHECK(hipMemcpy(ctx->d_params, ctx->params.data() + nd + nr, ng * sizeof(uint16_t), hipMemcpyHostToDevice));
	{
		std::vector<uint64_t> table;
		build_xor_layout(M.levels, d.n_levels, ctx->params.data() + nd + nr, table);
		NGP_HIP_CHECK(hipMalloc(&ctx->d_xgrid, table.size() * sizeof(uint64_t)));
		NGP_HIP_CHECK(hipMemcpy(ctx->d_xgrid, table.data(), table.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
		if (table.size() * sizeof(uint64_t) > 0x7FFFFFFFull || ng * sizeof(uint16_t) > 0x7FFFFFFFull) throw std::runtime_error("hash grid too large for 31-bit buffer-load offsets");
		M.xgrid_bytes = (uint32_t)(table.size() * sizeof(uint64_t));
		M.grid_bytes = (uint32_t)(ng * sizeof(uint16_t));
	}
	// weight fragments
	std::vector<uint16_t> frags((size_t)(N_FRAGS_MAX + N_NORMALS_FRAGS) * 64 * 8, 0); // (the Normals mode's four are permuted out of the forward ones on the device, below)
	const uint16_t* W = ctx->params.data();
	if (d.n_hidden_density == 0) {
		emit_fragments(frags, FRAG_D0, W, 16, 32); // configs/nerf/linear.json: the 16 x 32 output layer alone
	} else {
		emit_fragments(frags, FRAG_D0, W, 64, 32);
		emit_fragments(frags, FRAG_D1, W + 64 * 32, 16, 64);
	}
	const uint16_t* R = W + nd;
	const int rgb_mid = (int)d.n_hidden_rgb - 1; // 64x64 layers between the first and the output layer of the rgb head; -1: the output layer alone
	if (rgb_mid < 0) {
		emit_fragments(frags, FRAG_R0, R, (int)ws.rgb_out, 32);
	} else {
		emit_fragments(frags, FRAG_R0, R, 64, 32);
		for (int k = 0; k < rgb_mid; ++k) emit_fragments(frags, FRAG_R1 + 8 * k, R + 64 * 32 + (size_t)k * 64 * 64, 64, 64);
		emit_fragments(frags, FRAG_R1 + 8 * rgb_mid, R + 64 * 32 + (size_t)rgb_mid * 64 * 64, (int)ws.rgb_out, 64);
	}
	M.rgb_mid = rgb_mid;
	M.density_linear = d.n_hidden_density == 0 ? 1u : 0u;
	NGP_HIP_CHECK(hipMalloc((void**)&ctx->d_wfrags, frags.size() * sizeof(uint16_t)));
	NGP_HIP_CHECK(hipMemcpy(ctx->d_wfrags, frags.data(), frags.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
	launch_build_normals_fragments(ctx->d_wfrags, ctx->stream);
	}
	// occupancy: fp16 grid -> fp32 -> bitfield + mips on the device (K8/K9)
	const size_t bitfield_bytes = (size_t)NERF_GRID_N_CELLS / 8 * NERF_CASCADES;
	NGP_HIP_CHECK(hipMalloc((void**)&ctx->d_bitfield, bitfield_bytes));
	NGP_HIP_CHECK(hipMalloc((void**)&ctx->d_density_f32, n_grid_expected * sizeof(float)));
	NGP_HIP_CHECK(hipMalloc((void**)&ctx->d_partial, 256 * sizeof(double)));
	if (d.n_density_grid) {
		NGP_HIP_CHECK(hipMalloc((void**)&ctx->d_density_f16, d.n_density_grid * sizeof(uint16_t)));
		NGP_HIP_CHECK(hipMemcpy(ctx->d_density_f16, ctx->density_grid.data(), d.n_density_grid * sizeof(uint16_t), hipMemcpyHostToDevice));
	} else {
		// a snapshot whose grid was never populated renders as empty space (testbed.cu:5348-5351)
		NGP_HIP_CHECK(hipMemset(ctx->d_density_f32, 0, n_grid_expected * sizeof(float)));
	}
	launch_density_grid_to_bitfield(ctx->d_density_f16, (uint32_t)d.n_density_grid, max_cascade, ctx->d_density_f32, ctx->d_partial, ctx->d_bitfield,
	                                &ctx->bitfield_mean, ctx->stream);
	NGP_HIP_CHECK(hipMalloc((void**)&ctx->d_coarse, ((size_t)NERF_CASCADES * COARSE_WORDS_PER_MIP + NERF_CASCADES * 16) * sizeof(uint32_t)));
	launch_coarse_occupancy(ctx->d_bitfield, ctx->d_coarse, ctx->stream);
	NGP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	NGP_HIP_CHECK(hipGetLastError());

	M.grid = (const uint2*)ctx->d_params;
	M.xgrid = (const char*)ctx->d_xgrid;
	M.coarse = ctx->d_coarse;
	M.wfrags = wide ? nullptr : ctx->d_wfrags;
	M.bitfield = ctx->d_bitfield;
	for (int i = 0; i < 3; ++i) {
		M.aabb_min[i] = d.aabb_min[i];
		M.aabb_diag[i] = d.aabb_max[i] - d.aabb_min[i];
		M.raabb_min[i] = d.render_aabb_min[i];
		M.raabb_max[i] = d.render_aabb_max[i];
	}
	for (int i = 0; i < 9; ++i) M.r2l[i] = d.render_aabb_to_local[i];
	{
		const float ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
		M.r2l_identity = memcmp(M.r2l, ident, sizeof(ident)) == 0 ? 1u : 0u;
		M.diag_pow2 = 1u;
		for (int i = 0; i < 3; ++i) {
			int e;
			float m = frexpf(M.aabb_diag[i], &e);
			if (!(m == 0.5f) || e < -100 || e > 100) M.diag_pow2 = 0u; // power of two, comfortably inside the normal range
			M.aabb_inv_diag[i] = 1.0f / M.aabb_diag[i];
		}
	}
	M.max_cascade = max_cascade;
	M.cone_angle = d.cone_angle_constant;
	M.rgb_act = d.rgb_activation;
	M.density_act = d.density_activation;
	ctx->M = M;
	ctx->model_loaded = true;
	++ctx->model_generation;
	ctx->grid_generation = ctx->params_generation = 0;
}

// ------------------------------------------------------------------------------------------------ snapshot
void read_vec(const mj::Value& v, float* out, size_t n) {
	if (!v.is_array() || v.size() != n) throw std::runtime_error("snapshot: vector of unexpected size");
	for (size_t i = 0; i < n; ++i) out[i] = (float)v.at(i).num();
}
// tcnn vec_json.h: a tmat<T,N,M> is an array of M rows with N entries each; storage is column-major
void read_mat(const mj::Value& v, float* out, int n_cols, int n_rows) {
	if (!v.is_array() || (int)v.size() != n_rows) throw std::runtime_error("snapshot: matrix of unexpected size");
	for (int r = 0; r < n_rows; ++r) {
		const mj::Value& row = v.at((size_t)r);
		if (!row.is_array() || (int)row.size() != n_cols) throw std::runtime_error("snapshot: matrix of unexpected size");
		for (int c = 0; c < n_cols; ++c) out[c * n_rows + r] = (float)row.at((size_t)c).num();
	}
}
mj::Value write_vec(const float* v, size_t n) {
	mj::Value a = mj::Value::make_array();
	for (size_t i = 0; i < n; ++i) a.push(mj::Value::make_float(v[i]));
	return a;
}
mj::Value write_mat(const float* m, int n_cols, int n_rows) {
	mj::Value a = mj::Value::make_array();
	for (int r = 0; r < n_rows; ++r) {
		mj::Value row = mj::Value::make_array();
		for (int c = 0; c < n_cols; ++c) row.push(mj::Value::make_float(m[c * n_rows + r]));
		a.push(std::move(row));
	}
	return a;
}

// Lens <-> json, json_binding.h:37-93
void lens_from_json(const mj::Value& j, TrainingView& v) {
	auto num = [&](const char* k) { return (float)j.at(k).num(); };
	if (j.contains("k1")) {
		if (j.value("is_fisheye", false)) {
			v.lens_mode = NGP_LENS_OPENCV_FISHEYE;
			v.lens_params[0] = num("k1"); v.lens_params[1] = num("k2"); v.lens_params[2] = num("k3"); v.lens_params[3] = num("k4");
		} else {
			v.lens_mode = NGP_LENS_OPENCV;
			v.lens_params[0] = num("k1"); v.lens_params[1] = num("k2"); v.lens_params[2] = num("p1"); v.lens_params[3] = num("p2");
		}
	} else if (j.contains("ftheta_p0")) {
		v.lens_mode = NGP_LENS_FTHETA;
		const char* keys[7] = {"ftheta_p0", "ftheta_p1", "ftheta_p2", "ftheta_p3", "ftheta_p4", "w", "h"};
		for (int i = 0; i < 7; ++i) v.lens_params[i] = num(keys[i]);
	} else if (j.contains("latlong")) {
		v.lens_mode = NGP_LENS_LATLONG;
	} else if (j.contains("equirectangular")) {
		v.lens_mode = NGP_LENS_EQUIRECTANGULAR;
	} else {
		v.lens_mode = NGP_LENS_PERSPECTIVE;
	}
}
mj::Value lens_to_json(const TrainingView& v) {
	mj::Value j = mj::Value::make_object();
	auto put = [&](const char* k, float x) { j[k] = mj::Value::make_float(x); };
	if (v.lens_mode == NGP_LENS_OPENCV) {
		j["is_fisheye"] = mj::Value::make_bool(false);
		put("k1", v.lens_params[0]); put("k2", v.lens_params[1]); put("p1", v.lens_params[2]); put("p2", v.lens_params[3]);
	} else if (v.lens_mode == NGP_LENS_OPENCV_FISHEYE) {
		j["is_fisheye"] = mj::Value::make_bool(true);
		put("k1", v.lens_params[0]); put("k2", v.lens_params[1]); put("k3", v.lens_params[2]); put("k4", v.lens_params[3]);
	} else if (v.lens_mode == NGP_LENS_FTHETA) {
		const char* keys[7] = {"ftheta_p0", "ftheta_p1", "ftheta_p2", "ftheta_p3", "ftheta_p4", "w", "h"};
		for (int i = 0; i < 7; ++i) put(keys[i], v.lens_params[i]);
	} else if (v.lens_mode == NGP_LENS_LATLONG) {
		j["latlong"] = mj::Value::make_bool(true);
	} else if (v.lens_mode == NGP_LENS_EQUIRECTANGULAR) {
		j["equirectangular"] = mj::Value::make_bool(true);
	}
	return j;
}

// numbers out of untrusted files: a double that does not fit the integer type must not reach the cast (undefined behaviour); out-of-range values
// become ones that every later validation refuses
uint32_t to_u32(double v) { return v >= 0.0 && v < 4294967296.0 ? (uint32_t)v : 0xffffffffu; }
int to_int(double v) { return v > -2147483648.0 && v < 2147483648.0 ? (int)v : (v < 0.0 ? -2147483647 - 1 : 2147483647); }

void dataset_from_json(const mj::Value& j, Dataset& ds) { // json_binding.h:121-183
	const int64_t n_images = j.at("n_images").integer();
	if (n_images < 0 || !j.at("xforms").is_array() || (uint64_t)n_images != j.at("xforms").size()) throw std::runtime_error("snapshot dataset: n_images does not match the list of camera transforms");
	size_t n = (size_t)n_images;
	ds.views.assign(n, TrainingView{});
	for (size_t i = 0; i < n; ++i) {
		TrainingView& v = ds.views[i];
		v.principal_point[0] = v.principal_point[1] = 0.5f;
		v.focal_length[0] = v.focal_length[1] = 1000.f;
		v.resolution[0] = v.resolution[1] = 0;
		if (j.contains("principal_point")) read_vec(j.at("principal_point"), v.principal_point, 2);
		if (j.contains("focal_length")) read_vec(j.at("focal_length"), v.focal_length, 2);
		if (j.contains("image_resolution")) { float r[2]; read_vec(j.at("image_resolution"), r, 2); v.resolution[0] = to_int(r[0]); v.resolution[1] = to_int(r[1]); }
		read_mat(j.at("xforms").at(i).at("start"), v.xform.data(), 4, 3);
		if (j.contains("metadata")) {
			const mj::Value& ji = j.at("metadata").at(i);
			float r[2];
			read_vec(ji.at("resolution"), r, 2);
			v.resolution[0] = to_int(r[0]);
			v.resolution[1] = to_int(r[1]);
			read_vec(ji.at("focal_length"), v.focal_length, 2);
			read_vec(ji.at("principal_point"), v.principal_point, 2);
			if (ji.contains("lens")) lens_from_json(ji.at("lens"), v);
		}
		if (j.contains("paths") && i < j.at("paths").size()) v.path = j.at("paths").at(i).str();
	}
	const mj::Value& ra = j.at("render_aabb");
	read_vec(ra.at("min"), ds.render_aabb_min, 3);
	read_vec(ra.at("max"), ds.render_aabb_max, 3);
	ds.has_render_aabb = true;
	if (j.contains("render_aabb_to_local")) read_mat(j.at("render_aabb_to_local"), ds.render_aabb_to_local, 3, 3);
	read_vec(j.at("up"), ds.up, 3);
	read_vec(j.at("offset"), ds.offset, 3);
	ds.scale = (float)j.at("scale").num();
	ds.aabb_scale = (int)j.at("aabb_scale").integer();
	ds.from_mitsuba = j.at("from_mitsuba").boolean();
	ds.is_hdr = j.value("is_hdr", false);
	ds.n_extra_learnable_dims = to_int(j.value("n_extra_learnable_dims", 0.0));
}

mj::Value dataset_to_json(const Dataset& ds) { // json_binding.h:94-119
	mj::Value j = mj::Value::make_object();
	j["n_images"] = mj::Value::make_uint(ds.views.size());
	mj::Value paths = mj::Value::make_array(), metadata = mj::Value::make_array(), xforms = mj::Value::make_array();
	for (auto& v : ds.views) {
		paths.push(mj::Value::make_string(v.path));
		mj::Value m = mj::Value::make_object();
		m["focal_length"] = write_vec(v.focal_length, 2);
		m["lens"] = lens_to_json(v);
		m["principal_point"] = write_vec(v.principal_point, 2);
		float rs[4] = {0, 0, 0, 0};
		m["rolling_shutter"] = write_vec(rs, 4);
		mj::Value res = mj::Value::make_array();
		res.push(mj::Value::make_int(v.resolution[0]));
		res.push(mj::Value::make_int(v.resolution[1]));
		m["resolution"] = res;
		metadata.push(std::move(m));
		mj::Value x = mj::Value::make_object();
		x["start"] = write_mat(v.xform.data(), 4, 3);
		x["end"] = write_mat(v.xform.data(), 4, 3);
		xforms.push(std::move(x));
	}
	j["paths"] = paths;
	j["metadata"] = metadata;
	j["xforms"] = xforms;
	mj::Value ra = mj::Value::make_object();
	ra["min"] = write_vec(ds.render_aabb_min, 3);
	ra["max"] = write_vec(ds.render_aabb_max, 3);
	j["render_aabb"] = ra;
	j["render_aabb_to_local"] = write_mat(ds.render_aabb_to_local, 3, 3);
	j["up"] = write_vec(ds.up, 3);
	j["offset"] = write_vec(ds.offset, 3);
	mj::Value er = mj::Value::make_array();
	er.push(mj::Value::make_int(0));
	er.push(mj::Value::make_int(0));
	j["envmap_resolution"] = er;
	j["scale"] = mj::Value::make_float(ds.scale);
	j["aabb_scale"] = mj::Value::make_int(ds.aabb_scale);
	j["from_mitsuba"] = mj::Value::make_bool(ds.from_mitsuba);
	j["is_hdr"] = mj::Value::make_bool(ds.is_hdr);
	j["wants_importance_sampling"] = mj::Value::make_bool(true);
	j["n_extra_learnable_dims"] = mj::Value::make_int(ds.n_extra_learnable_dims);
	return j;
}

// Testbed::load_snapshot(nlohmann::json) (src/testbed.cu:5285-5463), Nerf mode, inference-relevant state
void load_snapshot_value(ngp_ctx* ctx, mj::Value root) {
	if (!root.contains("snapshot")) throw std::runtime_error("File does not contain a snapshot.");
	const mj::Value& snap = root.at("snapshot");
	if (snap.value("version", 0.0) < 1.0) throw std::runtime_error("Snapshot uses an old format and can not be loaded.");
	std::string mode = snap.value("mode", snap.contains("nerf") ? "nerf" : "none");
	for (auto& ch : mode) ch = (char)tolower(ch);
	if (mode != "nerf") throw std::runtime_error("Only NeRF snapshots are supported by this renderer (snapshot mode: " + mode + ").");
	if (snap.at("density_grid_size").integer() != (int64_t)NERF_GRIDSIZE) throw std::runtime_error("Incompatible grid size.");

	ngp_model_desc d{};
	const mj::Value& enc = root.at("encoding");
	const mj::Value& net = root.at("network");
	const mj::Value& rgb = root.at("rgb_network");
	std::string otype = enc.value("otype", "OneBlob");
	for (auto& ch : otype) ch = (char)tolower(ch);
	// tcnn GridEncoding: HashGrid, or DenseGrid (every level x + y*res + z*res^2, never hashed, not capped by a hash-map size) --
	// the latter is the former with a hash map that no level ever fills, which is how it is carried here (log2 = 31).
	// TiledGrid wraps coordinates per axis and drops axes whose stride exceeds the tile: not implemented.
	bool dense_grid = false;
	if (otype == "densegrid") dense_grid = true;
	else if (otype == "grid") {
		std::string gt = enc.value("type", "Hash");
		for (auto& ch : gt) ch = (char)tolower(ch);
		if (gt == "dense") dense_grid = true;
		else if (gt != "hash") throw std::runtime_error("unsupported grid type '" + gt + "' (Hash and Dense are implemented)");
	} else if (otype == "frequency") { // configs/nerf/frequency.json
		d.pos_encoding = 1;
		d.pos_n_frequencies = to_u32(enc.value("n_frequencies", 12.0));
	} else if (otype == "identity") { // configs/nerf/none.json: the position itself (tcnn Identity: in * scale + offset, padded with ones)
		if (enc.value("scale", 1.0) != 1.0 || enc.value("offset", 0.0) != 0.0) throw std::runtime_error("unsupported Identity encoding (scale 1, offset 0 are implemented)");
		d.pos_encoding = 2;
	} else if (otype != "hashgrid") {
		throw std::runtime_error("unsupported encoding '" + otype + "' (HashGrid, DenseGrid, Frequency and Identity are implemented)");
	}
	d.n_features_per_level = to_u32(enc.value("n_features_per_level", 2.0));
	d.n_levels = enc.contains("n_features") && enc.at("n_features").is_number() && enc.at("n_features").num() > 0 && d.n_features_per_level ? to_u32(enc.at("n_features").num()) / d.n_features_per_level
	                                                                                                                                       : to_u32(enc.value("n_levels", 16.0));
	d.log2_hashmap_size = dense_grid ? 31u : to_u32(enc.value("log2_hashmap_size", 15.0));
	d.base_resolution = to_u32(enc.value("base_resolution", 0.0));
	if (!d.base_resolution) d.base_resolution = d.log2_hashmap_size < 96u ? 1u << (d.log2_hashmap_size / 3) : 0u; // testbed.cu:3945-3949 (a larger value fails the validation below)

	const mj::Value& nerf = snap.at("nerf");
	Dataset ds;
	if (nerf.contains("dataset")) dataset_from_json(nerf.at("dataset"), ds);
	if (nerf.contains("aabb_scale")) ds.aabb_scale = (int)nerf.at("aabb_scale").integer();
	d.aabb_scale = (uint32_t)ds.aabb_scale;

	d.per_level_scale = (float)enc.value("per_level_scale", 0.0);
	if (d.pos_encoding >= 1) {
		d.n_levels = d.n_features_per_level = d.log2_hashmap_size = d.base_resolution = 0;
		d.per_level_scale = 0.0f;
	} else if (!(d.per_level_scale > 0.0f) && d.n_levels > 1) {
		// The fork derives it from m_geometry.nerf...aabb_scale, which is 1 in Nerf mode (testbed.cu:3959-3966).
		d.per_level_scale = std::exp(std::log(2048.0f * 1.0f / (float)d.base_resolution) / (float)(d.n_levels - 1));
	}
	auto fully_fused = [](const mj::Value& n) {
		std::string t = n.value("otype", "FullyFusedMLP");
		for (auto& ch : t) ch = (char)tolower(ch);
		return t == "fullyfusedmlp" || t == "megakernelmlp" || t == "cutlassmlp";
	};
	if (!fully_fused(net) || !fully_fused(rgb)) throw std::runtime_error("unsupported network otype");
	{ // encodings and the rgb network's input / output are padded to the networks' alignment: 16 for FullyFusedMLP, 8 for CutlassMLP
		// (nerf_network.h:81-100; the rgb network's own for its output). The implemented architectures use one kind for both networks.
		auto is_cutlass = [](const mj::Value& n) {
			std::string t = n.value("otype", "FullyFusedMLP");
			for (auto& ch : t) ch = (char)tolower(ch);
			return t == "cutlassmlp";
		};
		if (is_cutlass(net) != is_cutlass(rgb) && d.pos_encoding >= 1) throw std::runtime_error("unsupported network otype: density and rgb networks of different kinds");
		d.mlp_alignment = is_cutlass(rgb) ? 8u : 16u; // grid models (base_0layer.json mixes the kinds): the rgb network's, nerf_network.h:83
	}
	{ // what the kernels hard-wire beyond the shapes: ReLU hidden layers without an output activation, and a direction encoding of
	  // SphericalHarmonics degree 4 (bare, or first in a Composite whose remainder is Identity: configs/nerf/base.json). A snapshot
	  // with another choice has the same parameter count and would render silently wrong.
		auto lower = [](std::string t) { for (auto& ch : t) ch = (char)tolower(ch); return t; };
		for (const mj::Value* n : {&net, &rgb}) {
			if (lower(n->value("activation", "ReLU")) != "relu") throw std::runtime_error("unsupported network activation '" + n->value("activation", "ReLU") + "' (ReLU is implemented)");
			if (lower(n->value("output_activation", "None")) != "none") throw std::runtime_error("unsupported network output_activation '" + n->value("output_activation", "None") + "' (None is implemented)");
		}
		if (root.contains("dir_encoding")) {
			const mj::Value& de = root.at("dir_encoding");
			auto is_sh4 = [&](const mj::Value& e) { return lower(e.value("otype", "")) == "sphericalharmonics" && to_int(e.value("degree", 4.0)) == 4; };
			auto is_freq = [&](const mj::Value& e) { return d.pos_encoding >= 1 && lower(e.value("otype", "")) == "frequency"; };
			auto is_ident = [&](const mj::Value& e) { return d.pos_encoding >= 1 && lower(e.value("otype", "")) == "identity" && e.value("scale", 1.0) == 1.0 && e.value("offset", 0.0) == 0.0; };
			const mj::Value* first = &de;
			bool ok = is_sh4(de) || is_freq(de) || is_ident(de);
			if (!ok && lower(de.value("otype", "")) == "composite" && de.contains("nested") && de.at("nested").is_array() && de.at("nested").size() >= 1) {
				const mj::Value& nested = de.at("nested");
				first = &nested.at(0);
				ok = (is_sh4(*first) || is_freq(*first)) && (!first->contains("n_dims_to_encode") || first->at("n_dims_to_encode").integer() == 3);
				for (size_t i = 1; ok && i < nested.size(); ++i) ok = lower(nested.at(i).value("otype", "")) == "identity";
			}
			if (!ok) throw std::runtime_error("unsupported dir_encoding (SphericalHarmonics of degree 4 -- or Frequency beside a Frequency position encoding --, bare or first in a Composite with Identity for the extra dimensions, is implemented)");
			if (is_freq(*first)) {
				d.dir_encoding = 1;
				d.dir_n_frequencies = to_u32(first->value("n_frequencies", 12.0));
			} else if (first == &de && is_ident(de)) {
				d.dir_encoding = 2;
			}
		}
	}
	d.n_neurons = (uint32_t)net.at("n_neurons").integer();
	if ((uint32_t)rgb.at("n_neurons").integer() != d.n_neurons) throw std::runtime_error("density and rgb networks must have the same width");
	d.n_hidden_density = (uint32_t)net.at("n_hidden_layers").integer();
	d.n_hidden_rgb = (uint32_t)rgb.at("n_hidden_layers").integer();
	d.density_out_dims = to_u32(net.value("n_output_dims", 16.0));
	d.rgb_activation = ds.is_hdr ? NGP_ACT_EXPONENTIAL : NGP_ACT_LOGISTIC; // testbed_nerf.cu:2653
	d.density_activation = NGP_ACT_EXPONENTIAL;                           // nerf.h:151-152

	// m_aabb / m_render_aabb: load_nerf_post (testbed_nerf.cu:2720-2727), then the snapshot's own values (testbed.cu:5309,5422-5423)
	float half = 0.5f * (float)std::min<int>(1 << (NERF_CASCADES - 1), ds.aabb_scale);
	for (int i = 0; i < 3; ++i) {
		d.aabb_min[i] = 0.5f - half;
		d.aabb_max[i] = 0.5f + half;
		d.render_aabb_min[i] = d.aabb_min[i];
		d.render_aabb_max[i] = d.aabb_max[i];
	}
	for (int i = 0; i < 9; ++i) d.render_aabb_to_local[i] = ds.render_aabb_to_local[i];
	if (snap.contains("aabb")) {
		read_vec(snap.at("aabb").at("min"), d.aabb_min, 3);
		read_vec(snap.at("aabb").at("max"), d.aabb_max, 3);
	}
	if (snap.contains("render_aabb")) {
		read_vec(snap.at("render_aabb").at("min"), d.render_aabb_min, 3);
		read_vec(snap.at("render_aabb").at("max"), d.render_aabb_max, 3);
	}
	if (snap.contains("render_aabb_to_local")) read_mat(snap.at("render_aabb_to_local"), d.render_aabb_to_local, 3, 3);
	d.cone_angle_constant = ds.aabb_scale <= 1 ? 0.0f : (1.0f / 256.0f); // testbed_nerf.cu:2736
	d.linear_colors = 0;

	// Trainer::deserialize: params_binary (__half or float)
	const mj::Value& pb = snap.at("params_binary");
	if (pb.type != mj::Value::Binary) throw std::runtime_error("snapshot: params_binary is not binary");
	std::string ptype = snap.value("params_type", "__half");
	std::vector<uint16_t> params;
	if (ptype == "float") {
		size_t n = pb.s.size() / 4;
		params.resize(n);
		const float* src = (const float*)pb.s.data();
		for (size_t i = 0; i < n; ++i) params[i] = float_to_half(src[i]);
	} else {
		params.resize(pb.s.size() / 2);
		memcpy(params.data(), pb.s.data(), params.size() * 2);
	}
	if (snap.contains("n_params") && (uint64_t)snap.at("n_params").integer() != params.size()) throw std::runtime_error("snapshot: n_params does not match params_binary");
	d.params_fp16 = params.data();
	d.n_params = params.size();
	const mj::Value& gb = snap.at("density_grid_binary");
	if (gb.type != mj::Value::Binary) throw std::runtime_error("snapshot: density_grid_binary is not binary");
	d.density_grid_fp16 = (const uint16_t*)gb.s.data();
	d.n_density_grid = gb.s.size() / 2;

	set_model_impl(ctx, d);

	for (auto& v : ctx->dataset.views) // training images of the dataset being replaced
		if (v.d_pixels) (void)hipFree(v.d_pixels);
	if (ctx->train) ctx->train->images_dirty = true;
	ctx->dataset = ds;
	ctx->has_snapshot_camera = false;
	{ // src/testbed.cu:5395-5418
		ngp_session_state& st = ctx->session;
		st = ngp_session_state{};
		st.valid = 1;
		st.background_color[3] = 1.f;
		st.sun_dir[0] = st.sun_dir[1] = st.sun_dir[2] = 0.57735026f;
		st.up_dir[1] = 1.f;
		st.camera_scale = 1.5f;
		memcpy(st.up_dir, ds.up, sizeof(st.up_dir));
		if (snap.contains("background_color")) read_vec(snap.at("background_color"), st.background_color, 4);
		st.exposure = (float)snap.value("exposure", 0.0);
		if (snap.contains("sun_dir")) read_vec(snap.at("sun_dir"), st.sun_dir, 3);
		if (snap.contains("up_dir")) read_vec(snap.at("up_dir"), st.up_dir, 3);
		if (snap.contains("camera")) {
			const mj::Value& cam = snap.at("camera");
			st.camera_scale = (float)cam.value("scale", 1.5);
			st.aperture_size = (float)cam.value("aperture_size", 0.0);
			st.autofocus_depth = (float)cam.value("autofocus_depth", 0.0);
		}
	}
	if (snap.contains("camera")) {
		const mj::Value& cam = snap.at("camera");
		if (cam.contains("matrix")) {
			read_mat(cam.at("matrix"), ctx->snap_camera, 4, 3);
			ctx->has_snapshot_camera = true;
		}
		ctx->snap_fov_axis = (int32_t)cam.value("fov_axis", 1.0);
		if (cam.contains("relative_focal_length")) read_vec(cam.at("relative_focal_length"), ctx->snap_relative_focal_length, 2);
		if (cam.contains("screen_center")) read_vec(cam.at("screen_center"), ctx->snap_screen_center, 2);
		ctx->snap_zoom = (float)cam.value("zoom", 1.0);
	}
	// keep the network config (without the heavy binaries) for save_snapshot
	mj::Value cfg = mj::Value::make_object();
	for (auto& kv : root.obj)
		if (kv.first != "snapshot") cfg.set(kv.first, kv.second);
	ctx->config = std::move(cfg);
}

// ------------------------------------------------------------------------------------------------ transforms.json
// SI::natural::compare (dependencies/NaturalSort): digit runs compare by value
bool natural_less(const std::string& a, const std::string& b) {
	size_t i = 0, j = 0;
	while (i < a.size() && j < b.size()) {
		if (isdigit((unsigned char)a[i]) && isdigit((unsigned char)b[j])) {
			size_t i0 = i, j0 = j;
			while (i0 < a.size() && a[i0] == '0') ++i0;
			while (j0 < b.size() && b[j0] == '0') ++j0;
			size_t i1 = i0, j1 = j0;
			while (i1 < a.size() && isdigit((unsigned char)a[i1])) ++i1;
			while (j1 < b.size() && isdigit((unsigned char)b[j1])) ++j1;
			if (i1 - i0 != j1 - j0) return (i1 - i0) < (j1 - j0);
			int c = a.compare(i0, i1 - i0, b, j0, j1 - j0);
			if (c != 0) return c < 0;
			i = i1;
			j = j1;
		} else {
			if (a[i] != b[j]) return a[i] < b[j];
			++i;
			++j;
		}
	}
	return a.size() - i < b.size() - j;
}

float fov_to_focal_length(int resolution, float degrees) { return 0.5f * (float)resolution / tanf(0.5f * degrees * 3.14159265358979323846f / 180.0f); }

bool read_focal_length(const mj::Value& json, float* fl, const int* res) { // nerf_loader.cu:243-271
	auto read = [&](int resolution, const std::string& axis) -> float {
		if (json.contains(axis + "_fov")) return fov_to_focal_length(resolution, (float)json.at(axis + "_fov").num());
		if (json.contains("fl_" + axis)) return (float)json.at("fl_" + axis).num();
		if (json.contains("camera_angle_" + axis)) return fov_to_focal_length(resolution, (float)json.at("camera_angle_" + axis).num() * 180 / 3.14159265358979323846f);
		return 0.0f;
	};
	float x_fl = read(res[0], "x"), y_fl = read(res[1], "y");
	if (x_fl != 0) {
		fl[0] = fl[1] = x_fl;
		if (y_fl != 0) fl[1] = y_fl;
	} else if (y_fl != 0) {
		fl[0] = fl[1] = y_fl;
	} else {
		return false;
	}
	return true;
}

// NerfDataset::nerf_matrix_to_ngp (nerf_loader.h:101-120); m column-major 4x3 in place
void nerf_matrix_to_ngp(const Dataset& ds, float* m) {
	for (int r = 0; r < 3; ++r) {
		m[3 + r] *= -1.f;
		m[6 + r] *= -1.f;
		m[9 + r] = m[9 + r] * ds.scale + ds.offset[r];
	}
	if (ds.from_mitsuba) {
		for (int r = 0; r < 3; ++r) { m[0 + r] *= -1.f; m[6 + r] *= -1.f; }
	} else {
		for (int c = 0; c < 4; ++c) { // cycle rows xyz <- yzx
			float t = m[c * 3 + 0];
			m[c * 3 + 0] = m[c * 3 + 1];
			m[c * 3 + 1] = m[c * 3 + 2];
			m[c * 3 + 2] = t;
		}
	}
}

// ngp::load_nerf (src/nerf_loader.cu:273-743), camera metadata only: images are not decoded on the inference path,
// so per-view resolution comes from the json's "w"/"h".
void load_training_data_impl(ngp_ctx* ctx, const std::string& path) {
	std::vector<std::string> json_paths;
	if (is_directory(path)) {
		DIR* dir = opendir(path.c_str());
		if (!dir) throw std::runtime_error("cannot open directory '" + path + "'");
		while (dirent* e = readdir(dir)) {
			std::string name = e->d_name;
			if (ends_with_ci(name, ".json")) json_paths.push_back(path + "/" + name);
		}
		closedir(dir);
		std::sort(json_paths.begin(), json_paths.end());
	} else if (ends_with_ci(path, ".json")) {
		json_paths.push_back(path);
	} else {
		throw std::runtime_error("NeRF data path must either be a json file or a directory containing json files.");
	}
	if (json_paths.empty()) throw std::runtime_error("Cannot load NeRF data from an empty set of paths.");

	Dataset ds;
	ds.scale = 0.33f; // NERF_SCALE, nerf_loader.h:29
	ds.offset[0] = ds.offset[1] = ds.offset[2] = 0.5f;
	for (const std::string& jp : json_paths) {
		mj::Value json = mj::parse_json(read_file(jp));
		if (!json.contains("frames") || !json.at("frames").is_array()) continue;
		const std::string base = parent_dir(jp);
		std::vector<mj::Value> frames = json.at("frames").arr;
		std::stable_sort(frames.begin(), frames.end(), [](const mj::Value& a, const mj::Value& b) { return natural_less(a.at("file_path").str(), b.at("file_path").str()); });
		if (json.contains("n_frames")) frames.resize(std::min(frames.size(), (size_t)json.at("n_frames").integer()));
		auto resolve = [&](const std::string& local) {
			std::string p = (!local.empty() && local[0] == '/') ? local : base + "/" + local;
			if (p.find_last_of('.') == std::string::npos || p.find_last_of('.') < p.find_last_of('/')) {
				for (const char* ext : {"png", "jpg", "jpeg", "bmp", "gif", "tga", "pic", "pnm", "psd", "exr"})
					if (file_exists(p + "." + ext)) return p + "." + ext;
			}
			return p;
		};
		if (!frames.empty() && frames[0].contains("sharpness")) { // blurry / missing frames are dropped, nerf_loader.cu:364-388
			float thresh = (float)json.value("sharpness_discard_threshold", 0.0);
			std::vector<mj::Value> kept;
			for (int i = 0; i < (int)frames.size(); ++i) {
				float mean = 0.f;
				int s = std::max(0, i - 3), e = std::min(i + 3, (int)frames.size() - 1);
				for (int j = s; j < e; ++j) mean += (float)frames[j].value("sharpness", 1.0);
				mean /= (float)(e - s);
				if (file_exists(resolve(frames[i].at("file_path").str())) && (float)frames[i].value("sharpness", 1.0) > thresh * mean) kept.push_back(frames[i]);
			}
			frames.swap(kept);
		}
		if (json.contains("normal_mts_args")) ds.from_mitsuba = true;
		if (ds.from_mitsuba) { ds.scale = 0.66f; ds.offset[0] = ds.offset[1] = ds.offset[2] = 0.25f * ds.scale; }
		if (json.contains("render_aabb")) {
			read_vec(json.at("render_aabb").at(0), ds.render_aabb_min, 3);
			read_vec(json.at("render_aabb").at(1), ds.render_aabb_max, 3);
			ds.has_render_aabb = true;
		}
		if (json.contains("scale")) ds.scale = (float)json.at("scale").num();
		if (json.contains("n_extra_learnable_dims")) ds.n_extra_learnable_dims = (int)json.at("n_extra_learnable_dims").integer();
		if (json.contains("aabb_scale")) ds.aabb_scale = (int)json.at("aabb_scale").integer();
		if (json.contains("offset")) {
			const mj::Value& o = json.at("offset");
			if (o.is_array()) read_vec(o, ds.offset, 3);
			else ds.offset[0] = ds.offset[1] = ds.offset[2] = (float)o.num();
		}
		if (json.contains("aabb")) { // nerf_loader.cu:503-509
			const mj::Value& a = json.at("aabb");
			float lo[3], hi[3];
			read_vec(a.at(0), lo, 3);
			read_vec(a.at(1), hi, 3);
			float len = std::max(0.000001f, std::max(std::max(std::abs(hi[0] - lo[0]), std::abs(hi[1] - lo[1])), std::abs(hi[2] - lo[2])));
			ds.scale = 1.f / len;
			for (int i = 0; i < 3; ++i) ds.offset[i] = ((hi[i] + lo[i]) * 0.5f) * -ds.scale + 0.5f;
		}
		if (json.contains("up")) {
			ds.up[0] = (float)json.at("up").at(1).num();
			ds.up[1] = (float)json.at("up").at(2).num();
			ds.up[2] = (float)json.at("up").at(0).num();
		}
		float pp[2] = {0.5f, 0.5f};
		auto read_pp = [](const mj::Value& j, float* pp) {
			if (j.contains("cx")) pp[0] = (float)j.at("cx").num() / (float)j.at("w").num();
			if (j.contains("cy")) pp[1] = (float)j.at("cy").num() / (float)j.at("h").num();
		};
		read_pp(json, pp);
		// read_lens (src/nerf_loader.cu:175-240): OpenCV parameters switch the mode on when one of them is non-zero; an
		// outer (file-level) lens is kept unless the frame names its own
		auto read_lens = [](const mj::Value& j, TrainingView& v) {
			int mode = NGP_LENS_PERSPECTIVE;
			const int opencv_mode = j.value("is_fisheye", false) ? NGP_LENS_OPENCV_FISHEYE : NGP_LENS_OPENCV;
			auto rd = [&](const char* name, int idx) {
				if (j.contains(name)) {
					v.lens_params[idx] = (float)j.at(name).num();
					if (v.lens_params[idx] != 0.f) mode = opencv_mode;
				}
			};
			rd("k1", 0); rd("k2", 1); rd("k3", 2); rd("k4", 3);
			rd("p1", 2); rd("p2", 3);
			if (j.contains("ftheta_p0")) {
				const char* keys[7] = {"ftheta_p0", "ftheta_p1", "ftheta_p2", "ftheta_p3", "ftheta_p4", "w", "h"};
				for (int i = 0; i < 7; ++i) v.lens_params[i] = (float)j.at(keys[i]).num();
				mode = NGP_LENS_FTHETA;
			}
			if (j.contains("latlong")) mode = NGP_LENS_LATLONG;
			if (j.contains("equirectangular")) mode = NGP_LENS_EQUIRECTANGULAR;
			if (mode != NGP_LENS_PERSPECTIVE) v.lens_mode = mode;
		};
		TrainingView file_lens;
		read_lens(json, file_lens);
		for (const mj::Value& frame : frames) {
			TrainingView v;
			v.lens_mode = file_lens.lens_mode;
			memcpy(v.lens_params, file_lens.lens_params, sizeof(v.lens_params));
			read_lens(frame, v);
			v.path = frame.at("file_path").str();
			std::replace(v.path.begin(), v.path.end(), '\\', '/');
			v.abs_path = resolve(v.path);
			v.white_transparent = json.value("white_transparent", false);
			v.black_transparent = json.value("black_transparent", false);
			v.resolution[0] = to_int(frame.contains("w") ? frame.at("w").num() : json.value("w", 0.0));
			v.resolution[1] = to_int(frame.contains("h") ? frame.at("h").num() : json.value("h", 0.0));
			if ((v.resolution[0] <= 0 || v.resolution[1] <= 0) && !probe_image_size(v.abs_path, v.resolution[0], v.resolution[1]))
				throw std::runtime_error("transforms.json gives no 'w' / 'h' and the resolution of '" + v.abs_path + "' cannot be read (PNG or JPEG expected)");
			v.focal_length[0] = v.focal_length[1] = 1000.f;
			bool got = read_focal_length(json, v.focal_length, v.resolution);
			got |= read_focal_length(frame, v.focal_length, v.resolution);
			if (!got) throw std::runtime_error("Couldn't read fov.");
			const mj::Value& mat = frame.contains("transform_matrix_start") ? frame.at("transform_matrix_start") : frame.at("transform_matrix");
			for (int m = 0; m < 3; ++m)
				for (int n = 0; n < 4; ++n) v.xform[(size_t)n * 3 + m] = (float)mat.at((size_t)m).at((size_t)n).num();
			v.principal_point[0] = pp[0];
			v.principal_point[1] = pp[1];
			read_pp(frame, v.principal_point);
			nerf_matrix_to_ngp(ds, v.xform.data());
			ds.views.push_back(std::move(v));
		}
	}
	for (auto& v : ctx->dataset.views) // training images of the dataset being replaced
		if (v.d_pixels) (void)hipFree(v.d_pixels);
	if (ctx->train) ctx->train->images_dirty = true;
	ctx->dataset = std::move(ds);
	ctx->data_path = path;
}

// ------------------------------------------------------------------------------------------------ frame
// The scheduling knobs of the persistent render kernel (FrameParams::tune). They never change results except
// block_jumps (0 = the reference's voxel-by-voxel walk through empty space), but values outside these ranges would
// leave a wave spinning in fused_body's loop (a refill threshold above 64 never refills, zero march steps never
// advance a ray): a GPU hang, not an error. Hence one gate for every way of setting them.
void validate_schedule(const int32_t* t, int n) {
	static const struct { const char* name; int lo, hi; } range[8] = {{"refill_min", 16, 64}, {"skip_steps", 1, 64},  {"go_min", 1, 64},      {"max_stall", 0, 64},
	                                                                  {"k_busy", 1, 8},       {"k_drain", 1, 8},      {"block_jumps", 0, 1}, {"share", 0, 1}};
	if (n < 0 || n > 8) throw std::runtime_error("schedule: at most 8 knobs");
	for (int i = 0; i < n; ++i)
		if (t[i] < range[i].lo || t[i] > range[i].hi)
			throw std::runtime_error(std::string("schedule knob ") + range[i].name + " = " + std::to_string(t[i]) + " outside [" + std::to_string(range[i].lo) + ", " + std::to_string(range[i].hi) + "]");
}
// NGP_TUNE="refill_min,skip_steps,..." (experiments: tools/sweep_tune.sh): read ONCE, at context creation
void schedule_from_env(ngp_ctx* ctx) {
	const char* t = getenv("NGP_TUNE");
	if (!t || !*t) return;
	int32_t v[8];
	memcpy(v, ctx->tune, sizeof(v));
	int n = sscanf(t, "%d,%d,%d,%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6], &v[7]);
	if (n <= 0) throw std::runtime_error("NGP_TUNE: expected a comma-separated list of integers");
	validate_schedule(v, n);
	memcpy(ctx->tune, v, sizeof(v));
}

void ensure_frame_buffers(ngp_ctx* ctx, size_t n_pixels) {
	if (!ctx->d_sync) {
		NGP_HIP_CHECK(hipMalloc(&ctx->d_sync, ngp_ctx::SLOT_BYTES * ngp_ctx::HISTORY));
		NGP_HIP_CHECK(hipMemset(ctx->d_sync, 0, ngp_ctx::SLOT_BYTES * ngp_ctx::HISTORY));
		for (int i = 0; i < ngp_ctx::HISTORY; ++i) {
			NGP_HIP_CHECK(hipEventCreate(&ctx->ev_frame0[i]));
			NGP_HIP_CHECK(hipEventCreate(&ctx->ev_frame1[i]));
			NGP_HIP_CHECK(hipEventCreate(&ctx->ev_kern0[i]));
			NGP_HIP_CHECK(hipEventCreate(&ctx->ev_kern1[i]));
		}
	}
	if (n_pixels <= ctx->n_pixels_alloc) return;
	if (ctx->d_frame) (void)hipFree(ctx->d_frame);
	if (ctx->d_depth) (void)hipFree(ctx->d_depth);
	if (ctx->d_accum) (void)hipFree(ctx->d_accum);
	if (ctx->d_rgba) (void)hipFree(ctx->d_rgba);
	ctx->n_pixels_alloc = 0;
	NGP_HIP_CHECK(hipMalloc((void**)&ctx->d_frame, n_pixels * sizeof(float4)));
	NGP_HIP_CHECK(hipMalloc((void**)&ctx->d_depth, n_pixels * sizeof(float)));
	NGP_HIP_CHECK(hipMalloc((void**)&ctx->d_accum, n_pixels * sizeof(float4)));
	NGP_HIP_CHECK(hipMalloc((void**)&ctx->d_rgba, n_pixels * sizeof(float4)));
	ctx->n_pixels_alloc = n_pixels;
}

CameraParams make_camera_params(const ngp_camera& cam, uint32_t spp_index) {
	CameraParams C{};
	memcpy(C.m, cam.matrix, sizeof(C.m));
	C.width = cam.width;
	C.height = cam.height;
	C.focal[0] = cam.focal_length[0];
	C.focal[1] = cam.focal_length[1];
	C.screen_center[0] = cam.screen_center[0];
	C.screen_center[1] = cam.screen_center[1];
	C.spp = spp_index;
	C.near_distance = cam.near_distance;
	if (cam.lens_mode < 0 || cam.lens_mode > NGP_LENS_EQUIRECTANGULAR) throw std::runtime_error("unknown lens mode (Perspective, OpenCV, FTheta, LatLong, OpenCVFisheye, Equirectangular)");
	C.lens_mode = cam.lens_mode;
	C.aperture_size = cam.focus_z < 0.f ? 0.f : cam.aperture_size; // plane_z < 0 switches the aperture off (src/testbed_nerf.cu:1462-1464)
	C.focus_z = cam.focus_z;
	if (C.aperture_size != 0.f && !(C.focus_z > 0.f)) throw std::runtime_error("depth of field needs a positive focus distance");
	memcpy(C.lens_params, cam.lens_params, sizeof(C.lens_params));
	// camera_matrix1 + rolling shutter: a frame is "moving" only when camera1 differs from camera0 or the per-pixel time is not the
	// whole-frame constant the quaternion round trip would leave unchanged anyway
	C.moving = cam.has_matrix1 && memcmp(cam.matrix, cam.matrix1, sizeof(cam.matrix)) != 0 ? 1 : 0;
	memcpy(C.m1, cam.has_matrix1 ? cam.matrix1 : cam.matrix, sizeof(C.m1));
	memcpy(C.rolling_shutter, cam.rolling_shutter, sizeof(C.rolling_shutter));
	if (!cam.has_matrix1) { C.rolling_shutter[0] = C.rolling_shutter[1] = C.rolling_shutter[2] = 0.f; C.rolling_shutter[3] = 1.f; }
	ld_random_pixel_offset(cam.snap_to_pixel_centers ? 0u : spp_index, C.pixel_offset);
	return C;
}

// Testbed::render_frame (src/testbed.cu:4694-4721) for opts->spp samples; the final image lands in d_rgba_out.
void render_frames(ngp_ctx* ctx, const ngp_camera& cam, const ngp_render_opts& opts, float4* d_rgba_out, float* d_depth_out, hipStream_t stream) {
	if (ctx->device < 0) throw std::runtime_error("this context has no HIP device (host-only); rendering needs an MI355X -- there is no CPU fallback");
	if (!ctx->model_loaded && !(opts.testbed_mode == NGP_MODE_GEOMETRY && !ctx->meshes.empty())) throw std::runtime_error("No network available."); // testbed.cu:4735-4738
	ngp::sync_inference_model(ctx);
	if (cam.width <= 0 || cam.height <= 0 || cam.width > 65536 || cam.height > 65536) throw std::runtime_error("invalid render resolution"); // (tile counts stay inside 32 bits)
	if (opts.render_mode < NGP_RENDER_SHADE || opts.render_mode > NGP_RENDER_NORMALS) throw std::runtime_error("render modes implemented: Shade, ShadeEnvMap, ShadeGridEnvMap, AO, Normals, Positions, Depth, Cost");
	const bool gbuffer_mode = (opts.render_mode >= NGP_RENDER_AO && opts.render_mode <= NGP_RENDER_COST) || opts.render_mode == NGP_RENDER_NORMALS;
	if (gbuffer_mode && opts.testbed_mode == NGP_MODE_GEOMETRY) throw std::runtime_error("the G-buffer render modes (AO, Normals, Positions, Depth, Cost) apply to NeRF mode");
	if (opts.render_mode == NGP_RENDER_NORMALS && ctx->model_loaded && ctx->M.wide.width && (!ctx->M.wide.layers_t[0].n_mtiles || ctx->M.wide.enc_dims > ctx->M.wide.width))
		throw std::runtime_error("render_mode Normals on a Frequency / Identity-encoding model: implemented for up to 8 hidden density layers and an encoding no wider than the network");
	const uint32_t shard_count = opts.shard_count ? opts.shard_count : 1u;
	if (opts.shard_index >= shard_count) throw std::runtime_error("shard_index out of range");
	const uint32_t tiles_total = (uint32_t)((cam.width + 7) / 8) * (uint32_t)((cam.height + 7) / 8);
	const uint32_t local_tiles = tiles_total > opts.shard_index ? (tiles_total - opts.shard_index + shard_count - 1) / shard_count : 0;
	// frame-buffer extent: the whole image, or only this shard's tiles in tile-packed order
	const size_t n_pixels = opts.packed_output ? (size_t)local_tiles * 64 : (size_t)cam.width * cam.height;
	ensure_frame_buffers(ctx, n_pixels ? n_pixels : 1);
	const int spp = opts.spp > 0 ? opts.spp : 1;

	FrameParams F{};
	F.frame_buffer = ctx->d_frame;
	F.depth_buffer = d_depth_out ? d_depth_out : ctx->d_depth;
	const int slot = (int)(ctx->n_calls % ngp_ctx::HISTORY);
	// Call k reuses the queue word, exit counter and accumulators of call k - HISTORY, which may have been issued on another stream and,
	// in an unsynchronised loop of short frames, may still be running: two live launches on one slot would deal tiles twice and zero
	// the slot under each other. Order this call's stream behind that frame's end (a device-side wait, no host stall; free when the
	// old frame is long done, which is the usual case).
	if (ctx->n_calls >= (uint64_t)ngp_ctx::HISTORY) NGP_HIP_CHECK(hipStreamWaitEvent(stream, ctx->ev_frame1[slot], 0));
	ctx->bind_slot(F, slot); // every call has its own queue word and counters: frames on different streams may overlap
	if (shard_count > 1) F.xqueue = nullptr; // per-XCD bands pay for a whole frame (+1.4 %); a rank's interleaved share is too small for them (N = 4: -3 %, N = 8: -6 %)
	F.tiles_x = (uint32_t)(cam.width + 7) / 8;
	F.tiles_y = (uint32_t)(cam.height + 7) / 8;
	const uint32_t n_tiles = F.tiles_x * F.tiles_y;
	F.shard_index = opts.shard_index;
	F.shard_count = shard_count;
	F.n_local_tiles = n_tiles > opts.shard_index ? (n_tiles - opts.shard_index + shard_count - 1) / shard_count : 0;
	F.min_transmittance = opts.min_transmittance;
	F.linear_colors = ctx->desc.linear_colors;
	F.packed = opts.packed_output ? 1 : 0;
	F.prof = nullptr;
	memcpy(F.tune, ctx->tune, sizeof(F.tune));
	if (getenv("NGP_PROFILE_SECTIONS")) { // diagnostic: per-section cycle sums of the fused kernel, printed by ngp_get_render_stats
		if (!ctx->d_prof) NGP_HIP_CHECK(hipMalloc((void**)&ctx->d_prof, 1024));
		NGP_HIP_CHECK(hipMemsetAsync(ctx->d_prof, 0, 1024, stream));
		F.prof = ctx->d_prof;
		F.prof_level = atoi(getenv("NGP_PROFILE_SECTIONS"));
		if (const char* e = getenv("NGP_PROFILE_TRACE")) { // timelines of every stride-th working wave (tools/wave_trace.py)
			const int stride = atoi(e);
			if (stride > 0) {
				const size_t words = 16 + (size_t)ngp_ctx::TRACE_WAVES * 16 + (size_t)ngp_ctx::TRACE_WAVES * ngp_ctx::TRACE_ITERS * 16;
				if (!ctx->d_trace) NGP_HIP_CHECK(hipMalloc((void**)&ctx->d_trace, words * sizeof(uint32_t)));
				NGP_HIP_CHECK(hipMemsetAsync(ctx->d_trace, 0, words * sizeof(uint32_t), stream));
				F.trace = ctx->d_trace;
				F.trace_stride = (uint32_t)stride;
				F.trace_cap_waves = ngp_ctx::TRACE_WAVES;
				F.trace_cap_iters = ngp_ctx::TRACE_ITERS;
			}
		}
	}
	const bool geometry = opts.testbed_mode == NGP_MODE_GEOMETRY;
	const bool have_meshes = geometry && !ctx->meshes.empty();
	if (have_meshes && cam.has_matrix1 && memcmp(cam.matrix, cam.matrix1, sizeof(cam.matrix)) != 0) throw std::runtime_error("a moving camera (matrix1 / rolling shutter) renders NeRF mode");
	F.depth_test = geometry ? 1 : 0; // shade_kernel_nerf_geometry
	ModelParams M = ctx->M;
	if (have_meshes) { // load_scene sets m_render_aabb to the inflated mesh bb (testbed_geometry_training.cu:3185-3189)
		for (int i = 0; i < 3; ++i) { M.raabb_min[i] = ctx->mesh_scene.scene_min[i]; M.raabb_max[i] = ctx->mesh_scene.scene_max[i]; }
		const float ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
		memcpy(M.r2l, ident, sizeof(ident));
		M.r2l_identity = 1u;
	}

	// 1 spp and no mesh pass: the fused kernel writes finished pixels (clear + accumulate + tonemap folded in) straight
	// into the caller's image -- one 32-byte memset and one launch per frame
	// (a rank's share in image layout keeps the general path: the other ranks' pixels must read as an empty frame)
	F.direct = (spp == 1 && !have_meshes && !geometry && (shard_count == 1 || opts.packed_output)) ? 1 : 0;
	F.to_srgb = opts.to_srgb;
	F.render_mode = opts.render_mode == NGP_RENDER_SHADE_GRID_ENVMAP ? NGP_RENDER_SHADE_ENVMAP : opts.render_mode; // (the NeRF pass treats both like Shade)
	F.color_space = opts.color_space;
	if (opts.color_space != 0 && opts.color_space != 1) throw std::runtime_error("color_space: 0 (Linear) or 1 (SRGB)");
	F.depth_scale = opts.depth_scale != 0.f ? opts.depth_scale : 1.0f / 0.33f;
	memcpy(F.background, opts.background, sizeof(F.background));
	F.exposure_scale = powf(2.0f, opts.exposure);
	if (ctx->d_bg_envmap) {
		if (geometry) throw std::runtime_error("an environment map applies to NeRF mode (in the reference the Geometry-mode NeRF pass would paint it over the meshes, src/testbed_geometry_training.cu:1993-1995): clear it with ngp_set_envmap(ctx, 0, 0, NULL)");
		F.envmap = ctx->d_bg_envmap;
		F.env_w = ctx->bg_env_w;
		F.env_h = ctx->bg_env_h;
	}
	{ // a render box inside the outermost cascade's cube never puts a ray outside the occupancy grid (kernel selection)
		const float h = 0.5f * (float)(1u << M.max_cascade);
		bool inside = M.r2l_identity != 0;
		for (int i = 0; i < 3; ++i) inside = inside && M.raabb_min[i] >= 0.5f - h && M.raabb_max[i] <= 0.5f + h;
		F.outside_possible = inside ? 0 : 1;
	}
	ngp::order_after_model(ctx, stream); // a training step / grid refresh / peer copy that updated what this frame reads (a device-side wait)
	NGP_HIP_CHECK(hipEventRecord(ctx->ev_frame0[slot], stream));
	if (F.direct) {
		F.frame_buffer = d_rgba_out;
		CameraParams C = make_camera_params(cam, cam.spp_index);
		NGP_HIP_CHECK(hipEventRecord(ctx->ev_kern0[slot], stream));
		launch_render_nerf(M, C, F, ctx->n_cus, stream);
		NGP_HIP_CHECK(hipEventRecord(ctx->ev_kern1[slot], stream));
	}
	if (!F.direct && ctx->n_calls > 0 && (ctx->streams_mixed || (ctx->last_stream && ctx->last_stream != stream))) {
		// the general path goes through the context's own frame / accumulate buffers: frames on other streams must have
		// left them (only direct-output frames may overlap each other). Several may still be in flight, one event would
		// not cover them all: wait for the device.
		NGP_HIP_CHECK(hipDeviceSynchronize());
		ctx->streams_mixed = false;
	}
	if (ctx->last_stream && ctx->last_stream != stream) ctx->streams_mixed = true;
	for (int s = 0; s < spp && !F.direct; ++s) {
		CameraParams C = make_camera_params(cam, cam.spp_index + (uint32_t)s);
		// CudaRenderBufferView::clear (src/render_buffer.cu:603-607)
		NGP_HIP_CHECK(hipMemsetAsync(ctx->d_frame, 0, n_pixels * sizeof(float4), stream));
		NGP_HIP_CHECK(hipMemsetAsync(F.depth_buffer, 0, n_pixels * sizeof(float), stream));
		F.add_results = s > 0 ? 1 : 0; // the call's counters are the sums over its samples per pixel
		const bool last = s == spp - 1;
		if (have_meshes) {
			IrradianceMap I{};
			if (opts.render_mode == NGP_RENDER_SHADE_ENVMAP) {
				if (!ctx->d_irradiance || ctx->env_probe.mode == 3) throw std::runtime_error("render_mode ShadeEnvMap needs ngp_compute_envmap first");
				I = ngp::irradiance_map_of(ctx);
			} else if (opts.render_mode == NGP_RENDER_SHADE_GRID_ENVMAP) {
				if (!ctx->d_irradiance || ctx->env_probe.mode != 3) throw std::runtime_error("render_mode ShadeGridEnvMap needs ngp_compute_envmap_grid first");
				I = ngp::irradiance_map_of(ctx);
			}
			launch_render_mesh(ctx->mesh_scene, ctx->shade, I, C, ctx->d_frame, F.depth_buffer, F.shard_index, F.shard_count, F.packed, stream);
		}
		if (last) NGP_HIP_CHECK(hipEventRecord(ctx->ev_kern0[slot], stream));
		if (ctx->model_loaded) launch_render_nerf(M, C, F, ctx->n_cus, stream); // persistent grid sized by the launcher
		else if (s == 0) NGP_HIP_CHECK(hipMemsetAsync(F.results, 0, 24, stream)); // meshes only: no NeRF launch reports counters
		if (last) NGP_HIP_CHECK(hipEventRecord(ctx->ev_kern1[slot], stream));
		launch_accumulate_tonemap((uint32_t)n_pixels, ctx->d_frame, ctx->d_accum, (float)s, opts.background, opts.exposure, opts.to_srgb, opts.color_space, last ? d_rgba_out : nullptr, stream);
	}
	NGP_HIP_CHECK(hipEventRecord(ctx->ev_frame1[slot], stream));
	NGP_HIP_CHECK(hipGetLastError());
	ctx->last_stream = stream;
	ctx->hist_n_rays[slot] = (uint64_t)F.n_local_tiles * 64u * (uint64_t)spp;
	ctx->last_was_multi = false;
	++ctx->n_calls;
}

} // namespace

namespace ngp {
void ensure_sync_buffers(ngp_ctx* ctx) { ensure_frame_buffers(ctx, 0); }
void render_frames_on(ngp_ctx* ctx, const ngp_camera& cam, const ngp_render_opts& opts, float4* d_rgba, float* d_depth, hipStream_t stream) { render_frames(ctx, cam, opts, d_rgba, d_depth, stream); }
void ensure_frame_buffers_for(ngp_ctx* ctx, size_t n_pixels) { ensure_frame_buffers(ctx, n_pixels); }
void install_model(ngp_ctx* ctx, const ngp_model_desc& d) { set_model_impl(ctx, d); }
uint16_t half_from_float(float f) { return float_to_half(f); }
void update_density_grid_device(ngp_ctx* ctx, float decay, uint32_t n_uniform, uint32_t n_nonuniform, uint32_t n_iterations) {
	const uint32_t n_cascades = ctx->max_cascade + 1;
	const uint32_t n_elements = NERF_GRID_N_CELLS * n_cascades;
	hipStream_t stream = ctx->stream;
	ensure_frame_buffers(ctx, 0);
	order_after_frames(ctx, stream); // frames in flight on ANY stream read the bitfield and its summaries: the refresh waits for them on the device
	if (!ctx->d_density_tmp) NGP_HIP_CHECK(hipMalloc((void**)&ctx->d_density_tmp, (size_t)n_elements * sizeof(float)));
	Pcg32 rng;
	rng.state = ctx->grid_rng_state;
	rng.inc = ctx->grid_rng_inc;
	for (uint32_t it = 0; it < n_iterations; ++it) {
		uint32_t nu = n_uniform, nn = n_nonuniform;
		if (nu == 0 && nn == 0) { // training_prep_nerf's schedule (src/testbed_nerf.cu:3441-3445)
			if (ctx->grid_updates < 256) nu = NERF_GRID_N_CELLS * n_cascades;
			else nu = nn = NERF_GRID_N_CELLS / 4 * n_cascades;
		}
		NGP_HIP_CHECK(hipMemsetAsync(ctx->d_density_tmp, 0, (size_t)n_elements * sizeof(float), stream));
		if (ctx->M.wide.width) { // a network without a hash grid: positions -> NerfNetwork::inference (wide_kernels.hip) -> splat
			const size_t n_max = std::max(nu, nn);
			if (n_max > ctx->grid_scratch_samples) {
				if (ctx->d_grid_scratch) (void)hipFree(ctx->d_grid_scratch);
				ctx->d_grid_scratch = nullptr;
				ctx->grid_scratch_samples = 0;
				NGP_HIP_CHECK(hipMalloc(&ctx->d_grid_scratch, n_max * 24));
				ctx->grid_scratch_samples = n_max;
			}
			float* d_pos = (float*)ctx->d_grid_scratch;
			uint32_t* d_cell = (uint32_t*)((char*)ctx->d_grid_scratch + n_max * 12);
			uint16_t* d_out = (uint16_t*)((char*)ctx->d_grid_scratch + n_max * 16);
			launch_density_grid_update_wide(ctx->M, nu, rng, ctx->grid_ema_step, n_cascades, -0.01f, ctx->d_density_f32, ctx->d_density_tmp, d_pos, d_cell, d_out, ctx->n_cus, stream);
			rng.advance();
			launch_density_grid_update_wide(ctx->M, nn, rng, ctx->grid_ema_step, n_cascades, 0.01f, ctx->d_density_f32, ctx->d_density_tmp, d_pos, d_cell, d_out, ctx->n_cus, stream);
			rng.advance();
		} else {
			launch_density_grid_update(ctx->M, nu, rng, ctx->grid_ema_step, n_cascades, -0.01f, ctx->d_density_f32, ctx->d_density_tmp, stream);
			rng.advance();
			launch_density_grid_update(ctx->M, nn, rng, ctx->grid_ema_step, n_cascades, 0.01f /* NERF_MIN_OPTICAL_THICKNESS */, ctx->d_density_f32, ctx->d_density_tmp, stream);
			rng.advance();
		}
		launch_density_grid_ema(n_elements, decay, ctx->d_density_f32, ctx->d_density_tmp, stream);
		++ctx->grid_ema_step;
		++ctx->grid_updates;
	}
	ctx->grid_rng_state = rng.state;
	ctx->grid_rng_inc = rng.inc;
	// update_density_grid_mean_and_bitfield (:2863-2880) + the block summaries the march reads
	launch_density_grid_to_bitfield(nullptr, 0, ctx->max_cascade, ctx->d_density_f32, ctx->d_partial, ctx->d_bitfield, &ctx->bitfield_mean, stream);
	launch_coarse_occupancy(ctx->d_bitfield, ctx->d_coarse, stream);
	mark_model_updated(ctx, stream);
	ctx->density_grid_host_dirty = true;
	++ctx->grid_generation;
}
// keep the snapshot copy (fp16, as the reference serialises it) in step
void refresh_density_grid_host(ngp_ctx* ctx) {
	if (!ctx->density_grid_host_dirty || ctx->device < 0 || !ctx->model_loaded) return;
	const uint32_t n_elements = NERF_GRID_N_CELLS * (ctx->max_cascade + 1);
	std::vector<float> grid(n_elements);
	NGP_HIP_CHECK(hipMemcpyAsync(grid.data(), ctx->d_density_f32, (size_t)n_elements * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
	NGP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	NGP_HIP_CHECK(hipGetLastError());
	ctx->density_grid.resize(n_elements);
	for (uint32_t i = 0; i < n_elements; ++i) ctx->density_grid[i] = float_to_half(grid[i]);
	ctx->desc.n_density_grid = n_elements;
	ctx->density_grid_host_dirty = false;
}
void load_snapshot_path(ngp_ctx* ctx, const std::string& p) {
	std::string data = read_file(p);
	bool compressed = ends_with_ci(p, ".ingp"); // testbed.cu:262-266
	if (!compressed && !ends_with_ci(p, ".msgpack")) throw std::runtime_error("snapshot must be a .msgpack or .ingp file");
	if (compressed) data = inflate_all(data.data(), data.size());
	load_snapshot_value(ctx, mj::MsgpackReader((const uint8_t*)data.data(), data.size()).parse());
}
} // namespace ngp

// ================================================================================================== C ABI
extern "C" {

const char* ngp_version(void) { return "ngp_hip 0.1 (gfx950)"; }

ngp_ctx* ngp_create(int device) {
	if (device == -1) { // host-only context: file formats and validation, no rendering (there is no CPU renderer)
		ngp_ctx* ctx = new ngp_ctx();
		ctx->device = -1;
		return ctx;
	}
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return nullptr;
	if (hipSetDevice(device) != hipSuccess) return nullptr;
	ngp_ctx* ctx = new ngp_ctx();
	ctx->device = device;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->n_cus = prop.multiProcessorCount;
	if (hipStreamCreate(&ctx->stream) != hipSuccess) {
		delete ctx;
		return nullptr;
	}
	try {
		schedule_from_env(ctx);
	} catch (const std::exception& e) {
		fprintf(stderr, "ngp_create: %s\n", e.what());
		(void)hipStreamDestroy(ctx->stream);
		delete ctx;
		return nullptr;
	}
	return ctx;
}

void ngp_destroy(ngp_ctx* ctx) {
	if (!ctx) return;
	for (ngp_ctx* p : ctx->peers) ngp_destroy(p);
	ctx->peers.clear();
	if (ctx->device >= 0) ngp::free_multi_buffers(ctx);
	if (ctx->device < 0) { delete ctx->train; delete ctx; return; }
	(void)hipSetDevice(ctx->device);
	if (ctx->last_stream) (void)hipStreamSynchronize(ctx->last_stream);
	free_model(ctx);
	delete ctx->train;
	for (auto& v : ctx->dataset.views)
		if (v.d_pixels) (void)hipFree(v.d_pixels);
	for (auto& m : ctx->meshes) {
		if (m.d_tris) (void)hipFree(m.d_tris);
		if (m.d_nodes) (void)hipFree(m.d_nodes);
	}
	if (ctx->d_meshrefs) (void)hipFree(ctx->d_meshrefs);
	if (ctx->d_envmap) (void)hipFree(ctx->d_envmap);
	if (ctx->d_irradiance) (void)hipFree(ctx->d_irradiance);
	if (ctx->d_bg_envmap) (void)hipFree(ctx->d_bg_envmap);
	if (ctx->d_frame) (void)hipFree(ctx->d_frame);
	if (ctx->d_depth) (void)hipFree(ctx->d_depth);
	if (ctx->d_accum) (void)hipFree(ctx->d_accum);
	if (ctx->d_rgba) (void)hipFree(ctx->d_rgba);
	if (ctx->d_sync) (void)hipFree(ctx->d_sync);
	if (ctx->d_trace) (void)hipFree(ctx->d_trace);
	if (ctx->d_grid_scratch) (void)hipFree(ctx->d_grid_scratch);
	for (int i = 0; i < ngp_ctx::HISTORY; ++i) {
		if (ctx->ev_frame0[i]) (void)hipEventDestroy(ctx->ev_frame0[i]);
		if (ctx->ev_frame1[i]) (void)hipEventDestroy(ctx->ev_frame1[i]);
		if (ctx->ev_kern0[i]) (void)hipEventDestroy(ctx->ev_kern0[i]);
		if (ctx->ev_kern1[i]) (void)hipEventDestroy(ctx->ev_kern1[i]);
	}
	if (ctx->ev_model) (void)hipEventDestroy(ctx->ev_model);
	if (ctx->ev_synced) (void)hipEventDestroy(ctx->ev_synced);
	if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
	delete ctx;
}

const char* ngp_last_error(const ngp_ctx* ctx) { return ctx ? ctx->error.c_str() : "no HIP device / invalid context"; }

int ngp_set_model(ngp_ctx* ctx, const ngp_model_desc* desc) {
	return guarded(ctx, [&] {
		if (!desc) throw std::runtime_error("null model descriptor");
		set_model_impl(ctx, *desc);
		ctx->config = mj::Value();
	});
}

int ngp_load_snapshot(ngp_ctx* ctx, const void* bytes, size_t n_bytes, int is_compressed) {
	return guarded(ctx, [&] {
		if (!bytes || !n_bytes) throw std::runtime_error("empty snapshot");
		if (is_compressed) {
			std::string raw = inflate_all(bytes, n_bytes);
			load_snapshot_value(ctx, mj::MsgpackReader((const uint8_t*)raw.data(), raw.size()).parse());
		} else {
			load_snapshot_value(ctx, mj::MsgpackReader((const uint8_t*)bytes, n_bytes).parse());
		}
	});
}

int ngp_load_snapshot_file(ngp_ctx* ctx, const char* path) {
	return guarded(ctx, [&] {
		if (!path) throw std::runtime_error("null path");
		ngp::load_snapshot_path(ctx, path);
	});
}

int ngp_save_snapshot_file(ngp_ctx* ctx, const char* path, int compress) {
	return guarded(ctx, [&] {
		if (!ctx->have_desc) throw std::runtime_error("no model to save");
		if (!path) throw std::runtime_error("null path");
		ngp::sync_host_params(ctx);
		ngp::refresh_density_grid_host(ctx);
		const ngp_model_desc& d = ctx->desc;
		mj::Value root = ctx->config.is_object() ? ctx->config : mj::Value::make_object();
		if (!root.contains("encoding") && d.pos_encoding >= 1) { // configs/nerf/frequency.json, none.json
			mj::Value e = mj::Value::make_object();
			e["otype"] = mj::Value::make_string(d.pos_encoding == 2 ? "Identity" : "Frequency");
			if (d.pos_encoding == 1) e["n_frequencies"] = mj::Value::make_uint(d.pos_n_frequencies);
			root["encoding"] = e;
			auto mlp = [&](uint32_t hidden) {
				mj::Value n = mj::Value::make_object();
				n["otype"] = mj::Value::make_string(d.mlp_alignment == 8 ? "CutlassMLP" : "FullyFusedMLP");
				n["activation"] = mj::Value::make_string("ReLU");
				n["output_activation"] = mj::Value::make_string("None");
				n["n_neurons"] = mj::Value::make_uint(d.n_neurons);
				n["n_hidden_layers"] = mj::Value::make_uint(hidden);
				return n;
			};
			root["network"] = mlp(d.n_hidden_density);
			root["rgb_network"] = mlp(d.n_hidden_rgb);
			mj::Value de = mj::Value::make_object();
			if (d.dir_encoding == 1) {
				de["otype"] = mj::Value::make_string("Frequency");
				de["n_frequencies"] = mj::Value::make_uint(d.dir_n_frequencies);
			} else if (d.dir_encoding == 2) {
				de["otype"] = mj::Value::make_string("Identity");
			} else {
				de["otype"] = mj::Value::make_string("SphericalHarmonics");
				de["degree"] = mj::Value::make_uint(4);
			}
			root["dir_encoding"] = de;
		}
		if (!root.contains("encoding")) {
			mj::Value e = mj::Value::make_object();
			e["otype"] = mj::Value::make_string(d.log2_hashmap_size == 31 ? "DenseGrid" : "HashGrid");
			e["n_levels"] = mj::Value::make_uint(d.n_levels);
			e["n_features_per_level"] = mj::Value::make_uint(d.n_features_per_level);
			if (d.log2_hashmap_size != 31) e["log2_hashmap_size"] = mj::Value::make_uint(d.log2_hashmap_size);
			e["base_resolution"] = mj::Value::make_uint(d.base_resolution);
			root["encoding"] = e;
			auto mlp = [&](uint32_t hidden, bool cutlass) {
				mj::Value n = mj::Value::make_object();
				n["otype"] = mj::Value::make_string(cutlass ? "CutlassMLP" : "FullyFusedMLP");
				n["activation"] = mj::Value::make_string("ReLU");
				n["output_activation"] = mj::Value::make_string("None");
				n["n_neurons"] = mj::Value::make_uint(d.n_neurons);
				n["n_hidden_layers"] = mj::Value::make_uint(hidden);
				return n;
			};
			root["network"] = mlp(d.n_hidden_density, d.n_hidden_density == 0);
			root["rgb_network"] = mlp(d.n_hidden_rgb, d.mlp_alignment == 8);
			mj::Value de = mj::Value::make_object();
			de["otype"] = mj::Value::make_string("Composite");
			mj::Value nested = mj::Value::make_array();
			mj::Value sh = mj::Value::make_object();
			sh["n_dims_to_encode"] = mj::Value::make_uint(3);
			sh["otype"] = mj::Value::make_string("SphericalHarmonics");
			sh["degree"] = mj::Value::make_uint(4);
			nested.push(sh);
			mj::Value id = mj::Value::make_object();
			id["otype"] = mj::Value::make_string("Identity");
			nested.push(id);
			de["nested"] = nested;
			root["dir_encoding"] = de;
		}
		if (d.pos_encoding == 0) root["encoding"]["per_level_scale"] = mj::Value::make_float(d.per_level_scale);
		mj::Value snap = mj::Value::make_object();
		snap["n_params"] = mj::Value::make_uint(ctx->params.size());
		snap["params_type"] = mj::Value::make_string("__half");
		snap["params_binary"] = mj::Value::make_binary(ctx->params.data(), ctx->params.size() * 2);
		snap["version"] = mj::Value::make_uint(1);
		snap["mode"] = mj::Value::make_string("nerf");
		snap["density_grid_size"] = mj::Value::make_uint(NERF_GRIDSIZE);
		snap["density_grid_binary"] = mj::Value::make_binary(ctx->density_grid.data(), ctx->density_grid.size() * 2);
		mj::Value nerf = mj::Value::make_object();
		nerf["aabb_scale"] = mj::Value::make_uint(d.aabb_scale);
		mj::Value rgbc = mj::Value::make_object();
		rgbc["rays_per_batch"] = mj::Value::make_uint(4096);
		rgbc["measured_batch_size"] = mj::Value::make_uint(0);
		rgbc["measured_batch_size_before_compaction"] = mj::Value::make_uint(0);
		nerf["rgb"] = rgbc;
		Dataset ds = ctx->dataset;
		ds.aabb_scale = (int)d.aabb_scale;
		if (!ds.has_render_aabb) {
			for (int i = 0; i < 3; ++i) { ds.render_aabb_min[i] = d.render_aabb_min[i]; ds.render_aabb_max[i] = d.render_aabb_max[i]; }
		}
		nerf["dataset"] = dataset_to_json(ds);
		snap["nerf"] = nerf;
		snap["training_step"] = mj::Value::make_uint(0);
		snap["loss"] = mj::Value::make_float(0.0);
		mj::Value aabb = mj::Value::make_object();
		aabb["min"] = write_vec(d.aabb_min, 3);
		aabb["max"] = write_vec(d.aabb_max, 3);
		snap["aabb"] = aabb;
		mj::Value raabb = mj::Value::make_object();
		raabb["min"] = write_vec(d.render_aabb_min, 3);
		raabb["max"] = write_vec(d.render_aabb_max, 3);
		snap["render_aabb"] = raabb;
		snap["render_aabb_to_local"] = write_mat(d.render_aabb_to_local, 3, 3);
		snap["up_dir"] = write_vec(ctx->session.valid ? ctx->session.up_dir : ctx->dataset.up, 3);
		if (ctx->session.valid) { // src/testbed.cu:5249-5251
			snap["sun_dir"] = write_vec(ctx->session.sun_dir, 3);
			snap["exposure"] = mj::Value::make_float(ctx->session.exposure);
			snap["background_color"] = write_vec(ctx->session.background_color, 4);
		}
		if (ctx->has_snapshot_camera) {
			mj::Value cam = mj::Value::make_object();
			if (ctx->session.valid) {
				cam["scale"] = mj::Value::make_float(ctx->session.camera_scale);
				cam["aperture_size"] = mj::Value::make_float(ctx->session.aperture_size);
				cam["autofocus_depth"] = mj::Value::make_float(ctx->session.autofocus_depth);
			}
			cam["matrix"] = write_mat(ctx->snap_camera, 4, 3);
			cam["fov_axis"] = mj::Value::make_int(ctx->snap_fov_axis);
			cam["relative_focal_length"] = write_vec(ctx->snap_relative_focal_length, 2);
			cam["screen_center"] = write_vec(ctx->snap_screen_center, 2);
			cam["zoom"] = mj::Value::make_float(ctx->snap_zoom);
			snap["camera"] = cam;
		}
		root["snapshot"] = snap;
		mj::MsgpackWriter w;
		w.write(root);
		std::string p = path;
		std::ofstream f(p, std::ios::out | std::ios::binary);
		if (!f) throw std::runtime_error("cannot write '" + p + "'");
		if (ends_with_ci(p, ".ingp")) {
			std::string z = deflate_gzip(w.out, compress ? Z_DEFAULT_COMPRESSION : Z_NO_COMPRESSION);
			f.write(z.data(), (std::streamsize)z.size());
		} else {
			f.write(w.out.data(), (std::streamsize)w.out.size());
		}
	});
}

int ngp_get_model(ngp_ctx* ctx, ngp_model_desc* out) {
	if (!ctx || !out || !ctx->have_desc) return -1;
	return guarded(ctx, [&] {
		if (ctx->device >= 0) { // what was trained / refreshed on the device since is part of "the model as currently loaded"
			ngp::sync_host_params(ctx);
			ngp::refresh_density_grid_host(ctx);
		}
		*out = ctx->desc;
		out->params_fp16 = ctx->params.data();
		out->n_params = ctx->params.size();
		out->density_grid_fp16 = ctx->density_grid.data();
		out->n_density_grid = ctx->density_grid.size();
	});
}

int ngp_get_session_state(const ngp_ctx* ctx, ngp_session_state* out) {
	if (!ctx || !out) return -1;
	*out = ctx->session;
	return 0;
}

int ngp_set_session_state(ngp_ctx* ctx, const ngp_session_state* state, const float* matrix12, const float* rfl2, int32_t fov_axis, const float* sc2, float zoom) {
	if (!ctx || !state) return -1;
	ctx->session = *state;
	ctx->session.valid = 1;
	if (matrix12) {
		memcpy(ctx->snap_camera, matrix12, sizeof(float) * 12);
		if (rfl2) memcpy(ctx->snap_relative_focal_length, rfl2, sizeof(float) * 2);
		if (sc2) memcpy(ctx->snap_screen_center, sc2, sizeof(float) * 2);
		ctx->snap_fov_axis = fov_axis;
		ctx->snap_zoom = zoom;
		ctx->has_snapshot_camera = true;
	}
	return 0;
}

int ngp_get_snapshot_camera(const ngp_ctx* ctx, float* matrix12, float* rfl2, int32_t* fov_axis, float* sc2, float* zoom) {
	if (!ctx || !ctx->has_snapshot_camera) return -1;
	if (matrix12) memcpy(matrix12, ctx->snap_camera, sizeof(float) * 12);
	if (rfl2) memcpy(rfl2, ctx->snap_relative_focal_length, sizeof(float) * 2);
	if (fov_axis) *fov_axis = ctx->snap_fov_axis;
	if (sc2) memcpy(sc2, ctx->snap_screen_center, sizeof(float) * 2);
	if (zoom) *zoom = ctx->snap_zoom;
	return 0;
}

int ngp_load_training_data(ngp_ctx* ctx, const char* path) {
	if (!ctx) return -1;
	try { // needs no device
		if (!path) throw std::runtime_error("null path");
		load_training_data_impl(ctx, path);
		ctx->error.clear();
		return 0;
	} catch (const std::exception& e) {
		ctx->error = e.what();
		return -1;
	}
}

int ngp_get_training_view_lens(const ngp_ctx* ctx, int view, int32_t* lens_mode, float* lens_params7) {
	if (!ctx || view < 0 || (size_t)view >= ctx->dataset.views.size()) return 1;
	const TrainingView& v = ctx->dataset.views[(size_t)view];
	if (lens_mode) *lens_mode = v.lens_mode;
	if (lens_params7) memcpy(lens_params7, v.lens_params, sizeof(v.lens_params));
	return 0;
}

int ngp_n_training_views(const ngp_ctx* ctx) { return ctx ? (int)ctx->dataset.views.size() : -1; }

int ngp_get_training_view(const ngp_ctx* ctx, int view, float* matrix12, int32_t* res2, float* fl2, float* pp2) {
	if (!ctx || view < 0 || view >= (int)ctx->dataset.views.size()) return -1;
	const TrainingView& v = ctx->dataset.views[(size_t)view];
	if (matrix12) memcpy(matrix12, v.xform.data(), sizeof(float) * 12);
	if (res2) { res2[0] = v.resolution[0]; res2[1] = v.resolution[1]; }
	if (fl2) { fl2[0] = v.focal_length[0]; fl2[1] = v.focal_length[1]; }
	if (pp2) { pp2[0] = v.principal_point[0]; pp2[1] = v.principal_point[1]; }
	return 0;
}

int ngp_get_dataset_info(const ngp_ctx* ctx, int32_t* aabb_scale, float* scale, float* offset3, int32_t* is_hdr) {
	if (!ctx) return -1;
	if (aabb_scale) *aabb_scale = ctx->dataset.aabb_scale;
	if (scale) *scale = ctx->dataset.scale;
	if (offset3) memcpy(offset3, ctx->dataset.offset, sizeof(float) * 3);
	if (is_hdr) *is_hdr = ctx->dataset.is_hdr ? 1 : 0;
	return 0;
}

int ngp_render_device(ngp_ctx* ctx, const ngp_camera* cam, const ngp_render_opts* opts, void* d_rgba, void* d_depth, void* stream) {
	return guarded(ctx, [&] {
		if (!cam || !opts || !d_rgba) throw std::runtime_error("null argument");
		if (!ctx->peers.empty() && opts->shard_count <= 1) { // a multi-device context: every device renders its tiles, device 0 assembles
			ngp::render_frames_multi(ctx, *cam, *opts, (float4*)d_rgba, (float*)d_depth, stream ? (hipStream_t)stream : ctx->stream);
			return;
		}
		render_frames(ctx, *cam, *opts, (float4*)d_rgba, (float*)d_depth, stream ? (hipStream_t)stream : ctx->stream);
	});
}

uint32_t ngp_packed_tiles(int32_t width, int32_t height, uint32_t shard_index, uint32_t shard_count) {
	if (width <= 0 || height <= 0 || shard_count == 0 || shard_index >= shard_count) return 0;
	const uint32_t tiles_total = (uint32_t)((width + 7) / 8) * (uint32_t)((height + 7) / 8);
	return tiles_total > shard_index ? (tiles_total - shard_index + shard_count - 1) / shard_count : 0;
}

namespace {
void* pinned_device_alias(const void* host, size_t bytes); // below, with the pool
}

int ngp_render(ngp_ctx* ctx, const ngp_camera* cam, const ngp_render_opts* opts, float* rgba_out, float* depth_out) {
	return guarded(ctx, [&] {
		if (!cam || !opts) throw std::runtime_error("null argument");
		if (cam->width <= 0 || cam->height <= 0 || cam->width > 65536 || cam->height > 65536) throw std::runtime_error("invalid render resolution");
		if (!rgba_out) throw std::runtime_error("null argument");
		if (opts->packed_output) throw std::runtime_error("packed_output is for ngp_render_device (GPU-resident tiles); ngp_render returns images");
		if (ctx->device < 0) throw std::runtime_error("this context has no HIP device (host-only); rendering needs an MI355X -- there is no CPU fallback");
		const size_t n_pixels = (size_t)cam->width * cam->height;
		ensure_frame_buffers(ctx, n_pixels);
		// A destination from ngp_host_alloc is page-locked AND mapped into the device's address space: the image is write-only for
		// every kernel that produces it (the fused kernel's direct output, accumulate + tonemap, the multi-device tile scatter), so
		// they write it over the link while they run and no copy follows the frame -- the 33 MB of a 1080p frame would take the
		// copy engine 0.8 ms after a 2.4 ms render. NGP_HOST_DIRECT=0 restores render-then-copy (A/B measurements).
		static const bool host_direct = []() { const char* e = getenv("NGP_HOST_DIRECT"); return !e || atoi(e) != 0; }();
		float4* d_image = host_direct ? (float4*)pinned_device_alias(rgba_out, n_pixels * sizeof(float4)) : nullptr;
		float4* d_target = d_image ? d_image : ctx->d_rgba;
		if (!ctx->peers.empty() && opts->shard_count <= 1) ngp::render_frames_multi(ctx, *cam, *opts, d_target, ctx->d_depth, ctx->stream);
		else render_frames(ctx, *cam, *opts, d_target, nullptr, ctx->stream);
		// (otherwise: one DMA at the link's rate into page-locked memory, a staged copy into ordinary memory)
		if (!d_image) NGP_HIP_CHECK(hipMemcpyAsync(rgba_out, ctx->d_rgba, n_pixels * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
		if (depth_out) NGP_HIP_CHECK(hipMemcpyAsync(depth_out, ctx->d_depth, n_pixels * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
		NGP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	});
}

// ---- page-locked host images. Testbed::render_to_cpu returns a fresh numpy array per call (src/python_api.cu:124-202); a
// pageable destination makes the runtime stage the 33 MB of a 1080p frame through its own bounce buffers (1.7 ms, against
// 2.1 ms of rendering). Buffers from this pool are pinned once and recycled by size, so a binding can hand out "fresh"
// arrays that the copy engine writes directly.
namespace {
struct HostPool {
	std::mutex mu;
	std::multimap<size_t, void*> free_list;
	std::map<void*, size_t> live;
	~HostPool() { // (process exit: the runtime may already be gone; leave the pages to the OS)
	}
} g_host_pool;
constexpr size_t HOST_POOL_KEEP = 8; // buffers kept for reuse per process
} // namespace

namespace {
// the device-side address of [host, host + bytes) if that range lies inside a live page-locked buffer of the pool, else nullptr
void* pinned_device_alias(const void* host, size_t bytes) {
	std::lock_guard<std::mutex> lock(g_host_pool.mu);
	auto it = g_host_pool.live.upper_bound(const_cast<void*>(host));
	if (it == g_host_pool.live.begin()) return nullptr;
	--it;
	const char* base = (const char*)it->first;
	if (it->second == 0 || (const char*)host < base || (const char*)host + bytes > base + it->second) return nullptr;
	void* dev = nullptr;
	if (hipHostGetDevicePointer(&dev, it->first, 0) != hipSuccess || !dev) {
		(void)hipGetLastError();
		return nullptr;
	}
	return (char*)dev + ((const char*)host - base);
}
} // namespace

void* ngp_host_alloc(size_t bytes) {
	if (bytes == 0) return nullptr;
	const size_t rounded = (bytes + 4095) & ~(size_t)4095;
	{
		std::lock_guard<std::mutex> lock(g_host_pool.mu);
		auto it = g_host_pool.free_list.find(rounded);
		if (it != g_host_pool.free_list.end()) {
			void* p = it->second;
			g_host_pool.free_list.erase(it);
			g_host_pool.live[p] = rounded;
			return p;
		}
	}
	void* p = nullptr;
	if (hipHostMalloc(&p, rounded, hipHostMallocDefault) != hipSuccess) { // no device / no pinned memory left: plain memory still works, only slower
		(void)hipGetLastError();
		p = aligned_alloc(4096, rounded);
		if (!p) return nullptr;
		std::lock_guard<std::mutex> lock(g_host_pool.mu);
		g_host_pool.live[p] = 0; // 0: malloc'ed, never pooled
		return p;
	}
	std::lock_guard<std::mutex> lock(g_host_pool.mu);
	g_host_pool.live[p] = rounded;
	return p;
}

void ngp_host_free(void* p) {
	if (!p) return;
	size_t size = 0;
	bool release = false;
	{
		std::lock_guard<std::mutex> lock(g_host_pool.mu);
		auto it = g_host_pool.live.find(p);
		if (it == g_host_pool.live.end()) return; // not ours
		size = it->second;
		g_host_pool.live.erase(it);
		if (size != 0 && g_host_pool.free_list.size() < HOST_POOL_KEEP) g_host_pool.free_list.emplace(size, p);
		else release = true;
	}
	if (!release) return;
	if (size == 0) free(p);
	else (void)hipHostFree(p);
}

static void read_history_slot(ngp_ctx* ctx, uint64_t call, ngp_render_stats* out) {
	const int slot = (int)(call % ngp_ctx::HISTORY);
	unsigned long long c[4];
	NGP_HIP_CHECK(hipMemcpy(c, (char*)ctx->d_sync + ngp_ctx::SLOT_BYTES * (size_t)slot + 32, sizeof(c), hipMemcpyDeviceToHost)); // the slot's results (ngp_ctx::bind_slot)
	out->kernel_device_ms = (float)((double)c[3] * 1e-5); // 100 MHz ticks
	out->n_rays = ctx->hist_n_rays[slot];
	out->n_rays_alive_after_init = c[0];
	out->n_rays_hit = c[1];
	out->n_samples = c[2];
	NGP_HIP_CHECK(hipEventElapsedTime(&out->kernel_ms, ctx->ev_kern0[slot], ctx->ev_kern1[slot]));
	NGP_HIP_CHECK(hipEventElapsedTime(&out->frame_ms, ctx->ev_frame0[slot], ctx->ev_frame1[slot]));
}

int ngp_get_render_stats(ngp_ctx* ctx, ngp_render_stats* out) {
	return guarded(ctx, [&] {
		if (!out) throw std::runtime_error("null argument");
		if (!ctx->n_calls) throw std::runtime_error("nothing rendered yet");
		NGP_HIP_CHECK(hipStreamSynchronize(ctx->last_stream));
		read_history_slot(ctx, ctx->n_calls - 1, out);
		if (ctx->last_was_multi) { // a frame over several devices: totals over the devices' shares, the slowest share's times
			for (ngp_ctx* p : ctx->peers) {
				ngp_render_stats s{};
				NGP_HIP_CHECK(hipSetDevice(p->device));
				if (ngp_get_render_stats(p, &s) != 0) throw std::runtime_error(p->error);
				out->n_rays += s.n_rays; out->n_rays_alive_after_init += s.n_rays_alive_after_init; out->n_rays_hit += s.n_rays_hit; out->n_samples += s.n_samples;
				out->kernel_ms = std::max(out->kernel_ms, s.kernel_ms);
				out->frame_ms = std::max(out->frame_ms, s.frame_ms);
				out->kernel_device_ms = std::max(out->kernel_device_ms, s.kernel_device_ms);
			}
			NGP_HIP_CHECK(hipSetDevice(ctx->device));
		}
		if (ctx->d_prof && getenv("NGP_PROFILE_SECTIONS")) {
			unsigned long long p[128];
			NGP_HIP_CHECK(hipMemcpy(p, ctx->d_prof, sizeof(p), hipMemcpyDeviceToHost));
			{ // wave timeline on the 100 MHz chip clock: when the tile queue ran dry, when the last wave left
				const double us = 0.01, t_first = (double)(~p[8]);
				fprintf(stderr, "[ngp timeline] kernel %.1f us | queue empty seen first at %.1f us, last at %.1f us | wave exits per 0.1 ms:", ((double)p[9] - t_first) * us,
				        ((double)(~p[10]) - t_first) * us, ((double)p[11] - t_first) * us);
				for (int b = 0; b < 48; ++b) fprintf(stderr, " %llu", p[16 + b]);
				fprintf(stderr, "\n[ngp skips] lane-steps that left an empty cell %llu, an empty 4^3 block %llu, an empty 16^3 block %llu\n", p[12], p[13], p[14]);
			}
			double tot = (double)(p[0] + p[1] + p[2] + p[3]);
			fprintf(stderr, "[ngp profile] refill %.1f%% march %.1f%% network %.1f%% composite %.1f%% | wave-iterations %llu passes %llu | cycles/iter %.0f cycles/pass(network) %.0f | skip rounds %llu lane-steps %llu (%.1f lanes/round) cycles/round %.0f\n",
			        100.0 * p[0] / tot, 100.0 * p[1] / tot, 100.0 * p[2] / tot, 100.0 * p[3] / tot, p[4], p[5], tot / (double)p[4], (double)p[2] / (double)p[5], p[6], p[7], (double)p[7] / (double)p[6], (double)p[1] / (double)p[6]);
			if (ctx->M.wide.width && p[5]) // the wide kernel's finer sections (wide_kernels.hip), cycles per network round of one workgroup
				fprintf(stderr, "[ngp wide profile] per network round: hidden layers %.0f (-) %.0f output layers %.0f | march loop %.0f decision %.0f rows+prefetch %.0f encode %.0f composite %.0f | rounds %llu network rounds %llu\n",
				        (double)p[64] / p[5], (double)p[65] / p[5], (double)p[71] / p[5], (double)p[66] / p[5], (double)p[67] / p[5], (double)p[68] / p[5], (double)p[69] / p[5], (double)p[70] / p[5], p[4], p[5]);
		}
	});
}

// diagnostic (NGP_PROFILE_SECTIONS + NGP_PROFILE_TRACE): the wave timelines of the last frame; layout in csrc/ngp_kernels.h (FrameParams::trace)
int ngp_get_profile_trace(ngp_ctx* ctx, uint32_t* out, uint64_t n_words, uint32_t* cap_waves, uint32_t* cap_iters) {
	return guarded(ctx, [&] {
		if (!ctx->d_trace) throw std::runtime_error("no wave trace: set NGP_PROFILE_SECTIONS=1|2 and NGP_PROFILE_TRACE=<stride> before rendering");
		const size_t words = 16 + (size_t)ngp_ctx::TRACE_WAVES * 16 + (size_t)ngp_ctx::TRACE_WAVES * ngp_ctx::TRACE_ITERS * 16;
		if (cap_waves) *cap_waves = ngp_ctx::TRACE_WAVES;
		if (cap_iters) *cap_iters = ngp_ctx::TRACE_ITERS;
		if (!out) return;
		NGP_HIP_CHECK(hipDeviceSynchronize());
		NGP_HIP_CHECK(hipMemcpy(out, ctx->d_trace, std::min<size_t>(words, (size_t)n_words) * sizeof(uint32_t), hipMemcpyDeviceToHost));
	});
}

int ngp_set_schedule(ngp_ctx* ctx, const int32_t* knobs, int n) {
	return guarded(ctx, [&] {
		if (!knobs) throw std::runtime_error("schedule: null");
		validate_schedule(knobs, n);
		if (ctx->last_stream) NGP_HIP_CHECK(hipStreamSynchronize(ctx->last_stream));
		for (int i = 0; i < n; ++i) ctx->tune[i] = knobs[i];
	});
}

int ngp_get_render_history(ngp_ctx* ctx, int n, ngp_render_stats* out) {
	return guarded(ctx, [&] {
		if (!out || n <= 0) throw std::runtime_error("invalid argument");
		if ((uint64_t)n > ctx->n_calls || n > ngp_ctx::HISTORY) throw std::runtime_error("history holds fewer render calls than requested");
		NGP_HIP_CHECK(hipStreamSynchronize(ctx->last_stream));
		for (int i = 0; i < n; ++i) read_history_slot(ctx, ctx->n_calls - (uint64_t)n + (uint64_t)i, &out[i]);
	});
}

int ngp_grid_encode(ngp_ctx* ctx, uint32_t n, const float* pos01, uint16_t* out_fp16) {
	return guarded(ctx, [&] {
		if (ctx->device < 0) throw std::runtime_error("this context has no HIP device (host-only); there is no CPU fallback");
		if (!ctx->model_loaded) throw std::runtime_error("No network available.");
		ngp::sync_inference_model(ctx);
		if (n == 0) return;
		if (!pos01 || !out_fp16) throw std::runtime_error("null argument");
		float* d_pos = nullptr;
		uint16_t* d_out = nullptr;
		NGP_HIP_CHECK(hipMalloc((void**)&d_pos, (size_t)n * 3 * sizeof(float)));
		const size_t width = ctx->M.wide.width ? ctx->M.wide.enc_dims : 32; // the position encoding's (padded) width
		NGP_HIP_CHECK(hipMalloc((void**)&d_out, (size_t)n * width * sizeof(uint16_t)));
		NGP_HIP_CHECK(hipMemcpy(d_pos, pos01, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
		if (ctx->M.wide.width) launch_frequency_encode(ctx->M, n, d_pos, d_out, ctx->stream);
		else launch_grid_encode(ctx->M, n, d_pos, d_out, ctx->stream);
		NGP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
		NGP_HIP_CHECK(hipMemcpy(out_fp16, d_out, (size_t)n * width * sizeof(uint16_t), hipMemcpyDeviceToHost));
		(void)hipFree(d_pos);
		(void)hipFree(d_out);
		NGP_HIP_CHECK(hipGetLastError());
	});
}

int ngp_density_gradient(ngp_ctx* ctx, uint32_t n, const float* pos01, float* out_grad) {
	return guarded(ctx, [&] {
		if (ctx->device < 0) throw std::runtime_error("this context has no HIP device (host-only); there is no CPU fallback");
		if (!ctx->model_loaded) throw std::runtime_error("No network available.");
		if (ctx->M.wide.width && (!ctx->M.wide.layers_t[0].n_mtiles || ctx->M.wide.enc_dims > ctx->M.wide.width))
			throw std::runtime_error("the density gradient of a Frequency / Identity-encoding model: implemented for up to 8 hidden density layers and an encoding no wider than the network");
		ngp::sync_inference_model(ctx);
		if (n == 0) return;
		if (!pos01 || !out_grad) throw std::runtime_error("null argument");
		float *d_pos = nullptr, *d_out = nullptr;
		NGP_HIP_CHECK(hipMalloc((void**)&d_pos, (size_t)n * 3 * sizeof(float)));
		NGP_HIP_CHECK(hipMalloc((void**)&d_out, (size_t)n * 3 * sizeof(float)));
		NGP_HIP_CHECK(hipMemcpy(d_pos, pos01, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
		if (ctx->M.wide.width) launch_density_gradient_wide(ctx->M, n, d_pos, d_out, ctx->n_cus, ctx->stream);
		else launch_density_gradient(ctx->M, n, d_pos, d_out, ctx->stream);
		NGP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
		NGP_HIP_CHECK(hipMemcpy(out_grad, d_out, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost));
		(void)hipFree(d_pos);
		(void)hipFree(d_out);
		NGP_HIP_CHECK(hipGetLastError());
	});
}

int ngp_network_inference(ngp_ctx* ctx, uint32_t n, const float* pos01, const float* dir01, uint16_t* out_fp16) {
	return guarded(ctx, [&] {
		if (ctx->device < 0) throw std::runtime_error("this context has no HIP device (host-only); there is no CPU fallback");
		if (!ctx->model_loaded) throw std::runtime_error("No network available.");
		ngp::sync_inference_model(ctx);
		if (n == 0) return;
		if (!pos01 || !dir01 || !out_fp16) throw std::runtime_error("null argument");
		float *d_pos = nullptr, *d_dir = nullptr;
		uint16_t* d_out = nullptr;
		NGP_HIP_CHECK(hipMalloc((void**)&d_pos, (size_t)n * 3 * sizeof(float)));
		NGP_HIP_CHECK(hipMalloc((void**)&d_dir, (size_t)n * 3 * sizeof(float)));
		NGP_HIP_CHECK(hipMalloc((void**)&d_out, (size_t)n * 4 * sizeof(uint16_t)));
		NGP_HIP_CHECK(hipMemcpy(d_pos, pos01, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
		NGP_HIP_CHECK(hipMemcpy(d_dir, dir01, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
		if (ctx->M.wide.width) launch_network_inference_wide(ctx->M, n, d_pos, d_dir, d_out, ctx->n_cus, ctx->stream);
		else launch_network_inference(ctx->M, n, d_pos, d_dir, d_out, ctx->stream);
		NGP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
		NGP_HIP_CHECK(hipMemcpy(out_fp16, d_out, (size_t)n * 4 * sizeof(uint16_t), hipMemcpyDeviceToHost));
		(void)hipFree(d_pos);
		(void)hipFree(d_dir);
		(void)hipFree(d_out);
		NGP_HIP_CHECK(hipGetLastError());
	});
}

int ngp_get_density_bitfield(ngp_ctx* ctx, uint8_t* out, float* out_mean) {
	return guarded(ctx, [&] {
		if (ctx->device < 0) throw std::runtime_error("this context has no HIP device (host-only); there is no CPU fallback");
		if (!ctx->model_loaded) throw std::runtime_error("No network available.");
		ngp::sync_inference_model(ctx);
		if (out) NGP_HIP_CHECK(hipMemcpy(out, ctx->d_bitfield, (size_t)NERF_GRID_N_CELLS / 8 * NERF_CASCADES, hipMemcpyDeviceToHost));
		if (out_mean) *out_mean = ctx->bitfield_mean;
	});
}

int ngp_set_render_aabb(ngp_ctx* ctx, const float* min3, const float* max3, const float* to_local9) {
	return guarded(ctx, [&] {
		if (!ctx->have_desc) throw std::runtime_error("No network available.");
		if (!min3 || !max3) throw std::runtime_error("null argument");
		for (int i = 0; i < 3; ++i) if (!(min3[i] <= max3[i])) throw std::runtime_error("render_aabb: min must not exceed max");
		if (ctx->last_stream) NGP_HIP_CHECK(hipStreamSynchronize(ctx->last_stream));
		const float ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
		const float* r2l = to_local9 ? to_local9 : ident;
		memcpy(ctx->desc.render_aabb_min, min3, 12);
		memcpy(ctx->desc.render_aabb_max, max3, 12);
		memcpy(ctx->desc.render_aabb_to_local, r2l, 36);
		memcpy(ctx->M.raabb_min, min3, 12);
		memcpy(ctx->M.raabb_max, max3, 12);
		memcpy(ctx->M.r2l, r2l, 36);
		ctx->M.r2l_identity = memcmp(r2l, ident, 36) == 0 ? 1u : 0u;
	});
}

int ngp_set_envmap(ngp_ctx* ctx, int32_t width, int32_t height, const float* rgba) {
	return guarded(ctx, [&] {
		if (ctx->device < 0) throw std::runtime_error("this context has no HIP device (host-only)");
		NGP_HIP_CHECK(hipDeviceSynchronize()); // frames in flight read the map
		if (ctx->d_bg_envmap) (void)hipFree(ctx->d_bg_envmap);
		ctx->d_bg_envmap = nullptr;
		ctx->bg_env_w = ctx->bg_env_h = 0;
		if (!rgba || width <= 0 || height <= 0) return;
		if ((int64_t)width * height > (1ll << 28)) throw std::runtime_error("environment map too large");
		NGP_HIP_CHECK(hipMalloc((void**)&ctx->d_bg_envmap, (size_t)width * height * sizeof(float4)));
		NGP_HIP_CHECK(hipMemcpy(ctx->d_bg_envmap, rgba, (size_t)width * height * sizeof(float4), hipMemcpyHostToDevice));
		ctx->bg_env_w = width;
		ctx->bg_env_h = height;
		for (ngp_ctx* p : ctx->peers) { // the replicas of a multi-device context see the same background
			if (ngp_set_envmap(p, width, height, rgba) != 0) throw std::runtime_error(p->error);
		}
		NGP_HIP_CHECK(hipSetDevice(ctx->device));
	});
}

int ngp_set_cone_angle_constant(ngp_ctx* ctx, float cone_angle_constant) {
	return guarded(ctx, [&] {
		if (!ctx->have_desc) throw std::runtime_error("No network available.");
		if (!(cone_angle_constant >= 0.f)) throw std::runtime_error("cone_angle_constant must be >= 0");
		if (ctx->last_stream) NGP_HIP_CHECK(hipStreamSynchronize(ctx->last_stream));
		ctx->desc.cone_angle_constant = cone_angle_constant;
		ctx->M.cone_angle = cone_angle_constant;
	});
}

int ngp_update_density_grid(ngp_ctx* ctx, float decay, uint32_t n_uniform, uint32_t n_nonuniform, uint32_t n_iterations) {
	return guarded(ctx, [&] {
		if (ctx->device < 0) throw std::runtime_error("this context has no HIP device (host-only); there is no CPU fallback");
		if (!ctx->model_loaded) throw std::runtime_error("No network available.");
		ngp::sync_inference_model(ctx);
		ngp::update_density_grid_device(ctx, decay, n_uniform, n_nonuniform, n_iterations);
		ngp::refresh_density_grid_host(ctx);
	});
}

int ngp_get_density_grid(ngp_ctx* ctx, float* out, uint64_t n) {
	return guarded(ctx, [&] {
		if (ctx->device < 0) throw std::runtime_error("this context has no HIP device (host-only); there is no CPU fallback");
		if (!ctx->model_loaded) throw std::runtime_error("No network available.");
		ngp::sync_inference_model(ctx);
		const uint64_t n_elements = (uint64_t)NERF_GRID_N_CELLS * (ctx->max_cascade + 1);
		if (!out || n != n_elements) throw std::runtime_error("density grid holds " + std::to_string(n_elements) + " values");
		NGP_HIP_CHECK(hipMemcpy(out, ctx->d_density_f32, n_elements * sizeof(float), hipMemcpyDeviceToHost));
	});
}

int ngp_init_rays(ngp_ctx* ctx, const ngp_camera* cam, void* payloads_out) {
	return guarded(ctx, [&] {
		if (ctx->device < 0) throw std::runtime_error("this context has no HIP device (host-only); there is no CPU fallback");
		if (!ctx->model_loaded) throw std::runtime_error("No network available.");
		ngp::sync_inference_model(ctx);
		if (!cam || !payloads_out) throw std::runtime_error("null argument");
		const size_t n = (size_t)cam->width * cam->height;
		NerfPayload* d_p = nullptr;
		NGP_HIP_CHECK(hipMalloc((void**)&d_p, n * sizeof(NerfPayload)));
		CameraParams C = make_camera_params(*cam, cam->spp_index);
		launch_init_rays(ctx->M, C, d_p, ctx->stream);
		NGP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
		NGP_HIP_CHECK(hipMemcpy(payloads_out, d_p, n * sizeof(NerfPayload), hipMemcpyDeviceToHost));
		(void)hipFree(d_p);
		NGP_HIP_CHECK(hipGetLastError());
	});
}

} // extern "C"
