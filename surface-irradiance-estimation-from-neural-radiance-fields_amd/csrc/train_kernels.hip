// NeRF training step on gfx950. One step is five launches:
//
//   train_generate_samples_kernel   16 lanes per ray: pixel draw, speculative occupancy march, NerfCoordinates (:737-890)
//   network_inference (tcnn order)  the fused MLP on the training parameters                            (:3303)
//   train_loss_kernel               one wave per ray: scans for composite / loss / dL/d(rgb, sigma), compaction (:893-1213)
//   train_backward_kernel           forward again + backward + weight gradients + grid scatter, fused
//   train_optimizer_kernel          Adam (+ ExponentialDecay through the learning rate) + Ema, gradient reset
//
// plus train_build_fragments_kernel, which re-forms the MFMA weight fragments from the updated parameters.
//
// The reference materialises every activation of the batch in global memory, runs tcnn's backward kernels layer by
// layer and three CUTLASS split-k GEMMs for the weight gradients. Here a wave owns 16 samples at a time and keeps the
// whole chain in registers: samples sit on the MFMA N axis in the forward and the backward-data products (an
// accumulator tile is the next product's B operand, as in the render kernel); the weight gradients sum over SAMPLES, so
// each activation / gradient tile goes once through a per-wave LDS image and comes back transposed with
// ds_read_b64_tr_b16 as the 16x16x16 operands of dW += dY . X^T, accumulated in fp32 registers (160 per lane) for the
// whole launch and reduced block-wise at the end. No activation ever reaches HBM; gradients accumulate in fp32 (tcnn:
// fp16 atomics for the grid, fp16 GEMM outputs for the matrices).
#include "nerf_device.h"
#include "ngp_host.h"
#include "train_kernels.h"

namespace ngp {

namespace {
constexpr int BLOCK = 256;

typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------
// image access, common_device.cuh:753-829
NGP_DEV float4 read_training_pixel(const TrainImage& im, float u, float v) {
	int px = (int)(u * (float)im.res[0]), py = (int)(v * (float)im.res[1]);
	px = px < 0 ? 0 : (px > im.res[0] - 1 ? im.res[0] - 1 : px);
	py = py < 0 ? 0 : (py > im.res[1] - 1 ? im.res[1] - 1 : py);
	const size_t idx = (size_t)px + (size_t)py * (size_t)im.res[0];
	if (im.type == 1) {
		const uint32_t raw = ((const uint32_t*)im.pixels)[idx];
		if (raw == 0x00FF00FFu) return make_float4(-1.f, -1.f, -1.f, -1.f); // masked away
		const float a = (float)(raw >> 24) * (1.0f / 255.0f);
		return make_float4(srgb_to_linear((float)(raw & 255u) * (1.0f / 255.0f)) * a, srgb_to_linear((float)((raw >> 8) & 255u) * (1.0f / 255.0f)) * a,
		                   srgb_to_linear((float)((raw >> 16) & 255u) * (1.0f / 255.0f)) * a, a);
	}
	if (im.type == 3) return ((const float4*)im.pixels)[idx];
	return make_float4(5.0f, 0.0f, 0.0f, 1.0f);
}

// image_idx (nerf_device.cuh:617-638), no per-image CDF: neighbouring rays share an image
NGP_DEV uint32_t training_image_of(uint32_t base_idx, uint32_t n_rays, uint32_t n_images) { return ((base_idx * n_images) / n_rays) % n_images; }

NGP_DEV void training_uv(Pcg32& rng, const TrainImage& im, int snap, float& u, float& v) { // nerf_random_image_pos_training :592-615
	u = rng.next_float();
	v = rng.next_float();
	if (snap) {
		int px = (int)(u * (float)im.res[0]), py = (int)(v * (float)im.res[1]);
		px = px < 0 ? 0 : (px > im.res[0] - 1 ? im.res[0] - 1 : px);
		py = py < 0 ? 0 : (py > im.res[1] - 1 ? im.res[1] - 1 : py);
		u = ((float)px + 0.5f) / (float)im.res[0];
		v = ((float)py + 0.5f) / (float)im.res[1];
	}
}

NGP_DEV uint32_t mip_from_dt(float dt, f3 pos, uint32_t max_cascade) { // nerf_device.cuh:449-458
	uint32_t mip = mip_from_pos(pos, max_cascade);
	dt *= 2.0f * (float)NERF_GRIDSIZE;
	if (dt < 1.0f) return mip;
	int exponent;
	(void)__builtin_frexpf(dt, &exponent);
	int v = (int)mip < exponent ? exponent : (int)mip;
	v = v > (int)max_cascade ? (int)max_cascade : v;
	return (uint32_t)v;
}

NGP_DEV bool train_aabb_contains(const ModelParams& M, f3 p) {
	return p.x >= M.aabb_min[0] && p.x <= M.aabb_min[0] + M.aabb_diag[0] && p.y >= M.aabb_min[1] && p.y <= M.aabb_min[1] + M.aabb_diag[1] && p.z >= M.aabb_min[2] &&
	       p.z <= M.aabb_min[2] + M.aabb_diag[2];
}

// ---------------------------------------------------------------------------------------------------------
// generate_training_samples_nerf, src/testbed_nerf.cu:737-890 (no envmap, no error-map CDFs, no explicit rays, no
// random max level, static cameras).
//
// The reference marches a ray in one thread, twice (count, then write): up to 1024 dependent iterations of
// "look up the cell, then either step or jump". With the ~5000 rays of a batch that is a few dozen waves running a
// long serial chain each -- 2 ms of an otherwise 1 ms step. The chain of positions, however, only depends on memory
// through the yes/no answers, so a ROW of 16 lanes owns a ray (four rays per wave) and speculates:
//   dense mode   the 16 positions t_0 = t, t_{k+1} = t_k + calc_dt(t_k) that the loop visits if every cell is occupied
//                are formed once (a short recurrence, no memory), lane k tests position k, and the row's ballot bits
//                find the first empty or outside one: everything in front of it is accepted at once;
//   jump mode    after an empty cell the 16 positions of "every cell empty" (advance_to_next_voxel chained) are formed
//                and tested the same way; the first occupied one hands over to dense mode.
// The recurrences are the serial part; the four rows of a wave run theirs side by side in the same instructions.
// Accepted positions are exactly the reference loop's (the same functions applied in the same order to the same
// values); they wait in LDS until the ray's offset is known, then the row writes the coordinates.
constexpr int GEN_RAYS_PER_WAVE = 4, GEN_WAVES_PER_BLOCK = 2, GEN_CAND = 16;
__global__ __launch_bounds__(64 * GEN_WAVES_PER_BLOCK) void train_generate_samples_kernel(const ModelParams M, const TrainStepParams P, const TrainImage* __restrict__ images,
                                                                                         const TrainBatch B) {
	extern __shared__ char s_gen[];
	float* s_t = (float*)s_gen;                                                                    // [rays per block][NERF_STEPS]
	uint32_t* s_coarse = (uint32_t*)(s_gen + sizeof(float) * NERF_STEPS * GEN_RAYS_PER_WAVE * GEN_WAVES_PER_BLOCK); // [max_cascade + 1][1024]
	for (uint32_t k = threadIdx.x; k < (M.max_cascade + 1u) * COARSE_WORDS_PER_MIP; k += blockDim.x) s_coarse[k] = M.coarse[k];
	__syncthreads();
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
	const uint32_t i = (blockIdx.x * GEN_WAVES_PER_BLOCK + wave) * GEN_RAYS_PER_WAVE + g;
	float* my_ts = s_t + (size_t)(wave * GEN_RAYS_PER_WAVE + g) * NERF_STEPS;

	// ---- ray setup, identical in the 16 lanes of a row
	bool done = i >= P.n_rays;
	const uint32_t ic = done ? 0u : i;
	const TrainImage im = images[training_image_of(ic, P.n_rays, P.n_images)];
	Pcg32 rng = P.rng;
	rng.advance((uint64_t)(uint32_t)(ic * N_MAX_RANDOM_SAMPLES_PER_RAY));
	float u, v;
	training_uv(rng, im, P.snap_to_pixel_centers, u, v);
	if (read_training_pixel(im, u, v).x < 0.0f) done = true;
	(void)rng.next_float(); // motionblur_time
	CameraParams C;
	C.width = im.res[0]; C.height = im.res[1];
	C.focal[0] = im.focal[0]; C.focal[1] = im.focal[1];
	C.screen_center[0] = im.principal[0]; C.screen_center[1] = im.principal[1];
	C.lens_mode = im.lens_mode;
	for (int k = 0; k < 7; ++k) C.lens_params[k] = im.lens_params[k];
	f3 dir;
	lens_direction(C, u, v, dir);
	const f3 d_un = m3_mulv(im.xform, dir);
	const f3 o = mk3(im.xform[9], im.xform[10], im.xform[11]);
	const f3 d = normalize3(d_un);
	float bmax[3] = {M.aabb_min[0] + M.aabb_diag[0], M.aabb_min[1] + M.aabb_diag[1], M.aabb_min[2] + M.aabb_diag[2]};
	const float tmin = fmaxf(aabb_ray_entry(M.aabb_min, bmax, o, d), 0.0f);
	const Stepping stepping = make_stepping(M.cone_angle); // (once per thread: three logf / two expf the step functions would otherwise rebuild per call)
	const f3 idir = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);

	float t = advance_n_steps(tmin, stepping, rng.next_float()); // row-uniform from here on
	uint32_t j = 0;
	bool dense = true;
	const int row_shift = 16 * g;
	while (__ballot(!done) != 0ull) {
		float my_t = t, my_dt = 0.0f, next_t = t;
		if (__ballot(!done && dense) != 0ull) {
			if (!done && dense) {
				float tt = t;
				for (int k = 0; k < GEN_CAND; ++k) {
					const float dt = calc_dt(tt, stepping);
					if (c == k) { my_t = tt; my_dt = dt; }
					tt += dt;
				}
				next_t = tt;
			}
		}
		if (__ballot(!done && !dense) != 0ull) {
			if (!done && !dense) {
				float tt = t;
				for (int k = 0; k < GEN_CAND; ++k) {
					const float dt = calc_dt(tt, stepping);
					const f3 pos = add3(o, scale3(d, tt));
					const uint32_t mip = mip_from_dt(dt, pos, M.max_cascade);
					if (c == k) { my_t = tt; my_dt = dt; }
					tt = advance_to_next_voxel(tt, stepping, pos, d, idir, mip);
				}
				next_t = tt;
			}
		}
		const f3 my_pos = add3(o, scale3(d, my_t));
		const bool inside = !done && train_aabb_contains(M, my_pos);
		const uint32_t my_mip = mip_from_dt(my_dt, my_pos, M.max_cascade);
		const bool occupied = inside && density_grid_occupied_at_lds(my_pos, M.bitfield, s_coarse, my_mip);
		// the row's 16 answers
		const uint32_t in_mask = (uint32_t)(__ballot(inside) >> row_shift) & 0xFFFFu, occ_mask = (uint32_t)(__ballot(occupied) >> row_shift) & 0xFFFFu;
		const uint32_t out_mask = ~in_mask & 0xFFFFu;
		const int first_out = out_mask ? __builtin_ctz(out_mask) : GEN_CAND;
		// row broadcasts of the candidate a decision may need (executed by every lane: shuffles want the whole wave)
		const uint32_t empty_mask = ~occ_mask & in_mask;
		const int first_empty = empty_mask ? __builtin_ctz(empty_mask) : GEN_CAND;
		const int first_occ = occ_mask ? __builtin_ctz(occ_mask) : GEN_CAND;
		const int pick = dense ? (first_empty < GEN_CAND ? first_empty : 0) : (first_occ < GEN_CAND ? first_occ : 0);
		const float t_pick = __shfl(my_t, row_shift + pick, 64);
		const uint32_t mip_pick = (uint32_t)__shfl((int)my_mip, row_shift + pick, 64);
		if (done) continue;
		if (dense) {
			const int stop = first_out < first_empty ? first_out : first_empty;
			const uint32_t room = NERF_STEPS - j;
			const uint32_t n_acc = (uint32_t)stop < room ? (uint32_t)stop : room;
			if ((uint32_t)c < n_acc) my_ts[j + c] = my_t;
			j += n_acc;
			if (j >= NERF_STEPS || (stop == first_out && stop < GEN_CAND)) done = true; // full, or the ray left the box
			else if (stop == GEN_CAND) t = next_t;
			else { // position `stop` is inside and empty: one jump from it, then look for the next occupied cell
				t = advance_to_next_voxel(t_pick, stepping, add3(o, scale3(d, t_pick)), d, idir, mip_pick);
				dense = false;
			}
		} else {
			if (first_out < first_occ) done = true; // left the box before meeting anything
			else if (first_occ < GEN_CAND) { t = t_pick; dense = true; }
			else t = next_t;
		}
	}
	const uint32_t numsteps = j;
	uint32_t base = 0;
	const bool leader = c == 0 && numsteps > 0;
	if (leader) base = atomicAdd(&B.counters[0], numsteps);
	base = (uint32_t)__shfl((int)base, row_shift, 64);
	const bool fits = numsteps > 0 && base + numsteps <= P.max_samples;
	if (leader && fits) {
		const uint32_t ray_idx = atomicAdd(&B.counters[1], 1u);
		B.ray_indices[ray_idx] = i;
		float* r = B.rays + (size_t)ray_idx * 6;
		r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = d_un.x; r[4] = d_un.y; r[5] = d_un.z;
		B.numsteps[ray_idx * 2 + 0] = numsteps;
		B.numsteps[ray_idx * 2 + 1] = base;
	}
	if (!fits) return;
	const f3 wdir = mk3((d.x + 1.0f) * 0.5f, (d.y + 1.0f) * 0.5f, (d.z + 1.0f) * 0.5f);
	const f3 amin = mk3(M.aabb_min[0], M.aabb_min[1], M.aabb_min[2]), adiag = mk3(M.aabb_diag[0], M.aabb_diag[1], M.aabb_diag[2]);
	for (uint32_t k = (uint32_t)c; k < numsteps; k += 16) {
		const float tk = my_ts[k];
		const float dt = calc_dt(tk, stepping);
		const f3 w = div3(sub3(add3(o, scale3(d, tk)), amin), adiag);
		float* co = B.coords + (size_t)(base + k) * TRAIN_COORD_FLOATS;
		co[0] = w.x; co[1] = w.y; co[2] = w.z; co[3] = warp_dt(dt); co[4] = wdir.x; co[5] = wdir.y; co[6] = wdir.z;
	}
}

// ---------------------------------------------------------------------------------------------------------
// losses, nerf_device.cuh:61-142, 640-658
struct LossGrad { float loss, grad; };
NGP_DEV LossGrad loss_and_gradient(float target, float prediction, int type) {
	const float diff = prediction - target;
	const float sign = __builtin_copysignf(1.0f, diff);
	LossGrad r;
	switch (type) {
		case 1: r.loss = __builtin_fabsf(diff); r.grad = sign; break; // L1
		case 2: { float denom = __builtin_fabsf(prediction) + 1e-2f; r.loss = __builtin_fabsf(diff) / denom; r.grad = sign / denom; break; } // Mape
		case 3: { float denom = 0.5f * (__builtin_fabsf(prediction) + __builtin_fabsf(target)) + 1e-2f; r.loss = __builtin_fabsf(diff) / denom; r.grad = sign / denom; break; } // Smape
		case 4: { // Huber(alpha 0.1) / 5
			const float alpha = 0.1f, ad = __builtin_fabsf(diff);
			r.loss = (ad > alpha ? (ad - 0.5f * alpha) : (0.5f / alpha * diff * diff)) / 5.0f;
			r.grad = (ad > alpha ? (diff > 0.0f ? 1.0f : -1.0f) : (diff / alpha)) / 5.0f;
			break;
		}
		case 5: { float div = __builtin_fabsf(diff) + 1.0f; r.loss = __builtin_logf(div); r.grad = sign / div; break; } // LogL1
		case 6: { float denom = prediction * prediction + 1e-2f; r.loss = diff * diff / denom; r.grad = 2.0f * diff / denom; break; } // RelativeL2
		default: r.loss = diff * diff; r.grad = 2.0f * diff; break; // L2
	}
	return r;
}
NGP_DEV float network_to_rgb_derivative(float v, uint32_t act) { // nerf_device.cuh:214-223
	switch (act) {
		case 1: return v > 0.0f ? 1.0f : 0.0f;
		case 2: { float s = logistic(v); return s * (1.0f - s); }
		case 3: return fast_exp(fminf(fmaxf(v, -10.0f), 10.0f));
		default: return 1.0f;
	}
}
NGP_DEV float network_to_density_derivative(float v, uint32_t act) { // :245-254
	switch (act) {
		case 1: return v > 0.0f ? 1.0f : 0.0f;
		case 2: { float s = logistic(v); return s * (1.0f - s); }
		case 3: return fast_exp(fminf(fmaxf(v, -15.0f), 15.0f));
		default: return 1.0f;
	}
}
NGP_DEV float half_bits_to_float(uint16_t b) {
	union { uint16_t u; half_t h; } cv;
	cv.u = b;
	return (float)cv.h;
}
NGP_DEV uint16_t float_to_half_bits(float f) {
	union { uint16_t u; half_t h; } cv;
	cv.h = (half_t)f;
	return cv.u;
}

// compute_loss_kernel_train_nerf, src/testbed_nerf.cu:893-1213 (no envmap, exposure, depth supervision, error map,
// sharpness). The reference walks a ray's samples in one thread, three times; with ~5000 rays per batch that is a few
// dozen waves of serial, latency-bound loops. Here ONE WAVE owns a ray and its lanes own consecutive samples: the
// transmittance is a prefix product, the composited colour a prefix sum (6-step shuffle scans over the wave, carried
// from one 64-sample chunk to the next), early termination a ballot. Sums are formed pairwise instead of left to
// right, which moves results by an ulp or two of fp32.
NGP_DEV float wave_scan_mul(float v, int lane) {
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const float o = __shfl_up(v, d, 64);
		if (lane >= d) v *= o;
	}
	return v;
}
NGP_DEV float wave_scan_add(float v, int lane) {
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const float o = __shfl_up(v, d, 64);
		if (lane >= d) v += o;
	}
	return v;
}
NGP_DEV float wave_last(float v) { return __shfl(v, 63, 64); }

__global__ __launch_bounds__(BLOCK) void train_loss_kernel(const ModelParams M, const TrainStepParams P, const TrainImage* __restrict__ images, const TrainBatch B) {
	const int lane = threadIdx.x & 63;
	const uint32_t i = (blockIdx.x * BLOCK + threadIdx.x) >> 6; // one wave per ray
	if (i >= B.counters[1]) return;
	const uint32_t numsteps = B.numsteps[i * 2 + 0];
	const uint32_t base = B.numsteps[i * 2 + 1];
	const float* coords_in = B.coords + (size_t)base * TRAIN_COORD_FLOATS;
	const uint2* net = (const uint2*)(B.mlp_out + (size_t)base * 4);
	const f3 amin = mk3(M.aabb_min[0], M.aabb_min[1], M.aabb_min[2]), adiag = mk3(M.aabb_diag[0], M.aabb_diag[1], M.aabb_diag[2]);
	const f3 ray_o = mk3(B.rays[(size_t)i * 6 + 0], B.rays[(size_t)i * 6 + 1], B.rays[(size_t)i * 6 + 2]);
	const float EPSILON = 1e-4f;

	// ---- pass 1: composite until the transmittance falls below EPSILON
	float T_in = 1.0f;
	f3 rgb_ray = mk3(0.f, 0.f, 0.f);
	uint32_t compacted_numsteps = 0;
	for (uint32_t c0 = 0; c0 < numsteps; c0 += 64) {
		const uint32_t j = c0 + lane;
		const bool valid = j < numsteps;
		float alpha = 0.0f;
		f3 rgb = mk3(0.f, 0.f, 0.f);
		if (valid) {
			const uint2 o = net[j];
			rgb = mk3(network_to_rgb(half_bits_to_float((uint16_t)(o.x & 0xFFFFu)), M.rgb_act), network_to_rgb(half_bits_to_float((uint16_t)(o.x >> 16)), M.rgb_act),
			          network_to_rgb(half_bits_to_float((uint16_t)(o.y & 0xFFFFu)), M.rgb_act));
			const float dt = unwarp_dt(coords_in[(size_t)j * TRAIN_COORD_FLOATS + 3]);
			alpha = 1.0f - fast_exp(-network_to_density(half_bits_to_float((uint16_t)(o.y >> 16)), M.density_act) * dt);
		}
		const float incl = wave_scan_mul(1.0f - alpha, lane);
		float excl = __shfl_up(incl, 1, 64);
		if (lane == 0) excl = 1.0f;
		const float T = T_in * excl; // transmittance in front of sample j
		const bool processed = valid && T >= EPSILON;
		const unsigned long long pm = __ballot(processed);
		const float weight = processed ? alpha * T : 0.0f;
		rgb_ray.x += wave_last(wave_scan_add(weight * rgb.x, lane));
		rgb_ray.y += wave_last(wave_scan_add(weight * rgb.y, lane));
		rgb_ray.z += wave_last(wave_scan_add(weight * rgb.z, lane));
		const uint32_t n_proc = (uint32_t)__popcll(pm);
		compacted_numsteps += n_proc;
		// transmittance behind the last processed sample
		T_in = n_proc ? T_in * __shfl(incl, (int)n_proc - 1, 64) : T_in;
		if (n_proc < 64u) break; // terminated, or the ray ended inside this chunk
	}
	const float T_end = T_in;

	// the same random numbers as the generating thread (every lane computes them: wave-uniform)
	const uint32_t ray_idx = B.ray_indices[i];
	Pcg32 rng = P.rng;
	rng.advance((uint64_t)(uint32_t)(ray_idx * N_MAX_RANDOM_SAMPLES_PER_RAY));
	const uint32_t img = training_image_of(ray_idx, P.n_rays, P.n_images);
	const TrainImage im = images[img];
	float u, v;
	training_uv(rng, im, P.snap_to_pixel_centers, u, v);
	rng.advance(1); // motionblur_time
	f3 background = mk3(P.background[0], P.background[1], P.background[2]);
	if (P.random_bg_color) {
		background.x = rng.next_float();
		background.y = rng.next_float();
		background.z = rng.next_float();
	}
	background = mk3(srgb_to_linear(background.x), srgb_to_linear(background.y), srgb_to_linear(background.z));
	const float4 texsamp = read_training_pixel(im, u, v);
	f3 rgbtarget;
	if (P.linear_colors || P.color_space == 0) {
		rgbtarget = mk3(texsamp.x + (1.0f - texsamp.w) * background.x, texsamp.y + (1.0f - texsamp.w) * background.y, texsamp.z + (1.0f - texsamp.w) * background.z);
		if (!P.linear_colors) {
			rgbtarget = mk3(linear_to_srgb(rgbtarget.x), linear_to_srgb(rgbtarget.y), linear_to_srgb(rgbtarget.z));
			background = mk3(linear_to_srgb(background.x), linear_to_srgb(background.y), linear_to_srgb(background.z));
		}
	} else {
		background = mk3(linear_to_srgb(background.x), linear_to_srgb(background.y), linear_to_srgb(background.z));
		if (texsamp.w > 0.0f) {
			rgbtarget = mk3(linear_to_srgb(texsamp.x / texsamp.w) * texsamp.w + (1.0f - texsamp.w) * background.x,
			                linear_to_srgb(texsamp.y / texsamp.w) * texsamp.w + (1.0f - texsamp.w) * background.y,
			                linear_to_srgb(texsamp.z / texsamp.w) * texsamp.w + (1.0f - texsamp.w) * background.z);
		} else {
			rgbtarget = background;
		}
	}
	if (compacted_numsteps == numsteps) rgb_ray = add3(rgb_ray, scale3(background, T_end));

	uint32_t compacted_base = 0;
	if (lane == 0) compacted_base = atomicAdd(&B.counters[2], compacted_numsteps);
	compacted_base = (uint32_t)__shfl((int)compacted_base, 0, 64);
	const uint32_t room = P.target_batch - (P.target_batch < compacted_base ? P.target_batch : compacted_base);
	compacted_numsteps = room < compacted_numsteps ? room : compacted_numsteps;
	if (lane == 0) {
		B.numsteps[i * 2 + 0] = compacted_numsteps;
		B.numsteps[i * 2 + 1] = compacted_base;
	}
	if (compacted_numsteps == 0) return;

	const LossGrad lx = loss_and_gradient(rgbtarget.x, rgb_ray.x, P.loss_type), ly = loss_and_gradient(rgbtarget.y, rgb_ray.y, P.loss_type),
	               lz = loss_and_gradient(rgbtarget.z, rgb_ray.z, P.loss_type);
	const f3 lgrad = mk3(lx.grad, ly.grad, lz.grad);
	if (lane == 0) B.loss[i] = ((lx.loss + ly.loss + lz.loss) / 3.0f) / (float)P.n_rays;

	const float loss_scale = P.loss_scale / (float)P.n_rays;
	const float output_l2_reg = M.rgb_act == 3u ? 1e-4f : 0.0f;
	const float output_l1_reg_density = P.density_grid_mean < 0.01f ? 1e-4f : 0.0f; // NERF_MIN_OPTICAL_THICKNESS

	// ---- pass 2: gradients of the compacted prefix; lanes copy their sample's coordinates as they go
	float* coords_out = B.coords_compacted + (size_t)compacted_base * TRAIN_COORD_FLOATS;
	uint2* dloss = (uint2*)(B.dloss + (size_t)compacted_base * 4);
	f3 rgb_done = mk3(0.f, 0.f, 0.f); // rgb_ray2 behind the previous chunk
	T_in = 1.0f;
	for (uint32_t c0 = 0; c0 < compacted_numsteps; c0 += 64) {
		const uint32_t j = c0 + lane;
		const bool valid = j < compacted_numsteps;
		float alpha = 0.0f, dt = 0.0f, depth = 0.0f, o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
		f3 rgb = mk3(0.f, 0.f, 0.f);
		if (valid) {
			const float* cin = coords_in + (size_t)j * TRAIN_COORD_FLOATS;
			float* cout = coords_out + (size_t)j * TRAIN_COORD_FLOATS;
			float cv[TRAIN_COORD_FLOATS];
#pragma unroll
			for (int k = 0; k < (int)TRAIN_COORD_FLOATS; ++k) cv[k] = cin[k];
#pragma unroll
			for (int k = 0; k < (int)TRAIN_COORD_FLOATS; ++k) cout[k] = cv[k];
			const f3 pos = add3(mul3(mk3(cv[0], cv[1], cv[2]), adiag), amin); // unwarp_position
			const f3 dp = sub3(pos, ray_o);
			depth = __builtin_sqrtf(dot3(dp, dp));
			dt = unwarp_dt(cv[3]);
			const uint2 o = net[j];
			o0 = half_bits_to_float((uint16_t)(o.x & 0xFFFFu));
			o1 = half_bits_to_float((uint16_t)(o.x >> 16));
			o2 = half_bits_to_float((uint16_t)(o.y & 0xFFFFu));
			o3 = half_bits_to_float((uint16_t)(o.y >> 16));
			rgb = mk3(network_to_rgb(o0, M.rgb_act), network_to_rgb(o1, M.rgb_act), network_to_rgb(o2, M.rgb_act));
			alpha = 1.0f - fast_exp(-network_to_density(o3, M.density_act) * dt);
		}
		const float incl = wave_scan_mul(1.0f - alpha, lane);
		float excl = __shfl_up(incl, 1, 64);
		if (lane == 0) excl = 1.0f;
		const float T_before = T_in * excl, T_after = T_in * incl;
		const float weight = alpha * T_before;
		const f3 rgb_ray2 = mk3(rgb_done.x + wave_scan_add(weight * rgb.x, lane), rgb_done.y + wave_scan_add(weight * rgb.y, lane), rgb_done.z + wave_scan_add(weight * rgb.z, lane));
		if (valid) {
			// the suffix of the ray behind this sample is (1 - alpha) * something: d suffix / d alpha = -suffix / (1 - alpha)
			const f3 suffix = sub3(rgb_ray, rgb_ray2);
			const f3 dloss_by_drgb = scale3(lgrad, weight);
			const uint16_t d0 = float_to_half_bits(loss_scale * (dloss_by_drgb.x * network_to_rgb_derivative(o0, M.rgb_act) + fmaxf(0.0f, output_l2_reg * o0)));
			const uint16_t d1 = float_to_half_bits(loss_scale * (dloss_by_drgb.y * network_to_rgb_derivative(o1, M.rgb_act) + fmaxf(0.0f, output_l2_reg * o1)));
			const uint16_t d2 = float_to_half_bits(loss_scale * (dloss_by_drgb.z * network_to_rgb_derivative(o2, M.rgb_act) + fmaxf(0.0f, output_l2_reg * o2)));
			const float density_derivative = network_to_density_derivative(o3, M.density_act);
			const float dloss_by_dmlp = density_derivative * (dt * dot3(lgrad, sub3(scale3(rgb, T_after), suffix)));
			const uint16_t d3 = float_to_half_bits(loss_scale * dloss_by_dmlp + (o3 < 0.0f ? -output_l1_reg_density : 0.0f) + (o3 > -10.0f && depth < P.near_distance ? 1e-4f : 0.0f));
			dloss[j] = make_uint2((uint32_t)d0 | ((uint32_t)d1 << 16), (uint32_t)d2 | ((uint32_t)d3 << 16));
		}
		rgb_done = mk3(wave_last(rgb_ray2.x), wave_last(rgb_ray2.y), wave_last(rgb_ray2.z));
		T_in = wave_last(T_after);
	}
}

// fill_rollover_and_rescale + fill_rollover (tcnn common_device.h, called at src/testbed_nerf.cu:3362-3370): a batch
// short of the target is topped up with copies of its own samples, the copies' gradients scaled by n / target. The
// copies see the same coordinates, so their contribution is the original's times a constant: sample i of the n
// measured ones carries 1 + (copies of i) * n / target. Samples past n keep a zero gradient.
__global__ void train_rollover_kernel(uint32_t target_batch, const uint32_t* __restrict__ counters, uint16_t* __restrict__ dloss, float* __restrict__ coords) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= target_batch) return;
	const uint32_t n = counters[2] < target_batch ? counters[2] : target_batch;
	if (i >= n) {
		for (int k = 0; k < 4; ++k) dloss[(size_t)i * 4 + k] = 0;
		for (int k = 0; k < (int)TRAIN_COORD_FLOATS; ++k) coords[(size_t)i * TRAIN_COORD_FLOATS + k] = 0.5f;
		return;
	}
	if (n == target_batch || n == 0) return;
	const uint32_t copies = (target_batch - 1u - i) / n; // indices i + k n < target_batch, k >= 1
	if (copies == 0) return;
	const float rescale = (float)n / (float)target_batch;
	for (int k = 0; k < 4; ++k) {
		const float g = half_bits_to_float(dloss[(size_t)i * 4 + k]);
		const float copy = (float)(half_t)(g * rescale); // each copy is rounded to fp16 on its own
		dloss[(size_t)i * 4 + k] = float_to_half_bits(g + (float)copies * copy);
	}
}

// ---------------------------------------------------------------------------------------------------------
// MFMA fragments of the current parameters, on the device (the host forms the inference set once per model in
// ngp_api.cpp emit_fragments; during training the weights change every step).
// params: fp16 [W_D0 64x32 | W_D1 16x64 | W_R0 64x32 | W_R1 64x64 | W_R2 16x64], row-major [out][in].
NGP_DEV uint16_t frag_element(const uint16_t* W, int n_in, bool transposed, int f_local, int n_ksteps, int l, int j) {
	const int m = f_local / n_ksteps, s = f_local % n_ksteps;
	const int row = 16 * m + (l & 15), h = l >> 4;
	const int k = 32 * s + 16 * (j >> 2) + 4 * h + (j & 3);
	return transposed ? W[(size_t)k * n_in + row] : W[(size_t)row * n_in + k];
}
__global__ void train_build_fragments_kernel(const uint16_t* __restrict__ params, uint16_t* __restrict__ frags /* [N_TFRAGS][64][8] */, uint16_t* __restrict__ kfrags /* [N_KFRAGS][64][4] */) {
	const int t = threadIdx.x + blockIdx.x * blockDim.x;
	const uint16_t* WD0 = params;
	const uint16_t* WD1 = params + 64 * 32;
	const uint16_t* WR0 = params + 64 * 32 + 16 * 64;
	const uint16_t* WR1 = WR0 + 64 * 32;
	const uint16_t* WR2 = WR1 + 64 * 64;
	if (t < N_TFRAGS * 64 * 8) {
		const int f = t / 512, l = (t / 8) & 63, j = t & 7;
		uint16_t val;
		if (f < FRAG_D1) val = frag_element(WD0, 32, false, f - FRAG_D0, 1, l, j);
		else if (f < FRAG_R0) val = frag_element(WD1, 64, false, f - FRAG_D1, 2, l, j);
		else if (f < FRAG_R1) val = frag_element(WR0, 32, false, f - FRAG_R0, 1, l, j);
		else if (f < FRAG_R2) val = frag_element(WR1, 64, false, f - FRAG_R1, 2, l, j);
		else if (f < TFRAG_R1T) val = frag_element(WR2, 64, false, f - FRAG_R2, 2, l, j);
		else if (f < TFRAG_R0T) val = frag_element(WR1, 64, true, f - TFRAG_R1T, 2, l, j);  // (W_R1^T)[row][k] = W_R1[k][row]
		else if (f < TFRAG_D0T) val = frag_element(WR0, 32, true, f - TFRAG_R0T, 2, l, j);  // rows 0..15 of W_R0^T (32 x 64)
		else val = frag_element(WD0, 32, true, f - TFRAG_D0T, 2, l, j);                       // W_D0^T (32 x 64)
		frags[t] = val;
	}
	if (t < N_KFRAGS * 64 * 4) {
		// 16x16x16 A operand: lane (i = l & 15, kg = l >> 4), element j: A[16 m + i][k = 4 kg + j] = W[k][16 m + i]
		const int f = t / 256, l = (t / 4) & 63, j = t & 3;
		const int m = f & 3, row = 16 * m + (l & 15), k = 4 * (l >> 4) + j;
		kfrags[t] = (f < KFRAG_D1T ? WR2 : WD1)[(size_t)k * 64 + row];
	}
}

// ---------------------------------------------------------------------------------------------------------
// hash-grid encode from the tcnn-order table only (training parameters have no xor-layout copy)
NGP_DEV half8 encode_level_pair_tcnn(const uint2* __restrict__ grid, const LevelInfo* lv, int h, float x, float y, float z) {
	FeatureAcc acc[2] = {{{0, 0}, {0, 0}}, {{0, 0}, {0, 0}}};
#pragma unroll
	for (int l = 0; l < 2; ++l) {
		const LevelInfo& L = lv[h + 4 * l];
		const CellPos p = level_cell(L, x, y, z);
		CornerSet cs;
		level_corners(L, p, cs);
		float w[8];
		corner_weights(p, w);
#pragma unroll
		for (int c = 0; c < 8; ++c) accumulate_corner(*(const uint2*)((const char*)grid + cs.index[c]), w[c], acc[l]);
	}
	half8 out;
	store_features(acc[0], acc[1], out);
	return out;
}

__global__ __launch_bounds__(BLOCK) void train_inference_kernel(const ModelParams M, const uint4* __restrict__ frags, const uint32_t* __restrict__ counters, uint32_t max_samples,
                                                                const float* __restrict__ coords, uint16_t* __restrict__ out) {
	__shared__ uint4 s_w[N_FRAGS * 64];
	__shared__ LevelInfo s_lv[N_LEVELS];
	for (int i = threadIdx.x; i < N_FRAGS * 64; i += BLOCK) s_w[i] = frags[i];
	if (threadIdx.x < N_LEVELS) s_lv[threadIdx.x] = M.levels[threadIdx.x];
	__syncthreads();
	const uint32_t n = counters[0] < max_samples ? counters[0] : max_samples;
	const int lane = threadIdx.x & 63, c = lane & 15;
	for (uint32_t wave = (blockIdx.x * BLOCK + threadIdx.x) >> 6; wave * 64u < n; wave += gridDim.x * (BLOCK / 64)) {
		for (int p = 0; p < 4; ++p) {
			const uint32_t s = wave * 64u + 16u * p + c;
			const uint32_t sc = s < n ? s : n - 1;
			const float* co = coords + (size_t)sc * TRAIN_COORD_FLOATS;
			half8 enc = encode_level_pair_tcnn(M.grid, s_lv, lane >> 4, co[0], co[1], co[2]);
			MlpOut mo = mlp_pass(s_w, lane, enc, sh4_from_dir(lane >> 4, co[4], co[5], co[6]));
			if (s < n && lane < 16) {
				union { half_t h; uint16_t u; } cv;
				cv.h = mo.rgb[0]; out[(size_t)s * 4 + 0] = cv.u;
				cv.h = mo.rgb[1]; out[(size_t)s * 4 + 1] = cv.u;
				cv.h = mo.rgb[2]; out[(size_t)s * 4 + 2] = cv.u;
				cv.h = mo.sigma;  out[(size_t)s * 4 + 3] = cv.u;
			}
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
// The fused forward + backward + weight-gradient kernel.
constexpr int XROW = 256 * 2 + 16; // bytes per sample row of the activation image: enc 32 | hd 64 | rin 32 | h1 64 | h2 64 halves (+ pad)
constexpr int YROW = 64 * 2 + 16;  // bytes per sample row of the gradient image: up to 64 halves
constexpr int X_ENC = 0, X_HD = 32, X_RIN = 96, X_H1 = 128, X_H2 = 192;
constexpr int WAVE_SCRATCH = 16 * XROW + 16 * YROW;
constexpr int N_MLP_PARAMS = 64 * 32 + 16 * 64 + 64 * 32 + 64 * 64 + 16 * 64; // 10240
constexpr int OFF_D0 = 0, OFF_D1 = 2048, OFF_R0 = 3072, OFF_R1 = 5120, OFF_R2 = 9216;

NGP_DEV floatx4 mfma_k16(half4 a, half4 b, floatx4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }
NGP_DEV half4 ld_kfrag(const uint2* s_k, int f, int lane) {
	union { uint2 u; half4 h; } cv;
	cv.u = s_k[f * 64 + lane];
	return cv.h;
}
// transposed read of a [sample row][feature] image: the 16x16x16 operand whose 16 rows/columns are features
// f0..f0+15 and whose k index is the sample (cdna_hip_programming.md T10: lane 4q+p of group g addresses row 4g+q,
// columns 4p..4p+3; lane i receives column i of the four rows)
NGP_DEV half4 tr_read(const char* image, int row_bytes, int f0, int lane) {
	const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
	const char* addr = image + (4 * g + q) * row_bytes + (f0 + 4 * p) * 2;
	short4v r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)addr);
	return __builtin_bit_cast(half4, r);
}
// lane (h, c) owns rows 4h..4h+3 of a 16-feature tile for sample c: 8 bytes at [c][f0 + 4h]
NGP_DEV void st_tile4(char* image, int row_bytes, int f0, int lane, half4 v) {
	*(half4*)(image + (lane & 15) * row_bytes + (f0 + 4 * (lane >> 4)) * 2) = v;
}
NGP_DEV void st_pair(char* image, int row_bytes, int f0, int lane, half8 v) { // packed tiles f0.. and f0+16..
	half4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
	st_tile4(image, row_bytes, f0, lane, lo);
	st_tile4(image, row_bytes, f0 + 16, lane, hi);
}
NGP_DEV half8 pack_masked(floatx4 lo, floatx4 hi, half8 act) { // ReLU backward + round to fp16
	half8 r;
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		r[j] = act[j] > (half_t)0 ? (half_t)lo[j] : (half_t)0;
		r[4 + j] = act[4 + j] > (half_t)0 ? (half_t)hi[j] : (half_t)0;
	}
	return r;
}

__global__ __launch_bounds__(BLOCK, 1) void train_backward_kernel(const ModelParams M, const uint4* __restrict__ frags, const uint2* __restrict__ kfrags,
                                                                  const uint32_t* __restrict__ counters, uint32_t target_batch, const float* __restrict__ coords,
                                                                  const uint16_t* __restrict__ dloss, float* __restrict__ grad /* [n_params] fp32, tcnn parameter order */,
                                                                  uint32_t n_matrix_params, float* __restrict__ block_partials /* [gridDim.x][N_MLP_PARAMS] */) {
	extern __shared__ char s_dyn[];
	uint4* s_w = (uint4*)s_dyn;                                   // N_TFRAGS * 64 * 16 B
	uint2* s_k = (uint2*)(s_dyn + N_TFRAGS * 64 * 16);            // N_KFRAGS * 64 * 8 B
	char* s_scratch = s_dyn + N_TFRAGS * 64 * 16 + N_KFRAGS * 64 * 8;
	__shared__ LevelInfo s_lv[N_LEVELS];
	for (int i = threadIdx.x; i < N_TFRAGS * 64; i += BLOCK) s_w[i] = frags[i];
	for (int i = threadIdx.x; i < N_KFRAGS * 64; i += BLOCK) s_k[i] = kfrags[i];
	if (threadIdx.x < N_LEVELS) s_lv[threadIdx.x] = M.levels[threadIdx.x];
	__syncthreads();
	const int lane = threadIdx.x & 63, c = lane & 15, h = lane >> 4, wave_in_block = threadIdx.x >> 6;
	char* ximg = s_scratch + wave_in_block * WAVE_SCRATCH;
	char* yimg = ximg + 16 * XROW;
	const uint32_t n_raw = counters[2] < target_batch ? counters[2] : target_batch;
	const uint32_t n_passes = (n_raw + 15u) / 16u;
	const floatx4 zero = {0.f, 0.f, 0.f, 0.f};

	floatx4 accD0[4][2], accD1[4], accR0[4][2], accR1[4][4], accR2[4];
#pragma unroll
	for (int a = 0; a < 4; ++a) {
		accD1[a] = zero; accR2[a] = zero;
#pragma unroll
		for (int b = 0; b < 2; ++b) { accD0[a][b] = zero; accR0[a][b] = zero; }
#pragma unroll
		for (int b = 0; b < 4; ++b) accR1[a][b] = zero;
	}

	const uint32_t wave = (blockIdx.x * BLOCK + threadIdx.x) >> 6, n_waves = gridDim.x * (BLOCK / 64);
	for (uint32_t pass = wave; pass < n_passes; pass += n_waves) {
		const uint32_t s = pass * 16u + c; // < target_batch rounded up to 16 (buffers are padded by the host)
		const float* co = coords + (size_t)s * TRAIN_COORD_FLOATS;
		const float px = co[0], py = co[1], pz = co[2];
		// ---- forward, activations into the image
		const half8 enc = encode_level_pair_tcnn(M.grid, s_lv, h, px, py, pz);
		st_pair(ximg, XROW, X_ENC, lane, enc);
		floatx4 d0 = mfma16(ld_frag(s_w, FRAG_D0 + 0, lane), enc, zero);
		floatx4 d1 = mfma16(ld_frag(s_w, FRAG_D0 + 1, lane), enc, zero);
		floatx4 d2 = mfma16(ld_frag(s_w, FRAG_D0 + 2, lane), enc, zero);
		floatx4 d3 = mfma16(ld_frag(s_w, FRAG_D0 + 3, lane), enc, zero);
		const half8 hd0 = relu_pack(d0, d1), hd1 = relu_pack(d2, d3);
		st_pair(ximg, XROW, X_HD, lane, hd0);
		st_pair(ximg, XROW, X_HD + 32, lane, hd1);
		floatx4 dens = mfma16(ld_frag(s_w, FRAG_D1 + 0, lane), hd0, zero);
		dens = mfma16(ld_frag(s_w, FRAG_D1 + 1, lane), hd1, dens);
		const Sh4 shq = sh4_from_dir(h, co[4], co[5], co[6]);
		half8 rin;
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			rin[j] = (half_t)dens[j];
			rin[4 + j] = shq.v[j];
		}
		st_pair(ximg, XROW, X_RIN, lane, rin);
		d0 = mfma16(ld_frag(s_w, FRAG_R0 + 0, lane), rin, zero);
		d1 = mfma16(ld_frag(s_w, FRAG_R0 + 1, lane), rin, zero);
		d2 = mfma16(ld_frag(s_w, FRAG_R0 + 2, lane), rin, zero);
		d3 = mfma16(ld_frag(s_w, FRAG_R0 + 3, lane), rin, zero);
		const half8 h10 = relu_pack(d0, d1), h11 = relu_pack(d2, d3);
		st_pair(ximg, XROW, X_H1, lane, h10);
		st_pair(ximg, XROW, X_H1 + 32, lane, h11);
		d0 = mfma16(ld_frag(s_w, FRAG_R1 + 0, lane), h10, zero);
		d0 = mfma16(ld_frag(s_w, FRAG_R1 + 1, lane), h11, d0);
		d1 = mfma16(ld_frag(s_w, FRAG_R1 + 2, lane), h10, zero);
		d1 = mfma16(ld_frag(s_w, FRAG_R1 + 3, lane), h11, d1);
		d2 = mfma16(ld_frag(s_w, FRAG_R1 + 4, lane), h10, zero);
		d2 = mfma16(ld_frag(s_w, FRAG_R1 + 5, lane), h11, d2);
		d3 = mfma16(ld_frag(s_w, FRAG_R1 + 6, lane), h10, zero);
		d3 = mfma16(ld_frag(s_w, FRAG_R1 + 7, lane), h11, d3);
		const half8 h20 = relu_pack(d0, d1), h21 = relu_pack(d2, d3);
		st_pair(ximg, XROW, X_H2, lane, h20);
		st_pair(ximg, XROW, X_H2 + 32, lane, h21);

		// ---- backward. dL/d(rgb network output): rows 0..2 of the padded 16 (extract_rgb, nerf_network.h:206)
		const uint16_t* dl = dloss + (size_t)s * 4;
		half4 dout = {(half_t)0, (half_t)0, (half_t)0, (half_t)0};
		float dsigma = 0.0f;
		if (h == 0) {
			union { uint16_t u; half_t hv; } cv;
			cv.u = dl[0]; dout[0] = cv.hv;
			cv.u = dl[1]; dout[1] = cv.hv;
			cv.u = dl[2]; dout[2] = cv.hv;
			cv.u = dl[3]; dsigma = (float)cv.hv;
		}
		// W_R2: dW = dOut . h2^T
		st_tile4(yimg, YROW, 0, lane, dout);
		{
			const half4 a = tr_read(yimg, YROW, 0, lane);
#pragma unroll
			for (int tj = 0; tj < 4; ++tj) accR2[tj] = mfma_k16(a, tr_read(ximg, XROW, X_H2 + 16 * tj, lane), accR2[tj]);
		}
		// dH2 = W_R2^T . dOut, masked by h2 > 0
		floatx4 g0 = mfma_k16(ld_kfrag(s_k, KFRAG_R2T + 0, lane), dout, zero);
		floatx4 g1 = mfma_k16(ld_kfrag(s_k, KFRAG_R2T + 1, lane), dout, zero);
		floatx4 g2 = mfma_k16(ld_kfrag(s_k, KFRAG_R2T + 2, lane), dout, zero);
		floatx4 g3 = mfma_k16(ld_kfrag(s_k, KFRAG_R2T + 3, lane), dout, zero);
		half8 p0 = pack_masked(g0, g1, h20), p1 = pack_masked(g2, g3, h21);
		st_pair(yimg, YROW, 0, lane, p0);
		st_pair(yimg, YROW, 32, lane, p1);
		// W_R1: dW = dH2 . h1^T
		{
			half4 b[4];
#pragma unroll
			for (int tj = 0; tj < 4; ++tj) b[tj] = tr_read(ximg, XROW, X_H1 + 16 * tj, lane);
#pragma unroll
			for (int ti = 0; ti < 4; ++ti) {
				const half4 a = tr_read(yimg, YROW, 16 * ti, lane);
#pragma unroll
				for (int tj = 0; tj < 4; ++tj) accR1[ti][tj] = mfma_k16(a, b[tj], accR1[ti][tj]);
			}
		}
		// dH1 = W_R1^T . dH2, masked by h1 > 0
		g0 = mfma16(ld_frag(s_w, TFRAG_R1T + 0, lane), p0, zero);
		g0 = mfma16(ld_frag(s_w, TFRAG_R1T + 1, lane), p1, g0);
		g1 = mfma16(ld_frag(s_w, TFRAG_R1T + 2, lane), p0, zero);
		g1 = mfma16(ld_frag(s_w, TFRAG_R1T + 3, lane), p1, g1);
		g2 = mfma16(ld_frag(s_w, TFRAG_R1T + 4, lane), p0, zero);
		g2 = mfma16(ld_frag(s_w, TFRAG_R1T + 5, lane), p1, g2);
		g3 = mfma16(ld_frag(s_w, TFRAG_R1T + 6, lane), p0, zero);
		g3 = mfma16(ld_frag(s_w, TFRAG_R1T + 7, lane), p1, g3);
		p0 = pack_masked(g0, g1, h10);
		p1 = pack_masked(g2, g3, h11);
		st_pair(yimg, YROW, 0, lane, p0);
		st_pair(yimg, YROW, 32, lane, p1);
		// W_R0: dW = dH1 . rin^T
		{
			const half4 b0 = tr_read(ximg, XROW, X_RIN, lane), b1 = tr_read(ximg, XROW, X_RIN + 16, lane);
#pragma unroll
			for (int ti = 0; ti < 4; ++ti) {
				const half4 a = tr_read(yimg, YROW, 16 * ti, lane);
				accR0[ti][0] = mfma_k16(a, b0, accR0[ti][0]);
				accR0[ti][1] = mfma_k16(a, b1, accR0[ti][1]);
			}
		}
		// d(density network output) = rows 0..15 of W_R0^T . dH1, plus dL/d sigma on row 0 (add_density_gradient, :235)
		floatx4 u = mfma16(ld_frag(s_w, TFRAG_R0T + 0, lane), p0, zero);
		u = mfma16(ld_frag(s_w, TFRAG_R0T + 1, lane), p1, u);
		half4 ddens;
#pragma unroll
		for (int j = 0; j < 4; ++j) ddens[j] = (half_t)u[j];
		if (h == 0) ddens[0] = (half_t)((float)ddens[0] + dsigma); // fp16 + fp16 like add_density_gradient
		st_tile4(yimg, YROW, 0, lane, ddens);
		// W_D1: dW = dDens . hd^T
		{
			const half4 a = tr_read(yimg, YROW, 0, lane);
#pragma unroll
			for (int tj = 0; tj < 4; ++tj) accD1[tj] = mfma_k16(a, tr_read(ximg, XROW, X_HD + 16 * tj, lane), accD1[tj]);
		}
		// dHd = W_D1^T . dDens, masked by hd > 0
		g0 = mfma_k16(ld_kfrag(s_k, KFRAG_D1T + 0, lane), ddens, zero);
		g1 = mfma_k16(ld_kfrag(s_k, KFRAG_D1T + 1, lane), ddens, zero);
		g2 = mfma_k16(ld_kfrag(s_k, KFRAG_D1T + 2, lane), ddens, zero);
		g3 = mfma_k16(ld_kfrag(s_k, KFRAG_D1T + 3, lane), ddens, zero);
		p0 = pack_masked(g0, g1, hd0);
		p1 = pack_masked(g2, g3, hd1);
		st_pair(yimg, YROW, 0, lane, p0);
		st_pair(yimg, YROW, 32, lane, p1);
		// W_D0: dW = dHd . enc^T
		{
			const half4 b0 = tr_read(ximg, XROW, X_ENC, lane), b1 = tr_read(ximg, XROW, X_ENC + 16, lane);
#pragma unroll
			for (int ti = 0; ti < 4; ++ti) {
				const half4 a = tr_read(yimg, YROW, 16 * ti, lane);
				accD0[ti][0] = mfma_k16(a, b0, accD0[ti][0]);
				accD0[ti][1] = mfma_k16(a, b1, accD0[ti][1]);
			}
		}
		// dEnc = W_D0^T . dHd: tile 0 rows 4h+r = level h feature r, tile 1 = level h+4 feature r
		floatx4 e0 = mfma16(ld_frag(s_w, TFRAG_D0T + 0, lane), p0, zero);
		e0 = mfma16(ld_frag(s_w, TFRAG_D0T + 1, lane), p1, e0);
		floatx4 e1 = mfma16(ld_frag(s_w, TFRAG_D0T + 2, lane), p0, zero);
		e1 = mfma16(ld_frag(s_w, TFRAG_D0T + 3, lane), p1, e1);
		// grid scatter (tcnn kernel_grid_backward): corner k of level l receives weight_k * dL/d feature. The 16 lanes of a
		// DPP row are 16 consecutive samples of (mostly) one ray: on the coarse levels they sit in one cell, where per-lane
		// atomics would all land on the same 8 entries and serialise. Rows whose lanes share the cell add their 32 products
		// with row rotations first and issue 2 atomics per lane instead of 32.
		const bool live = s < n_raw;
#pragma unroll
		for (int l = 0; l < 2; ++l) {
			const floatx4 g = l ? e1 : e0;
			// the fp16 gradient the reference's MLP backward hands to the encoding; padding samples contribute nothing
			float gf[4];
#pragma unroll
			for (int f = 0; f < 4; ++f) gf[f] = live ? (float)(half_t)g[f] : 0.0f;
			const LevelInfo& L = s_lv[h + 4 * l];
			const CellPos p = level_cell(L, px, py, pz);
			CornerSet cs;
			level_corners(L, p, cs);
			float w[8];
			corner_weights(p, w);
			float* gbase = grad + n_matrix_params;
			// same cell as the lane one to the right within the row <=> the whole row shares the cell
			const uint32_t nx = (uint32_t)__builtin_amdgcn_mov_dpp((int)p.gx, 0x121, 0xF, 0xF, false), ny = (uint32_t)__builtin_amdgcn_mov_dpp((int)p.gy, 0x121, 0xF, 0xF, false),
			               nz = (uint32_t)__builtin_amdgcn_mov_dpp((int)p.gz, 0x121, 0xF, 0xF, false);
			const unsigned long long same = __ballot(nx == p.gx && ny == p.gy && nz == p.gz);
			const bool row_uniform = ((same >> (16 * h)) & 0xFFFFull) == 0xFFFFull;
			// Float atomics run at the memory side at a fixed rate of 64-B requests (MI355X_MICROARCH.md, Global float atomics):
			// the four features of an entry are one 16-B segment, so four neighbouring lanes add them in ONE request.
			if (row_uniform) {
				// after four rotate-adds every lane of the row holds the row's sum of w_k * g_f; two instructions, lane c adds
				// feature c & 3 of corner (c >> 2) and of corner 4 + (c >> 2)
				float mine_a = 0.0f, mine_b = 0.0f;
#pragma unroll
				for (int k = 0; k < 8; ++k) {
#pragma unroll
					for (int f = 0; f < 4; ++f) {
						float v = w[k] * gf[f];
						v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false)); // row_ror:8
						v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false)); // row_ror:4
						v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x122, 0xF, 0xF, false)); // row_ror:2
						v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x121, 0xF, 0xF, false)); // row_ror:1
						if ((c & 3) == f && (c >> 2) == (k & 3)) {
							if (k & 4) mine_b = v;
							else mine_a = v;
						}
					}
				}
				uint32_t idx_a = cs.index[0], idx_b = cs.index[4];
#pragma unroll
				for (int k = 1; k < 4; ++k) {
					idx_a = (c >> 2) == k ? cs.index[k] : idx_a;
					idx_b = (c >> 2) == k ? cs.index[4 + k] : idx_b;
				}
				if (mine_a != 0.0f) atomicAdd(gbase + (size_t)(idx_a >> 1) + (c & 3), mine_a);
				if (mine_b != 0.0f) atomicAdd(gbase + (size_t)(idx_b >> 1) + (c & 3), mine_b);
			} else {
				// An octet (8 consecutive samples) takes turns: in turn t its lanes (e = which corner of an x-pair, f = feature)
				// add sample t's corners two at a time. The corners of an x-pair are neighbouring entries whenever x is even
				// (hashed levels: index ^ 1) or the level is dense, so about half of the pairs leave as ONE request.
				// ds_swizzle (bit mode) broadcasts lane t of every octet: lane' = (lane & 0x18) | t.
#define NGP_OCT_BCAST_F(x, t) __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, (x)), 0x18 | ((t) << 5)))
#define NGP_OCT_BCAST_U(x, t) (uint32_t)__builtin_amdgcn_ds_swizzle((int)(x), 0x18 | ((t) << 5))
#define NGP_OCT_TURN(t)                                                                                                  \
				{                                                                                                        \
					const float g0 = NGP_OCT_BCAST_F(gf[0], t), g1 = NGP_OCT_BCAST_F(gf[1], t), g2 = NGP_OCT_BCAST_F(gf[2], t), g3 = NGP_OCT_BCAST_F(gf[3], t); \
					const float gm = (c & 2) ? ((c & 1) ? g3 : g2) : ((c & 1) ? g1 : g0);                                 \
					if (__builtin_amdgcn_ballot_w64(gm != 0.0f) != 0ull) {                                                \
						_Pragma("unroll") for (int pr = 0; pr < 4; ++pr) {                                                \
							const uint32_t i0 = NGP_OCT_BCAST_U(cs.index[2 * pr], t), i1 = NGP_OCT_BCAST_U(cs.index[2 * pr + 1], t); \
							const float w0 = NGP_OCT_BCAST_F(w[2 * pr], t), w1 = NGP_OCT_BCAST_F(w[2 * pr + 1], t);       \
							const float v = ((c & 4) ? w1 : w0) * gm;                                                     \
							if (v != 0.0f) atomicAdd(gbase + (size_t)(((c & 4) ? i1 : i0) >> 1) + (c & 3), v);            \
						}                                                                                                 \
					}                                                                                                     \
				}
				NGP_OCT_TURN(0)
				NGP_OCT_TURN(1)
				NGP_OCT_TURN(2)
				NGP_OCT_TURN(3)
				NGP_OCT_TURN(4)
				NGP_OCT_TURN(5)
				NGP_OCT_TURN(6)
				NGP_OCT_TURN(7)
#undef NGP_OCT_TURN
#undef NGP_OCT_BCAST_F
#undef NGP_OCT_BCAST_U
			}
		}
	}

	// ---- block-wise reduction of the weight gradients, then one atomic per weight and block
	__syncthreads();
	float* s_red = (float*)s_scratch; // 4 * WAVE_SCRATCH >= 10240 floats (checked by the host launcher)
	for (int i = threadIdx.x; i < N_MLP_PARAMS; i += BLOCK) s_red[i] = 0.0f;
	__syncthreads();
	// accumulator tile (ti, tj) of a layer with n_in inputs: D[row 4h+r][col c] = dW[16 ti + 4h + r][16 tj + c]
#pragma unroll
	for (int ti = 0; ti < 4; ++ti) {
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const int row = 16 * ti + 4 * h + r;
#pragma unroll
			for (int tj = 0; tj < 2; ++tj) {
				atomicAdd(&s_red[OFF_D0 + row * 32 + 16 * tj + c], accD0[ti][tj][r]);
				atomicAdd(&s_red[OFF_R0 + row * 32 + 16 * tj + c], accR0[ti][tj][r]);
			}
#pragma unroll
			for (int tj = 0; tj < 4; ++tj) atomicAdd(&s_red[OFF_R1 + row * 64 + 16 * tj + c], accR1[ti][tj][r]);
		}
	}
#pragma unroll
	for (int tj = 0; tj < 4; ++tj) {
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			atomicAdd(&s_red[OFF_D1 + (4 * h + r) * 64 + 16 * tj + c], accD1[tj][r]);
			atomicAdd(&s_red[OFF_R2 + (4 * h + r) * 64 + 16 * tj + c], accR2[tj][r]);
		}
	}
	__syncthreads();
	// every block leaves its sums in its own row; train_reduce_partials_kernel adds the rows (256 blocks hammering the
	// same 10240 addresses with atomics serialise at the memory side)
	float* mine = block_partials + (size_t)blockIdx.x * N_MLP_PARAMS;
	for (int i = threadIdx.x; i < N_MLP_PARAMS; i += BLOCK) mine[i] = s_red[i];
}

__global__ void train_reduce_partials_kernel(const float* __restrict__ block_partials, uint32_t n_blocks, float* __restrict__ grad) {
	// block = 64 consecutive weights x 4 row groups: coalesced 256-B reads, rows split over the four waves
	__shared__ float s_part[4][64];
	const uint32_t w = blockIdx.x * 64u + (threadIdx.x & 63u), g = threadIdx.x >> 6;
	float acc = 0.0f;
	for (uint32_t b = g; b < n_blocks; b += 4) acc += block_partials[(size_t)b * N_MLP_PARAMS + w];
	s_part[g][threadIdx.x & 63u] = acc;
	__syncthreads();
	if (g == 0) grad[w] += (s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + (s_part[2][threadIdx.x] + s_part[3][threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------------------
// tcnn Adam (optimizers/adam.h adam_step) + Ema (optimizers/ema.h ema_step) in one pass over the parameters; the
// gradient is cleared for the next step. Hash-grid entries no sample touched (gradient exactly 0) are skipped, as
// in tcnn, and keep their own step count for the bias correction.
__global__ void train_optimizer_kernel(const AdamParams A, float* __restrict__ weights_fp32, uint16_t* __restrict__ weights, float* __restrict__ grad,
                                       float* __restrict__ first_moments, float* __restrict__ second_moments, uint32_t* __restrict__ param_steps,
                                       float* __restrict__ ema_tmp, uint16_t* __restrict__ weights_ema) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= A.n_params) return;
	float gradient = grad[i] / A.loss_scale;
	grad[i] = 0.0f;
	bool update = true;
	if (i >= A.n_matrix) {
		if (!A.optimize_non_matrix || gradient == 0.0f) update = false;
	} else if (!A.optimize_matrix) {
		update = false;
	}
	if (update) {
		const float weight_fp = weights_fp32[i];
		if (i < A.n_matrix) gradient += A.l2_reg * weight_fp; // no L2 regularisation of the encoding
		const float first_moment = first_moments[i] = A.beta1 * first_moments[i] + (1.0f - A.beta1) * gradient;
		const float second_moment = second_moments[i] = A.beta2 * second_moments[i] + (1.0f - A.beta2) * gradient * gradient;
		const uint32_t current_step = ++param_steps[i];
		const float lr = A.learning_rate * sqrtf(1.0f - powf(A.beta2, (float)current_step)) / (1.0f - powf(A.beta1, (float)current_step));
		const float effective_lr = lr / (sqrtf(second_moment) + A.epsilon);
		const float new_weight = weight_fp - effective_lr * first_moment;
		weights_fp32[i] = new_weight;
		weights[i] = float_to_half_bits(new_weight);
	}
	if (weights_ema) {
		const float filtered = (ema_tmp[i] * A.ema_decay * A.ema_debias_old + half_bits_to_float(weights[i]) * (1.0f - A.ema_decay)) * A.ema_debias_new;
		ema_tmp[i] = filtered;
		weights_ema[i] = float_to_half_bits(filtered);
	}
}

// the xor layout of one level of a parameter table (ngp_api.cpp build_xor_layout), on the device: one thread per
// destination entry -- a copy for hashed levels, the padded lattice [0, res]^3 for dense ones
__global__ void train_xor_layout_kernel(const LevelInfo L, const uint2* __restrict__ src, char* __restrict__ dst, uint32_t n) {
	const uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
	if (t >= n) return;
	if (L.mask8 != 0xFFFFFFFFu) { // hashed levels, and dense indices whose uint32 strides wrapped (build_xor_layout): entry for entry
		*(uint2*)(dst + L.base8 + (size_t)t * 8u) = src[L.offset + t];
		return;
	}
	const uint32_t r1 = L.res + 1u;
	const uint32_t x = t % r1, y = (t / r1) % r1, z = t / (r1 * r1);
	const uint32_t e = (x + y * L.res + z * L.res * L.res) % L.size;
	*(uint2*)(dst + L.base8 + ((x << 3) ^ (y * L.mul_y8) ^ (z * L.mul_z8))) = src[L.offset + e];
}

// CudaRenderBuffer::overlay_image at alpha 1 (src/render_buffer.cu:344-414; Testbed::render_frame_epilogue shows the
// training image of the current view this way when m_render_ground_truth is set, src/testbed.cu:4979-4994): the image
// resampled (nearest) around the screen centre, blended over the background, exposure, output colour space.
__global__ void overlay_image_kernel(int width, int height, float exposure, float4 background, const TrainImage im, int color_space, int to_srgb, int fov_axis, float zoom,
                                     float4* __restrict__ out) {
	const int x = threadIdx.x + blockDim.x * blockIdx.x, y = threadIdx.y + blockDim.y * blockIdx.y;
	if (x >= width || y >= height) return;
	const float scale = (float)im.res[fov_axis] / (float)(fov_axis ? height : width);
	float fx = (float)x + 0.5f, fy = (float)y + 0.5f;
	fx -= (float)width * 0.5f; fx /= zoom; fx += 0.5f * (float)width;
	fy -= (float)height * 0.5f; fy /= zoom; fy += 0.5f * (float)height;
	const float u = (fx - (float)width * 0.5f) * scale + (float)im.res[0] * 0.5f;
	const float v = (fy - (float)height * 0.5f) * scale + (float)im.res[1] * 0.5f;
	const int srcx = (int)__builtin_floorf(u), srcy = (int)__builtin_floorf(v);
	float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
	if (srcx >= 0 && srcy >= 0 && srcx < im.res[0] && srcy < im.res[1]) val = read_training_pixel(im, ((float)srcx + 0.5f) / (float)im.res[0], ((float)srcy + 0.5f) / (float)im.res[1]);
	f3 color = mk3(val.x, val.y, val.z);
	f3 bg = mk3(background.x, background.y, background.z);
	if (color_space != 1) { // the background colour is given in sRGB
		bg = mk3(srgb_to_linear(bg.x), srgb_to_linear(bg.y), srgb_to_linear(bg.z));
	} else if (val.w > 0.0f) {
		color = mk3(linear_to_srgb(color.x / val.w) * val.w, linear_to_srgb(color.y / val.w) * val.w, linear_to_srgb(color.z / val.w) * val.w);
	} else {
		color = mk3(0.f, 0.f, 0.f);
	}
	const float weight = (1.0f - val.w) * background.w;
	color = add3(color, scale3(bg, weight));
	const float alpha = val.w + weight;
	if (color_space == 1) color = mk3(srgb_to_linear(color.x), srgb_to_linear(color.y), srgb_to_linear(color.z)); // tonemap(): to linear,
	color = scale3(color, __builtin_exp2f(exposure));                                                              // exposure,
	if (to_srgb) color = mk3(linear_to_srgb(color.x), linear_to_srgb(color.y), linear_to_srgb(color.z));           // output space
	out[(size_t)y * width + x] = make_float4(color.x, color.y, color.z, alpha);
}

__global__ void train_loss_sum_kernel(const float* __restrict__ loss, uint32_t n, float* __restrict__ out) {
	__shared__ float s[256];
	float acc = 0.0f;
	for (uint32_t i = threadIdx.x; i < n; i += 256) acc += loss[i];
	s[threadIdx.x] = acc;
	__syncthreads();
	for (int k = 128; k > 0; k >>= 1) {
		if ((int)threadIdx.x < k) s[threadIdx.x] += s[threadIdx.x + k];
		__syncthreads();
	}
	if (threadIdx.x == 0) *out = s[0];
}

} // namespace

// ---------------------------------------------------------------------------------------------------------
void launch_train_generate_samples(const ModelParams& M, const TrainStepParams& P, const TrainImage* images, const TrainBatch& B, hipStream_t stream) {
	const uint32_t rays_per_block = GEN_RAYS_PER_WAVE * GEN_WAVES_PER_BLOCK;
	const size_t lds = sizeof(float) * NERF_STEPS * rays_per_block + (size_t)(M.max_cascade + 1u) * COARSE_WORDS_PER_MIP * sizeof(uint32_t);
	hipLaunchKernelGGL(train_generate_samples_kernel, dim3((P.n_rays + rays_per_block - 1) / rays_per_block), dim3(64 * GEN_WAVES_PER_BLOCK), lds, stream, M, P, images, B);
}
void launch_train_inference(const ModelParams& M, const uint4* frags, const uint32_t* counters, uint32_t max_samples, const float* coords, uint16_t* out, int n_cus,
                            hipStream_t stream) {
	uint32_t blocks = (max_samples + 255) / 256;
	const uint32_t cap = (uint32_t)n_cus * 8u;
	hipLaunchKernelGGL(train_inference_kernel, dim3(blocks < cap ? blocks : cap), dim3(BLOCK), 0, stream, M, frags, counters, max_samples, coords, out);
}
void launch_train_loss(const ModelParams& M, const TrainStepParams& P, const TrainImage* images, const TrainBatch& B, hipStream_t stream) {
	hipLaunchKernelGGL(train_loss_kernel, dim3((P.n_rays + 3) / 4), dim3(BLOCK), 0, stream, M, P, images, B); // one wave per ray
	hipLaunchKernelGGL(train_rollover_kernel, dim3((P.target_batch + 255) / 256), dim3(256), 0, stream, P.target_batch, B.counters, B.dloss, B.coords_compacted);
}
void launch_train_build_fragments(const uint16_t* params, uint4* frags, uint2* kfrags, hipStream_t stream) {
	hipLaunchKernelGGL(train_build_fragments_kernel, dim3((N_TFRAGS * 64 * 8 + 255) / 256), dim3(256), 0, stream, params, (uint16_t*)frags, (uint16_t*)kfrags);
}
void launch_train_backward(const ModelParams& M, const uint4* frags, const uint2* kfrags, const uint32_t* counters, uint32_t target_batch, const float* coords,
                           const uint16_t* dloss, float* grad, uint32_t n_matrix_params, float* block_partials, int n_blocks, hipStream_t stream) {
	static_assert(4 * WAVE_SCRATCH >= N_MLP_PARAMS * 4, "the block reduction reuses the waves' scratch");
	const size_t lds = (size_t)N_TFRAGS * 64 * 16 + (size_t)N_KFRAGS * 64 * 8 + 4 * (size_t)WAVE_SCRATCH;
	static bool configured[64] = {}; // per device: a process may hold contexts on several GPUs
	int device = 0;
	NGP_HIP_CHECK(hipGetDevice(&device));
	if (device < 0 || device >= 64 || !configured[device]) {
		NGP_HIP_CHECK(hipFuncSetAttribute((const void*)train_backward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		if (device >= 0 && device < 64) configured[device] = true;
	}
	hipLaunchKernelGGL(train_backward_kernel, dim3((uint32_t)n_blocks), dim3(BLOCK), lds, stream, M, frags, kfrags, counters, target_batch, coords, dloss, grad, n_matrix_params,
	                   block_partials);
	hipLaunchKernelGGL(train_reduce_partials_kernel, dim3(N_MLP_PARAMS / 64), dim3(256), 0, stream, block_partials, (uint32_t)n_blocks, grad);
}
size_t train_backward_partials_floats(int n_blocks) { return (size_t)n_blocks * N_MLP_PARAMS; }
void launch_train_optimizer(const AdamParams& A, float* weights_fp32, uint16_t* weights, float* grad, float* m1, float* m2, uint32_t* steps, float* ema_tmp,
                            uint16_t* weights_ema, hipStream_t stream) {
	hipLaunchKernelGGL(train_optimizer_kernel, dim3((A.n_params + 255) / 256), dim3(256), 0, stream, A, weights_fp32, weights, grad, m1, m2, steps, ema_tmp, weights_ema);
}
void launch_train_xor_layout(const ModelParams& M, const uint2* src, char* dst, hipStream_t stream) {
	for (int l = 0; l < N_LEVELS; ++l) {
		const LevelInfo& L = M.levels[l];
		if (L.xor_disabled) continue; // no xor form: the render kernels read this level from the tcnn-order table
		const uint32_t n = L.mask8 != 0xFFFFFFFFu ? L.size : (L.res + 1u) * (L.res + 1u) * (L.res + 1u);
		hipLaunchKernelGGL(train_xor_layout_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, L, src, dst, n);
	}
}
void launch_overlay_image(int width, int height, float exposure, const float* background4, const TrainImage& im, int color_space, int to_srgb, int fov_axis, float zoom, float4* out,
                          hipStream_t stream) {
	hipLaunchKernelGGL(overlay_image_kernel, dim3((width + 15) / 16, (height + 7) / 8), dim3(16, 8), 0, stream, width, height, exposure,
	                   make_float4(background4[0], background4[1], background4[2], background4[3]), im, color_space, to_srgb, fov_axis, zoom, out);
}
void launch_train_loss_sum(const float* loss, uint32_t n, float* out, hipStream_t stream) { hipLaunchKernelGGL(train_loss_sum_kernel, dim3(1), dim3(256), 0, stream, loss, n, out); }

} // namespace ngp
