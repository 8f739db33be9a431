// Host side of the training step: Testbed::reset_network, Testbed::train, Testbed::train_nerf,
// train_nerf_step, NerfCounters (src/testbed.cu:3820-4210, 4364-4470; src/testbed_nerf.cu:2914-3431) and the optimizer
// chain of configs/nerf/base.json (tcnn Ema > ExponentialDecay > Adam). Kernels: train_kernels.hip.
#include "ngp_host.h"
#include "jpeg_decode.h"
#include "png_decode.h"

#include <cmath>
#include <cstring>

using namespace ngp;

namespace {

constexpr uint32_t BATCH_SIZE_GRANULARITY = 128; // tcnn

uint32_t next_multiple(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

void require_device_model(ngp_ctx* ctx) {
	if (ctx->device < 0) throw std::runtime_error("this context has no HIP device (host-only); there is no CPU fallback");
	if (!ctx->model_loaded) throw std::runtime_error("No network available.");
}

void default_opts(ngp_training_opts& o) {
	memset(&o, 0, sizeof(o));
	o.struct_size = sizeof(o);
	o.loss_type = NGP_LOSS_HUBER; // configs/nerf/base.json "loss"
	o.random_bg_color = 1;
	o.linear_colors = 0;
	o.snap_to_pixel_centers = 1;
	o.near_distance = 0.1f;
	o.density_grid_decay = 0.95f;
	o.train_network = o.train_encoding = 1;
	o.learning_rate = 1e-2f;
	o.beta1 = 0.9f;
	o.beta2 = 0.99f;
	o.epsilon = 1e-15f;
	o.l2_reg = 1e-6f;
	o.ema_decay = 0.95f;
	o.decay_start = 20000;
	o.decay_interval = 10000;
	o.decay_base = 0.33f;
	o.background_color[0] = o.background_color[1] = o.background_color[2] = 0.f;
	o.color_space = 1; // EColorSpace::SRGB
}

template <typename T>
void dev_alloc(T*& p, size_t n) {
	NGP_HIP_CHECK(hipMalloc((void**)&p, n * sizeof(T)));
}
template <typename T>
void dev_free(T*& p) {
	if (p) (void)hipFree((void*)p);
	p = nullptr;
}

ModelParams training_model(const ngp_ctx* ctx) {
	ModelParams M = ctx->M;
	const TrainState& T = *ctx->train;
	M.grid = (const uint2*)(T.d_weights + T.n_matrix);
	M.xgrid = nullptr; // training kernels read the tcnn-order table only
	M.wfrags = T.d_tfrags;
	return M;
}

// Trainer construction: parameters from the current model (ctx->params), optimizer state zeroed
TrainState& ensure_training(ngp_ctx* ctx) {
	require_device_model(ctx);
	if (ctx->M.wide.width) throw std::runtime_error("training is built for the configs/nerf/base.json network; a Frequency-encoding model (configs/nerf/frequency.json) is inference only");
	if (ctx->train && ctx->train->d_weights) return *ctx->train;
	if (!ctx->train) ctx->train = new TrainState();
	if (ctx->train->opts.struct_size == 0) default_opts(ctx->train->opts);
	TrainState& T = *ctx->train;
	const ngp_model_desc& d = ctx->desc;
	T.n_params = (uint32_t)d.n_params;
	uint64_t ng = 0;
	for (int l = 0; l < N_LEVELS; ++l) ng += (uint64_t)ctx->M.levels[l].size * N_FEATURES;
	T.n_matrix = (uint32_t)(d.n_params - ng);
	if (T.n_matrix != 10240u) throw std::runtime_error("training is built for the configs/nerf/base.json network (10240 matrix weights)");
	const size_t n = T.n_params;
	dev_alloc(T.d_weights_fp32, n);
	dev_alloc(T.d_weights, n);
	dev_alloc(T.d_weights_ema, n);
	dev_alloc(T.d_ema_tmp, n);
	dev_alloc(T.d_grad, n);
	dev_alloc(T.d_m1, n);
	dev_alloc(T.d_m2, n);
	dev_alloc(T.d_steps, n);
	dev_alloc(T.d_tfrags, (size_t)N_TFRAGS * 64);
	dev_alloc(T.d_kfrags, (size_t)N_KFRAGS * 64);
	dev_alloc(T.d_tfrags_inference, (size_t)N_TFRAGS * 64);
	dev_alloc(T.d_kfrags_inference, (size_t)N_KFRAGS * 64);
	dev_alloc(T.d_loss_sum, 1);
	dev_alloc(T.d_partials, train_backward_partials_floats(ctx->n_cus));
	NGP_HIP_CHECK(hipMemcpy(T.d_weights, ctx->params.data(), n * sizeof(uint16_t), hipMemcpyHostToDevice));
	NGP_HIP_CHECK(hipMemcpy(T.d_weights_ema, ctx->params.data(), n * sizeof(uint16_t), hipMemcpyHostToDevice));
	{
		std::vector<float> w(n);
		for (size_t i = 0; i < n; ++i) {
			// fp16 -> fp32 (the snapshot holds fp16; a freshly reset network is exactly representable as well)
			const uint16_t hbits = ctx->params[i];
			const uint32_t sign = (uint32_t)(hbits & 0x8000u) << 16;
			uint32_t exp = (hbits >> 10) & 0x1Fu, man = hbits & 0x3FFu, bits;
			if (exp == 0) {
				if (man == 0) bits = sign;
				else {
					int e = -1;
					do { ++e; man <<= 1; } while (!(man & 0x400u));
					bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13);
				}
			} else if (exp == 31) bits = sign | 0x7F800000u | (man << 13);
			else bits = sign | ((exp + 112u) << 23) | (man << 13);
			memcpy(&w[i], &bits, 4);
		}
		NGP_HIP_CHECK(hipMemcpy(T.d_weights_fp32, w.data(), n * sizeof(float), hipMemcpyHostToDevice));
		NGP_HIP_CHECK(hipMemcpy(T.d_ema_tmp, w.data(), n * sizeof(float), hipMemcpyHostToDevice));
	}
	NGP_HIP_CHECK(hipMemset(T.d_grad, 0, n * sizeof(float)));
	NGP_HIP_CHECK(hipMemset(T.d_m1, 0, n * sizeof(float)));
	NGP_HIP_CHECK(hipMemset(T.d_m2, 0, n * sizeof(float)));
	NGP_HIP_CHECK(hipMemset(T.d_steps, 0, n * sizeof(uint32_t)));
	launch_train_build_fragments(T.d_weights, T.d_tfrags, T.d_kfrags, ctx->stream);
	NGP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	T.rng.seed(1337u); // m_rng = default_rng_t{m_seed}; the density-grid generator took its first draw (set_model_impl)
	(void)T.rng.next_uint();
	T.training_step = 0;
	T.optimizer_step = 0;
	T.lr_factor = 1.0f;
	T.rays_per_batch = 1u << 12;
	T.n_rays_total = 0;
	T.measured_batch_size = T.measured_batch_size_before_compaction = 0;
	T.loss_scalar = 0.f;
	T.images_dirty = true;
	return T;
}

void upload_images(ngp_ctx* ctx, TrainState& T) {
	if (!T.images_dirty && T.d_images) return;
	std::vector<TrainImage> meta;
	for (const TrainingView& v : ctx->dataset.views) {
		if (!v.d_pixels) continue; // views without pixels take no part (n_images_for_training counts loaded images)
		TrainImage im{};
		im.pixels = v.d_pixels;
		im.type = v.image_type;
		im.res[0] = v.resolution[0]; im.res[1] = v.resolution[1];
		im.focal[0] = v.focal_length[0]; im.focal[1] = v.focal_length[1];
		im.principal[0] = v.principal_point[0]; im.principal[1] = v.principal_point[1];
		im.lens_mode = v.lens_mode;
		for (int k = 0; k < 7; ++k) im.lens_params[k] = v.lens_params[k];
		for (int k = 0; k < 12; ++k) im.xform[k] = v.xform[k];
		meta.push_back(im);
	}
	if (meta.empty()) throw std::runtime_error("No training data available."); // Testbed::train, src/testbed.cu:4365-4369
	dev_free(T.d_images);
	dev_alloc(T.d_images, meta.size());
	NGP_HIP_CHECK(hipMemcpy(T.d_images, meta.data(), meta.size() * sizeof(TrainImage), hipMemcpyHostToDevice));
	T.n_images = (uint32_t)meta.size();
	T.images_dirty = false;
}

void ensure_gen_set(TrainState::GenSet& G, uint32_t n_rays, uint32_t max_samples) {
	if (!G.counters) dev_alloc(G.counters, 4);
	if (n_rays > G.cap_rays) {
		dev_free(G.ray_indices); dev_free(G.rays); dev_free(G.numsteps);
		G.cap_rays = n_rays;
		dev_alloc(G.ray_indices, n_rays);
		dev_alloc(G.rays, (size_t)n_rays * 6);
		dev_alloc(G.numsteps, (size_t)n_rays * 2);
	}
	if (max_samples > G.cap_samples) {
		dev_free(G.coords);
		G.cap_samples = max_samples;
		dev_alloc(G.coords, ((size_t)max_samples + 64) * TRAIN_COORD_FLOATS);
	}
}
void ensure_workspace(ngp_ctx* ctx, TrainState& T, uint32_t n_rays, uint32_t max_samples, uint32_t target) {
	TrainBatch& B = T.B;
	if (!T.stream2) {
		NGP_HIP_CHECK(hipStreamCreateWithFlags(&T.stream2, hipStreamNonBlocking));
		NGP_HIP_CHECK(hipEventCreateWithFlags(&T.ev_gen, hipEventDisableTiming));
		NGP_HIP_CHECK(hipEventCreateWithFlags(&T.ev_loss, hipEventDisableTiming));
		NGP_HIP_CHECK(hipHostMalloc((void**)&T.h_counters, 8 * sizeof(uint32_t)));
	}
	if (n_rays > T.cap_loss) { // read by the previous step's loss kernels only: complete (the host waited for ev_loss)
		dev_free(B.loss);
		T.cap_loss = n_rays;
		dev_alloc(B.loss, n_rays);
	}
	if (max_samples > T.cap_out) {
		dev_free(B.mlp_out);
		T.cap_out = max_samples;
		dev_alloc(B.mlp_out, ((size_t)max_samples + 64) * 4);
	}
	if (target > T.cap_target) {
		NGP_HIP_CHECK(hipStreamSynchronize(ctx->stream)); // a backward pass may still read the old buffers
		dev_free(B.coords_compacted); dev_free(B.dloss);
		T.cap_target = target;
		dev_alloc(B.coords_compacted, ((size_t)target + 64) * TRAIN_COORD_FLOATS);
		dev_alloc(B.dloss, ((size_t)target + 64) * 4);
		NGP_HIP_CHECK(hipMemset(B.coords_compacted, 0, ((size_t)target + 64) * TRAIN_COORD_FLOATS * sizeof(float)));
		NGP_HIP_CHECK(hipMemset(B.dloss, 0, ((size_t)target + 64) * 4 * sizeof(uint16_t)));
	}
}

// training_prep_nerf on Testbed::train's schedule (src/testbed.cu:4412-4434)
void training_prep(ngp_ctx* ctx, TrainState& T);

bool prep_due(const TrainState& T, uint32_t step) {
	uint32_t n_prep_to_skip = step / 16u;
	n_prep_to_skip = n_prep_to_skip < 1u ? 1u : (n_prep_to_skip > 16u ? 16u : n_prep_to_skip);
	return step % n_prep_to_skip == 0;
}

// the parameters of the step the state is about to run (train_nerf_step's head, :3183-3260)
TrainStepParams step_params(const ngp_ctx* ctx, const TrainState& T, uint32_t target_batch) {
	const uint32_t max_samples = target_batch * 16; // "somewhat of a worst case", :3185
	const uint32_t max_inference = T.measured_batch_size_before_compaction == 0 ? max_samples
	                                                                           : next_multiple(std::min(T.measured_batch_size_before_compaction, max_samples), BATCH_SIZE_GRANULARITY);
	TrainStepParams P{};
	P.n_rays = T.rays_per_batch;
	P.n_rays_total = T.training_step == 0 ? 0u : T.n_rays_total;
	P.n_images = T.n_images;
	P.max_samples = max_inference;
	P.target_batch = target_batch;
	P.rng = T.rng;
	P.snap_to_pixel_centers = T.opts.snap_to_pixel_centers;
	P.random_bg_color = T.opts.random_bg_color;
	P.linear_colors = T.opts.linear_colors;
	P.color_space = T.opts.color_space;
	P.loss_type = T.opts.loss_type;
	for (int k = 0; k < 3; ++k) P.background[k] = T.opts.background_color[k];
	P.near_distance = T.opts.near_distance;
	P.loss_scale = TRAIN_LOSS_SCALE;
	P.density_grid_mean = ctx->bitfield_mean;
	return P;
}

TrainBatch gen_batch(const TrainState::GenSet& G) {
	TrainBatch B{};
	B.counters = G.counters;
	B.ray_indices = G.ray_indices;
	B.rays = G.rays;
	B.numsteps = G.numsteps;
	B.coords = G.coords;
	return B;
}
void launch_generate(ngp_ctx* ctx, TrainState& T, int set, const TrainStepParams& P, hipStream_t stream) {
	TrainState::GenSet& G = T.gen[set];
	ensure_gen_set(G, P.n_rays, P.max_samples);
	NGP_HIP_CHECK(hipMemsetAsync(G.counters, 0, 4 * sizeof(uint32_t), stream));
	launch_train_generate_samples(training_model(ctx), P, T.d_images, gen_batch(G), stream);
}

// generate_training_samples_nerf + inference + compute_loss_kernel_train_nerf of train_nerf_step
void prepare_batch(ngp_ctx* ctx, TrainState& T, uint32_t target_batch, bool get_loss_scalar) {
	if (target_batch == 0 || target_batch % BATCH_SIZE_GRANULARITY != 0) throw std::runtime_error("the training batch size must be a positive multiple of 128");
	upload_images(ctx, T);
	hipStream_t stream = ctx->stream;
	TrainStepParams P;
	if (T.pregenerated && T.pre_P.target_batch == target_batch) {
		P = T.pre_P;
		T.cur ^= 1;
		NGP_HIP_CHECK(hipStreamWaitEvent(stream, T.ev_gen, 0));
	} else {
		if (T.pregenerated) NGP_HIP_CHECK(hipStreamSynchronize(T.stream2)); // samples for another batch size: dropped
		if (T.measured_batch_size_before_compaction == 0) T.measured_batch_size_before_compaction = target_batch * 16;
		P = step_params(ctx, T, target_batch);
	}
	ensure_workspace(ctx, T, P.n_rays, P.max_samples, target_batch);
	if (!T.pregenerated || T.pre_P.target_batch != target_batch) launch_generate(ctx, T, T.cur, P, stream);
	T.pregenerated = false;
	if (T.training_step == 0) T.n_rays_total = 0;
	T.n_rays_total += P.n_rays;
	const TrainState::GenSet& G = T.gen[T.cur];
	T.B.counters = G.counters;
	T.B.ray_indices = G.ray_indices;
	T.B.rays = G.rays;
	T.B.numsteps = G.numsteps;
	T.B.coords = G.coords;
	const ModelParams M = training_model(ctx);
	NGP_HIP_CHECK(hipMemsetAsync(T.B.loss, 0, (size_t)P.n_rays * sizeof(float), stream));
	launch_train_inference(M, T.d_tfrags, T.B.counters, P.max_samples, T.B.coords, T.B.mlp_out, ctx->n_cus, stream);
	launch_train_loss(M, P, T.d_images, T.B, stream);
	// what the host needs to plan the next step, as soon as the loss kernel is through
	if (get_loss_scalar) launch_train_loss_sum(T.B.loss, P.n_rays, T.d_loss_sum, stream);
	NGP_HIP_CHECK(hipMemcpyAsync(T.h_counters, T.B.counters, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
	if (get_loss_scalar) NGP_HIP_CHECK(hipMemcpyAsync(T.h_counters + 4, T.d_loss_sum, sizeof(float), hipMemcpyDeviceToHost, stream));
	NGP_HIP_CHECK(hipEventRecord(T.ev_loss, stream));
	T.last_step = P;
	T.batch_ready = true;
}

void backward(ngp_ctx* ctx, TrainState& T, uint32_t target_batch) {
	if (!T.batch_ready) throw std::runtime_error("no training batch prepared");
	launch_train_backward(training_model(ctx), T.d_tfrags, T.d_kfrags, T.B.counters, target_batch, T.B.coords_compacted, T.B.dloss, T.d_grad, T.n_matrix, T.d_partials, ctx->n_cus, ctx->stream);
}

// m_trainer->optimizer_step (:3002): ExponentialDecay > Adam > Ema, then the fragments of the new weights. Asynchronous.
void launch_optimizer(ngp_ctx* ctx, TrainState& T) {
	hipStream_t stream = ctx->stream;
	AdamParams A{};
	A.n_params = T.n_params;
	A.n_matrix = T.n_matrix;
	A.learning_rate = T.opts.learning_rate * T.lr_factor; // the rate of this step uses the factor accumulated so far
	A.beta1 = T.opts.beta1;
	A.beta2 = T.opts.beta2;
	A.epsilon = T.opts.epsilon;
	A.l2_reg = T.opts.l2_reg;
	A.loss_scale = TRAIN_LOSS_SCALE;
	A.optimize_matrix = T.opts.train_network;
	A.optimize_non_matrix = T.opts.train_encoding;
	++T.optimizer_step;
	const bool ema = T.opts.ema_decay > 0.f;
	A.ema_decay = T.opts.ema_decay;
	A.ema_debias_old = 1.0f - std::pow(T.opts.ema_decay, (float)(T.optimizer_step - 1));
	A.ema_debias_new = 1.0f / (1.0f - std::pow(T.opts.ema_decay, (float)T.optimizer_step));
	launch_train_optimizer(A, T.d_weights_fp32, T.d_weights, T.d_grad, T.d_m1, T.d_m2, T.d_steps, T.d_ema_tmp, ema ? T.d_weights_ema : nullptr, stream);
	launch_train_build_fragments(T.d_weights, T.d_tfrags, T.d_kfrags, stream);
	if (T.opts.decay_interval > 0 && T.optimizer_step >= T.opts.decay_start && (T.optimizer_step - T.opts.decay_start) % T.opts.decay_interval == 0) T.lr_factor *= T.opts.decay_base;
	T.inference_dirty = true;
	T.host_params_dirty = true;
	T.batch_ready = false;
}

// ++m_training_step + NerfCounters::update_after_training (:3004-3020, 2923-2947): waits for the loss kernel only --
// the backward pass and the optimizer keep running
float finish_step(TrainState& T, uint32_t target_batch, bool get_loss_scalar) {
	NGP_HIP_CHECK(hipEventSynchronize(T.ev_loss));
	const uint32_t* counters = T.h_counters;
	float loss_sum = 0.f;
	memcpy(&loss_sum, T.h_counters + 4, sizeof(float));
	++T.training_step;
	T.rng.advance();
	T.measured_batch_size = 0;
	T.measured_batch_size_before_compaction = 0;
	if (counters[0] == 0 || counters[2] == 0) {
		T.loss_scalar = 0.f;
		throw std::runtime_error("Nerf training generated 0 samples. Aborting training."); // :3016-3020
	}
	T.measured_batch_size_before_compaction = counters[0];
	T.measured_batch_size = counters[2];
	float loss_scalar = 0.f;
	if (get_loss_scalar) {
		loss_scalar = loss_sum * (float)T.measured_batch_size / (float)target_batch;
		T.loss_scalar = loss_scalar;
	}
	T.rays_per_batch = (uint32_t)((float)T.last_step.n_rays * (float)target_batch / (float)T.measured_batch_size);
	T.rays_per_batch = std::min(next_multiple(T.rays_per_batch, BATCH_SIZE_GRANULARITY), 1u << 18);
	return loss_scalar;
}

void training_prep(ngp_ctx* ctx, TrainState& T) {
	if (!prep_due(T, T.training_step)) return;
	sync_inference_model(ctx); // NerfNetwork::density runs on the inference parameters
	const uint32_t n_cascades = ctx->max_cascade + 1;
	if (T.training_step < 256) update_density_grid_device(ctx, T.opts.density_grid_decay, NERF_GRID_N_CELLS * n_cascades, 0, 1);
	else update_density_grid_device(ctx, T.opts.density_grid_decay, NERF_GRID_N_CELLS / 4 * n_cascades, NERF_GRID_N_CELLS / 4 * n_cascades, 1);
}

// step N+1's samples, on the second stream, while step N's backward pass and optimizer run
void pregenerate(ngp_ctx* ctx, TrainState& T, uint32_t target_batch) {
	T.pre_P = step_params(ctx, T, target_batch);
	launch_generate(ctx, T, T.cur ^ 1, T.pre_P, T.stream2);
	NGP_HIP_CHECK(hipEventRecord(T.ev_gen, T.stream2));
	T.pregenerated = true;
}

bool decode_image(const std::string& bytes, std::vector<uint8_t>& rgba, int& width, int& height, std::string& why) {
	if (bytes.size() >= 8 && (uint8_t)bytes[0] == 0x89 && bytes[1] == 'P') return decode_png(bytes, rgba, width, height, why);
	if (bytes.size() >= 3 && (uint8_t)bytes[0] == 0xFF && (uint8_t)bytes[1] == 0xD8) return decode_jpeg(bytes, rgba, width, height, why);
	why = "only PNG and JPEG images are decoded";
	return false;
}

} // namespace

namespace ngp {

// resolution of a PNG / JPEG file from its header (the loader needs it when transforms.json gives no "w" / "h",
// as in the NeRF-synthetic scenes; the reference takes it from the decoded image, src/nerf_loader.cu:560-600)
bool probe_image_size(const std::string& path, int& width, int& height) {
	std::ifstream f(path, std::ios::binary);
	if (!f) return false;
	std::string head(1 << 16, '\0');
	f.read(&head[0], (std::streamsize)head.size());
	head.resize((size_t)f.gcount());
	const uint8_t* d = (const uint8_t*)head.data();
	const size_t n = head.size();
	if (n >= 24 && d[0] == 0x89 && d[1] == 'P' && !memcmp(d + 12, "IHDR", 4)) {
		width = (int)be32(d + 16);
		height = (int)be32(d + 20);
		return width > 0 && height > 0;
	}
	if (n >= 4 && d[0] == 0xFF && d[1] == 0xD8) {
		size_t pos = 2;
		while (pos + 9 < n) {
			if (d[pos] != 0xFF) { ++pos; continue; }
			const int marker = d[pos + 1];
			if (marker == 0xFF) { ++pos; continue; }
			if (marker == 0x01 || (marker >= 0xD0 && marker <= 0xD8)) { pos += 2; continue; }
			const size_t len = ((size_t)d[pos + 2] << 8) | d[pos + 3];
			if (marker >= 0xC0 && marker <= 0xCF && marker != 0xC4 && marker != 0xC8 && marker != 0xCC) {
				height = (d[pos + 5] << 8) | d[pos + 6];
				width = (d[pos + 7] << 8) | d[pos + 8];
				return width > 0 && height > 0;
			}
			pos += 2 + len;
		}
	}
	return false;
}

void free_training(ngp_ctx* ctx) {
	if (!ctx->train) return;
	TrainState& T = *ctx->train;
	dev_free(T.d_weights_fp32); dev_free(T.d_weights); dev_free(T.d_weights_ema); dev_free(T.d_ema_tmp); dev_free(T.d_grad); dev_free(T.d_m1); dev_free(T.d_m2);
	dev_free(T.d_steps); dev_free(T.d_tfrags); dev_free(T.d_kfrags); dev_free(T.d_tfrags_inference); dev_free(T.d_kfrags_inference); dev_free(T.d_images); dev_free(T.d_loss_sum); dev_free(T.d_partials);
	if (T.stream2) (void)hipStreamSynchronize(T.stream2);
	for (auto& G : T.gen) { dev_free(G.counters); dev_free(G.ray_indices); dev_free(G.rays); dev_free(G.numsteps); dev_free(G.coords); }
	dev_free(T.B.mlp_out); dev_free(T.B.coords_compacted); dev_free(T.B.dloss); dev_free(T.B.loss);
	if (T.stream2) (void)hipStreamDestroy(T.stream2);
	if (T.ev_gen) (void)hipEventDestroy(T.ev_gen);
	if (T.ev_loss) (void)hipEventDestroy(T.ev_loss);
	if (T.h_counters) (void)hipHostFree(T.h_counters);
	const ngp_training_opts keep = T.opts; // settings outlive a model (they belong to the Testbed, not to the network)
	delete ctx->train;
	ctx->train = new TrainState();
	ctx->train->opts = keep;
}

// The render model (grid table in both layouts + weight fragments) follows the inference parameters: Ema's shadow
// weights, or the training weights without an Ema.
void sync_inference_model(ngp_ctx* ctx) {
	if (!ctx->train || !ctx->train->inference_dirty || ctx->device < 0 || !ctx->model_loaded) return;
	TrainState& T = *ctx->train;
	hipStream_t stream = ctx->stream;
	ensure_sync_buffers(ctx);
	order_after_frames(ctx, stream); // frames in flight on any stream read the tables: the update waits for them on the device, the host does not
	const uint16_t* src = T.opts.ema_decay > 0.f ? T.d_weights_ema : T.d_weights;
	const size_t ng = (size_t)T.n_params - T.n_matrix;
	NGP_HIP_CHECK(hipMemcpyAsync(ctx->d_params, src + T.n_matrix, ng * sizeof(uint16_t), hipMemcpyDeviceToDevice, stream));
	launch_train_xor_layout(ctx->M, (const uint2*)ctx->d_params, (char*)ctx->d_xgrid, stream);
	launch_train_build_fragments(src, T.d_tfrags_inference, T.d_kfrags_inference, stream);
	NGP_HIP_CHECK(hipMemcpyAsync(ctx->d_wfrags, T.d_tfrags_inference, (size_t)N_FRAGS * 64 * sizeof(uint4), hipMemcpyDeviceToDevice, stream));
	launch_build_normals_fragments(ctx->d_wfrags, stream);
	mark_model_updated(ctx, stream); // (frames issued from here on wait for it: order_after_model in render_frames)
	NGP_HIP_CHECK(hipGetLastError());
	T.inference_dirty = false;
	++ctx->params_generation;
}

// Trainer::serialize writes the training parameters (fp16), like the reference's snapshots without optimizer state
void sync_host_params(ngp_ctx* ctx) {
	if (!ctx->train || !ctx->train->host_params_dirty || ctx->device < 0 || !ctx->train->d_weights) return;
	TrainState& T = *ctx->train;
	NGP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	NGP_HIP_CHECK(hipMemcpy(ctx->params.data(), T.d_weights, (size_t)T.n_params * sizeof(uint16_t), hipMemcpyDeviceToHost));
	T.host_params_dirty = false;
}

} // namespace ngp

extern "C" {

void ngp_default_training_opts(ngp_training_opts* opts) {
	if (opts) default_opts(*opts);
}

int ngp_set_training_opts(ngp_ctx* ctx, const ngp_training_opts* opts) {
	return guarded(ctx, [&] {
		if (!opts || opts->struct_size != sizeof(ngp_training_opts)) throw std::runtime_error("ngp_training_opts: struct_size mismatch");
		if (opts->loss_type < 0 || opts->loss_type > NGP_LOSS_RELATIVE_L2) throw std::runtime_error("unknown loss type");
		if (!(opts->learning_rate > 0.f) || !(opts->beta1 >= 0.f && opts->beta1 < 1.f) || !(opts->beta2 >= 0.f && opts->beta2 < 1.f) || !(opts->ema_decay >= 0.f && opts->ema_decay < 1.f))
			throw std::runtime_error("invalid optimizer settings");
		if (!ctx->train) ctx->train = new TrainState();
		ctx->train->opts = *opts;
		if (ctx->have_desc) ctx->desc.linear_colors = opts->linear_colors; // m_nerf.training.linear_colors is one setting for training and shade_kernel_nerf
	});
}

int ngp_get_training_opts(const ngp_ctx* ctx, ngp_training_opts* opts) {
	if (!ctx || !opts) return -1;
	if (ctx->train && ctx->train->opts.struct_size) *opts = ctx->train->opts;
	else default_opts(*opts);
	return 0;
}

int ngp_reset_network(ngp_ctx* ctx, uint32_t log2_hashmap_size, uint64_t seed) {
	return guarded(ctx, [&] {
		ngp_model_desc d{};
		d.n_levels = N_LEVELS;
		d.n_features_per_level = N_FEATURES;
		d.log2_hashmap_size = log2_hashmap_size;
		d.base_resolution = 16;
		// the fork derives per_level_scale with aabb_scale = 1 (src/testbed.cu:3951-3966): desired resolution 2048 at the last level
		d.per_level_scale = std::exp(std::log(2048.0f * 1.0f / (float)d.base_resolution) / (float)(d.n_levels - 1));
		d.n_neurons = MLP_WIDTH;
		d.n_hidden_density = 1;
		d.n_hidden_rgb = 2;
		d.density_out_dims = 16;
		d.rgb_activation = ctx->dataset.is_hdr ? 3u : 2u; // Exponential for HDR data, else Logistic (load_nerf_post, src/testbed_nerf.cu:2652-2653)
		d.density_activation = 3; // Exponential
		const uint32_t aabb_scale = ctx->dataset.views.empty() ? 1u : (uint32_t)ctx->dataset.aabb_scale;
		d.aabb_scale = aabb_scale;
		const float half = 0.5f * (float)aabb_scale; // load_nerf_post, src/testbed_nerf.cu:2720-2727
		for (int i = 0; i < 3; ++i) {
			d.aabb_min[i] = 0.5f - half;
			d.aabb_max[i] = 0.5f + half;
			d.render_aabb_min[i] = ctx->dataset.has_render_aabb ? ctx->dataset.render_aabb_min[i] : d.aabb_min[i];
			d.render_aabb_max[i] = ctx->dataset.has_render_aabb ? ctx->dataset.render_aabb_max[i] : d.aabb_max[i];
		}
		for (int i = 0; i < 9; ++i) d.render_aabb_to_local[i] = ctx->dataset.render_aabb_to_local[i];
		d.cone_angle_constant = aabb_scale <= 1 ? 0.0f : (1.0f / 256.0f); // :2736
		d.linear_colors = (ctx->train && ctx->train->opts.struct_size) ? ctx->train->opts.linear_colors : 0;
		// parameter count: the same level table the loader builds
		uint64_t ng = 0;
		{
			const float log2_pls = std::log2(d.per_level_scale);
			for (uint32_t l = 0; l < d.n_levels; ++l) {
				const float scale = std::exp2((float)l * log2_pls) * (float)d.base_resolution - 1.0f;
				const uint32_t res = (uint32_t)std::ceil(scale) + 1u;
				uint64_t n = std::min<uint64_t>((uint64_t)res * res * res, 0xFFFFFFFFull / 2);
				n = (n + 7) / 8 * 8;
				n = std::min<uint64_t>(n, 1ull << log2_hashmap_size);
				ng += n * d.n_features_per_level;
			}
		}
		const uint64_t n_matrix = 64 * 32 + 16 * 64 + 64 * 32 + 64 * 64 + 16 * 64;
		std::vector<uint16_t> params(n_matrix + ng);
		// tcnn Trainer::initialize_params: xavier-uniform matrices (scale sqrt(6 / (fan_in + fan_out))), grid in +-1e-4
		Pcg32 rng;
		rng.seed(seed);
		const struct { uint32_t n_out, n_in; } mats[5] = {{64, 32}, {16, 64}, {64, 32}, {64, 64}, {16, 64}};
		size_t k = 0;
		for (const auto& m : mats) {
			const float scale = std::sqrt(6.0f / (float)(m.n_in + m.n_out));
			for (uint32_t i = 0; i < m.n_out * m.n_in; ++i) params[k++] = half_from_float((rng.next_float() * 2.0f - 1.0f) * scale);
		}
		for (; k < params.size(); ++k) params[k] = half_from_float((rng.next_float() * 2.0f - 1.0f) * 1e-4f);
		d.params_fp16 = params.data();
		d.n_params = params.size();
		d.density_grid_fp16 = nullptr;
		d.n_density_grid = 0;
		install_model(ctx, d);
	});
}

int ngp_set_training_image(ngp_ctx* ctx, int view, int32_t width, int32_t height, const void* rgba, int32_t image_type) {
	return guarded(ctx, [&] {
		if (ctx->device < 0) throw std::runtime_error("this context has no HIP device (host-only); there is no CPU fallback");
		if (view < 0 || (size_t)view >= ctx->dataset.views.size()) throw std::runtime_error("NerfDataset::set_training_image: invalid frame index");
		if (!rgba || width <= 0 || height <= 0) throw std::runtime_error("image should be (H,W,C) where C=4");
		if (image_type != NGP_IMAGE_BYTE && image_type != NGP_IMAGE_FLOAT) throw std::runtime_error("unknown image type in set_training_image");
		TrainingView& v = ctx->dataset.views[(size_t)view];
		const size_t bytes = (size_t)width * height * (image_type == NGP_IMAGE_BYTE ? 4 : 16);
		if (v.d_pixels) (void)hipFree(v.d_pixels);
		v.d_pixels = nullptr;
		NGP_HIP_CHECK(hipMalloc(&v.d_pixels, bytes));
		NGP_HIP_CHECK(hipMemcpy(v.d_pixels, rgba, bytes, hipMemcpyHostToDevice));
		if (v.resolution[0] != width || v.resolution[1] != height) {
			// intrinsics follow the pixel grid (the loader scales them with the image it finds)
			const float sx = (float)width / (float)v.resolution[0], sy = (float)height / (float)v.resolution[1];
			v.focal_length[0] *= sx;
			v.focal_length[1] *= sy;
			v.resolution[0] = width;
			v.resolution[1] = height;
		}
		v.image_type = image_type;
		if (ctx->train) ctx->train->images_dirty = true;
	});
}

int ngp_decode_image(const void* bytes, size_t n_bytes, int32_t* width, int32_t* height, uint8_t* rgba_out, size_t rgba_capacity, char* error_out, size_t error_capacity) {
	std::string why;
	try {
		if (!bytes || !width || !height) throw std::runtime_error("null argument");
		std::vector<uint8_t> rgba;
		int w = 0, h = 0;
		if (!decode_image(std::string((const char*)bytes, n_bytes), rgba, w, h, why)) throw std::runtime_error(why);
		*width = w;
		*height = h;
		if (rgba_out) {
			if (rgba_capacity < rgba.size()) throw std::runtime_error("output buffer too small");
			memcpy(rgba_out, rgba.data(), rgba.size());
		}
		return 0;
	} catch (const std::exception& e) {
		if (error_out && error_capacity) snprintf(error_out, error_capacity, "%s", e.what());
		return -1;
	}
}

int ngp_load_training_images(ngp_ctx* ctx, int32_t* n_loaded_out) {
	return guarded(ctx, [&] {
		if (ctx->device < 0) throw std::runtime_error("this context has no HIP device (host-only); there is no CPU fallback");
		int n_loaded = 0;
		std::string first_problem;
		for (size_t i = 0; i < ctx->dataset.views.size(); ++i) {
			TrainingView& v = ctx->dataset.views[i];
			if (v.d_pixels) { ++n_loaded; continue; }
			std::string why;
			if (v.abs_path.empty() || !file_exists(v.abs_path)) why = "no such file";
			else {
				std::vector<uint8_t> rgba;
				int w = 0, h = 0;
				if (decode_image(read_file(v.abs_path), rgba, w, h, why)) {
					// convert_rgba32 (src/nerf_loader.cu:41-63): "white = transparent" / "black = transparent" datasets, and the
					// dynamic mask beside the image (:596-615): masked pixels become the hot-pink sentinel no ray is drawn from
					if (v.white_transparent || v.black_transparent) {
						for (size_t k = 0; k < rgba.size(); k += 4) {
							if (v.white_transparent && rgba[k] == 255 && rgba[k + 1] == 255 && rgba[k + 2] == 255) rgba[k + 3] = 0;
							if (v.black_transparent && rgba[k] == 0 && rgba[k + 1] == 0 && rgba[k + 2] == 0) rgba[k + 3] = 0;
						}
					}
					{
						const size_t slash = v.abs_path.find_last_of('/');
						const std::string dir = slash == std::string::npos ? std::string(".") : v.abs_path.substr(0, slash);
						std::string stem = slash == std::string::npos ? v.abs_path : v.abs_path.substr(slash + 1);
						const size_t dot = stem.find_last_of('.');
						if (dot != std::string::npos) stem = stem.substr(0, dot);
						const std::string mask_path = dir + "/dynamic_mask_" + stem + ".png";
						if (file_exists(mask_path)) {
							std::vector<uint8_t> mask;
							int mw = 0, mh = 0;
							std::string mwhy;
							if (!decode_image(read_file(mask_path), mask, mw, mh, mwhy)) throw std::runtime_error("Dynamic mask " + mask_path + " could not be loaded.");
							if (mw != w || mh != h) throw std::runtime_error("Dynamic mask " + mask_path + " has wrong resolution.");
							for (size_t k = 0; k < rgba.size(); k += 4)
								if (mask[k] != 0 || mask[k + 1] != 0 || mask[k + 2] != 0) { rgba[k] = 0xFF; rgba[k + 1] = 0x00; rgba[k + 2] = 0xFF; rgba[k + 3] = 0x00; }
						}
					}
					if (ngp_set_training_image(ctx, (int)i, w, h, rgba.data(), NGP_IMAGE_BYTE) != 0) throw std::runtime_error(ctx->error);
					++n_loaded;
					continue;
				}
			}
			if (first_problem.empty()) first_problem = "'" + (v.abs_path.empty() ? v.path : v.abs_path) + "': " + why;
		}
		if (n_loaded_out) *n_loaded_out = n_loaded;
		if (n_loaded == 0 && !ctx->dataset.views.empty()) throw std::runtime_error("no training image could be loaded (" + first_problem + ")");
	});
}

int ngp_render_ground_truth(ngp_ctx* ctx, int view, int32_t width, int32_t height, const float* background_rgba, float exposure, int32_t color_space, int32_t to_srgb,
                            int32_t fov_axis, float zoom, float* rgba_out) {
	return guarded(ctx, [&] {
		if (ctx->device < 0) throw std::runtime_error("this context has no HIP device (host-only); there is no CPU fallback");
		if (view < 0 || (size_t)view >= ctx->dataset.views.size()) throw std::runtime_error("Invalid training view.");
		const TrainingView& v = ctx->dataset.views[(size_t)view];
		if (!v.d_pixels) throw std::runtime_error("training view " + std::to_string(view) + " has no image (ngp_load_training_images / ngp_set_training_image)");
		if (width <= 0 || height <= 0 || !rgba_out || !background_rgba || !(zoom > 0.f) || (fov_axis != 0 && fov_axis != 1)) throw std::runtime_error("invalid ground-truth render arguments");
		TrainImage im{};
		im.pixels = v.d_pixels;
		im.type = v.image_type;
		im.res[0] = v.resolution[0]; im.res[1] = v.resolution[1];
		ensure_sync_buffers(ctx);
		float4* d_out = nullptr;
		NGP_HIP_CHECK(hipMalloc((void**)&d_out, (size_t)width * height * sizeof(float4)));
		launch_overlay_image(width, height, exposure, background_rgba, im, color_space, to_srgb, fov_axis, zoom, d_out, ctx->stream);
		hipError_t e = hipMemcpyAsync(rgba_out, d_out, (size_t)width * height * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream);
		if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
		(void)hipFree(d_out);
		NGP_HIP_CHECK(e);
		NGP_HIP_CHECK(hipGetLastError());
	});
}

int ngp_train(ngp_ctx* ctx, uint32_t n_steps, uint32_t batch_size, float* loss_out) {
	return guarded(ctx, [&] {
		TrainState& T = ensure_training(ctx);
		for (uint32_t it = 0; it < n_steps; ++it) {
			upload_images(ctx, T);
			training_prep(ctx, T);
			const bool get_loss_scalar = T.training_step % 16 == 0;
			prepare_batch(ctx, T, batch_size, get_loss_scalar);
			backward(ctx, T, batch_size);
			launch_optimizer(ctx, T);
			finish_step(T, batch_size, get_loss_scalar);
			// the next step's rays do not depend on the weights: generate them beside this step's backward pass, unless
			// the occupancy grid is refreshed first
			if (it + 1 < n_steps && !prep_due(T, T.training_step) && !T.images_dirty) pregenerate(ctx, T, batch_size);
		}
		NGP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
		NGP_HIP_CHECK(hipGetLastError());
		if (loss_out) *loss_out = T.loss_scalar;
	});
}

int ngp_get_training_state(const ngp_ctx* ctx, ngp_training_state* out) {
	if (!ctx || !out) return -1;
	memset(out, 0, sizeof(*out));
	out->struct_size = sizeof(*out);
	out->rays_per_batch = 1u << 12; // NerfCounters' initial value
	if (!ctx->train) return 0;
	const TrainState& T = *ctx->train;
	out->training_step = T.training_step;
	out->rays_per_batch = T.rays_per_batch;
	out->measured_batch_size = T.measured_batch_size;
	out->measured_batch_size_before_compaction = T.measured_batch_size_before_compaction;
	out->n_rays_total = T.n_rays_total;
	out->loss = T.loss_scalar;
	out->learning_rate = T.opts.learning_rate * T.lr_factor;
	out->n_params = T.n_params;
	out->n_matrix_params = T.n_matrix;
	return 0;
}

int ngp_train_prepare_batch(ngp_ctx* ctx, uint32_t batch_size, uint32_t* counters3, uint32_t* ray_indices, uint32_t* numsteps, float* coords_compacted, uint16_t* dloss_fp16,
                            float* loss) {
	return guarded(ctx, [&] {
		TrainState& T = ensure_training(ctx);
		prepare_batch(ctx, T, batch_size, true);
		NGP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
		NGP_HIP_CHECK(hipGetLastError());
		uint32_t c[4];
		NGP_HIP_CHECK(hipMemcpy(c, T.B.counters, sizeof(c), hipMemcpyDeviceToHost));
		if (counters3) { counters3[0] = c[0]; counters3[1] = c[1]; counters3[2] = c[2]; }
		const uint32_t n_rays = T.last_step.n_rays;
		if (ray_indices) NGP_HIP_CHECK(hipMemcpy(ray_indices, T.B.ray_indices, (size_t)n_rays * sizeof(uint32_t), hipMemcpyDeviceToHost));
		if (numsteps) NGP_HIP_CHECK(hipMemcpy(numsteps, T.B.numsteps, (size_t)n_rays * 2 * sizeof(uint32_t), hipMemcpyDeviceToHost));
		if (coords_compacted) NGP_HIP_CHECK(hipMemcpy(coords_compacted, T.B.coords_compacted, (size_t)batch_size * TRAIN_COORD_FLOATS * sizeof(float), hipMemcpyDeviceToHost));
		if (dloss_fp16) NGP_HIP_CHECK(hipMemcpy(dloss_fp16, T.B.dloss, (size_t)batch_size * 4 * sizeof(uint16_t), hipMemcpyDeviceToHost));
		if (loss) NGP_HIP_CHECK(hipMemcpy(loss, T.B.loss, (size_t)n_rays * sizeof(float), hipMemcpyDeviceToHost));
	});
}

int ngp_train_gradients(ngp_ctx* ctx, uint32_t batch_size, float* grad_out) {
	return guarded(ctx, [&] {
		TrainState& T = ensure_training(ctx);
		NGP_HIP_CHECK(hipMemsetAsync(T.d_grad, 0, (size_t)T.n_params * sizeof(float), ctx->stream));
		backward(ctx, T, batch_size);
		NGP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
		NGP_HIP_CHECK(hipGetLastError());
		if (grad_out) NGP_HIP_CHECK(hipMemcpy(grad_out, T.d_grad, (size_t)T.n_params * sizeof(float), hipMemcpyDeviceToHost));
	});
}

int ngp_train_apply(ngp_ctx* ctx) {
	return guarded(ctx, [&] {
		TrainState& T = ensure_training(ctx);
		if (!T.batch_ready) throw std::runtime_error("no training batch prepared");
		launch_optimizer(ctx, T);
		finish_step(T, T.last_step.target_batch, true);
		NGP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
		NGP_HIP_CHECK(hipGetLastError());
	});
}

int ngp_get_training_params(ngp_ctx* ctx, float* params_out, float* ema_out) {
	return guarded(ctx, [&] {
		TrainState& T = ensure_training(ctx);
		NGP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
		if (params_out) NGP_HIP_CHECK(hipMemcpy(params_out, T.d_weights_fp32, (size_t)T.n_params * sizeof(float), hipMemcpyDeviceToHost));
		if (ema_out) NGP_HIP_CHECK(hipMemcpy(ema_out, T.d_ema_tmp, (size_t)T.n_params * sizeof(float), hipMemcpyDeviceToHost));
	});
}

} // extern "C"
