// Host-side state behind the C ABI (include/ngp_hip.h): model, dataset, frame buffers.
#pragma once

#include "../../include/ngp_hip.h"
#include "minijson.h"
#include "ngp_kernels.h"
#include "pcg32.h"
#include "train_kernels.h"

#include <hip/hip_runtime.h>

#include <array>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <sys/stat.h>
#include <stdexcept>
#include <string>
#include <vector>

namespace ngp {

#define NGP_HIP_CHECK(expr)                                                                                        \
	do {                                                                                                           \
		hipError_t _e = (expr);                                                                                    \
		if (_e != hipSuccess) throw std::runtime_error(std::string(#expr " failed: ") + hipGetErrorString(_e));   \
	} while (0)

// kernel launchers, nerf_kernels.hip
void launch_render_nerf(const ModelParams& M, const CameraParams& C, const FrameParams& F, int n_cus, hipStream_t stream);
void launch_grid_encode(const ModelParams& M, uint32_t n, const float* pos01, uint16_t* out, hipStream_t stream);
void launch_build_normals_fragments(uint4* wfrags, hipStream_t stream); // after every change of the forward fragments
void launch_density_gradient(const ModelParams& M, uint32_t n, const float* pos01, float* out, hipStream_t stream);
// wide_kernels.hip (ModelParams::wide): the launchers above that take a model forward to these when M.wide.width != 0
void launch_render_nerf_wide(const ModelParams& M, const CameraParams& C, const FrameParams& F, int n_cus, hipStream_t stream);
void launch_trace_probe_wide(const ModelParams& M, const FrameParams& F, const ProbeParams& P, int n_cus, hipStream_t stream);
void launch_network_inference_wide(const ModelParams& M, uint32_t n, const float* pos01, const float* dir01, uint16_t* out, int n_cus, hipStream_t stream);
void launch_density_gradient_wide(const ModelParams& M, uint32_t n, const float* pos01, float* out, int n_cus, hipStream_t stream);
void launch_frequency_encode(const ModelParams& M, uint32_t n, const float* pos01, uint16_t* out, hipStream_t stream);
void launch_network_inference(const ModelParams& M, uint32_t n, const float* pos01, const float* dir01, uint16_t* out, hipStream_t stream);
void launch_density_grid_update(const ModelParams& M, uint32_t n_samples, const Pcg32& rng, uint32_t step, uint32_t n_cascades, float thresh, const float* grid,
                                float* grid_tmp, hipStream_t stream);
void launch_density_grid_update_wide(const ModelParams& M, uint32_t n_samples, const Pcg32& rng, uint32_t step, uint32_t n_cascades, float thresh, const float* grid,
                                     float* grid_tmp, float* d_pos01, uint32_t* d_cell, uint16_t* d_out, int n_cus, hipStream_t stream);
void launch_density_grid_ema(uint32_t n_elements, float decay, float* grid, const float* grid_tmp, hipStream_t stream);
void launch_init_rays(const ModelParams& M, const CameraParams& C, NerfPayload* payloads, hipStream_t stream);
void launch_density_grid_to_bitfield(const uint16_t* d_grid_fp16, uint32_t n_grid, uint32_t max_cascade, float* d_grid_f32, double* d_partial,
                                     uint8_t* d_bitfield, float* out_mean, hipStream_t stream);
void launch_coarse_occupancy(const uint8_t* bitfield, uint32_t* coarse, hipStream_t stream);
void launch_accumulate_tonemap(uint32_t n_pixels, const float4* frame_buffer, float4* accumulate_buffer, float sample_count, const float* background,
                               float exposure, int to_srgb, int color_space, float4* rgba_out, hipStream_t stream);

void launch_render_mesh(const MeshSceneParams& S, const MeshShadeParams& P, const IrradianceMap& I, const CameraParams& C, float4* frame_buffer, float* depth_buffer,
                        uint32_t shard_index, uint32_t shard_count, int packed, hipStream_t stream);
void launch_trace_probe(const ModelParams& M, const FrameParams& F, const ProbeParams& P, int n_cus, hipStream_t stream);
void launch_probe_reduce(const ProbeParams& P, float4* envmap, hipStream_t stream);
void launch_irradiance(const ProbeParams& P, const float4* envmap, uint32_t n, const float* normals, float4* out, hipStream_t stream);
void launch_irradiance_lookup(const IrradianceMap& I, uint32_t n, const float* positions, const float* normals, float4* out, hipStream_t stream);
void launch_trace_mesh_rays(const MeshSceneParams& S, uint32_t n, float* positions, float* directions, hipStream_t stream);

// kernel launchers, train_kernels.hip
void launch_train_generate_samples(const ModelParams& M, const TrainStepParams& P, const TrainImage* images, const TrainBatch& B, hipStream_t stream);
void launch_train_inference(const ModelParams& M, const uint4* frags, const uint32_t* counters, uint32_t max_samples, const float* coords, uint16_t* out, int n_cus,
                            hipStream_t stream);
void launch_train_loss(const ModelParams& M, const TrainStepParams& P, const TrainImage* images, const TrainBatch& B, hipStream_t stream);
void launch_train_build_fragments(const uint16_t* params, uint4* frags, uint2* kfrags, hipStream_t stream);
void launch_train_backward(const ModelParams& M, const uint4* frags, const uint2* kfrags, const uint32_t* counters, uint32_t target_batch, const float* coords,
                           const uint16_t* dloss, float* grad, uint32_t n_matrix_params, float* block_partials, int n_blocks, hipStream_t stream);
size_t train_backward_partials_floats(int n_blocks);
void launch_train_optimizer(const AdamParams& A, float* weights_fp32, uint16_t* weights, float* grad, float* m1, float* m2, uint32_t* steps, float* ema_tmp,
                            uint16_t* weights_ema, hipStream_t stream);
void launch_train_xor_layout(const ModelParams& M, const uint2* src, char* dst, hipStream_t stream);
void launch_train_loss_sum(const float* loss, uint32_t n, float* out, hipStream_t stream);
void launch_overlay_image(int width, int height, float exposure, const float* background4, const TrainImage& im, int color_space, int to_srgb, int fov_axis, float zoom, float4* out,
                          hipStream_t stream);

struct HostMesh { // MeshData (mesh.h:18-24) after load_mesh
	std::vector<Triangle> tris;        // reordered by the BVH build
	std::vector<TriangleBvhNode> nodes;
	float bmin[3], bmax[3];
	float center[3];
	Triangle* d_tris = nullptr;
	TriangleBvhNode* d_nodes = nullptr;
};

// NerfDataset subset (nerf_loader.h:60-170): what rendering and the harness read
struct TrainingView {
	std::array<float, 12> xform; // ngp-space camera-to-world, column-major 4x3
	int32_t resolution[2];
	float focal_length[2];
	float principal_point[2];
	int32_t lens_mode = 0; // ELensMode
	float lens_params[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
	std::string path;
	std::string abs_path; // the image file the loader resolved (extension probing, nerf_loader.cu), may not exist
	bool white_transparent = false, black_transparent = false; // transforms.json flags (NSVF-style data), src/nerf_loader.cu:462-468
	void* d_pixels = nullptr; // training image on the device (ngp_set_training_image), RGBA
	int32_t image_type = 0;   // ngp_image_type
};
struct Dataset {
	std::vector<TrainingView> views;
	int32_t aabb_scale = 1;
	float scale = 1.0f;
	float offset[3] = {0.f, 0.f, 0.f};
	float up[3] = {0.f, 1.f, 0.f};
	bool from_mitsuba = false;
	bool is_hdr = false;
	bool has_render_aabb = false;
	float render_aabb_min[3], render_aabb_max[3];
	float render_aabb_to_local[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
	int32_t n_extra_learnable_dims = 0;
};


// Everything Testbed::train touches beyond the inference model (m_trainer, m_optimizer, NerfCounters, m_rng ...)
struct TrainState {
	ngp_training_opts opts{};
	uint32_t n_params = 0, n_matrix = 0;
	float* d_weights_fp32 = nullptr;   // Trainer::m_params_full_precision
	uint16_t* d_weights = nullptr;     // m_params (fp16, what the training kernels read; tcnn order: density MLP, rgb MLP, grid)
	uint16_t* d_weights_ema = nullptr; // Ema's m_weights_ema = the inference parameters
	float* d_ema_tmp = nullptr;
	float* d_grad = nullptr;           // fp32, cleared by the optimizer kernel
	float* d_m1 = nullptr;
	float* d_m2 = nullptr;
	uint32_t* d_steps = nullptr;
	uint4* d_tfrags = nullptr;         // MFMA fragments of the training parameters (forward + transposed)
	uint2* d_kfrags = nullptr;
	uint4* d_tfrags_inference = nullptr;
	uint2* d_kfrags_inference = nullptr;
	TrainImage* d_images = nullptr;
	uint32_t n_images = 0;
	bool images_dirty = true;
	TrainBatch B{}; // the step in flight: gen[cur]'s buffers + the shared ones
	// sample generation only reads the images, the occupancy bitfield and the step's rng: step N+1's runs on a second
	// stream beside step N's backward pass, into the other of two buffer sets
	struct GenSet {
		uint32_t* counters = nullptr;
		uint32_t* ray_indices = nullptr;
		float* rays = nullptr;
		uint32_t* numsteps = nullptr;
		float* coords = nullptr;
		uint32_t cap_rays = 0, cap_samples = 0;
	} gen[2];
	int cur = 0;
	bool pregenerated = false; // gen[cur ^ 1] holds the next step's samples (ev_gen marks them complete)
	TrainStepParams pre_P{};
	hipStream_t stream2 = nullptr;
	hipEvent_t ev_gen = nullptr, ev_loss = nullptr;
	uint32_t* h_counters = nullptr; // pinned: counters[4] + loss sum
	uint32_t cap_loss = 0, cap_out = 0, cap_target = 0;
	float* d_loss_sum = nullptr;
	float* d_partials = nullptr; // per-block weight-gradient sums of train_backward_kernel
	// NerfCounters + Testbed members
	uint32_t training_step = 0;
	uint32_t rays_per_batch = 1u << 12;
	uint32_t n_rays_total = 0;
	uint32_t measured_batch_size = 0, measured_batch_size_before_compaction = 0;
	float loss_scalar = 0.f;
	Pcg32 rng{};
	uint32_t optimizer_step = 0; // Adam::m_current_step
	float lr_factor = 1.0f;      // ExponentialDecay
	bool inference_dirty = false; // the render model lags the training parameters
	bool host_params_dirty = false;
	bool batch_ready = false;
	TrainStepParams last_step{};
};
} // namespace ngp

struct ngp_ctx {
	int device = 0;
	int n_cus = 256;
	std::string error;
	hipStream_t stream = nullptr;

	// ---- model
	bool model_loaded = false; // uploaded to the device
	bool have_desc = false;    // parsed + validated on the host
	ngp_model_desc desc{};
	std::vector<uint16_t> params;
	std::vector<uint16_t> density_grid;
	uint32_t max_cascade = 0;
	void* d_params = nullptr;
	void* d_xgrid = nullptr; // the grid table again, in the xor layout (ngp_api.cpp build_xor_layout)
	uint4* d_wfrags = nullptr;
	uint8_t* d_bitfield = nullptr;
	uint32_t* d_coarse = nullptr;
	float* d_density_tmp = nullptr; // density_grid_tmp of update_density_grid_nerf
	uint64_t grid_rng_state = 0, grid_rng_inc = 0; // m_nerf.training.density_grid_rng
	uint32_t grid_ema_step = 0, grid_updates = 0;
	uint16_t* d_density_f16 = nullptr;
	float* d_density_f32 = nullptr;
	double* d_partial = nullptr;
	float bitfield_mean = 0.f;
	ngp::ModelParams M{};

	// ---- snapshot extras (src/testbed.cu:5396-5424) and dataset
	mj::Value config; // network config (+ "snapshot" on load)
	ngp_session_state session{}; // background, exposure, sun / up direction, camera scale / aperture / focus of the snapshot
	bool has_snapshot_camera = false;
	float snap_camera[12];
	float snap_relative_focal_length[2] = {1.f, 1.f};
	int32_t snap_fov_axis = 1;
	float snap_screen_center[2] = {0.5f, 0.5f};
	float snap_zoom = 1.f;
	ngp::Dataset dataset;
	std::string data_path;

	// ---- geometry mode
	std::vector<ngp::HostMesh> meshes;
	ngp::MeshRef* d_meshrefs = nullptr;
	ngp::MeshSceneParams mesh_scene{};
	ngp::MeshShadeParams shade{{0.57735026f, 0.57735026f, 0.57735026f}, {0.f, 1.f, 0.f}, 0.f, 0.f, 1.f, 0.5f, 0.f, 0.f, 0.f, {0.8f, 0.8f, 0.8f}, {0.f, 0.f, 0.f}};

	// ---- irradiance probe texture(s) (m_envmap_tex / gridSize, testbed.h:949-950) and E(n) tabulated at their texels
	float4* d_envmap = nullptr;
	float4* d_irradiance = nullptr;
	uint32_t env_n_theta = 0, env_n_phi = 0;
	ngp::ProbeParams env_probe{}; // what was traced: mode, shell position(s), grid

	// ---- environment map behind the NeRF (m_envmap.inference_view(), testbed.h:1297-1316)
	float4* d_bg_envmap = nullptr;
	int32_t bg_env_w = 0, bg_env_h = 0;

	// ---- frame
	size_t n_pixels_alloc = 0;
	float4* d_frame = nullptr;
	float* d_depth = nullptr;
	float4* d_accum = nullptr;
	float4* d_rgba = nullptr;
	// a ring of per-call slots of 80 B: accumulators [alive, hit, samples], {tile queue, exited waves}, results [alive, hit, samples, device ticks], start stamp.
	// Zeroed once; every launch's last wave leaves its slot's first four words zero again (nerf_kernels.hip fused_body)
	static constexpr int HISTORY = 256;
	static constexpr size_t SLOT_BYTES = 128;
	void* d_sync = nullptr;
	void bind_slot(ngp::FrameParams& F, int slot) const {
		unsigned long long* w = (unsigned long long*)((char*)d_sync + SLOT_BYTES * (size_t)slot);
		F.counters = w;
		F.queue = (uint32_t*)(w + 3);
		F.done = F.queue + 1;
		F.results = w + 4;
		F.add_results = 0;
		static const bool xcd_queues = []() { const char* e = getenv("NGP_XCD_QUEUES"); return !e || atoi(e) != 0; }(); // 0: one queue (A/B)
		F.xqueue = xcd_queues ? (uint32_t*)(w + 9) : nullptr; // bytes 72..103 of the 128-byte slot
	}
	hipEvent_t ev_frame0[HISTORY] = {}, ev_frame1[HISTORY] = {}, ev_kern0[HISTORY] = {}, ev_kern1[HISTORY] = {};
	uint64_t hist_n_rays[HISTORY] = {};
	uint64_t n_calls = 0; // render calls so far; call k uses slot k % HISTORY
	hipStream_t last_stream = nullptr;
	// Ordering between frames (any stream) and updates of what they read (render tables after training steps, the occupancy grid after a
	// refresh, peer copies into a replica) without stalling the host: an update first orders ITS stream behind every frame issued
	// since the previous update (ngp::order_after_frames: device-side waits on the frames' end events), does its work, and records
	// ev_model; every later frame orders its stream behind ev_model (ngp::order_after_model).
	uint64_t fenced_calls = 0;        // frames [0, fenced_calls) are already ordered before the last update
	hipEvent_t ev_model = nullptr;    // end of the last update of the render model / occupancy grid on this device
	bool ev_model_valid = false;
	hipEvent_t ev_synced = nullptr;   // peer: its copies out of the primary's buffers are done
	unsigned long long* d_prof = nullptr;
	void* d_grid_scratch = nullptr; // occupancy-grid refresh of a Frequency-encoding model: positions, cells, network outputs of a batch of samples
	size_t grid_scratch_samples = 0;
	uint32_t* d_trace = nullptr; // wave timelines of the diagnostic build (NGP_PROFILE_TRACE)
	static constexpr uint32_t TRACE_WAVES = 64, TRACE_ITERS = 1024;
	int32_t tune[8] = {64, 4, 32, 1, 1, 4, 1, 1}; // FrameParams::tune; changed only through validate_schedule (ngp_api.cpp)

	// ---- several devices behind this context (ngp_multi.cpp): replicas on the auxiliary devices, tile gather at the primary
	std::vector<ngp_ctx*> peers;  // owned; empty for a single-device context
	ngp_ctx* primary = nullptr;   // set on a peer
	uint64_t model_generation = 0, synced_generation = 0;               // the model was replaced (set_model, snapshot)
	uint64_t grid_generation = 0, synced_grid_generation = 0;           // the occupancy grid was refreshed from the network
	uint64_t params_generation = 0, synced_params_generation = 0;       // the inference parameters followed a training step
	uint64_t mesh_generation = 0, synced_mesh_generation = 0;           // the mesh list / BVHs changed (Geometry mode)
	uint64_t probe_generation = 0, synced_probe_generation = 0;         // the irradiance probe textures were (re)computed
	float4* d_pack_rgba = nullptr;   // this device's tiles of the current frame, tile-packed
	float* d_pack_depth = nullptr;
	size_t pack_alloc = 0;
	float4* d_gather_rgba = nullptr; // primary: [device][slots * 64]
	float* d_gather_depth = nullptr;
	size_t gather_alloc = 0;
	hipEvent_t ev_pack = nullptr, ev_unpacked = nullptr;
	uint64_t n_multi_frames = 0;
	bool last_was_multi = false; // the last frame was rendered over all devices (ngp_get_render_stats sums the shares)
	bool streams_mixed = false; // frames were issued on more than one stream since the last device-wide wait

	// ---- training (ngp_train.cpp)
	ngp::TrainState* train = nullptr;
	bool density_grid_host_dirty = false; // ctx->density_grid lags d_density_f32
};

namespace ngp {
// ------------------------------------------------------------------------------------------------ helpers
template <typename F>
inline int guarded(ngp_ctx* ctx, F&& f) {
	if (!ctx) return -1;
	try {
		if (ctx->device >= 0) NGP_HIP_CHECK(hipSetDevice(ctx->device));
		f();
		ctx->error.clear();
		return 0;
	} catch (const std::exception& e) {
		ctx->error = e.what();
		return -1;
	}
}

inline std::string read_file(const std::string& path) {
	std::ifstream f(path, std::ios::in | std::ios::binary);
	if (!f) throw std::runtime_error("cannot open '" + path + "'");
	std::stringstream ss;
	ss << f.rdbuf();
	return ss.str();
}

inline bool file_exists(const std::string& p) {
	struct stat st;
	return stat(p.c_str(), &st) == 0;
}
inline bool is_directory(const std::string& p) {
	struct stat st;
	return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}
inline std::string parent_dir(const std::string& p) {
	size_t k = p.find_last_of('/');
	return k == std::string::npos ? std::string(".") : p.substr(0, k);
}
inline bool ends_with_ci(const std::string& s, const std::string& suffix) {
	if (s.size() < suffix.size()) return false;
	for (size_t i = 0; i < suffix.size(); ++i)
		if (tolower(s[s.size() - suffix.size() + i]) != tolower(suffix[i])) return false;
	return true;
}


inline void order_after_frames(ngp_ctx* ctx, hipStream_t stream) {
	const uint64_t pending = ctx->n_calls - ctx->fenced_calls, n = pending < (uint64_t)ngp_ctx::HISTORY ? pending : (uint64_t)ngp_ctx::HISTORY;
	for (uint64_t k = 0; k < n; ++k) NGP_HIP_CHECK(hipStreamWaitEvent(stream, ctx->ev_frame1[(ctx->n_calls - 1 - k) % ngp_ctx::HISTORY], 0));
	ctx->fenced_calls = ctx->n_calls;
}
inline void mark_model_updated(ngp_ctx* ctx, hipStream_t stream) {
	if (!ctx->ev_model) NGP_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_model, hipEventDisableTiming));
	NGP_HIP_CHECK(hipEventRecord(ctx->ev_model, stream));
	ctx->ev_model_valid = true;
}
inline void order_after_model(ngp_ctx* ctx, hipStream_t stream) {
	if (ctx->ev_model_valid) NGP_HIP_CHECK(hipStreamWaitEvent(stream, ctx->ev_model, 0));
}

void load_snapshot_path(ngp_ctx* ctx, const std::string& path);
void install_model(ngp_ctx* ctx, const ngp_model_desc& d); // set_model_impl of ngp_api.cpp
void update_density_grid_device(ngp_ctx* ctx, float decay, uint32_t n_uniform, uint32_t n_nonuniform, uint32_t n_iterations);
void refresh_density_grid_host(ngp_ctx* ctx);
uint16_t half_from_float(float f);
// ngp_train.cpp
void free_training(ngp_ctx* ctx);
bool probe_image_size(const std::string& path, int& width, int& height);
void sync_inference_model(ngp_ctx* ctx); // render what has been trained (no-op when nothing changed)
void sync_host_params(ngp_ctx* ctx);     // ctx->params <- training parameters, for snapshots

void ensure_sync_buffers(ngp_ctx* ctx);
// ngp_multi.cpp / ngp_api.cpp
void render_frames_on(ngp_ctx* ctx, const ngp_camera& cam, const ngp_render_opts& opts, float4* d_rgba, float* d_depth, hipStream_t stream);
void ensure_frame_buffers_for(ngp_ctx* ctx, size_t n_pixels);
void render_frames_multi(ngp_ctx* ctx, const ngp_camera& cam, const ngp_render_opts& opts, float4* d_rgba, float* d_depth, hipStream_t stream);
void free_multi_buffers(ngp_ctx* ctx);
// ngp_mesh.cpp: Geometry mode on an auxiliary device -- the primary's meshes (BVHs as built), shading parameters and irradiance tables
void sync_peer_geometry(ngp_ctx* primary, ngp_ctx* peer);
inline IrradianceMap irradiance_map_of(const ngp_ctx* ctx) {
	IrradianceMap I{};
	I.irradiance = ctx->d_irradiance;
	I.n_theta = ctx->env_n_theta;
	I.n_phi = ctx->env_n_phi;
	if (ctx->env_probe.mode == 3) {
		I.grid_x = ctx->env_probe.grid_x;
		I.grid_y = ctx->env_probe.grid_y;
	}
	for (int i = 0; i < 3; ++i) I.center[i] = ctx->env_probe.center[i];
	return I;
}
} // namespace ngp
