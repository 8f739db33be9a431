// Host-side state behind the C ABI (include/ngp_hip.h): model, dataset, frame buffers.
#pragma once

#include "../../include/ngp_hip.h"
#include "minijson.h"
#include "ngp_kernels.h"

#include <hip/hip_runtime.h>

#include <array>
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
void launch_render_nerf(const ModelParams& M, const CameraParams& C, const FrameParams& F, int n_blocks, hipStream_t stream);
void launch_grid_encode(const ModelParams& M, uint32_t n, const float* pos01, uint16_t* out, hipStream_t stream);
void launch_network_inference(const ModelParams& M, uint32_t n, const float* pos01, const float* dir01, uint16_t* out, hipStream_t stream);
void launch_init_rays(const ModelParams& M, const CameraParams& C, NerfPayload* payloads, hipStream_t stream);
void launch_density_grid_to_bitfield(const uint16_t* d_grid_fp16, uint32_t n_grid, uint32_t max_cascade, float* d_grid_f32, double* d_partial,
                                     uint8_t* d_bitfield, float* out_mean, hipStream_t stream);
void launch_accumulate_tonemap(uint32_t n_pixels, const float4* frame_buffer, float4* accumulate_buffer, float sample_count, const float* background,
                               float exposure, int to_srgb, float4* rgba_out, hipStream_t stream);

// NerfDataset subset (nerf_loader.h:60-170): what rendering and the harness read
struct TrainingView {
	std::array<float, 12> xform; // ngp-space camera-to-world, column-major 4x3
	int32_t resolution[2];
	float focal_length[2];
	float principal_point[2];
	std::string path;
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
	uint4* d_wfrags = nullptr;
	uint8_t* d_bitfield = nullptr;
	uint16_t* d_density_f16 = nullptr;
	float* d_density_f32 = nullptr;
	double* d_partial = nullptr;
	float bitfield_mean = 0.f;
	ngp::ModelParams M{};

	// ---- snapshot extras (src/testbed.cu:5396-5424) and dataset
	mj::Value config; // network config (+ "snapshot" on load)
	bool has_snapshot_camera = false;
	float snap_camera[12];
	float snap_relative_focal_length[2] = {1.f, 1.f};
	int32_t snap_fov_axis = 1;
	float snap_screen_center[2] = {0.5f, 0.5f};
	float snap_zoom = 1.f;
	ngp::Dataset dataset;
	std::string data_path;

	// ---- frame
	size_t n_pixels_alloc = 0;
	float4* d_frame = nullptr;
	float* d_depth = nullptr;
	float4* d_accum = nullptr;
	float4* d_rgba = nullptr;
	// queue word (64 B) followed by a ring of per-call counter slots (32 B each: alive, hit, samples, pad)
	static constexpr int HISTORY = 256;
	void* d_sync = nullptr;
	hipEvent_t ev_frame0[HISTORY] = {}, ev_frame1[HISTORY] = {}, ev_kern0[HISTORY] = {}, ev_kern1[HISTORY] = {};
	uint64_t hist_n_rays[HISTORY] = {};
	uint64_t n_calls = 0; // render calls so far; call k uses slot k % HISTORY
	hipStream_t last_stream = nullptr;
};
