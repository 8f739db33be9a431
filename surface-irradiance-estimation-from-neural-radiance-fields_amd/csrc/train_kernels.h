// Training step of the NeRF: device-side declarations shared by train_kernels.hip and the host.
// Reference: Testbed::train_nerf / train_nerf_step (src/testbed_nerf.cu:2949-3431), generate_training_samples_nerf
// (:737-890), compute_loss_kernel_train_nerf (:893-1213), NerfNetwork::backward_impl (nerf_network.h:189-268); the
// tiny-cuda-nn pieces behind them (FullyFusedMLP / GridEncoding backward, Adam, ExponentialDecay, Ema) are an
// un-vendored submodule and restated from their published algorithms.
#pragma once

#include "ngp_kernels.h"
#include "pcg32.h"

namespace ngp {

constexpr uint32_t N_MAX_RANDOM_SAMPLES_PER_RAY = 16; // nerf_device.cuh:39
constexpr float TRAIN_LOSS_SCALE = 128.0f;            // tcnn default_loss_scale<__half>()
constexpr uint32_t TRAIN_COORD_FLOATS = 7;            // NerfCoordinate: pos (3), dt, dir (3)

// TrainingImageMetadata + TrainingXForm (nerf_loader.h), the fields the default training path reads
struct TrainImage {
	const void* pixels; // RGBA: uint8 sRGB straight alpha (type 1) or float linear premultiplied (type 3)
	int32_t type;       // EImageDataType: 0 None, 1 Byte, 3 Float
	int32_t res[2];
	float focal[2];
	float principal[2];
	int32_t lens_mode;
	float lens_params[7];
	float xform[12]; // camera-to-world, ngp space, column-major 4x3
};

struct TrainStepParams {
	uint32_t n_rays;          // counters.rays_per_batch
	uint32_t n_rays_total;    // rays generated before this step (unused by image_idx, kept for the interface)
	uint32_t n_images;
	uint32_t max_samples;     // capacity of coords / mlp_out for this step (max_inference)
	uint32_t target_batch;    // max_samples_compacted
	Pcg32 rng;                // m_rng of this step
	int32_t snap_to_pixel_centers, random_bg_color, linear_colors, color_space;
	int32_t loss_type;        // ELossType
	float background[3];
	float near_distance;
	float loss_scale;
	float density_grid_mean;  // *mean_density_ptr
};

struct TrainBatch { // workspace of one step (device pointers)
	uint32_t* counters;     // [0] numsteps_counter, [1] ray_counter, [2] numsteps_counter_compacted
	uint32_t* ray_indices;  // [n_rays]
	float* rays;            // [n_rays][6] origin, unnormalized direction
	uint32_t* numsteps;     // [n_rays][2] count, base
	float* coords;          // [max_samples][7]
	uint16_t* mlp_out;      // [max_samples][4] fp16 rgb + density logit
	float* coords_compacted; // [target_batch][7]
	uint16_t* dloss;        // [target_batch][4] fp16, loss-scaled
	float* loss;            // [n_rays]
};

// MFMA A-operand fragments of one parameter set: the 20 forward fragments of ngp_kernels.h followed by the
// transposed matrices of the backward pass.
constexpr int TFRAG_R1T = N_FRAGS;          // W_R1^T 64x64: 8 fragments
constexpr int TFRAG_R0T = TFRAG_R1T + 8;    // rows 0..15 of W_R0^T (the density-output half of the rgb input): 2
constexpr int TFRAG_D0T = TFRAG_R0T + 2;    // W_D0^T 32x64: 4
constexpr int N_TFRAGS = TFRAG_D0T + 4;     // 34 fragments of 64 lanes x 16 B
constexpr int KFRAG_R2T = 0, KFRAG_D1T = 4, N_KFRAGS = 8; // 16x16x16 fragments (64 lanes x 8 B): W_R2^T, W_D1^T (64x16 each)

struct AdamParams {
	uint32_t n_params, n_matrix;
	float learning_rate, beta1, beta2, epsilon, l2_reg, loss_scale;
	int32_t optimize_matrix, optimize_non_matrix;
	float ema_decay, ema_debias_old, ema_debias_new;
};

} // namespace ngp
