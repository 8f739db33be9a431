/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_common.h). PARITY UNPINNED.
 *
 * Public C interface of the CPU restatement. Plain pointers and sizes so that
 * tests can bind it with ctypes.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 32
#define ORC_NERF_GRIDSIZE 128u
#define ORC_NERF_CASCADES 8u
#define ORC_MAX_DEPTH 16384.0f
#define ORC_MARCH_ITER 10000u /* src/testbed_nerf.cu:46 */

enum { ORC_GRID_ACC_LEGACY = 0, ORC_GRID_ACC_FMA = 1 };
enum { ORC_MLP_ACC_EXACT = 0, ORC_MLP_ACC_FP16_K16 = 1, ORC_MLP_ACC_IDEAL = 2 };
enum { ORC_ACT_NONE = 0, ORC_ACT_RELU = 1, ORC_ACT_LOGISTIC = 2, ORC_ACT_EXPONENTIAL = 3 };

/* Model descriptor: what Testbed::reset_network + load_snapshot leave behind
 * for the render path (src/testbed.cu:3928-3977, 5285-5463; nerf_network.h:81-101). */
typedef struct orc_nerf_model {
	/* multiresolution hash grid (tcnn GridEncoding, Hash / Linear) */
	uint32_t n_levels;
	uint32_t n_features_per_level;
	uint32_t log2_hashmap_size;
	uint32_t base_resolution;
	float per_level_scale;
	/* fully fused MLPs (bias-free, ReLU hidden, no output activation) */
	uint32_t n_neurons;
	uint32_t n_hidden_density; /* configs/nerf/base.json: 1 */
	uint32_t n_hidden_rgb;     /* configs/nerf/base.json: 2 */
	uint32_t density_out_dims; /* 16, nerf_network.h:88-90 */
	uint32_t rgb_activation;   /* Logistic for LDR data, testbed_nerf.cu:2653 */
	uint32_t density_activation; /* Exponential, nerf.h:151-152 */
	/* parameters in tcnn order: density MLP, rgb MLP, grid (nerf_network.h:356-371); fp16 */
	const uint16_t* params;
	uint64_t n_params;
	/* scene */
	float aabb_min[3], aabb_max[3];               /* m_aabb (train aabb) */
	float render_aabb_min[3], render_aabb_max[3]; /* m_render_aabb */
	float render_aabb_to_local[9];                /* column-major mat3 */
	uint32_t max_cascade;
	float cone_angle_constant;
	const uint8_t* density_grid_bitfield; /* 8 levels x 128^3 bits */
	/* How kernel_grid sums the 8 corners of a level (tiny-cuda-nn is un-vendored and un-pinned, .gitmodules:13-15; both
	 * published sequences are restated): ORC_GRID_ACC_FMA = `result = fma((T)weight, grid_val(...), result)`, the tvec-era
	 * kernel (the tcnn that has tcnn::vec3 / mat4x3, which the reference's sources use throughout); ORC_GRID_ACC_LEGACY =
	 * `result[f] += (T)(weight * (float)val[f])`, the kernel before the tvec refactor. */
	uint32_t grid_accumulate;
	/* configs/nerf/frequency.json (the original NeRF's architecture): tcnn Frequency encodings in place of the hash grid and
	 * of the spherical harmonics -- out[j] = sin(2^((j / 2) % n_freq) * pi * x[j / (2 n_freq)] + (j % 2) * pi / 2), no
	 * parameters -- feeding CutlassMLPs (256 wide, 7 + 1 hidden layers in that config). pos_encoding / dir_encoding: 0 = grid / SH
	 * degree 4 (base.json), 1 = Frequency with that many frequencies. mlp_alignment: 16 for FullyFusedMLP, 8 for CutlassMLP --
	 * NerfNetwork pads encoding widths, the rgb network's input and output to it (nerf_network.h:81-100). 0 = 16. */
	uint32_t pos_encoding, pos_n_frequencies;
	uint32_t dir_encoding, dir_n_frequencies;
	uint32_t mlp_alignment;
	/* How a layer of the MLPs sums (orc_nerf.c mlp_layer*): ORC_MLP_ACC_EXACT = every dot product exact, rounded to fp32 and then to
	 * fp16 (the default: what fp32 MFMA accumulators give up to summation order); ORC_MLP_ACC_FP16_K16 = the running sum rounded to
	 * fp16 after every 16-wide block of the inner dimension -- tcnn FullyFusedMLP's __half accumulator fragments, the reference's
	 * own arithmetic as far as it can be restated; ORC_MLP_ACC_IDEAL = the whole network (interpolation, encodings, MLPs) in float64
	 * without any intermediate rounding. EXACT and FP16_K16 bracket the reference; IDEAL is the yardstick both are measured against. */
	uint32_t mlp_accumulate;
	/* derived, filled by orc_nerf_prepare */
	void* prepared;
} orc_nerf_model;
/* tcnn FrequencyEncoding of n x n_dims inputs: out n x (n_dims * 2 * n_frequencies) fp16 */
void orc_frequency_encode(uint32_t n, uint32_t n_dims, uint32_t n_frequencies, const float* x, uint16_t* out);

typedef struct orc_camera {
	float matrix[12]; /* camera-to-world 4x3, column-major: x axis, y axis, z axis (fwd), position */
	int32_t width, height;
	float focal_length[2];  /* in pixels: relative_focal_length * res[fov_axis] * zoom */
	float screen_center[2]; /* after Testbed::render_screen_center */
	uint32_t spp_index;     /* render_buffer.spp */
	int32_t snap_to_pixel_centers;
	float near_distance;
	int32_t lens_mode;    /* ELensMode (common.h:223-230): 0 Perspective, 1 OpenCV, 3 LatLong, 4 OpenCVFisheye, 5 Equirectangular */
	float lens_params[7]; /* OpenCV: k1 k2 p1 p2; fisheye: k1 k2 k3 k4 */
	float aperture_size;  /* depth of field (uv_to_ray, common_device.cuh:471-477); 0 = pinhole */
	float focus_z;        /* plane_z = m_slice_plane_z + m_scale */
	/* camera_matrix1 + rolling_shutter of Testbed::render_frame: when has_matrix1 is set (and matrix1 differs from matrix), pixel (u, v)
	 * is rendered by camera_slerp(matrix, matrix1, rs.x + rs.y u + rs.z v + rs.w ld_random_val(spp, idx * 72239731))
	 * (get_xform_given_rolling_shutter, common_device.cuh:651-659); depth is measured along matrix1 (src/testbed_nerf.cu:2412).
	 * tcnn's slerp(mat3, mat3, t) is not in the mount: restated from the GLM-derived quat_cast / slerp / mat3_cast its vec.h is built on. */
	int32_t has_matrix1;
	float matrix1[12];
	float rolling_shutter[4];
} orc_camera;
/* the camera of one pixel of such a frame (out12: column-major 4x3) */
void orc_camera_at_pixel(const orc_camera* cam, float u, float v, uint32_t idx, float* out12);

/* camera-space direction of image coordinate (u, v) under the camera's lens (uv_to_ray, common_device.cuh:441-462) */
void orc_lens_direction(const orc_camera* cam, float u, float v, float* dir3);
/* the depth-of-field step of uv_to_ray on a camera-space origin / direction pair already rotated into world space */
void orc_apply_aperture(const orc_camera* cam, float u, float v, float* origin3, float* dir3);

typedef struct orc_render_opts {
	float min_transmittance;      /* m_nerf.render_min_transmittance */
	int32_t train_in_linear_colors; /* m_nerf.training.linear_colors */
	int32_t depth_test;           /* 1: shade_kernel_nerf_geometry depth test against depth_buffer */
	int32_t capped_skip;          /* 1: the 200-iteration skip of nerf_device.cuh:497-534 (trace_mesh) */
	int32_t n_threads;            /* OpenMP threads, <=0: all */
	int32_t render_mode;          /* 0/1 Shade; 2 AO, 3 Positions, 4 Depth (composite_kernel_nerf, testbed_nerf.cu:689-702) */
	float depth_scale;            /* Depth mode: 1 / dataset scale */
	const float* envmap;          /* m_envmap.inference_view(): env_w x env_h RGBA, lat-long (read_envmap, envmap.cuh:24-50); NULL = none */
	int32_t env_w, env_h;
} orc_render_opts;
void orc_read_envmap(const float* envmap, int32_t res_x, int32_t res_y, const float* dir3, float* out4);

typedef struct orc_render_stats {
	uint64_t n_rays;
	uint64_t n_rays_alive_after_init;
	uint64_t n_rays_hit;     /* rays that reached shading (alpha > 0.001) */
	uint64_t n_samples;      /* network queries that were composited or generated (sum over rays of marched samples) */
} orc_render_stats;

/* NerfPayload, nerf_device.cuh:144-152 (same member order, natural alignment). */
typedef struct orc_payload {
	float origin[3];
	float dir[3];
	float t;
	float max_weight;
	uint32_t idx;
	uint16_t n_steps;
	uint8_t alive;
	uint8_t pad;
} orc_payload;

int orc_nerf_prepare(orc_nerf_model* m);
void orc_nerf_release(orc_nerf_model* m);

/* grid layout: offsets (in entries), resolutions and scales per level. */
int orc_grid_layout(const orc_nerf_model* m, uint32_t* offsets /*n_levels+1*/, uint32_t* resolutions, float* scales);
uint64_t orc_n_params(const orc_nerf_model* m); /* total fp16 parameter count */

/* K5a: hash-grid encode. pos01: n x 3 floats in [0,1]; out: n x (L*F) fp16 */
void orc_grid_encode(const orc_nerf_model* m, uint32_t n, const float* pos01, uint16_t* out);
/* K5c: SH degree 4 of dir01 (n x 3, in [0,1]); out n x 16 fp16 */
void orc_sh4_encode(uint32_t n, const float* dir01, uint16_t* out);
/* K5: full network. out: n x 4 fp16 (rgb logits, density logit) */
void orc_nerf_network(const orc_nerf_model* m, uint32_t n, const float* pos01, const float* dir01, uint16_t* out);
/* d density logit / d (warped) position, what ERenderMode::Normals composites (tcnn input_gradient, src/testbed_nerf.cu:2106-2107): n x 3 */
void orc_density_gradient(const orc_nerf_model* m, uint32_t n, const float* pos01, float* grad);

/* K8/K9: density grid (float, Morton order, (max_cascade+1) x 128^3) -> bitfield (8 x 128^3 / 8 bytes) */
void orc_density_grid_to_bitfield(const float* grid, uint32_t max_cascade, uint8_t* bitfield, float* out_mean);

/* Density-grid refresh from the network: one iteration of Testbed::update_density_grid_nerf (src/testbed_nerf.cu:2772-2861)
 * on `grid` ((max_cascade + 1) x 128^3 floats, Morton order), without the mean / bitfield step. */
typedef struct orc_pcg32 { uint64_t state, inc; } orc_pcg32;
void orc_pcg32_seed(orc_pcg32* r, uint64_t initstate, uint64_t initseq);
uint32_t orc_pcg32_next_uint(orc_pcg32* r);
void orc_pcg32_advance(orc_pcg32* r, uint64_t delta);
void orc_update_density_grid(const orc_nerf_model* m, float* grid, uint32_t max_cascade, orc_pcg32* rng, uint32_t* ema_step, float decay, uint32_t n_uniform,
                             uint32_t n_nonuniform);

/* ---- training step, orc_nerf.c */
typedef struct orc_train_image { /* TrainingImageMetadata + TrainingXForm, the fields the default path reads */
	const void* pixels; /* RGBA: uint8 sRGB straight alpha (type 1) or float linear premultiplied (type 3) */
	int32_t type;
	int32_t res[2];
	float focal[2];
	float principal[2];
	int32_t lens_mode;
	float lens_params[7];
	float xform[12];
} orc_train_image;
typedef struct orc_train_opts {
	uint32_t n_rays, n_images;
	orc_pcg32 rng;
	int32_t snap_to_pixel_centers, random_bg_color, linear_colors, color_space, loss_type;
	float background[3];
	float near_distance, loss_scale, density_grid_mean;
} orc_train_opts;
uint32_t orc_train_generate_samples(const orc_nerf_model* m, const orc_train_image* images, const orc_train_opts* o, uint32_t max_samples, uint32_t* numsteps,
                                    uint32_t* base_out, float* rays6, float* coords);
void orc_train_loss(const orc_nerf_model* m, const orc_train_image* images, const orc_train_opts* o, const uint32_t* numsteps, const uint32_t* base_in,
                    const float* rays6, const float* coords, const uint16_t* network_output, uint32_t* compacted_numsteps, float* loss_out, uint16_t* dloss);


/* sampling sequences, random_val.cuh:207-370 */
float orc_ld_random_val(uint32_t index, uint32_t seed, uint32_t dim);
void orc_ld_random_pixel_offset(uint32_t spp, float* out2);

/* K1: primary ray for pixel (x,y) */
void orc_init_ray(const orc_nerf_model* m, const orc_camera* cam, uint32_t x, uint32_t y, orc_payload* out);
/* K2: start-of-ray jitter + skip to first occupied voxel */
void orc_advance_pos(const orc_nerf_model* m, const orc_camera* cam, orc_payload* p);

/* K3-K6 for one ray: marches until termination. rgba[4]/depth updated in place. returns #samples. */
uint32_t orc_trace_ray(const orc_nerf_model* m, const float* cam_matrix, const orc_render_opts* o, orc_payload* p, float* rgba, float* depth);

/* K1-K7 for a whole frame; frame_buffer (W*H*4) and depth_buffer (W*H) are read-modify-write
 * exactly like CudaRenderBufferView after clear() (or after the mesh pass in Geometry mode). */
void orc_render_nerf(const orc_nerf_model* m, const orc_camera* cam, const orc_render_opts* o,
                     float* frame_buffer, float* depth_buffer, orc_render_stats* stats);

/* Generic tracer over prepared payloads (irradiance probes): out rgba n x 4, depth n. */
void orc_trace_payloads(const orc_nerf_model* m, const float* cam_matrix, const orc_render_opts* o,
                        uint32_t n, orc_payload* payloads, float* rgba, float* depth, orc_render_stats* stats);

/* P1: accumulate (linear colour space) + tonemap (identity curve) -> rgba_out (W*H*4) */
void orc_accumulate(uint32_t n_pixels, const float* frame_buffer, float* accumulate_buffer, float sample_count);
void orc_tonemap(uint32_t n_pixels, const float* accumulate_buffer, const float* background_rgba, float exposure, int32_t to_srgb, float* rgba_out);
/* the same with EColorSpace: 0 Linear (as above), 1 SRGB (samples averaged as sRGB values, src/render_buffer.cu:241-248, 324-340, 537-541) */
void orc_accumulate_cs(uint32_t n_pixels, const float* frame_buffer, float* accumulate_buffer, float sample_count, int32_t color_space);
void orc_tonemap_cs(uint32_t n_pixels, const float* accumulate_buffer, const float* background_rgba, float exposure, int32_t to_srgb, int32_t color_space,
                    float* rgba_out);

float orc_srgb_to_linear(float v);
float orc_linear_to_srgb(float v);

#ifdef __cplusplus
}
#endif
#endif
