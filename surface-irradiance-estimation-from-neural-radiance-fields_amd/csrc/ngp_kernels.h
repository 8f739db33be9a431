// Host/device shared declarations for libngp_hip's kernels (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ngp {

constexpr uint32_t NERF_GRIDSIZE = 128;
constexpr uint32_t NERF_GRID_N_CELLS = NERF_GRIDSIZE * NERF_GRIDSIZE * NERF_GRIDSIZE;
constexpr uint32_t NERF_CASCADES = 8;
constexpr uint32_t NERF_STEPS = 1024;
constexpr float MAX_DEPTH = 16384.0f;
constexpr uint32_t MARCH_ITER = 10000; // reference src/testbed_nerf.cu:46

// The fused kernel is specialised for the configs/nerf/base.json architecture:
// 8 levels x 4 features -> 32, density 32->64->16, rgb 32->64->64->16, all on 16x16x32 f16 MFMA tiles.
constexpr int N_LEVELS = 8;
constexpr int N_FEATURES = 4;
constexpr int MLP_WIDTH = 64;
// MFMA A-operand fragments (weights): density L1 (4), density out (2), rgb L1 (4), rgb L2 (8), rgb out (2)
constexpr int FRAG_D0 = 0, FRAG_D1 = 4, FRAG_R0 = 6, FRAG_R1 = 10, FRAG_R2 = 18, N_FRAGS = 20;
// other rgb heads (n_hidden_layers 1 or 3, configs/nerf/base_1layer.json / base_3layer.json): 0 or 2 layers of 64x64 at
// FRAG_R1, the output layer behind them; the fragment buffer always holds N_FRAGS_MAX
constexpr int MAX_RGB_MID = 2, N_FRAGS_MAX = FRAG_R1 + 8 * MAX_RGB_MID + 2;
// rgb_mid -1 (configs/nerf/linear.json, base_0layer.json): the rgb head is its output layer alone, one fragment at FRAG_R0; a density
// head without a hidden layer (linear.json) is one fragment at FRAG_D0
constexpr int n_frags_for(int rgb_mid) { return rgb_mid < 0 ? FRAG_R0 + 1 : FRAG_R1 + 8 * rgb_mid + 2; }
// behind them, four fragments of the density head's first layer transposed: the backward pass of ERenderMode::Normals (nerf_device.h)
constexpr int N_NORMALS_FRAGS = 4, FRAG_NORMALS = N_FRAGS_MAX;

struct LevelInfo {
	float scale;
	uint32_t res;
	uint32_t size;   // entries in this level
	uint32_t offset; // first entry
	uint32_t hashed; // 1: spatial hash, 0: dense x + y*res + z*res^2
	uint32_t mask;   // size-1 if size is a power of two, else 0
	// xor layout (ModelParams::xgrid): byte offset of corner (x, y, z) = ((8x ^ y*mul_y8 ^ z*mul_z8) & mask8) | base8,
	// valid while max(x, y, z) of the cell's low corner <= coord_max (always for hashed levels; see ngp_api.cpp
	// build_xor_layout)
	uint32_t coord_max;
	uint32_t mul_y8, mul_z8, mask8, base8;
	uint32_t xor_disabled; // 1: the level has no xor form (a dense index whose uint32 strides wrapped, not a power-of-two shape)
};
static_assert(sizeof(LevelInfo) == 48, "LevelInfo layout");

struct NerfPayload { // nerf_device.cuh:144-152
	float origin[3];
	float dir[3];
	float t;
	float max_weight;
	uint32_t idx;
	uint16_t n_steps;
	uint8_t alive;
	uint8_t pad;
};
static_assert(sizeof(NerfPayload) == 40, "NerfPayload layout");

// The "wide" architecture (configs/nerf/frequency.json, the original NeRF's: Frequency encodings, CutlassMLPs 128 or 256 wide with
// any number of hidden layers) runs in wide_kernels.hip on v_mfma_f32_32x32x16_f16 tiles. Every layer's weights are stored as MFMA
// A fragments: fragment (m, kb) of a layer = rows TM m .. (output neurons), columns TK kb .. (inputs) of its row-major [out][in]
// matrix (TM x TK = WIDE_TILE_M x WIDE_TILE_K: 16 x 32 or 32 x 16), 64 lanes x 8 fp16 with lane (r = lane % TM, h = lane / TM) holding
// W[TM m + r][TK kb + 8h + j]; rows and columns beyond the matrix are zeros. Layer l's fragments start at uint4 index frag_offset and
// are ordered [m][kb][lane].
// WIDE_MFMA16 = 1 builds the wide-MLP kernels on v_mfma_f32_16x16x32_f16 (A fragments of 16 rows x 32 columns) instead of
// v_mfma_f32_32x32x16_f16 (32 x 16): the same FLOP per cycle on paper, less power per FLOP under a clock that this kernel holds at
// its power limit (MI355X_MICROARCH.md, MFMA shapes). Host fragment builder (ngp_api.cpp) and kernels (wide_kernels.hip) switch
// together. Measured (round 3, same box, same run): the network alone 3.95 instead of 4.16 ms for 4.2 M samples (+5 %), the render
// kernel 933 instead of 952 TFLOP/s (-2 %: twice the address / LDS instructions per FLOP, and more of the ray state spilled) -- the
// 32x32x16 form ships; all of tests/test_frequency_gpu.py passes either way (python build.py with NGP_BUILD_DEFINES=-DWIDE_MFMA16=1).
#ifndef WIDE_MFMA16
#define WIDE_MFMA16 0
#endif
constexpr int WIDE_TILE_M = WIDE_MFMA16 ? 16 : 32; // rows (neurons) of an A fragment
constexpr int WIDE_TILE_K = WIDE_MFMA16 ? 32 : 16; // its columns (inputs)
constexpr int WIDE_MAX_LAYERS = 24;
struct WideLayer {
	uint32_t frag_offset;
	uint16_t n_kblocks; // WIDE_TILE_K-wide blocks of the input, zero-padded to K = 128 or 256 (what wide_kernels.hip is instantiated for)
	uint16_t n_mtiles;  // WIDE_TILE_M-row tiles of the (zero-padded) output
};
struct WideModel {
	const uint4* frags;
	uint32_t width;                        // n_neurons: 128 or 256; 0 = the model is not of this architecture
	uint32_t pos_freqs, dir_freqs;         // n_frequencies of the position / direction encodings; dir_freqs 0 = SphericalHarmonics degree 4 ...
	uint32_t pos_identity, dir_identity;   // ... unless the encoding is tcnn's Identity (configs/nerf/none.json): the inputs themselves, padded with ones
	uint32_t enc_dims, dir_dims, rgb_in;   // padded widths (nerf_network.h:81-100): position encoding, direction encoding, rgb network input
	uint32_t n_hidden_density, n_hidden_rgb;
	// density: layers [0, n_hidden_density] (the last one is the 16-wide output layer); rgb: the n_hidden_rgb + 1 layers behind them
	WideLayer layers[WIDE_MAX_LAYERS];
	// ERenderMode::Normals (the density network's input gradient): its hidden layers TRANSPOSED -- layers_t[l] maps the gradient at layer l's
	// neurons to layer l's inputs (width x width fragments; layer 0's rows beyond the encoding are zeros) -- for up to WIDE_MAX_NORMALS_LAYERS
	// hidden layers (0 tiles in layers_t[0]: not prepared), and row 0 of the density output layer (`width` halves at frags + out_row0_offset)
	WideLayer layers_t[8];
	uint32_t out_row0_offset;
};
constexpr int WIDE_MAX_NORMALS_LAYERS = 8;

struct ModelParams {
	const uint2* grid;       // fp16 x4 per entry, tcnn order (level-major, entry-major)
	const char* xgrid;       // the same entries in the xor layout: one index formula for dense and hashed levels
	const uint4* wfrags;     // [N_FRAGS][64] x 8 fp16, MFMA A fragments in lane order
	const uint8_t* bitfield; // 8 x 128^3 bits
	const uint32_t* coarse;  // 8 x 32^3 bits: bit (morton >> 6) of mip m is set iff any cell of that 4x4x4 block is occupied
	LevelInfo levels[N_LEVELS];
	float aabb_min[3], aabb_diag[3];
	float raabb_min[3], raabb_max[3];
	float r2l[9]; // render_aabb_to_local, column-major
	uint32_t max_cascade;
	float cone_angle;
	uint32_t rgb_act, density_act;
	int32_t rgb_mid;       // 64x64 layers of the rgb head (n_hidden_layers - 1): 1 for configs/nerf/base.json, -1 for a head without a hidden layer
	uint32_t density_linear; // the density head has no hidden layer (configs/nerf/linear.json; rgb_mid is -1 then)
	uint32_t r2l_identity; // render_aabb_to_local is the identity (the usual case): skip the matrix product
	uint32_t diag_pow2;    // every component of aabb_diag is a power of two: x / diag == x * (1/diag) bit for bit
	float aabb_inv_diag[3];
	uint32_t grid_bytes, xgrid_bytes; // sizes of the two tables (buffer-load descriptors: out-of-range gathers read zeros)
	WideModel wide;                   // wide.width != 0: grid / xgrid / wfrags / levels are unused, wide_kernels.hip renders
};
constexpr uint32_t COARSE_WORDS_PER_MIP = 32 * 32 * 32 / 32;

struct CameraParams {
	float m[12]; // column-major 4x3
	int32_t width, height;
	float focal[2];
	float screen_center[2];
	float pixel_offset[2]; // ld_random_pixel_offset(snap ? 0 : spp), per-frame constant
	uint32_t spp;
	float near_distance;
	int32_t lens_mode; // ELensMode: 0 Perspective, 1 OpenCV, 3 LatLong, 4 OpenCVFisheye, 5 Equirectangular
	float lens_params[7];
	float aperture_size, focus_z; // depth of field: 0 = pinhole
	// a frame whose camera moves (camera_matrix0 -> camera_matrix1, Testbed::render_frame): every pixel gets the camera of its time
	// rolling_shutter.x + .y u + .z v + .w ld_random_val(spp, idx * 72239731); m is camera0. moving = 0: m alone (camera0 == camera1)
	float m1[12];
	float rolling_shutter[4];
	int32_t moving;
};

struct FrameParams {
	float4* frame_buffer;
	float* depth_buffer;
	uint32_t* queue;               // [0]: next strip (quarter tile) of this rank's share
	uint32_t* xqueue;              // [0..7]: one queue per XCD over an eighth of the share each (nullptr: the single queue); nerf_kernels.hip
	unsigned long long* counters;  // [0] alive after init, [1] hit, [2] samples: accumulators of the launch; its last wave moves them to `results` and leaves the slot zeroed
	unsigned long long* results;   // [0..2]: what ngp_get_render_stats reads; [3] the launch(es) in ticks of the 100 MHz device clock; [4] scratch (start stamp, left zero)
	uint32_t* done;                // waves of this launch that have left
	uint32_t n_waves;              // waves launched (set by the launcher)
	int32_t add_results;           // 0: the launch's totals replace results (first launch of a call), 1: they are added (further samples per pixel)
	uint32_t tiles_x, tiles_y;
	uint32_t n_local_tiles;
	uint32_t shard_index, shard_count;
	float min_transmittance;
	int32_t linear_colors;
	int32_t depth_test;
	int32_t packed;           // 1: pixel (local tile q, slot s) is written at q*64+s (tile-packed layout for the RCCL gather) instead of x+W*y
	int32_t tune[8];          // refill_min, skip_steps, go_min, max_stall, samples a ray may emit per round while most of a wave's ray slots are live /
	                          // at most, block_jumps (0: the reference's one-voxel steps only), share (a wave without work takes over half the
	                          // rays of a busy wave of its workgroup) -- nerf_kernels.hip; validated by ngp_set_schedule
	// direct output (1 spp, no mesh pass): the kernel writes the final pixel -- accumulate_kernel + tonemap_kernel
	// (src/render_buffer.cu:228-262, 529-561) folded into ray setup / shading -- into frame_buffer = the caller's image
	int32_t outside_possible; // the render box is not contained in the outermost cascade's cube (kernel selection)
	int32_t render_mode;      // ngp_render_mode: 0/1 Shade, 2 AO, 3 Positions, 4 Depth (composite_kernel_nerf :689-702)
	float depth_scale;
	int32_t direct, to_srgb, color_space;
	float background[4];
	float exposure_scale;
	const float4* envmap;     // m_envmap.inference_view(): lat-long radiance behind the NeRF (read_envmap), nullptr = none
	int32_t env_w, env_h;
	unsigned long long* prof; // diagnostic build only (NGP_PROFILE_SECTIONS=1): [refill, march, network, composite, iterations, passes] cycle sums
	// diagnostic build, NGP_PROFILE_TRACE=stride: timelines of single waves (nerf_kernels.hip fused_body; read back by ngp_get_profile_trace,
	// decoded by tools/wave_trace.py). trace[0] counts the waves that were dealt rays; every stride-th of them gets a header of 16 words at
	// trace + 16 + 16 slot and records of 16 words at trace + 16 + 16 cap_waves + 16 (slot cap_iters + round)
	uint32_t* trace;
	uint32_t trace_stride, trace_cap_waves, trace_cap_iters;
	int32_t prof_level;       // NGP_PROFILE_SECTIONS: 1 section stamps, 2 also inside the network section
};

// ---- irradiance probes
struct ProbeParams {
	int32_t mode; // 0 centre fan (K10), 1 shell position looking inward (K11), 2 Halton-jittered centres (K12), 3 a grid_x x grid_y lattice of K11 probes
	uint32_t n_theta, n_phi, n_origin;
	float origin[3];
	float center[3]; // render_aabb.center()
	uint32_t n_rays;
	float4* ray_rgba; // one shaded RGBA per probe ray (zero when the ray saw nothing)
	uint32_t grid_x, grid_y; // mode 3: probe g = i + grid_x * j sits at center + shell_radius * cylindrical_to_dir_nerf((i + .5) / grid_x, (j + .5) / grid_y)
	float shell_radius;
};

// ---- geometry mode (meshes)
struct Triangle { // triangle.cuh:163 -- 36 B
	float a[3], b[3], c[3];
};
struct TriangleBvhNode { // triangle_bvh.cuh:28-32 -- 32 B
	float bmin[3], bmax[3];
	int left_idx; // negative: leaf, triangles [-left_idx-1, -right_idx-1)
	int right_idx;
};
struct MeshRef {
	const TriangleBvhNode* nodes;
	const Triangle* tris;
	float bmin[3], bmax[3];
	uint32_t n_tris, n_nodes;
};
struct MeshSceneParams {
	const MeshRef* meshes;
	uint32_t n_meshes;
	float scene_min[3], scene_max[3]; // root bb inflated by 4 (load_scene, testbed_geometry_training.cu:3185-3189)
};
struct MeshShadeParams { // BRDFParams (common.h:167-177) + m_sun_dir / m_up_dir
	float sun_dir[3], up_dir[3];
	float metallic, subsurface, specular, roughness, sheen, clearcoat, clearcoat_gloss;
	float basecolor[3], ambientcolor[3];
};
struct IrradianceMap { // E(n) tabulated at the probe texture's texel directions; nullptr = not computed
	const float4* irradiance; // grid_x * grid_y tables (one when grid_x == 0) of n_theta * n_phi texels
	uint32_t n_theta, n_phi;
	uint32_t grid_x, grid_y;  // ShadeGridEnvMap: the probes around the direction of (surface point - center) are blended
	float center[3];
};

} // namespace ngp
