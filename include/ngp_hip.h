/*
 * libngp_hip.so -- C ABI of the MI355X-native NeRF volume renderer.
 *
 * The reference (fnysalehi/Surface-Irradiance-Estimation-from-Neural-Radiance-Fields) has no FFI seam:
 * its public surface is the C++ class ngp::Testbed (include/neural-graphics-primitives/testbed.h)
 * re-exported by pybind11 (src/python_api.cu). This header is the seam the build inserts UNDER that
 * class: every entry point names the Testbed member (file:line in the reference) it replaces. Plain
 * pointers and sizes only; no exceptions cross the boundary (0 = ok, otherwise ngp_last_error()).
 *
 * Threading: a context is not re-entrant -- one host thread per context. All inputs are copied during
 * the call; no borrowed pointer survives a call.
 */
#ifndef NGP_HIP_H
#define NGP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NGP_API __attribute__((visibility("default")))

typedef struct ngp_ctx ngp_ctx;

enum ngp_activation { NGP_ACT_NONE = 0, NGP_ACT_RELU = 1, NGP_ACT_LOGISTIC = 2, NGP_ACT_EXPONENTIAL = 3 };

/* ERenderMode subset (common.h:58-72). Only Shade is on the hot path; AO..Cost are "next" (SURVEY 8f-3). */
/* ELensMode, common.h:223-230 */
enum ngp_lens_mode { NGP_LENS_PERSPECTIVE = 0, NGP_LENS_OPENCV = 1, NGP_LENS_FTHETA = 2, NGP_LENS_LATLONG = 3, NGP_LENS_OPENCV_FISHEYE = 4, NGP_LENS_EQUIRECTANGULAR = 5 };

enum ngp_render_mode {
	NGP_RENDER_SHADE = 0,
	NGP_RENDER_SHADE_ENVMAP = 1, /* ERenderMode::ShadeEnvMap: meshes lit by the NeRF-derived irradiance probe (ngp_compute_envmap) */
	/* G-buffer modes of composite_kernel_nerf (src/testbed_nerf.cu:689-702): the per-sample colour is replaced, the compositing is unchanged,
	 * and shade_kernel_nerf skips the sRGB -> linear conversion (:1393). NeRF mode only. */
	NGP_RENDER_AO = 2,        /* rgb = alpha of the sample */
	NGP_RENDER_POSITIONS = 3, /* rgb = (pos - 0.5) / 2 + 0.5 */
	NGP_RENDER_DEPTH = 4,     /* rgb = dot(cam_fwd, pos - ray origin) * depth_scale */
	NGP_RENDER_COST = 5,      /* ERenderMode::Cost (shade_kernel_nerf :1382-1384): grey = samples composited on the ray / 128, opaque. The reference's
	                           * payload.n_steps holds that count for rays that saturate and the last compaction batch's for rays that leave the
	                           * volume (:466, :730); here it is the ray's total in both cases */
	NGP_RENDER_NORMALS = 7,   /* ERenderMode::Normals (composite_kernel_nerf :688-693, shade_kernel_nerf :1379-1381): every sample's colour is the unit
	                           * vector opposite to the density's gradient w.r.t. the position -- tcnn's input_gradient (src/testbed_nerf.cu:2106-2107):
	                           * a backward pass through the density MLP and the grid encoding per sample; the pixel is (0.5 n + 0.5) alpha. Grid models. */
	NGP_RENDER_SHADE_GRID_ENVMAP = 6 /* ERenderMode::ShadeGridEnvMap, the fork's default (testbed.h:880): meshes lit by the GRID of NeRF-derived
	                           * irradiance probes (ngp_compute_envmap_grid), position-dependent */
};

/* ETestbedMode subset (common.h:35-43): Nerf, and the fork's Geometry mode (meshes + NeRF, depth composited) */
enum ngp_testbed_mode { NGP_MODE_NERF = 0, NGP_MODE_GEOMETRY = 1 };

/* What Testbed::reset_network (src/testbed.cu:3844-4212) + load_nerf_post (src/testbed_nerf.cu:2652-2739)
 * + the snapshot (src/testbed.cu:5285-5463) leave behind for rendering. */
typedef struct ngp_model_desc {
	/* tcnn HashGrid encoding */
	uint32_t n_levels;
	uint32_t n_features_per_level;
	uint32_t log2_hashmap_size; /* 31 = tcnn DenseGrid: no level is ever hashed or capped */
	uint32_t base_resolution;
	float per_level_scale; /* explicit: the fork derives it with aabb_scale = 1, src/testbed.cu:3959-3966 */
	/* tcnn FullyFusedMLP / CutlassMLP density and rgb heads (nerf_network.h:81-101). Grid architecture: n_hidden_density 1 with
	 * n_hidden_rgb 0..3 (configs/nerf/base_0layer .. base_3layer.json; base.json: 2), or both 0 (linear.json). A head without a
	 * hidden layer is tcnn's CutlassMLP with one (padded output) x (input) matrix. */
	uint32_t n_neurons;
	uint32_t n_hidden_density;
	uint32_t n_hidden_rgb;
	uint32_t density_out_dims;
	uint32_t rgb_activation;     /* ngp_activation */
	uint32_t density_activation; /* ngp_activation */
	/* Trainer::serialize "params_binary": density MLP, rgb MLP, grid (nerf_network.h:356-371), fp16 */
	const uint16_t* params_fp16;
	uint64_t n_params;
	/* snapshot "density_grid_binary": (max_cascade+1) x 128^3 fp16, Morton order (src/testbed.cu:5339-5347) */
	const uint16_t* density_grid_fp16;
	uint64_t n_density_grid;
	float aabb_min[3], aabb_max[3];               /* m_aabb */
	float render_aabb_min[3], render_aabb_max[3]; /* m_render_aabb */
	float render_aabb_to_local[9];                /* column-major mat3 */
	uint32_t aabb_scale;                          /* dataset.aabb_scale -> max_cascade, src/testbed_nerf.cu:2729-2732 */
	float cone_angle_constant;                    /* src/testbed_nerf.cu:2736 */
	int32_t linear_colors;                        /* m_nerf.training.linear_colors */
	/* The second architecture the renderer implements: configs/nerf/frequency.json, the original NeRF's network (tcnn Frequency
	 * encodings, CutlassMLPs 128 or 256 wide with any number of hidden layers). All zero = the grid architecture above.
	 *   pos_encoding  0: the grid encoding described by the fields above; 1: Frequency with pos_n_frequencies (the grid fields are
	 *                 ignored, params_fp16 holds the two MLPs only); 2: Identity (configs/nerf/none.json: the position itself, padded
	 *                 with ones to mlp_alignment)
	 *   dir_encoding  0: SphericalHarmonics degree 4; 1: Frequency with dir_n_frequencies; 2: Identity (with pos_encoding 1 or 2)
	 *   mlp_alignment 16: FullyFusedMLP, 8: CutlassMLP (0 = 16) -- what encodings, the rgb network's input and its output are padded
	 *                 to (nerf_network.h:81-100). For the grid architecture it is the RGB network's alignment (:83): 8 makes the rgb
	 *                 output layer 8 rows (linear.json, base_0layer.json)
	 * Inference only: ngp_train_* refuse such a model. */
	uint32_t pos_encoding, pos_n_frequencies;
	uint32_t dir_encoding, dir_n_frequencies;
	uint32_t mlp_alignment;
} ngp_model_desc;

/* Arguments of Testbed::render_frame (testbed.h:561-575) that the NeRF path consumes. */
typedef struct ngp_camera {
	float matrix[12];       /* camera-to-world 4x3 column-major (m_camera) */
	int32_t width, height;
	float focal_length[2];  /* pixels: calc_focal_length, src/testbed.cu:4474-4476 */
	float screen_center[2]; /* render_screen_center */
	uint32_t spp_index;     /* render_buffer.spp() */
	int32_t snap_to_pixel_centers;
	float near_distance;    /* m_render_near_distance */
	/* m_nerf.render_lens when m_nerf.render_with_lens_distortion (uv_to_ray, common_device.cuh:416-483): ngp_lens_mode +
	 * parameters (OpenCV: k1 k2 p1 p2; OpenCVFisheye: k1 k2 k3 k4; FTheta: r0..r4 of the angle polynomial, then the resolution
	 * x, y the intrinsics refer to -- f_theta_undistortion, common_device.cuh:361-375). All zero = perspective. */
	int32_t lens_mode;
	float lens_params[7];
	/* depth of field (uv_to_ray, common_device.cuh:471-477): m_aperture_size and the focus distance plane_z = m_slice_plane_z + m_scale
	 * (src/testbed_nerf.cu:2342); aperture_size 0 or focus_z < 0 = pinhole */
	float aperture_size, focus_z;
	/* camera_matrix1 and rolling_shutter of Testbed::render_frame (testbed.h:561-575): when has_matrix1 is set and matrix1 differs
	 * from matrix (= camera_matrix0), every pixel is rendered by the camera of its own time
	 *   t = rolling_shutter[0] + [1] u + [2] v + [3] ld_random_val(spp_index, pixel * 72239731),  camera_slerp(matrix, matrix1, t)
	 * (get_xform_given_rolling_shutter, common_device.cuh:651-659; src/testbed_nerf.cu:1468) and depth is measured along matrix1.
	 * rolling_shutter (0, 0, 0, 1) -- what Testbed::render passes -- is motion blur over the whole interval. NeRF mode. */
	int32_t has_matrix1;
	float matrix1[12];
	float rolling_shutter[4];
} ngp_camera;

typedef struct ngp_render_opts {
	int32_t render_mode;      /* ngp_render_mode */
	float min_transmittance;  /* m_nerf.render_min_transmittance */
	float background[4];      /* m_background_color (sRGB, straight) */
	float exposure;           /* m_exposure */
	int32_t to_srgb;          /* !linear of Testbed::render (python_api.cu:124,189) */
	int32_t spp;              /* samples accumulated by ngp_render */
	/* camera-tile sharding across GPUs: this context renders 8x8-pixel tiles t with t % shard_count == shard_index */
	uint32_t shard_index, shard_count;
	int32_t testbed_mode;     /* ngp_testbed_mode: Geometry = render_geometry_mesh then render_geometry_nerf (src/testbed.cu:4833-4889) */
	/* 0: rgba/depth are W*H images (pixel x + W*y). 1: tile-packed -- only this shard's tiles, local tile q (global tile
	 * shard_index + q*shard_count, tiles numbered row-major over ceil(W/8) x ceil(H/8)) occupies pixels [64q, 64q+64),
	 * slot (x&7) + 8*(y&7); buffers hold 64 * ngp_packed_tiles(...) pixels. This is the layout the per-frame RCCL
	 * all_gather moves, so no pack pass is needed on the sending side. */
	int32_t packed_output;
	float depth_scale;        /* NGP_RENDER_DEPTH: 1 / dataset scale in the reference (src/testbed_nerf.cu:2478); 0 selects 1 / 0.33 (NERF_SCALE) */
	int32_t color_space;      /* EColorSpace in which samples are averaged and the background is blended: 0 Linear, 1 SRGB
	                           * (accumulate_kernel / tonemap_kernel, src/render_buffer.cu:241-248, 324-340, 537-541; run.py --nerf_compatibility) */
} ngp_render_opts;

/* BRDFParams (common.h:167-177) + m_sun_dir / m_up_dir (testbed.h:875-876) used by shade_kernel_mesh_geometry */
typedef struct ngp_geometry_opts {
	float sun_dir[3], up_dir[3];
	float metallic, subsurface, specular, roughness, sheen, clearcoat, clearcoat_gloss;
	float basecolor[3], ambientcolor[3];
} ngp_geometry_opts;

typedef struct ngp_render_stats {
	uint64_t n_rays;
	uint64_t n_rays_alive_after_init;
	uint64_t n_rays_hit;
	uint64_t n_samples;      /* network queries composited */
	float kernel_ms;         /* duration of the fused march/encode/MLP/composite kernel, HIP events on its stream */
	float frame_ms;          /* whole frame on the device (clear .. tonemap), HIP events */
	float kernel_device_ms;  /* the fused kernel from its first wave's start to its last wave's exit, read inside the kernel from the chip's
	                          * 100 MHz clock (s_memrealtime): unlike the HIP events it does not include the time a launch waits for CUs
	                          * behind another frame's kernel when frames overlap on several streams */
} ngp_render_stats;

/* --- lifetime: Testbed::Testbed / ~Testbed (testbed.h:80-95). device = HIP device ordinal; -1 creates a host-only
 * context that can read/validate/write the file formats but cannot render (there is no CPU renderer). NULL on failure. */
NGP_API ngp_ctx* ngp_create(int device);
/* Several GPUs behind one context: Testbed's device list (m_devices, src/testbed.cu:5490-5616 -- a replica per device kept in step
 * by sync_device, auxiliary devices render on their own streams, peer copies return frame + depth to the primary). devices[0] is
 * the primary; every other call takes the returned context as if it had one device. ngp_render / ngp_render_device deal the camera's
 * 8x8 tiles round-robin to the devices, each renders its share tile-packed and pushes it to device 0 (hipMemcpyPeerAsync over xGMI),
 * which scatters the tiles into the image; nothing waits on the host. NeRF mode. The same ordinal may be listed twice (a rehearsal
 * of the path on one GPU). NULL on failure. */
NGP_API ngp_ctx* ngp_create_multi(const int* devices, int n_devices);
NGP_API int ngp_n_devices(const ngp_ctx* ctx);
NGP_API void ngp_destroy(ngp_ctx* ctx);
NGP_API const char* ngp_last_error(const ngp_ctx* ctx);
NGP_API const char* ngp_version(void);

/* --- model: Testbed::reset_network + Trainer::deserialize (src/testbed.cu:3844,5428-5433) */
NGP_API int ngp_set_model(ngp_ctx* ctx, const ngp_model_desc* desc);
/* Testbed::load_snapshot(std::istream&, bool is_compressed) (src/testbed.cu:5477-5488): msgpack, optionally zlib (.ingp) */
NGP_API int ngp_load_snapshot(ngp_ctx* ctx, const void* bytes, size_t n_bytes, int is_compressed);
/* Testbed::load_snapshot(const fs::path&) (src/testbed.cu:5465-5475) */
NGP_API int ngp_load_snapshot_file(ngp_ctx* ctx, const char* path);
/* Testbed::save_snapshot (src/testbed.cu:5219-5283), inference state only */
NGP_API int ngp_save_snapshot_file(ngp_ctx* ctx, const char* path, int compress);
/* the model as currently loaded, trained parameters and a refreshed occupancy grid included; params_fp16 / density_grid_fp16 point at the
 * context's own host copies (read-only, valid until the next call that changes the model) */
NGP_API int ngp_get_model(ngp_ctx* ctx, ngp_model_desc* out);
/* session state a snapshot carries beside the model and the camera (save_snapshot / load_snapshot, src/testbed.cu:5245-5263,
 * 5395-5418): m_background_color, m_exposure, m_sun_dir, m_up_dir, camera scale / aperture_size / autofocus_depth (= m_slice_plane_z).
 * get: what the loaded snapshot held (valid = 0 when nothing was loaded); set: what the next ngp_save_snapshot_file writes,
 * together with the camera (matrix12 may be NULL to leave the camera as it is). */
typedef struct ngp_session_state {
	int32_t valid;
	float background_color[4];
	float exposure;
	float sun_dir[3], up_dir[3];
	float camera_scale, aperture_size, autofocus_depth;
} ngp_session_state;
NGP_API int ngp_get_session_state(const ngp_ctx* ctx, ngp_session_state* out);
NGP_API int ngp_set_session_state(ngp_ctx* ctx, const ngp_session_state* state, const float* matrix12, const float* relative_focal_length2, int32_t fov_axis,
                                  const float* screen_center2, float zoom);
/* camera stored in the snapshot: m_camera, relative focal length, fov axis, screen center, zoom */
NGP_API int ngp_get_snapshot_camera(const ngp_ctx* ctx, float* matrix12, float* relative_focal_length2, int32_t* fov_axis, float* screen_center2, float* zoom);

/* --- data: Testbed::load_training_data (src/testbed.cu:125-152) -> ngp::load_nerf (src/nerf_loader.cu:273):
 * camera metadata of a transforms.json (or a directory of them); images are not decoded (inference path). */
NGP_API int ngp_load_training_data(ngp_ctx* ctx, const char* path);
NGP_API int ngp_n_training_views(const ngp_ctx* ctx);
/* per-view: ngp-space camera matrix (nerf_matrix_to_ngp applied), resolution, focal length (pixels), principal point */
NGP_API int ngp_get_training_view(const ngp_ctx* ctx, int view, float* matrix12, int32_t* resolution2, float* focal_length2, float* principal_point2);
/* the view's lens (read_lens, src/nerf_loader.cu:175-240): ngp_lens_mode and 7 parameters */
NGP_API int ngp_get_training_view_lens(const ngp_ctx* ctx, int view, int32_t* lens_mode, float* lens_params7);
NGP_API int ngp_get_dataset_info(const ngp_ctx* ctx, int32_t* aabb_scale, float* scale, float* offset3, int32_t* is_hdr);

/* --- render: Testbed::render_frame (src/testbed.cu:4694-4721) = clear + render_nerf (src/testbed_nerf.cu:2328-2488)
 * + accumulate/tonemap (src/render_buffer.cu:631-694) for opts->spp samples, then the copy that
 * Testbed::render_to_cpu (src/python_api.cu:197-201) does. rgba_out: host, H*W*4 floats, premultiplied alpha.
 * depth_out (nullable): host, H*W floats. */
NGP_API int ngp_render(ngp_ctx* ctx, const ngp_camera* cam, const ngp_render_opts* opts, float* rgba_out, float* depth_out);
/* Page-locked host memory for images, pooled by size (thread-safe; usable before any context exists). ngp_render copies
 * into such a buffer with one DMA at the link's rate; into ordinary memory the runtime stages the copy (about 2.5x slower for
 * a 1080p frame). pyngp's Testbed.render returns arrays that own such buffers. No counterpart in the reference (its
 * render_to_cpu reads back from a CUDA array, src/python_api.cu:197-201). */
NGP_API void* ngp_host_alloc(size_t bytes);
NGP_API void ngp_host_free(void* p);
/* Same frame, results left in device memory (d_rgba: W*H*4 floats, d_depth nullable: W*H floats) and enqueued on
 * `stream` (a hipStream_t, NULL = default stream) without synchronising: for callers that keep the image on the GPU
 * (RCCL gather of tiles, benchmarks). */
NGP_API int ngp_render_device(ngp_ctx* ctx, const ngp_camera* cam, const ngp_render_opts* opts, void* d_rgba, void* d_depth, void* stream);
/* number of local tiles (hence 64x that many pixels) of a tile-packed frame */
NGP_API uint32_t ngp_packed_tiles(int32_t width, int32_t height, uint32_t shard_index, uint32_t shard_count);
/* counters + timings of the last ngp_render / ngp_render_device (synchronises the stream) */
NGP_API int ngp_get_render_stats(ngp_ctx* ctx, ngp_render_stats* out);
/* per device of a multi-device context (0 = primary): the last frame's counters and timings of that device's share */
NGP_API int ngp_get_device_render_stats(ngp_ctx* ctx, int device_index, ngp_render_stats* out);
/* the same for the last n calls (oldest first; the context keeps 256), read once after a batch of asynchronous
 * ngp_render_device calls so that measuring does not serialise them */
NGP_API int ngp_get_render_history(ngp_ctx* ctx, int n, ngp_render_stats* out);

/* Scheduling of the persistent render kernel: knobs[0..n) = refill_min [16,64], skip_steps [1,64], go_min [1,64], max_stall [0,64],
 * samples a ray may emit per round while most of a wave's ray slots are live [1,8], at most once fewer are [1,8] (the reference's n_steps
 * between two compactions, src/testbed_nerf.cu:2080-2086), block_jumps {0,1}, share {0,1} (a wave that has run out of work takes over half
 * the rays of a busy wave of its workgroup). Performance only, except block_jumps: 1 (default) leaves empty 4^3 / 16^3 blocks of the occupancy grid in
 * one step, 0 walks them voxel by voxel exactly as if_unoccupied_advance_to_next_occupied_voxel (nerf_device.cuh:461-494) does.
 * Out-of-range values are refused (they would hang the kernel). No counterpart in the reference; the environment variable
 * NGP_TUNE sets the same list at ngp_create. */
NGP_API int ngp_set_schedule(ngp_ctx* ctx, const int32_t* knobs, int n);
/* Diagnostic, no counterpart in the reference: with NGP_PROFILE_SECTIONS=1|2 and NGP_PROFILE_TRACE=<stride> in the environment the render
 * kernel's stamped twin records the timeline of every stride-th wave that was dealt rays (one record per loop round: s_memtime at the
 * top of the round and after refill / march / network / composite, what the round carried). Copies the last frame's trace
 * (n_words 32-bit words at most; layout: csrc/ngp_kernels.h FrameParams::trace, decoder tools/wave_trace.py); out == NULL only
 * reports the capacities. */
NGP_API int ngp_get_profile_trace(ngp_ctx* ctx, uint32_t* out, uint64_t n_words, uint32_t* cap_waves, uint32_t* cap_iters);

/* --- stage entry points (what the reference launches as separate kernels; used by parity tests and tools)
 * K5a tcnn GridEncoding::inference (call site nerf_network.h:113-118): host pos01 n x 3 -> host fp16 n x (L*F) */
NGP_API int ngp_grid_encode(ngp_ctx* ctx, uint32_t n, const float* pos01, uint16_t* out_fp16);
/* tcnn DifferentiableObject::input_gradient(stream, 3, ...) as ERenderMode::Normals calls it (src/testbed_nerf.cu:2106-2107): d density
 * logit / d position for n positions, host n x 3 floats. Grid models. */
NGP_API int ngp_density_gradient(ngp_ctx* ctx, uint32_t n, const float* pos01, float* out_grad);
/* K5 NerfNetwork::inference_mixed_precision (nerf_network.h:105-139): pos01/dir01 n x 3 -> fp16 n x 4 (rgb logits, density logit) */
NGP_API int ngp_network_inference(ngp_ctx* ctx, uint32_t n, const float* pos01, const float* dir01, uint16_t* out_fp16);
/* K8/K9 update_density_grid_mean_and_bitfield (src/testbed_nerf.cu:2863-2880): the bitfield in use, 8 x 128^3 / 8 bytes */
NGP_API int ngp_get_density_bitfield(ngp_ctx* ctx, uint8_t* out, float* out_mean);
/* m_render_aabb / m_render_aabb_to_local (python_api.cu: testbed.render_aabb, .render_aabb_to_local): the crop box of the render,
 * min/max in ngp space; to_local9 column-major mat3 or NULL for identity */
NGP_API int ngp_set_render_aabb(ngp_ctx* ctx, const float* min3, const float* max3, const float* to_local9);
/* m_envmap (testbed.h:1297-1316; filled from the dataset's environment image or trained, src/testbed.cu:4194-4208): a lat-long RGBA
 * radiance map of width x height texels behind the NeRF. Every pixel with a valid ray starts from read_envmap(map, ray direction)
 * (envmap.cuh:24-50: bilinear, x periodic, y clamped; src/testbed_nerf.cu:1526-1528) and the NeRF is composited over it. NeRF mode.
 * rgba = NULL (or a zero size) removes the map. */
NGP_API int ngp_set_envmap(ngp_ctx* ctx, int32_t width, int32_t height, const float* rgba);
/* m_nerf.cone_angle_constant (python_api.cu: testbed.nerf.cone_angle_constant, run.py:167): 0 = fixed step size */
NGP_API int ngp_set_cone_angle_constant(ngp_ctx* ctx, float cone_angle_constant);
/* Testbed::update_density_grid_nerf (src/testbed_nerf.cu:2772-2861; kernels :185-232, :253-276): refresh the occupancy
 * grid from the density network -- n_uniform samples in random cells + n_nonuniform samples in cells above
 * NERF_MIN_OPTICAL_THICKNESS, density MLP, max-splat, decayed maximum into the grid -- n_iterations times, then the
 * mean / bitfield / max-pool of K8/K9. The generator (pcg32, seeded like reset_network does, testbed.cu:3848-3861) and
 * the EMA step counter live in the context. n_uniform = n_nonuniform = 0 selects training_prep_nerf's schedule
 * (:3432-3446): 128^3 x cascades uniform samples for the first 256 steps, then a quarter of that of each kind. */
NGP_API int ngp_update_density_grid(ngp_ctx* ctx, float decay, uint32_t n_uniform, uint32_t n_nonuniform, uint32_t n_iterations);
/* the float density grid in use: (max_cascade + 1) x 128^3 values, Morton order per cascade */
NGP_API int ngp_get_density_grid(ngp_ctx* ctx, float* out, uint64_t n);
/* K1+K2 init_rays_with_payload_kernel_nerf + advance_pos_nerf_kernel (src/testbed_nerf.cu:1428-1544,333-381):
 * out: W*H NerfPayload records of 40 bytes (nerf_device.cuh:144-152) */
NGP_API int ngp_init_rays(ngp_ctx* ctx, const ngp_camera* cam, void* payloads_out);


/* --- geometry mode: Testbed::load_scene / load_mesh (src/testbed_geometry_training.cu:3101-3210, 2786-2866)
 * load_scene: {"geometry":[{"center":[x,y,z],"path":..., "type":"Mesh"|"Nerf"}]}; Mesh -> load_mesh (.obj/.stl),
 * Nerf -> load_snapshot. Relative paths resolve against the json's directory. */
NGP_API int ngp_load_scene(ngp_ctx* ctx, const char* json_path);
/* load_mesh on triangles already in memory: vertices = n_tris*9 floats in file space; normalised into the unit cube
 * around `center`, BVH4 (8 triangles per leaf) built on the host, uploaded. Rebuilds the scene AABB. */
NGP_API int ngp_add_mesh(ngp_ctx* ctx, const float* vertices, uint32_t n_tris, const float* center3);
NGP_API int ngp_load_mesh_file(ngp_ctx* ctx, const char* path, const float* center3);
NGP_API int ngp_clear_meshes(ngp_ctx* ctx);
NGP_API int ngp_n_meshes(const ngp_ctx* ctx);
/* per mesh: triangle / node counts and the mesh AABB; mesh = -1: the scene AABB (root inflated by 4) */
NGP_API int ngp_get_mesh_info(const ngp_ctx* ctx, int mesh, uint32_t* n_tris, uint32_t* n_nodes, float* aabb6);
/* the built BVH: nodes (n_nodes x 32 B: bb.min, bb.max, left_idx, right_idx) and reordered triangles (n_tris x 36 B) */
NGP_API int ngp_get_mesh_bvh(const ngp_ctx* ctx, int mesh, void* nodes_out, void* triangles_out);
NGP_API int ngp_set_geometry_opts(ngp_ctx* ctx, const ngp_geometry_opts* opts);
/* M2 mesh_raytrace_kernel (src/geometry_bvh.cu:646-676): host positions / directions n x 3, updated in place */
NGP_API int ngp_trace_mesh_rays(ngp_ctx* ctx, uint32_t n, float* positions, float* directions);


/* --- irradiance probes: Testbed::computeEnvmap / computeEnvmapMultiple (declared testbed.h:709-743, no body in the
 * reference; ray generators src/testbed_nerf.cu:1559-1773, tracer trace_mesh :2146-2262). Traces n_theta x n_phi
 * (x n_origin^2) rays through the NeRF and stores the lat-long RGBA texture (texel idx = i_theta + n_theta*j_phi,
 * mean over a texel's rays) in the context, plus E(n) tabulated at the texel directions. */
enum ngp_probe_mode { NGP_PROBE_CENTER = 0, NGP_PROBE_CENTER_OUTWARD = 1, NGP_PROBE_MULTI_CENTER = 2 };
typedef struct ngp_probe_desc {
	int32_t mode; /* ngp_probe_mode */
	uint32_t n_theta, n_phi, n_origin;
	float origin[3];         /* NGP_PROBE_CENTER_OUTWARD: shell position */
	float min_transmittance;
} ngp_probe_desc;
NGP_API int ngp_compute_envmap(ngp_ctx* ctx, const ngp_probe_desc* desc, float* rgba_out /* nullable: n_theta*n_phi*4 */);
/* the probe texture(s) and E(n) tabulated at the texel directions; after ngp_compute_envmap_grid: grid_x*grid_y textures back to back */
NGP_API int ngp_get_envmap(ngp_ctx* ctx, uint32_t* n_theta, uint32_t* n_phi, float* rgba_out, float* irradiance_rgba_out);
/* E(n) = sum_texels L(w) max(0, n.w) dOmega, dOmega = 4 pi/(n_theta n_phi), w = the direction the texel's ray travelled (for an
 * outward probe: -frame(normalize(origin)) * texel direction): host normals n x 3 -> host rgb n x 3. One probe only. */
NGP_API int ngp_irradiance(ngp_ctx* ctx, uint32_t n, const float* normals, float* rgb_out);

/* Testbed::computeEnvmapGrid (declared testbed.h:743, called src/main.cu:187-188 when m_render_mode == ShadeGridEnvMap -- the
 * fork's default, testbed.h:880 -- no body in the reference) with gridSize / m_envmap_tex (testbed.h:949-950). Definition used
 * here: grid_x x grid_y shell positions c + shell_radius * cylindrical_to_dir_nerf(
 * (i + .5) / grid_x, (j + .5) / grid_y) around c = render_aabb.center(); at each the K11 fan
 * (init_rays_from_center_outward_with_payload_kernel_nerf, src/testbed_nerf.cu:1611-1673) of n_theta x n_phi rays, all probes
 * traced in ONE launch; E_g(n) tabulated per probe. Mesh shading (NGP_RENDER_SHADE_GRID_ENVMAP) and ngp_irradiance_at blend
 * the four probes around the direction of (surface point - c), each read bilinearly at the normal in the manner of read_envmap
 * (envmap.cuh:24-50). */
typedef struct ngp_probe_grid_desc {
	uint32_t grid_x, grid_y; /* gridSize */
	uint32_t n_theta, n_phi; /* texels per probe */
	float shell_radius;
	float min_transmittance;
} ngp_probe_grid_desc;
NGP_API int ngp_compute_envmap_grid(ngp_ctx* ctx, const ngp_probe_grid_desc* desc, float* rgba_out /* nullable: grid_x*grid_y*n_theta*n_phi*4 */);
/* the grid as computed and its shell positions (nullable: grid_x*grid_y*3) */
NGP_API int ngp_get_envmap_grid(ngp_ctx* ctx, ngp_probe_grid_desc* desc_out, float* origins_out);
/* the lookup mesh shading does, at explicit surface points: positions / normals n x 3 -> rgb n x 3 (one probe: position unused) */
NGP_API int ngp_irradiance_at(ngp_ctx* ctx, uint32_t n, const float* positions, const float* normals, float* rgb_out);


/* --- training: Testbed::reset_network (src/testbed.cu:3820-4210), Testbed::train (:4364-4470),
 * Testbed::train_nerf / train_nerf_step (src/testbed_nerf.cu:2949-3431), training_prep_nerf (:3432-3446). The default
 * path of configs/nerf/base.json: no envmap, no camera / exposure / latent optimisation, no error-map sampling,
 * no depth supervision. */
enum ngp_loss_type { NGP_LOSS_L2 = 0, NGP_LOSS_L1, NGP_LOSS_MAPE, NGP_LOSS_SMAPE, NGP_LOSS_HUBER, NGP_LOSS_LOGL1, NGP_LOSS_RELATIVE_L2 }; /* ELossType, common.h:84-92 */
enum ngp_image_type { NGP_IMAGE_NONE = 0, NGP_IMAGE_BYTE = 1, NGP_IMAGE_HALF = 2, NGP_IMAGE_FLOAT = 3 };                                    /* EImageDataType */
typedef struct ngp_training_opts {
	uint32_t struct_size;
	int32_t loss_type;              /* m_nerf.training.loss_type; configs/nerf/base.json: Huber */
	int32_t random_bg_color;        /* nerf.h defaults: true */
	int32_t linear_colors;          /* false */
	int32_t snap_to_pixel_centers;  /* true */
	float near_distance;            /* 0.1 */
	float density_grid_decay;       /* 0.95 */
	int32_t train_network, train_encoding; /* m_train_network / m_train_encoding -> optimize_matrix_params / optimize_non_matrix_params */
	/* configs/nerf/base.json "optimizer": Ema{decay} > ExponentialDecay{decay_start, decay_interval, decay_base} > Adam */
	float learning_rate, beta1, beta2, epsilon, l2_reg;
	float ema_decay;                /* 0: no Ema, inference uses the training parameters */
	uint32_t decay_start, decay_interval;
	float decay_base;
	float background_color[3];      /* m_background_color.rgb() when !random_bg_color */
	int32_t color_space;            /* m_color_space: 0 Linear, 1 SRGB */
} ngp_training_opts;
typedef struct ngp_training_state {
	uint32_t struct_size;
	uint32_t training_step;                          /* m_training_step */
	uint32_t rays_per_batch;                         /* counters_rgb.rays_per_batch */
	uint32_t measured_batch_size;                    /* samples after compaction in the last step */
	uint32_t measured_batch_size_before_compaction;  /* samples marched in the last step */
	uint32_t n_rays_total;
	float loss;                                      /* m_loss_scalar.val(): refreshed every 16th step like Testbed::train */
	float learning_rate;                             /* Adam's rate after ExponentialDecay */
	uint64_t n_params, n_matrix_params;
} ngp_training_state;
/* Testbed::reset_network for configs/nerf/base.json: fresh random parameters (xavier-uniform matrices, grid in
 * +-1e-4, pcg32 seeded with `seed`; the reference's m_seed is 1337), an empty occupancy grid, training counters at 0.
 * The boxes and aabb_scale come from the loaded dataset (ngp_load_training_data) or, without one, aabb_scale 1. */
NGP_API int ngp_reset_network(ngp_ctx* ctx, uint32_t log2_hashmap_size, uint64_t seed);
NGP_API void ngp_default_training_opts(ngp_training_opts* opts);
NGP_API int ngp_set_training_opts(ngp_ctx* ctx, const ngp_training_opts* opts);
NGP_API int ngp_get_training_opts(const ngp_ctx* ctx, ngp_training_opts* opts);
/* NerfDataset::set_training_image (src/nerf_loader.cu:745-): pixels of training view `view`, RGBA, NGP_IMAGE_BYTE
 * (sRGB, straight alpha) or NGP_IMAGE_FLOAT (linear, premultiplied -- python_api.cu nerf.training.set_image);
 * width/height replace the view's resolution */
NGP_API int ngp_set_training_image(ngp_ctx* ctx, int view, int32_t width, int32_t height, const void* rgba, int32_t image_type);
/* Decode the dataset's image files into training images (the reference does this inside load_nerf with stb_image,
 * src/nerf_loader.cu:520-640): PNG and baseline JPEG files -- views whose file is missing or of another format (EXR,
 * progressive JPEG ...) stay without pixels and take no part in training. Fails when no view could be loaded. */
NGP_API int ngp_load_training_images(ngp_ctx* ctx, int32_t* n_loaded_out);
/* The decoder behind it, on its own (needs no device): PNG (8/16-bit, non-interlaced) and baseline JPEG -> RGBA8.
 * rgba_out may be NULL to query the size; error_out (nullable) receives the reason on failure. */
NGP_API int ngp_decode_image(const void* bytes, size_t n_bytes, int32_t* width, int32_t* height, uint8_t* rgba_out, size_t rgba_capacity, char* error_out, size_t error_capacity);
/* m_render_ground_truth (python_api.cu:490-491; CudaRenderBuffer::overlay_image at alpha 1, src/render_buffer.cu:344-414, called
 * from Testbed::render_frame_epilogue, src/testbed.cu:4979-4994): the training image of `view`, resampled (nearest) to
 * width x height around the screen centre, over background_rgba (sRGB values), times 2^exposure, in linear or sRGB output --
 * what scripts/run.py --test_transforms compares the render with (run.py:236-241). color_space: m_color_space (0 Linear,
 * 1 SRGB); fov_axis / zoom: m_fov_axis / m_zoom. Host float RGBA out. */
NGP_API int ngp_render_ground_truth(ngp_ctx* ctx, int view, int32_t width, int32_t height, const float* background_rgba, float exposure, int32_t color_space, int32_t to_srgb,
                                    int32_t fov_axis, float zoom, float* rgba_out);
/* n_steps x Testbed::train(batch_size): occupancy-grid refresh on training_prep_nerf's schedule, one train_nerf_step,
 * the optimizer, the counters; loss_out (nullable) receives the running loss. batch_size: a multiple of 128, 2^18 in the
 * reference's GUI and scripts/run.py */
NGP_API int ngp_train(ngp_ctx* ctx, uint32_t n_steps, uint32_t batch_size, float* loss_out);
NGP_API int ngp_get_training_state(const ngp_ctx* ctx, ngp_training_state* out);
/* The pieces of one step, for parity tests against the oracle. ngp_train_prepare_batch: sample generation + network +
 * loss for the current step (no parameter changes); the buffers (host pointers, nullable) receive ray_indices[n_rays],
 * numsteps[n_rays][2] (after compaction), coords_compacted[target][7], dloss fp16 [target][4], loss[n_rays];
 * counters3 = {samples marched, rays kept, samples after compaction}. ngp_train_gradients: the fused backward on that
 * batch, gradient of every parameter in snapshot order (loss-scaled by 128 like the reference). ngp_train_apply: the
 * optimizer step on the gradient currently held + the step bookkeeping. */
NGP_API int ngp_train_prepare_batch(ngp_ctx* ctx, uint32_t batch_size, uint32_t* counters3, uint32_t* ray_indices, uint32_t* numsteps, float* coords_compacted,
                                    uint16_t* dloss_fp16, float* loss);
NGP_API int ngp_train_gradients(ngp_ctx* ctx, uint32_t batch_size, float* grad_out /* n_params */);
NGP_API int ngp_train_apply(ngp_ctx* ctx);
/* current training parameters (fp32 master copy) and their Ema (fp16 -> fp32), n_params each, nullable */
NGP_API int ngp_get_training_params(ngp_ctx* ctx, float* params_out, float* ema_out);

#ifdef __cplusplus
}
#endif
#endif
