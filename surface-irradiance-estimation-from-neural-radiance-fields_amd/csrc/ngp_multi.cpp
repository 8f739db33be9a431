// Several GPUs behind ONE context: what Testbed's device list does in the reference (src/testbed.cu:5490-5616 -- a model
// replica per device, sync_device copies parameters and the occupancy bitfield to a device that lags, every auxiliary device
// renders on its own stream / worker, peer copies bring frame and depth buffers to the primary, :5604-5605) -- with one
// difference of substance: the reference hands whole VIEWS to devices (VR eyes, :2487), which does nothing for a single
// camera; here the camera's 8x8-pixel tiles are dealt round-robin to the devices (the sharding bench.py uses across
// processes), each device renders its share straight into the tile-packed layout, pushes it to device 0 over xGMI with one
// hipMemcpyPeerAsync, and device 0 scatters the tiles into the image. The renderer has no host synchronisation inside a
// frame, so one host thread enqueues the work of every device; nothing waits until the caller does.
#include "ngp_host.h"

#include <cstring>

namespace ngp {

void launch_unpack_tiles(const float4* gathered_rgba, const float* gathered_depth, uint32_t n_devices, uint32_t n_slots, int width, int height, float4* rgba, float* depth,
                         hipStream_t stream);

namespace {
struct DeviceGuard {
	int prev = 0;
	explicit DeviceGuard(int dev) {
		(void)hipGetDevice(&prev);
		NGP_HIP_CHECK(hipSetDevice(dev));
	}
	~DeviceGuard() { (void)hipSetDevice(prev); }
};

// sync_device (src/testbed.cu:5523-5563): a device whose replica lags the primary's gets what changed, device to device
// where the shapes stand (parameters after training steps, the occupancy grid after a refresh: the reference's
// cudaMemcpyPeerAsync of params + bitfield, :5542-5555), through the host descriptor when the model itself was replaced
void sync_peer_model(ngp_ctx* primary, ngp_ctx* peer) {
	sync_inference_model(primary); // what was trained is what gets rendered (bumps params_generation when it had work to do)
	if (peer->synced_generation != primary->model_generation || !peer->model_loaded) {
		sync_host_params(primary);
		refresh_density_grid_host(primary);
		ngp_model_desc d = primary->desc;
		d.params_fp16 = primary->params.data();
		d.n_params = primary->params.size();
		d.density_grid_fp16 = primary->density_grid.data();
		d.n_density_grid = primary->density_grid.size();
		{
			DeviceGuard g(peer->device);
			install_model(peer, d);
		}
		peer->synced_generation = primary->model_generation;
		// (install_model derives the occupancy bits from the fp16 grid of the descriptor; if the primary has refreshed its grid since
		// it was loaded, its fp32 grid is the truth -- copied below)
		peer->synced_grid_generation = primary->grid_generation == 0 ? 0 : ~0ull;
		peer->synced_params_generation = primary->params_generation == 0 ? 0 : ~0ull;
	}
	const bool grid = peer->synced_grid_generation != primary->grid_generation, params = peer->synced_params_generation != primary->params_generation;
	if (!grid && !params) return;
	// Device to device, ordered by events (the reference's sync_device uses cudaMemcpyPeerAsync on the device's stream, :5542-5555):
	//   the peer's stream waits for the primary's update (ev_model) -- its own earlier frames are on that stream, hence before the copies;
	//   the copies run on the peer's stream; the primary's update stream then waits for them (ev_synced), so the next training step /
	//   refresh cannot overwrite a source that is still being read. No host wait, no device-wide wait on either side.
	{
		DeviceGuard g(primary->device);
		if (!primary->ev_model_valid) mark_model_updated(primary, primary->stream); // (a model that was never updated on the device: everything before now)
	}
	DeviceGuard g(peer->device);
	hipStream_t s = peer->stream;
	NGP_HIP_CHECK(hipStreamWaitEvent(s, primary->ev_model, 0));
	order_after_frames(peer, s); // (frames a caller put on another stream of the peer, if any)
	if (grid) {
		const size_t n_cells = (size_t)NERF_GRID_N_CELLS * (primary->max_cascade + 1);
		NGP_HIP_CHECK(hipMemcpyPeerAsync(peer->d_bitfield, peer->device, primary->d_bitfield, primary->device, (size_t)NERF_GRID_N_CELLS / 8 * NERF_CASCADES, s));
		NGP_HIP_CHECK(hipMemcpyPeerAsync(peer->d_coarse, peer->device, primary->d_coarse, primary->device, ((size_t)NERF_CASCADES * COARSE_WORDS_PER_MIP + NERF_CASCADES * 16) * sizeof(uint32_t), s));
		NGP_HIP_CHECK(hipMemcpyPeerAsync(peer->d_density_f32, peer->device, primary->d_density_f32, primary->device, n_cells * sizeof(float), s));
		peer->bitfield_mean = primary->bitfield_mean;
		peer->grid_rng_state = primary->grid_rng_state;
		peer->grid_rng_inc = primary->grid_rng_inc;
		peer->grid_ema_step = primary->grid_ema_step;
		peer->grid_updates = primary->grid_updates;
		peer->synced_grid_generation = primary->grid_generation;
	}
	if (params) {
		NGP_HIP_CHECK(hipMemcpyPeerAsync(peer->d_params, peer->device, primary->d_params, primary->device, primary->M.grid_bytes, s));
		NGP_HIP_CHECK(hipMemcpyPeerAsync(peer->d_xgrid, peer->device, primary->d_xgrid, primary->device, primary->M.xgrid_bytes, s));
		NGP_HIP_CHECK(hipMemcpyPeerAsync(peer->d_wfrags, peer->device, primary->d_wfrags, primary->device, (size_t)(N_FRAGS_MAX + N_NORMALS_FRAGS) * 64 * sizeof(uint4), s));
		peer->synced_params_generation = primary->params_generation;
	}
	mark_model_updated(peer, s); // the peer's frames read the new tables
	if (!peer->ev_synced) NGP_HIP_CHECK(hipEventCreateWithFlags(&peer->ev_synced, hipEventDisableTiming));
	NGP_HIP_CHECK(hipEventRecord(peer->ev_synced, s));
	{
		DeviceGuard gp(primary->device);
		NGP_HIP_CHECK(hipStreamWaitEvent(primary->stream, peer->ev_synced, 0)); // the primary's next update of these buffers comes after the copies
	}
}

void ensure_pack_buffers(ngp_ctx* ctx, size_t n_pixels_packed) {
	if (n_pixels_packed <= ctx->pack_alloc) return;
	if (ctx->d_pack_rgba) (void)hipFree(ctx->d_pack_rgba);
	if (ctx->d_pack_depth) (void)hipFree(ctx->d_pack_depth);
	ctx->pack_alloc = 0;
	NGP_HIP_CHECK(hipMalloc((void**)&ctx->d_pack_rgba, n_pixels_packed * sizeof(float4)));
	NGP_HIP_CHECK(hipMalloc((void**)&ctx->d_pack_depth, n_pixels_packed * sizeof(float)));
	ctx->pack_alloc = n_pixels_packed;
	if (!ctx->ev_pack) NGP_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_pack, hipEventDisableTiming));
}
} // namespace

void free_multi_buffers(ngp_ctx* ctx) {
	if (ctx->d_pack_rgba) (void)hipFree(ctx->d_pack_rgba);
	if (ctx->d_pack_depth) (void)hipFree(ctx->d_pack_depth);
	if (ctx->d_gather_rgba) (void)hipFree(ctx->d_gather_rgba);
	if (ctx->d_gather_depth) (void)hipFree(ctx->d_gather_depth);
	if (ctx->ev_pack) (void)hipEventDestroy(ctx->ev_pack);
	ctx->d_pack_rgba = ctx->d_gather_rgba = nullptr;
	ctx->d_pack_depth = ctx->d_gather_depth = nullptr;
	ctx->ev_pack = nullptr;
	ctx->pack_alloc = ctx->gather_alloc = 0;
}

// One frame over every device of a multi-device context; the assembled image lands in d_rgba / d_depth (device 0), enqueued
// on `stream` of device 0. Nothing here waits for the GPUs.
void render_frames_multi(ngp_ctx* ctx, const ngp_camera& cam, const ngp_render_opts& opts, float4* d_rgba, float* d_depth, hipStream_t stream) {
	if (opts.packed_output) throw std::runtime_error("packed_output addresses one shard: render it through a single-device context");
	const uint32_t n_dev = 1u + (uint32_t)ctx->peers.size();
	const uint32_t tiles = (uint32_t)((cam.width + 7) / 8) * (uint32_t)((cam.height + 7) / 8);
	const uint32_t n_slots = (tiles + n_dev - 1) / n_dev; // tiles per device, rounded up: the stride of a device's block at the primary
	const size_t packed = (size_t)n_slots * 64;
	{ // the primary's landing zone for everybody's tiles
		DeviceGuard g(ctx->device);
		if (packed * n_dev > ctx->gather_alloc) {
			if (ctx->d_gather_rgba) (void)hipFree(ctx->d_gather_rgba);
			if (ctx->d_gather_depth) (void)hipFree(ctx->d_gather_depth);
			ctx->gather_alloc = 0;
			NGP_HIP_CHECK(hipMalloc((void**)&ctx->d_gather_rgba, packed * n_dev * sizeof(float4)));
			NGP_HIP_CHECK(hipMalloc((void**)&ctx->d_gather_depth, packed * n_dev * sizeof(float)));
			ctx->gather_alloc = packed * n_dev;
		}
		if (!ctx->ev_pack) NGP_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_pack, hipEventDisableTiming));
		// the landing zone is rewritten by every frame: the previous frame's unpack (on `stream` or another stream of this device) must be done with it
		if (ctx->n_multi_frames > 0) NGP_HIP_CHECK(hipStreamWaitEvent(stream, ctx->ev_unpacked, 0));
		NGP_HIP_CHECK(hipEventRecord(ctx->ev_pack, stream)); // "the landing zone is free as of here"
	}
	for (uint32_t i = 0; i < n_dev; ++i) {
		ngp_ctx* dev = i == 0 ? ctx : ctx->peers[i - 1];
		if (i > 0) {
			if (ctx->model_loaded) sync_peer_model(ctx, dev); // (a Geometry session may hold meshes only)
			// by-value render state follows the primary every frame (m_render_aabb, cone angle: src/testbed.cu:5529-5563 copies them on sync)
			memcpy(dev->M.raabb_min, ctx->M.raabb_min, sizeof(ctx->M.raabb_min));
			memcpy(dev->M.raabb_max, ctx->M.raabb_max, sizeof(ctx->M.raabb_max));
			memcpy(dev->M.r2l, ctx->M.r2l, sizeof(ctx->M.r2l));
			dev->M.r2l_identity = ctx->M.r2l_identity;
			dev->M.cone_angle = ctx->M.cone_angle;
			memcpy(dev->tune, ctx->tune, sizeof(ctx->tune));
			if (opts.testbed_mode == NGP_MODE_GEOMETRY) sync_peer_geometry(ctx, dev); // meshes + BVHs, BRDF parameters, irradiance tables
		}
		DeviceGuard g(dev->device);
		ensure_pack_buffers(dev, packed);
		hipStream_t s = i == 0 ? stream : dev->stream;
		ngp_render_opts o = opts;
		o.shard_index = i;
		o.shard_count = n_dev;
		o.packed_output = 1;
		if (i > 0) NGP_HIP_CHECK(hipStreamWaitEvent(s, ctx->ev_pack, 0)); // do not push into a landing zone the previous frame still reads
		render_frames_on(dev, cam, o, dev->d_pack_rgba, dev->d_pack_depth, s);
		// push this device's tiles to the primary: one peer copy each for colour and depth (5.2 MB per GPU at 1080p / 8 GPUs), on the
		// rendering device's stream so that it follows the kernel without a host round trip
		NGP_HIP_CHECK(hipMemcpyPeerAsync(ctx->d_gather_rgba + packed * i, ctx->device, dev->d_pack_rgba, dev->device, packed * sizeof(float4), s));
		NGP_HIP_CHECK(hipMemcpyPeerAsync(ctx->d_gather_depth + packed * i, ctx->device, dev->d_pack_depth, dev->device, packed * sizeof(float), s));
		if (i > 0) NGP_HIP_CHECK(hipEventRecord(dev->ev_pack, s));
	}
	DeviceGuard g(ctx->device);
	for (ngp_ctx* p : ctx->peers) NGP_HIP_CHECK(hipStreamWaitEvent(stream, p->ev_pack, 0));
	if (!d_depth) {
		ensure_frame_buffers_for(ctx, (size_t)cam.width * cam.height);
		d_depth = ctx->d_depth;
	}
	launch_unpack_tiles(ctx->d_gather_rgba, ctx->d_gather_depth, n_dev, n_slots, cam.width, cam.height, d_rgba, d_depth, stream);
	if (!ctx->ev_unpacked) NGP_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_unpacked, hipEventDisableTiming));
	NGP_HIP_CHECK(hipEventRecord(ctx->ev_unpacked, stream));
	++ctx->n_multi_frames;
	ctx->last_was_multi = true;
	NGP_HIP_CHECK(hipGetLastError());
}

} // namespace ngp

extern "C" {

// Testbed's device list (src/testbed.cu:5490-5521, m_devices): devices[0] is the primary, the context the caller talks to
ngp_ctx* ngp_create_multi(const int* devices, int n_devices) {
	if (!devices || n_devices < 1 || n_devices > 64) return nullptr;
	ngp_ctx* primary = ngp_create(devices[0]);
	if (!primary) return nullptr;
	for (int i = 1; i < n_devices; ++i) {
		ngp_ctx* p = ngp_create(devices[i]);
		if (!p) {
			ngp_destroy(primary);
			return nullptr;
		}
		p->primary = primary;
		primary->peers.push_back(p);
		if (devices[i] != devices[0]) { // xGMI peer access both ways (the same device twice is a rehearsal on one GPU)
			int can = 0;
			(void)hipDeviceCanAccessPeer(&can, devices[0], devices[i]);
			if (can) {
				(void)hipSetDevice(devices[0]);
				(void)hipDeviceEnablePeerAccess(devices[i], 0);
				(void)hipSetDevice(devices[i]);
				(void)hipDeviceEnablePeerAccess(devices[0], 0);
				(void)hipGetLastError(); // "already enabled" is fine
			}
		}
	}
	(void)hipSetDevice(devices[0]);
	return primary;
}

int ngp_n_devices(const ngp_ctx* ctx) { return ctx ? 1 + (int)ctx->peers.size() : 0; }

// counters of the last frame per device (device 0 first): the frame's totals are the sums, its kernel time the maximum
int ngp_get_device_render_stats(ngp_ctx* ctx, int device_index, ngp_render_stats* out) {
	return ngp::guarded(ctx, [&] {
		if (!out || device_index < 0 || device_index > (int)ctx->peers.size()) throw std::runtime_error("invalid argument");
		ngp_ctx* dev = device_index == 0 ? ctx : ctx->peers[device_index - 1];
		ngp::DeviceGuard g(dev->device);
		const bool multi = dev->last_was_multi;
		dev->last_was_multi = false; // this device's own share, not the frame's totals
		const int rc = ngp_get_render_stats(dev, out);
		dev->last_was_multi = multi;
		if (rc != 0) throw std::runtime_error(dev->error);
	});
}

} // extern "C"
