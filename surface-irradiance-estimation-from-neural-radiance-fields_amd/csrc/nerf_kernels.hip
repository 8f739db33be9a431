// Kernels of the NeRF inference hot path for MI355X (gfx950).
//
// render_nerf_fused: ONE persistent launch per sample-per-pixel replaces the reference's per-iteration chain
//   compact_kernel_nerf -> (D2H counter + stream sync) -> generate_next_nerf_network_inputs -> 4 tcnn launches +
//   extract_density -> composite_kernel_nerf        (reference src/testbed_nerf.cu:2056-2138)
// and the ray setup / shading kernels around it (:1878-1921, :2463). Rays never leave registers: no NerfPayload,
// NerfCoordinate or network-output round trip through HBM; the only global traffic is hash-table gathers,
// occupancy bits and one frame-buffer write per pixel.
//
// Work distribution: 8x8-pixel camera tiles are dealt from a global atomic queue (one atomic per 64 rays). Each
// wave64 owns 64 ray slots; finished slots are refilled from the wave's current tile using a wave ballot +
// prefix count (mbcnt) -- the wave64 counterpart of the reference's global-atomic compaction, with no host sync.
// Every iteration the live slots are compacted (ds_permute) onto 16-sample MFMA passes: 4 lanes cooperate on one
// sample's hash-grid levels (2 levels each) and the MLPs run with samples on the MFMA N axis (nerf_device.h).
#include "render_common.h"
#include "pcg32.h"

namespace ngp {

constexpr int BLOCK = 256;

// UNIT: unit-cube scenes (aabb_scale 1 => one cascade, cone angle 0 => fixed step sqrt(3)/1024; load_nerf_post,
// src/testbed_nerf.cu:2729-2736). The instantiation folds away the cascade climb, the mip arithmetic and both
// exponential-stepping branches; the arithmetic that remains is the same expression for expression.
// OUTSIDE: the render box may reach beyond the occupancy grid (geometry mode, a hand-set render box)
// PLAIN: static pinhole camera, no depth of field, no environment map (FrameParams::plain, decided by the host)
// NORMALS: ERenderMode::Normals -- every sample's colour is the unit vector opposite to the density's input gradient (one backward
// pass through the density head and the encoding per sample, density_gradient_pass); an instantiation of its own, so that no other
// kernel carries its registers
template <bool PROBE, int PROF = 0, bool UNIT = false, int MIPS = (UNIT ? 1 : (int)NERF_CASCADES), bool OUTSIDE = true, int RGB_MID = 1, bool PLAIN = false, bool NORMALS = false, int FB = BLOCK>
NGP_DEV void fused_body(const ModelParams& M, const CameraParams& C, const FrameParams& F, const ProbeParams& P) {
	const uint32_t max_cascade = UNIT ? 0u : M.max_cascade;
	const float cone_angle = UNIT ? 0.0f : M.cone_angle;
	const Stepping stepping = make_stepping(cone_angle); // (wave-uniform: the exponential stepping's constants, formed once)
	__shared__ uint4 s_w[n_frags_for(RGB_MID) * 64];
	__shared__ LevelInfo s_lv[N_LEVELS];
	__shared__ uint32_t s_coarse[MIPS * COARSE_WORDS_PER_MIP]; // 4 KB per cascade: empty-space summary of the occupancy grid, cascades 0 .. MIPS - 1 (outer ones: M.coarse)
	__shared__ uint32_t s_coarse16[(UNIT ? 1 : (int)NERF_CASCADES) * 16];
	__shared__ uint2 s_sh[FB * 4]; // per ray slot: 16 fp16 SH coefficients of its direction, written once per ray
	constexpr int SLOTS = 64; // a wave's sample list: 4 network passes of 16
	__shared__ float4 s_samp[FB / 64 * SLOTS]; // samples that wait for the network, in emission order: warped position, warped dt
	__shared__ uint2 s_res[FB / 64 * SLOTS];   // .x = the lane that owns the sample; after the pass: the network's 4 fp16 outputs (rgb, density)
	__shared__ float4 s_nrm[NORMALS ? FB / 64 * SLOTS : 1]; // Normals: d logit / d warped position and the logit
	// Ray sharing inside a workgroup (knob 7): a wave that has run out of work asks through s_xstate, a busy wave hands it every second
	// one of its live rays (16 at most at a time) through s_xray (16 words per ray; the SH coefficients go straight into the receiver's s_sh rows), so the last
	// tiles of a frame -- or a small frame's heavy tiles -- are finished by four waves instead of one.
	//   s_xstate: 0 free | 0x100 + w: wave w asks | 0x200 + w: a donor is writing for w | 0x300 + w: s_xcount rays are ready for w
	//   s_active: waves of this workgroup that hold rays or may still be dealt some; an asking wave leaves when it reaches 0
	__shared__ uint32_t s_xstate, s_xcount, s_active;
	constexpr uint32_t XRAYS = 16; // rays per hand-over
	__shared__ float s_xray[XRAYS * 16];
	if (threadIdx.x == 0) {
		s_xstate = 0u;
		s_xcount = 0u;
		s_active = FB / 64;
	}
	unsigned long long t_entry = 0, rt_entry = 0; // diagnostic build: the wave's arrival, before the workgroup stages weights and occupancy summaries
	if (PROF) { t_entry = stamp(); rt_entry = realtime(); }
	for (int i = threadIdx.x; i < n_frags_for(RGB_MID) * 64; i += FB) s_w[i] = M.wfrags[i];
	if (threadIdx.x < N_LEVELS) s_lv[threadIdx.x] = M.levels[threadIdx.x];
	for (uint32_t i = threadIdx.x; i < (max_cascade + 1 < (uint32_t)MIPS ? max_cascade + 1 : (uint32_t)MIPS) * COARSE_WORDS_PER_MIP; i += FB) s_coarse[i] = M.coarse[i];
	if (threadIdx.x < (UNIT ? 1 : (int)NERF_CASCADES) * 16) s_coarse16[threadIdx.x] = M.coarse[NERF_CASCADES * COARSE_WORDS_PER_MIP + threadIdx.x];
	__syncthreads();

	const int lane = threadIdx.x & 63;
	const int c = lane & 15;
	if (lane == 0 && (threadIdx.x >> 6) == 0) atomicMax(&F.results[4], ~realtime()); // start stamp (one per workgroup): max of the complements = the earliest
	const GridRsrc t_grid = make_grid_rsrc(M.grid, M.grid_bytes), t_xgrid = make_grid_rsrc(M.xgrid, M.xgrid_bytes);
	const float* cam_last = (!PLAIN && C.moving) ? C.m1 : C.m; // depth is measured along camera_matrix1 (src/testbed_nerf.cu:2412)
	const f3 cam_fwd = mk3(cam_last[6], cam_last[7], cam_last[8]);
	// direct output: the background's trip through the tonemap is the same for every pixel
	// (the background colour is sRGB: linearised unless the frame is averaged in sRGB, src/render_buffer.cu:537-541)
	const f3 bg_linear = (PROBE || !F.direct) ? mk3(0.f, 0.f, 0.f)
	                     : F.color_space == 1 ? mk3(F.background[0], F.background[1], F.background[2])
	                                          : mk3(srgb_to_linear(F.background[0]), srgb_to_linear(F.background[1]), srgb_to_linear(F.background[2]));
	const float4 empty_pixel = (!PROBE && F.direct) ? tonemap_pixel(F, bg_linear, 0.f, 0.f, 0.f, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
	const f3 cam_pos = mk3(cam_last[9], cam_last[10], cam_last[11]);
	const f3 amin = mk3(M.aabb_min[0], M.aabb_min[1], M.aabb_min[2]);
	const f3 adiag = mk3(M.aabb_diag[0], M.aabb_diag[1], M.aabb_diag[2]);

	// per-lane ray slot. A slot is free (!alive) or carries a ray that marches through empty space and emits samples into the wave's list.
	RayState ray;
	ray.alive = false;
	ray.o = ray.d = mk3(0.f, 0.f, 0.f);
	ray.t = 0.f;
	ray.idx = 0;
	ray.out = 0;
	f3 idir = mk3(0.f, 0.f, 0.f);
	Accum acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
	uint32_t step = 1;
	uint32_t skip_i = 1;
	bool counted = false;
	// A ray may have several samples in the wave's list at once (emitted this round, not yet through the network): n_pend of them,
	// their slots packed 8 bits each in emission order. left_box: the ray has run out of the render box behind its last waiting sample.
	uint32_t n_pend = 0, sl_lo = 0, sl_hi = 0;
	bool left_box = false;
	uint32_t n_slots = 0; // wave-uniform: samples in the list
	const int wave_base = threadIdx.x & ~63;
	OccBlock occ_cache;
	occ_cache.key = 0xffffffffu;
	occ_cache.bits = make_uint2(0u, 0u);
	bool finished = false; // the ray has ended and waits to be shaded (once per round, with every other finished ray)
	bool idle = false;     // wave-uniform: this wave has told the workgroup that it is out of work (s_active)
	const uint32_t my_wave = threadIdx.x >> 6;

	// wave-uniform tile reservoir
	bool exhausted = false;
	uint32_t qsel = (uint32_t)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u; // HW_REG_XCC_ID[3:0]: the XCD this workgroup runs on
	int stall = 0;
	uint32_t n_alive_init = 0, n_hit = 0, n_samples = 0; // wave-uniform (ballot counts)
	unsigned long long p_skip[3] = {0, 0, 0};
	unsigned long long pt[4] = {0, 0, 0, 0}, p_iters = 0, p_passes = 0, p_rounds = 0, p_lane_steps = 0, t0 = 0, t1 = 0;

	unsigned long long rt_start = 0;
	if (PROF && F.prof) {
		rt_start = realtime();
		if (lane == 0) atomicMax(&F.prof[8], ~rt_start); // = min over waves of the start time
	}
	// ---- wave timeline (diagnostic build, NGP_PROFILE_TRACE=stride): every stride-th wave that is dealt rays writes one 32-byte
	// record per loop round -- s_memtime at the top of the round and after each section, what the round carried -- and a header
	// (where it ran, when it started on the chip's 100 MHz clock). Layout: tools/wave_trace.py.
	int tr_slot = -1; // -1: no rays dealt yet, -2: not a traced wave
	uint32_t tr_it = 0, tr_info0 = 0, tr_info1 = 0, tr_march = 0;
	uint32_t tr_net[5] = {0, 0, 0, 0, 0}; // [address arithmetic + gather issue, gather wait, corner sums, MFMA chains, hand-back] of the round's passes
	unsigned long long tr_t[5] = {0, 0, 0, 0, 0};
	auto trace_emit = [&](int upto) { // sections after `upto` did not run this round: their stamps repeat the last one taken
		if (!PROF || tr_slot < 0) return;
		if (tr_it < F.trace_cap_iters && lane == 0) {
			uint32_t* r = F.trace + 16 + F.trace_cap_waves * 16u + ((size_t)tr_slot * F.trace_cap_iters + tr_it) * 16u;
			r[0] = (uint32_t)tr_t[0];
			for (int k = 1; k < 5; ++k) r[k] = (uint32_t)(tr_t[k <= upto ? k : upto] - tr_t[0]);
			r[5] = tr_info0;
			r[6] = tr_info1;
			r[7] = tr_march;
			for (int k = 0; k < 5; ++k) r[8 + k] = upto >= 3 ? tr_net[k] : 0u; // PROF 2: inside the network section
		}
		++tr_it;
	};
	for (;;) {
		if (PROF) { t0 = stamp(); tr_t[0] = t0; tr_info0 = tr_info1 = tr_march = 0; for (int k = 0; k < 5; ++k) tr_net[k] = 0; }
		// ---- refill free slots from the tile queue: K1 and the start-of-ray jitter of K2. The skip to the first
		// occupied voxel that K2 also does (advance_pos_nerf, :356) is the same loop as K4's and runs below with every
		// other marching lane -- a ray with nothing in front of it must not stall the 63 other slots of its wave.
		unsigned long long dead_mask = __ballot(!ray.alive);
		int n_dead = __popcll(dead_mask);
		// ---- retire: K7 for the rays that ended since the last refill, all at once (sRGB->linear is three powf and a
		// frame-buffer read-modify-write; run per round it would execute with one or two live lanes)
		if ((exhausted || n_dead >= F.tune[0]) && __any(finished)) {
			bool hit = false;
			if (finished) {
				hit = shade_ray<PROBE, PLAIN, NORMALS>(F, P, bg_linear, ray.out, acc, step - 1u, ray.d); // step counts from 1 like the reference's march loop
				finished = false;
			}
			n_hit += (uint32_t)__popcll(__ballot(hit));
			if (PROF) tr_info1 |= 4u << 24;
		}
		if (!exhausted && n_dead >= (F.tune[0] > 16 ? F.tune[0] : 16)) {
			// the queue deals 4x4-pixel strips (quarters of the 8x8 tiles): one atomic hands this wave n_dead / 16 of them
			const uint32_t want = (uint32_t)n_dead >> 4, n_strips = F.n_local_tiles * 4u;
			uint32_t first = 0, limit = n_strips;
			if (!PROBE && F.xqueue) {
				// one queue per XCD over an eighth of the share's tiles (a band of the image): the workgroups of an XCD march through
				// neighbouring tiles and share their hash-grid lines in that XCD's L2; a wave whose band is dealt out moves on to the next
				// XCD's (and stays there), so the bands need not be equally heavy
				for (int tries = 0; tries < 8; ++tries) {
					const uint32_t lo = (uint32_t)(((unsigned long long)n_strips * qsel) >> 3), hi = (uint32_t)(((unsigned long long)n_strips * (qsel + 1u)) >> 3);
					uint32_t off = 0;
					if (lane == 0) off = atomicAdd(F.xqueue + qsel, want);
					off = __builtin_amdgcn_readfirstlane(off);
					first = lo + off;
					limit = hi;
					if (off < hi - lo) break;
					first = limit = n_strips; // (nothing here)
					qsel = (qsel + 1u) & 7u;
				}
			} else {
				if (lane == 0) first = atomicAdd(F.queue, want);
				first = __builtin_amdgcn_readfirstlane(first);
			}
			if (first >= limit) {
				exhausted = true;
			} else {
				const uint32_t got = limit - first < want ? limit - first : want;
				if (PROF && F.trace && tr_slot == -1) {
					uint32_t n = 0;
					if (lane == 0) n = atomicAdd(F.trace, 1u);
					n = __builtin_amdgcn_readfirstlane(n);
					tr_slot = (n % F.trace_stride == 0u && n / F.trace_stride < F.trace_cap_waves) ? (int)(n / F.trace_stride) : -2;
					if (tr_slot >= 0 && lane == 0) {
						uint32_t* hd = F.trace + 16 + (size_t)tr_slot * 16u;
						hd[0] = (uint32_t)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)); // HW_REG_HW_ID: wave, SIMD, CU, SH, SE
						hd[1] = (uint32_t)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u; // XCD
						hd[2] = blockIdx.x;
						hd[3] = threadIdx.x >> 6;
						hd[5] = (uint32_t)rt_entry; hd[6] = (uint32_t)(rt_entry >> 32);
						hd[9] = (uint32_t)t_entry; hd[10] = (uint32_t)tr_t[0];
						hd[11] = n;
					}
				}
				if (PROF) tr_info1 |= 8u << 24;
				const uint32_t r = lanes_below(dead_mask);
				const uint32_t strip = first + (r >> 4);
				const bool take = !ray.alive && r < got * 16u;
				// a strip is a 4 x 4 quarter of the tile (16 neighbouring rays share more hash-grid lines than two rows of 8: +1 %);
				// slot = the pixel's index y * 8 + x in the tile
				const uint32_t tile_local = strip >> 2, s4 = strip & 3u, i16 = r & 15u;
				const uint32_t slot = ((s4 >> 1) * 4u + (i16 >> 2)) * 8u + (s4 & 1u) * 4u + (i16 & 3u);
				const uint32_t tile = F.shard_index + F.shard_count * tile_local;
				bool fresh = false;
				if (PROBE) {
					uint32_t q = tile * 64u + slot;
					if (take && q < P.n_rays) {
						init_probe_ray(P, q, ray);
						fresh = true;
					}
				} else if (take) {
					uint32_t x = (tile % F.tiles_x) * 8u + (slot & 7u);
					uint32_t y = (tile / F.tiles_x) * 8u + (slot >> 3);
					if (x < (uint32_t)C.width && y < (uint32_t)C.height) {
						init_ray<PLAIN>(M, C, x, y, ray);
						if (F.packed) ray.out = tile_local * 64u + slot;
						if (PLAIN || !F.envmap) {
							if (F.direct) { // CudaRenderBufferView::clear + the untouched pixel's trip through accumulate / tonemap
								F.frame_buffer[ray.out] = empty_pixel;
								F.depth_buffer[ray.out] = MAX_DEPTH;
							} else if (F.depth_buffer[ray.out] < 0.01f) { // src/testbed_nerf.cu:1490-1493
								F.depth_buffer[ray.out] = MAX_DEPTH;
							}
						} else { // frame_buffer[idx] = read_envmap(envmap, ray.d) for every valid ray (src/testbed_nerf.cu:1526-1528)
							const bool valid = ray.d.x != 0.0f || ray.d.y != 0.0f || ray.d.z != 0.0f;
							float env[4] = {0.f, 0.f, 0.f, 0.f};
							if (valid) {
								float d3[3] = {ray.d.x, ray.d.y, ray.d.z};
								read_envmap(F.envmap, F.env_w, F.env_h, d3, env);
							}
							if (F.direct) {
								F.frame_buffer[ray.out] = valid ? tonemap_pixel(F, bg_linear, env[0], env[1], env[2], env[3]) : empty_pixel;
								F.depth_buffer[ray.out] = MAX_DEPTH;
							} else {
								if (valid) F.frame_buffer[ray.out] = make_float4(env[0], env[1], env[2], env[3]);
								if (F.depth_buffer[ray.out] < 0.01f) F.depth_buffer[ray.out] = MAX_DEPTH;
							}
						}
						if (ray.alive) {
							ray.t = advance_n_steps(ray.t, stepping, ld_random_val_dim0(C.spp, ray.idx * 786433u)); // :355
							fresh = true;
						}
					}
				}
				if (fresh) {
					{ // K5c once per ray: the direction (and so its SH encoding) is constant along the ray
						union { half_t h[16]; uint2 u[4]; } sh;
						sh4_all((ray.d.x + 1.0f) * 0.5f, (ray.d.y + 1.0f) * 0.5f, (ray.d.z + 1.0f) * 0.5f, sh.h);
#pragma unroll
						for (int q = 0; q < 4; ++q) s_sh[threadIdx.x * 4 + q] = sh.u[q];
					}
					idir = mk3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
					acc = Accum{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
					step = 1;
					skip_i = 1;
					counted = PROBE; // probe rays count as alive from the start (there is no K2 for them)
				}
				if (PROBE) n_alive_init += (uint32_t)__popcll(__ballot(fresh));
			}
		}

		if (!PROBE && F.tune[7] && n_slots == 0u) {
			// ---- a sibling without work is asking: hand it every second live ray (none of them has a sample in the list at this point)
			const uint32_t st = __hip_atomic_load(&s_xstate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			const unsigned long long live_mask = __ballot(ray.alive);
			if ((st & 0xf00u) == 0x100u && __popcll(live_mask) >= 16) {
				const uint32_t to = st & 0xffu;
				uint32_t won = 0;
				if (lane == 0) won = atomicCAS(&s_xstate, st, 0x200u | to) == st ? 1u : 0u;
				won = __builtin_amdgcn_readfirstlane(won);
				if (won) {
					const uint32_t rank = lanes_below(live_mask);
					if (ray.alive && (rank & 1u) && (rank >> 1) < XRAYS) {
						const uint32_t k = rank >> 1;
						float* x = s_xray + k * 16u;
						x[0] = ray.o.x; x[1] = ray.o.y; x[2] = ray.o.z;
						x[3] = ray.d.x; x[4] = ray.d.y; x[5] = ray.d.z;
						x[6] = ray.t;
						x[7] = __uint_as_float(ray.out);
						x[8] = acc.r; x[9] = acc.g; x[10] = acc.b; x[11] = acc.a; x[12] = acc.depth; x[13] = acc.max_weight;
						x[14] = __uint_as_float(step | (counted ? 0x80000000u : 0u));
						x[15] = __uint_as_float(ray.idx);
#pragma unroll
						for (int q = 0; q < 4; ++q) s_sh[(to * 64u + k) * 4u + q] = s_sh[threadIdx.x * 4 + q];
						ray.alive = false; // (not finished: the ray lives on in the other wave)
					}
					if (lane == 0) {
						atomicAdd(&s_active, 1u); // the receiver counts as busy again before it can see its rays
						const uint32_t n_out = (uint32_t)__popcll(live_mask) >> 1;
						s_xcount = n_out < XRAYS ? n_out : XRAYS;
						__hip_atomic_store(&s_xstate, 0x300u | to, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
					}
				}
			}
		}
		if (PROF) { t1 = stamp(); pt[0] += t1 - t0; t0 = t1; tr_t[1] = t1; }
		// ---- K4 / K2: if_unoccupied_advance_to_next_occupied_voxel (nerf_device.cuh:461-494). A lane that reaches an occupied voxel
		// EMITS a sample -- warped position and step into the next free slot of the wave's sample list in LDS, in emission order -- and
		// marches on: up to k_max samples per ray and round (the reference's n_steps > 1 between two compactions,
		// src/testbed_nerf.cu:2080-2086), as long as the list (SLOTS = 4 network passes) has room. While the wave's 64 ray slots are
		// all live that is one sample per ray; as rays end, the survivors fill the passes, so a tile's tail needs a fraction of the
		// rounds and every pass, every march iteration and every composite runs with most lanes in use. Samples past the one that
		// terminates a ray are dropped by the compositor, exactly as the reference drops them.
		// (how many: k_busy while more than half of the wave's ray slots are live, else as many as fit the list, at most k_drain)
		const int n_live = __popcll(__ballot(ray.alive));
		const int k_fit = n_live <= 8 ? 8 : (n_live <= 10 ? 6 : (n_live <= 12 ? 5 : (n_live <= 16 ? 4 : (n_live <= 21 ? 3 : 2)))); // SLOTS / n_live without the division (wave-uniform compares)
		const int k_max = PROBE ? 1 : (n_live > SLOTS / 2 ? F.tune[4] : (k_fit < F.tune[5] ? k_fit : F.tune[5]));
		const int max_it = F.tune[1] > k_max ? F.tune[1] : k_max;
		bool blocked = false; // found a sample but the list is full: the lane stands still until the next round
		for (int k = 0; k < max_it; ++k) {
			const bool marching = ray.alive && !left_box && !blocked && (int)n_pend < k_max;
			if (!__any(marching) || n_slots >= (uint32_t)SLOTS) break;
			if (PROF) { const uint32_t nm = (uint32_t)__popcll(__ballot(marching)); ++p_rounds; p_lane_steps += (unsigned long long)nm; tr_march += 1u + (nm << 8); }
			// One iteration, written flat: every marching lane forms its position and looks its cell up (a lane that is still in the 4^3
			// block it read last needs one bit test); only the lanes that stand in an empty cell go through the voxel-exit arithmetic,
			// and the wave skips that part altogether while all of its rays are inside the object.
			const f3 pos = add3(ray.o, scale3(ray.d, ray.t));
			bool out = ray.t >= MAX_DEPTH || !raabb_contains(M, pos);
			if (PROBE && skip_i >= 200) out = true; // the 200-iteration variant of trace_mesh (:497-534)
			const bool inside = marching && !out;
			uint32_t mip = 0, empty = 1u;
			if (inside) {
				if (!UNIT) {
					mip = mip_from_pos(pos, NERF_CASCADES - 1);
					mip = mip > max_cascade ? max_cascade : mip;
				}
				empty = occupancy_state_at(pos, M.bitfield, s_coarse, s_coarse16, mip, occ_cache, MIPS, M.coarse);
			}
			const bool emit = inside && empty == 0u;
			const bool skip = inside && empty != 0u;
			float e_dt = 0.f;
			f3 e_w = mk3(0.f, 0.f, 0.f);
			if (emit) {
				e_dt = calc_dt(ray.t, stepping);
				e_w = sub3(pos, amin); // warp_position: (pos - min) / diag
				if (M.diag_pow2) e_w = mul3(e_w, mk3(M.aabb_inv_diag[0], M.aabb_inv_diag[1], M.aabb_inv_diag[2]));
				else e_w = div3(e_w, adiag);
			}
			bool ends = marching && out;
			if (skip) {
				// climb to the largest empty cascade cell around pos (nerf_device.cuh:488-490); each level doubles
				// the cell, so the block summary of the final level is looked up again
				if (!UNIT) {
					// (with block jumps the step is `empty` cells of cascade `mip`: a 4^3 block of this cascade reaches further than one cell of the
					// next -- keep the pair that spans most, not simply the coarsest empty cell)
					if (PROBE || !F.tune[6]) { // the reference's walk: the coarsest cascade whose cell around pos is empty, one cell at a time
						while (mip < max_cascade) {
							uint32_t e = occupancy_state_at(pos, M.bitfield, s_coarse, s_coarse16, mip + 1, occ_cache, MIPS, M.coarse);
							if (e == 0u) break;
							++mip;
							empty = e;
						}
					} else { // whole empty blocks of coarser cascades, from their summaries (no bitfield word, the block cache keeps this cascade's block)
						uint32_t best_mip = mip, best_empty = empty;
						for (uint32_t m = mip + 1u; m <= max_cascade; ++m) {
							const uint32_t e = empty_block_summary_at(pos, s_coarse, s_coarse16, m, MIPS, M.coarse);
							if (e == 0u) break;
							if ((e << m) >= (best_empty << best_mip)) { best_mip = m; best_empty = e; }
						}
						mip = best_mip;
						empty = best_empty;
					}
				}
				if (PROF) p_skip[empty == 16u ? 2 : (empty == 4u ? 1 : 0)] += 1ull;
				const float grid_half = 0.5f * (float)(1u << max_cascade);
				const bool outside = OUTSIDE && !PROBE && empty == 1u && mip == max_cascade &&
				                     fmaxf(fmaxf(__builtin_fabsf(pos.x - 0.5f), __builtin_fabsf(pos.y - 0.5f)), __builtin_fabsf(pos.z - 0.5f)) > grid_half;
				const float to_grid = outside ? grid_cube_entry(pos, idir, grid_half) : 0.0f;
				if (outside && to_grid < 0.0f) { // the ray never reaches the occupancy grid: it would leave the render box without a sample
					ends = true;
				} else if (outside && to_grid > 0.0f) {
					ray.t = advance_by_distance(ray.t, stepping, to_grid);
				} else {
					ray.t = advance_to_next_voxel(ray.t, stepping, pos, ray.d, idir, mip, (PROBE || !F.tune[6]) ? 1u : empty);
				}
				++skip_i;
			}
			if (ends) {
				if (n_pend == 0) {
					ray.alive = false;
					finished = true;
				} else {
					left_box = true; // its waiting samples are composited first
				}
			}
			// ---- K3: the wave64 counterpart of compact_kernel_nerf -- emitting lanes take consecutive slots (ballot + prefix count)
			const unsigned long long emit_mask = __ballot(emit);
			if (emit_mask) {
				const uint32_t slot = n_slots + lanes_below(emit_mask);
				bool newly_counted = false;
				if (emit) {
					newly_counted = !counted;
					counted = true;
					if (slot < (uint32_t)SLOTS) {
						s_samp[wave_base + slot] = make_float4(e_w.x, e_w.y, e_w.z, warp_dt(e_dt));
						s_res[wave_base + slot].x = (uint32_t)lane;
						if (n_pend < 4u) sl_lo |= slot << (8u * n_pend);
						else sl_hi |= slot << (8u * (n_pend - 4u));
						++n_pend;
						ray.t = ray.t + e_dt;
						skip_i = 1;
					} else {
						blocked = true;
					}
				}
				n_alive_init += (uint32_t)__popcll(__ballot(newly_counted));
				n_slots += (uint32_t)__popcll(emit_mask);
				n_slots = n_slots > (uint32_t)SLOTS ? (uint32_t)SLOTS : n_slots;
			}
		}
		const bool can_march = __any(ray.alive && !left_box && !blocked && (int)n_pend < k_max);
		if (PROF) {
			t1 = stamp(); pt[1] += t1 - t0; t0 = t1; ++p_iters; tr_t[2] = t1;
			tr_info0 = n_slots | ((uint32_t)stall << 24);
			tr_info1 |= (uint32_t)__popcll(__ballot(ray.alive)) | ((uint32_t)__popcll(__ballot(ray.alive && !left_box && n_pend == 0)) << 8) | ((uint32_t)k_max << 16) |
			            ((exhausted ? 1u : 0u) << 24);
		}
		if (n_slots == 0) {
			trace_emit(2);
			if (exhausted && !__any(ray.alive) && !__any(finished)) {
				if (PROBE || !F.tune[7]) break;
				// ---- out of work: ask a busy wave of this workgroup for half of its rays; leave when no wave of the workgroup has any
				if (!idle) {
					idle = true;
					if (lane == 0) atomicSub(&s_active, 1u);
				}
				bool got = false;
				for (;;) {
					uint32_t st = 0, act = 0;
					if (lane == 0) {
						st = __hip_atomic_load(&s_xstate, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
						act = __hip_atomic_load(&s_active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					}
					st = __builtin_amdgcn_readfirstlane(st);
					act = __builtin_amdgcn_readfirstlane(act);
					if (st == (0x300u | my_wave)) {
						got = true;
						break;
					}
					if (act == 0u) break; // (a donor raises s_active before it publishes, so rays on their way to this wave keep it above 0)
					if (st == 0u && lane == 0) atomicCAS(&s_xstate, 0u, 0x100u | my_wave);
					__builtin_amdgcn_s_sleep(32);
				}
				if (!got) break;
				const uint32_t n_in = s_xcount;
				if ((uint32_t)lane < n_in) {
					const float* x = s_xray + (uint32_t)lane * 16u;
					ray.o = mk3(x[0], x[1], x[2]);
					ray.d = mk3(x[3], x[4], x[5]);
					ray.t = x[6];
					ray.out = __float_as_uint(x[7]);
					acc = Accum{x[8], x[9], x[10], x[11], x[12], x[13]};
					const uint32_t sw = __float_as_uint(x[14]);
					step = sw & 0x7fffffffu;
					counted = (sw >> 31) != 0u;
					ray.idx = __float_as_uint(x[15]);
					idir = mk3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
					ray.alive = true;
					finished = false;
					left_box = false;
					skip_i = 1;
					n_pend = 0;
					sl_lo = sl_hi = 0;
					occ_cache.key = 0xffffffffu;
				}
				idle = false;
				if (lane == 0) __hip_atomic_store(&s_xstate, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); // (after the reads above: LDS serves a wave's requests in order)
			}
			continue;
		}
		// run the network once enough samples wait, or nothing else can make progress; never starve a waiting sample
		if ((int)n_slots < F.tune[2] && can_march && stall < F.tune[3]) {
			++stall;
			trace_emit(2);
			continue;
		}
		stall = 0;

		// ---- K5: network, 16 samples per pass straight from the list; two passes are run together whenever 17+ samples wait, so that
		// 32 gathers per lane and two independent MFMA chains are in flight (the loop is latency-, not issue-bound)
		const int n_pass = (int)((n_slots + 15u) >> 4);
		const int hq = lane >> 4;
		auto sample_of = [&](int p, float& sx, float& sy, float& sz, Sh4& shq) {
			const float4 a = s_samp[wave_base + 16 * p + c];
			const uint32_t owner = s_res[wave_base + 16 * p + c].x;
			sx = a.x; sy = a.y; sz = a.z;
			union { uint2 u; half_t h[4]; } cv; // the 4 SH coefficients this lane group feeds to the rgb head
			cv.u = s_sh[(wave_base + (int)owner) * 4 + hq];
#pragma unroll
			for (int j = 0; j < 4; ++j) shq.v[j] = cv.h[j];
		};
		auto deliver = [&](int p, const MlpOut& mo) { // results live in lanes 0..15 (h == 0): slot 16p + c
			union { half_t h[4]; uint2 u; } o;
			o.h[0] = mo.rgb[0]; o.h[1] = mo.rgb[1]; o.h[2] = mo.rgb[2]; o.h[3] = mo.sigma;
			if (lane < 16) s_res[wave_base + 16 * p + lane] = o.u;
		};
		int p = 0;
		if (NORMALS) {
			for (; p < n_pass; ++p) {
				float ax, ay, az;
				Sh4 sha;
				sample_of(p, ax, ay, az, sha);
				EncodeInFlight e;
				encode_issue(t_grid, t_xgrid, s_lv, hq, ax, ay, az, e);
				const half8 enc = encode_finish(e);
				DensityGrad dg = density_gradient_pass(s_w, M.wfrags, lane, e, enc, s_lv[hq].scale, s_lv[hq + 4].scale, M.density_linear != 0);
#pragma unroll
				for (int k = 0; k < 3; ++k) { // the four lanes of a sample hold two levels each
					dg.g[k] += __shfl_xor(dg.g[k], 16, 64);
					dg.g[k] += __shfl_xor(dg.g[k], 32, 64);
				}
				if (lane < 16) s_nrm[wave_base + 16 * p + lane] = make_float4(dg.g[0] * (1.0f / 128.0f), dg.g[1] * (1.0f / 128.0f), dg.g[2] * (1.0f / 128.0f), (float)dg.sigma);
			}
		}
		unsigned long long u0 = 0, u1 = 0;
		auto lap = [&](int k, bool drain_gathers) { // PROF 2 only: the stamps (and the explicit wait) serialise what the shipped kernel overlaps
			if (drain_gathers) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			u1 = stamp();
			tr_net[k] += (uint32_t)(u1 - u0);
			u0 = u1;
		};
		if (PROF >= 2) u0 = stamp();
		for (; p + 1 < n_pass; p += 2) {
			float ax, ay, az, bx, by, bz;
			Sh4 sha, shb;
			sample_of(p, ax, ay, az, sha);
			sample_of(p + 1, bx, by, bz, shb);
			EncodeInFlight ea, eb;
			encode_issue(t_grid, t_xgrid, s_lv, hq, ax, ay, az, ea);
			encode_issue(t_grid, t_xgrid, s_lv, hq, bx, by, bz, eb);
			if (PROF >= 2) { lap(0, false); lap(1, true); }
			half8 enca = encode_finish(ea);
			half8 encb = encode_finish(eb);
			if (PROF >= 2) lap(2, false);
			MlpOut moa = mlp_pass<RGB_MID>(s_w, lane, enca, sha);
			MlpOut mob = mlp_pass<RGB_MID>(s_w, lane, encb, shb);
			if (PROF >= 2) lap(3, false);
			deliver(p, moa);
			deliver(p + 1, mob);
			if (PROF >= 2) lap(4, false);
		}
		if (p < n_pass) {
			float ax, ay, az;
			Sh4 sha;
			sample_of(p, ax, ay, az, sha);
			EncodeInFlight e1;
			encode_issue(t_grid, t_xgrid, s_lv, hq, ax, ay, az, e1);
			if (PROF >= 2) { lap(0, false); lap(1, true); }
			half8 enc = encode_finish(e1);
			if (PROF >= 2) lap(2, false);
			MlpOut mo = mlp_pass<RGB_MID>(s_w, lane, enc, sha);
			if (PROF >= 2) lap(3, false);
			deliver(p, mo);
			if (PROF >= 2) lap(4, false);
		}

		if (PROF) { t1 = stamp(); pt[2] += t1 - t0; t0 = t1; p_passes += (unsigned long long)n_pass; tr_t[3] = t1; tr_info0 |= (n_slots << 8) | ((uint32_t)n_pass << 16); }
		// ---- K6: composite_kernel_nerf (:569-726): every lane takes its own samples in emission order and stops where its ray ends
		for (uint32_t k = 0;; ++k) {
			const bool have = k < n_pend && ray.alive;
			if (!__any(have)) break;
			if (have) {
				const uint32_t slot = ((k < 4u ? sl_lo >> (8u * k) : sl_hi >> (8u * (k - 4u))) & 0xffu);
				const float4 a = s_samp[wave_base + slot];
				union { uint2 u; half_t h[4]; } o;
				o.u = s_res[wave_base + slot];
				float gx = 0.f, gy = 0.f, gz = 0.f;
				if (NORMALS) {
					const float4 g = s_nrm[wave_base + slot];
					gx = g.x; gy = g.y; gz = g.z;
					o.h[3] = (half_t)g.w;
				}
				const f3 pos = add3(amin, mul3(mk3(a.x, a.y, a.z), adiag)); // unwarp_position
				const float sdepth = dot3(cam_fwd, sub3(pos, cam_pos));
				float T = 1.0f - acc.a;
				float dt = unwarp_dt(a.w);
				float alpha = 1.0f - fast_exp(-network_to_density((float)o.h[3], M.density_act) * dt);
				float weight = alpha * T;
				float cr = network_to_rgb((float)o.h[0], M.rgb_act), cg = network_to_rgb((float)o.h[1], M.rgb_act), cb = network_to_rgb((float)o.h[2], M.rgb_act);
				if (!PROBE && F.render_mode > 1) { // src/testbed_nerf.cu:689-702
					if (NORMALS) { // :688-693: opposite to the density gradient
						const float dd = network_to_density_derivative((float)o.h[3], M.density_act);
						const f3 nrm = normalize3(mk3(-dd * gx, -dd * gy, -dd * gz));
						cr = nrm.x; cg = nrm.y; cb = nrm.z;
					} else if (F.render_mode == 2) {
						cr = cg = cb = alpha;
					} else if (F.render_mode == 3) {
						cr = (pos.x - 0.5f) / 2.0f + 0.5f; cg = (pos.y - 0.5f) / 2.0f + 0.5f; cb = (pos.z - 0.5f) / 2.0f + 0.5f;
					} else {
						cr = cg = cb = dot3(cam_fwd, sub3(pos, ray.o)) * F.depth_scale;
					}
				}
				acc.r += cr * weight;
				acc.g += cg * weight;
				acc.b += cb * weight;
				acc.a += weight;
				if (weight > acc.max_weight) {
					acc.max_weight = weight;
					acc.depth = sdepth;
				}
				++step;
				if (acc.a > (1.0f - F.min_transmittance)) {
					acc.r /= acc.a; acc.g /= acc.a; acc.b /= acc.a; acc.a /= acc.a;
					ray.alive = false;
					finished = true;
				} else if (step >= MARCH_ITER) {
					ray.alive = false; // never compacted into the hit buffer by the reference loop (:2056)
				}
			}
			n_samples += (uint32_t)__popcll(__ballot(have));
		}
		if (left_box && ray.alive) { // the ray had left the render box behind its last sample
			ray.alive = false;
			finished = true;
		}
		left_box = false;
		n_pend = 0;
		sl_lo = sl_hi = 0;
		n_slots = 0;
		if (PROF) { t1 = stamp(); pt[3] += t1 - t0; tr_t[4] = t1; trace_emit(4); }
	}
	if (PROF) { // per-lane skip steps by jump size: reduce over the wave first
		for (int k = 0; k < 3; ++k)
			for (int off = 32; off > 0; off >>= 1) p_skip[k] += __shfl_down(p_skip[k], off, 64);
	}
	if (PROF && tr_slot >= 0 && lane == 0) {
		uint32_t* hd = F.trace + 16 + (size_t)tr_slot * 16u;
		const unsigned long long rt_end = realtime(), t_end = stamp();
		hd[4] = tr_it;
		hd[7] = (uint32_t)rt_end; hd[8] = (uint32_t)(rt_end >> 32);
		hd[12] = (uint32_t)t_end;
		hd[13] = n_samples;
	}
	if (PROF && lane == 0 && F.prof) {
		for (int k = 0; k < 3; ++k) atomicAdd(&F.prof[12 + k], p_skip[k]);
		for (int k = 0; k < 4; ++k) atomicAdd(&F.prof[k], pt[k]);
		atomicAdd(&F.prof[4], p_iters);
		atomicAdd(&F.prof[5], p_passes);
		atomicAdd(&F.prof[6], p_rounds);
		atomicAdd(&F.prof[7], p_lane_steps);
		unsigned long long t_end = realtime(), t_first = ~F.prof[8];
		atomicMax(&F.prof[9], t_end);
		unsigned long long bucket = (t_end - t_first) / 10000ull; // 100 MHz * 0.1 ms = 10000 ticks
		if (bucket > 47ull) bucket = 47ull;
		atomicAdd(&F.prof[16 + bucket], 1ull);
	}

	finish_launch(F, lane, n_alive_init, n_hit, n_samples);
}

__global__ __launch_bounds__(BLOCK, 2) void render_nerf_fused(const ModelParams M, const CameraParams C, const FrameParams F) {
	ProbeParams P{};
	fused_body<false>(M, C, F, P);
}
#ifndef NGP_UNIT_NONPLAIN_WAVES
#define NGP_UNIT_NONPLAIN_WAVES 3
#endif
__global__ __launch_bounds__(BLOCK, NGP_UNIT_NONPLAIN_WAVES) void render_nerf_fused_unit(const ModelParams M, const CameraParams C, const FrameParams F) {
	ProbeParams P{};
	fused_body<false, false, true>(M, C, F, P);
}
// the same for a static pinhole camera without depth of field or environment map -- the frame a benchmark or a screenshot renders
// threads per workgroup of the benchmark instantiation. Waves share rays inside a workgroup (knob 7), so a larger one balances better:
// 768 (one 12-wave workgroup per CU) renders one 1080p frame at a time 2 % faster, but such a workgroup holds its CU until its last
// wave is done, and overlapped frames / a rank's share lose 2-10 %; 384 or 512 leave SIMDs half empty (wave placement). Measured, 256 stays.
constexpr int FB_UNIT_PLAIN = BLOCK;
__global__ __launch_bounds__(FB_UNIT_PLAIN, 3) void render_nerf_fused_unit_plain(const ModelParams M, const CameraParams C, const FrameParams F) {
	ProbeParams P{};
	fused_body<false, 0, true, 1, true, 1, true, false, FB_UNIT_PLAIN>(M, C, F, P);
}
__global__ __launch_bounds__(BLOCK, 3) void render_nerf_fused_c5_plain(const ModelParams M, const CameraParams C, const FrameParams F) {
	ProbeParams P{};
	fused_body<false, 0, false, 4, false, 1, true>(M, C, F, P);
}
// scenes of up to 5 cascades (aabb_scale <= 16: fox, garden) rendered inside their occupancy grid: the block summaries of four cascades in
// LDS (16 KB instead of 32; the fifth cascade's 4 KB table is read through the vector L1) leave room for a third workgroup per CU
__global__ __launch_bounds__(BLOCK, 3) void render_nerf_fused_c5(const ModelParams M, const CameraParams C, const FrameParams F) {
	ProbeParams P{};
	fused_body<false, 0, false, 4, false>(M, C, F, P);
}
// rgb heads with 1 or 3 hidden layers (configs/nerf/base_1layer.json, base_3layer.json): the general kernel with no / two 64x64 layers
__global__ __launch_bounds__(BLOCK, 2) void render_nerf_fused_mid0(const ModelParams M, const CameraParams C, const FrameParams F) {
	ProbeParams P{};
	fused_body<false, false, false, (int)NERF_CASCADES, true, 0>(M, C, F, P);
}
__global__ __launch_bounds__(BLOCK, 2) void render_nerf_fused_mid2(const ModelParams M, const CameraParams C, const FrameParams F) {
	ProbeParams P{};
	fused_body<false, false, false, (int)NERF_CASCADES, true, 2>(M, C, F, P);
}
// heads without a hidden layer: configs/nerf/base_0layer.json (the rgb head is one matrix) and linear.json (both are)
__global__ __launch_bounds__(BLOCK, 2) void render_nerf_fused_lin_rgb(const ModelParams M, const CameraParams C, const FrameParams F) {
	ProbeParams P{};
	fused_body<false, false, false, (int)NERF_CASCADES, true, -1>(M, C, F, P);
}
__global__ __launch_bounds__(BLOCK, 2) void render_nerf_fused_lin(const ModelParams M, const CameraParams C, const FrameParams F) {
	ProbeParams P{};
	fused_body<false, false, false, (int)NERF_CASCADES, true, -2>(M, C, F, P);
}
__global__ __launch_bounds__(BLOCK, 2) void render_nerf_fused_normals(const ModelParams M, const CameraParams C, const FrameParams F) {
	ProbeParams P{};
	fused_body<false, false, false, (int)NERF_CASCADES, true, 1, false, true>(M, C, F, P);
}
__global__ __launch_bounds__(BLOCK, 2) void render_nerf_fused_prof(const ModelParams M, const CameraParams C, const FrameParams F) {
	ProbeParams P{};
	fused_body<false, 1>(M, C, F, P);
}
// the stamped twins of the kernel a benchmark frame runs (unit scene, static pinhole camera), at its occupancy: section stamps (1) and,
// in addition, stamps inside the network section (2: they serialise gather wait and MFMA chain, so that build's totals are an upper bound)
__global__ __launch_bounds__(BLOCK, 3) void render_nerf_fused_unit_plain_prof(const ModelParams M, const CameraParams C, const FrameParams F) {
	ProbeParams P{};
	fused_body<false, 1, true, 1, true, 1, true>(M, C, F, P);
}
__global__ __launch_bounds__(BLOCK, 3) void render_nerf_fused_unit_plain_prof2(const ModelParams M, const CameraParams C, const FrameParams F) {
	ProbeParams P{};
	fused_body<false, 2, true, 1, true, 1, true>(M, C, F, P);
}

// the same machinery fed by the probe ray fans instead of the camera (Testbed::computeEnvmap*, testbed.h:709-743)
__global__ __launch_bounds__(BLOCK, 2) void trace_probe_fused(const ModelParams M, const FrameParams F, const ProbeParams P) {
	CameraParams C{};
	fused_body<true>(M, C, F, P);
}

// probe texture: texel = mean of its rays' shaded RGBA, summed in increasing ray index (mode 3: one texture per probe of the grid)
__global__ void probe_reduce_kernel(const ProbeParams P, float4* __restrict__ envmap) {
	uint32_t texel = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t per_probe = P.n_theta * P.n_phi, n_probes = P.mode == 3 ? P.grid_x * P.grid_y : 1u;
	if (texel >= per_probe * n_probes) return;
	if (P.mode == 3) { // one ray per texel
		envmap[texel] = P.ray_rgba[texel];
		return;
	}
	const uint32_t no = P.mode == 2 ? P.n_origin : 1u;
	uint32_t i = texel % P.n_theta, j = texel / P.n_theta;
	float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
	for (uint32_t pr = 0; pr < no; ++pr) {
		for (uint32_t tr = 0; tr < no; ++tr) {
			float4 v = P.ray_rgba[(i * no + tr) + P.n_theta * no * (j * no + pr)];
			acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
		}
	}
	float inv = 1.0f / (float)(no * no);
	envmap[texel] = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
}

// E(n) = sum_texels L(w) max(0, n.w) dOmega, dOmega = 4 pi / (n_theta n_phi), w = the direction the texel's ray travelled:
// the texel direction for the centre fans, -frame(normalize(origin)) * texel direction for an outward (K11) probe.
// One block per query; queries are explicit normals (one probe) or every texel direction of every probe (tabulation).
struct IrradianceQuery {
	uint32_t n_theta, n_phi;
	const float4* envmap;  // n_probes textures
	uint32_t n;            // queries
	const float* normals;  // n x 3, or nullptr: query q = texel q % (n_theta n_phi) of probe q / (n_theta n_phi)
	int32_t outward;       // 0: centre fans; 1: one outward probe at origin; 3: the grid
	float origin[3], center[3];
	uint32_t grid_x, grid_y;
	float shell_radius;
	float4* out;
};
__global__ void irradiance_kernel(const IrradianceQuery Q) {
	__shared__ double s[3][256];
	const uint32_t q = blockIdx.x;
	if (q >= Q.n) return;
	const uint32_t texels = Q.n_theta * Q.n_phi;
	const uint32_t probe = Q.normals ? 0u : q / texels;
	f3 nrm;
	if (!Q.normals) nrm = cylindrical_to_dir_nerf((float)((q % texels) % Q.n_theta) / (float)Q.n_theta, (float)((q % texels) / Q.n_theta) / (float)Q.n_phi);
	else nrm = mk3(Q.normals[3 * (size_t)q], Q.normals[3 * (size_t)q + 1], Q.normals[3 * (size_t)q + 2]);
	float frame[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
	if (Q.outward) {
		const f3 origin = Q.outward == 3 ? probe_grid_origin(Q.center, Q.grid_x, Q.grid_y, Q.shell_radius, probe) : mk3(Q.origin[0], Q.origin[1], Q.origin[2]);
		local_frame(normalize3(origin), frame);
	}
	const float4* envmap = Q.envmap + (size_t)texels * probe;
	double a0 = 0, a1 = 0, a2 = 0;
	for (uint32_t t = threadIdx.x; t < texels; t += blockDim.x) {
		f3 w = cylindrical_to_dir_nerf((float)(t % Q.n_theta) / (float)Q.n_theta, (float)(t / Q.n_theta) / (float)Q.n_phi);
		if (Q.outward) w = scale3(normalize3(m3_mulv(frame, w)), -1.0f);
		float c = dot3(nrm, w);
		if (c > 0.0f) {
			float4 L = envmap[t];
			a0 += (double)(L.x * c); a1 += (double)(L.y * c); a2 += (double)(L.z * c);
		}
	}
	s[0][threadIdx.x] = a0; s[1][threadIdx.x] = a1; s[2][threadIdx.x] = a2;
	__syncthreads();
	for (int off = 128; off > 0; off >>= 1) {
		if ((int)threadIdx.x < off) {
			s[0][threadIdx.x] += s[0][threadIdx.x + off];
			s[1][threadIdx.x] += s[1][threadIdx.x + off];
			s[2][threadIdx.x] += s[2][threadIdx.x + off];
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		const double d_omega = 4.0 * 3.14159265358979323846 / ((double)Q.n_theta * (double)Q.n_phi);
		Q.out[q] = make_float4((float)(s[0][0] * d_omega), (float)(s[1][0] * d_omega), (float)(s[2][0] * d_omega), 0.f);
	}
}

// ---------------------------------------------------------------------------------------------------------
// Stage kernels (parity tests / tools): same device functions, one wave = 64 samples in 4 passes.
__global__ __launch_bounds__(BLOCK) void grid_encode_kernel(const ModelParams M, uint32_t n, const float* __restrict__ pos01, uint16_t* __restrict__ out) {
	__shared__ LevelInfo s_lv[N_LEVELS];
	if (threadIdx.x < N_LEVELS) s_lv[threadIdx.x] = M.levels[threadIdx.x];
	__syncthreads();
	const int lane = threadIdx.x & 63, c = lane & 15, h = lane >> 4;
	const GridRsrc t_grid = make_grid_rsrc(M.grid, M.grid_bytes), t_xgrid = make_grid_rsrc(M.xgrid, M.xgrid_bytes);
	const uint32_t wave = (blockIdx.x * BLOCK + threadIdx.x) >> 6;
	for (int p = 0; p < 4; ++p) {
		uint32_t s = wave * 64u + 16u * p + c;
		uint32_t sc = s < n ? s : n - 1;
		half8 enc = encode_level_pair(t_grid, t_xgrid, s_lv, h, pos01[3 * sc], pos01[3 * sc + 1], pos01[3 * sc + 2]);
		if (s < n) {
			union { half_t h; uint16_t u; } cv;
			for (int j = 0; j < 8; ++j) {
				cv.h = enc[j];
				out[(size_t)s * 32 + 16 * (j >> 2) + 4 * h + (j & 3)] = cv.u;
			}
		}
	}
}

template <int RGB_MID>
__global__ __launch_bounds__(BLOCK) void network_inference_kernel(const ModelParams M, uint32_t n, const float* __restrict__ pos01, const float* __restrict__ dir01, uint16_t* __restrict__ out) {
	__shared__ uint4 s_w[n_frags_for(RGB_MID) * 64];
	__shared__ LevelInfo s_lv[N_LEVELS];
	for (int i = threadIdx.x; i < n_frags_for(RGB_MID) * 64; i += BLOCK) s_w[i] = M.wfrags[i];
	if (threadIdx.x < N_LEVELS) s_lv[threadIdx.x] = M.levels[threadIdx.x];
	__syncthreads();
	const int lane = threadIdx.x & 63, c = lane & 15;
	const GridRsrc t_grid = make_grid_rsrc(M.grid, M.grid_bytes), t_xgrid = make_grid_rsrc(M.xgrid, M.xgrid_bytes);
	const uint32_t wave = (blockIdx.x * BLOCK + threadIdx.x) >> 6;
	for (int p = 0; p < 4; ++p) {
		uint32_t s = wave * 64u + 16u * p + c;
		uint32_t sc = s < n ? s : n - 1;
		half8 enc = encode_level_pair(t_grid, t_xgrid, s_lv, lane >> 4, pos01[3 * sc], pos01[3 * sc + 1], pos01[3 * sc + 2]);
		MlpOut mo = mlp_pass<RGB_MID>(s_w, lane, enc, sh4_from_dir(lane >> 4, dir01[3 * sc], dir01[3 * sc + 1], dir01[3 * sc + 2]));
		if (s < n && lane < 16) {
			union { half_t h; uint16_t u; } cv;
			cv.h = mo.rgb[0]; out[(size_t)s * 4 + 0] = cv.u;
			cv.h = mo.rgb[1]; out[(size_t)s * 4 + 1] = cv.u;
			cv.h = mo.rgb[2]; out[(size_t)s * 4 + 2] = cv.u;
			cv.h = mo.sigma;  out[(size_t)s * 4 + 3] = cv.u;
		}
	}
}

// The A fragments of W1^T for the Normals mode's backward pass, permuted out of the forward fragments of W1 (FRAG_D0..+3: lane
// (h, row), element j = W1[16 m + row][16 (j >> 2) + 4 h + (j & 3)]): fragment 2t + s, lane (h', row'), element j' =
// W1[neuron 32 s + 16 (j' >> 2) + 4 h' + (j' & 3)][encoding row 16 t + row']. One workgroup of 256 threads; runs whenever the forward
// fragments change (model load, the training loop's hand-over to the renderer).
__global__ void build_normals_fragments_kernel(uint16_t* __restrict__ wfrags) {
	const int i = threadIdx.x; // fragment 2t + s = i >> 6, lane = i & 63
	const int t = i >> 7, s = (i >> 6) & 1, lane = i & 63, h1 = lane >> 4, row1 = lane & 15;
	for (int j1 = 0; j1 < 8; ++j1) {
		const int neuron = 32 * s + 16 * (j1 >> 2) + 4 * h1 + (j1 & 3), e = 16 * t + row1;
		const int m = neuron >> 4, row = neuron & 15, h = (e & 15) >> 2, j = 4 * (e >> 4) + (e & 3);
		wfrags[((size_t)(FRAG_NORMALS + 2 * t + s) * 64 + lane) * 8 + j1] = wfrags[((size_t)(FRAG_D0 + m) * 64 + 16 * h + row) * 8 + j];
	}
}
// ngp_density_gradient: d density logit / d position for explicit positions (the stage behind ERenderMode::Normals)
__global__ __launch_bounds__(BLOCK) void density_gradient_kernel(const ModelParams M, uint32_t n, const float* __restrict__ pos01, float* __restrict__ out) {
	__shared__ uint4 s_w[FRAG_R0 * 64];
	__shared__ LevelInfo s_lv[N_LEVELS];
	for (int i = threadIdx.x; i < FRAG_R0 * 64; i += BLOCK) s_w[i] = M.wfrags[i];
	if (threadIdx.x < N_LEVELS) s_lv[threadIdx.x] = M.levels[threadIdx.x];
	__syncthreads();
	const int lane = threadIdx.x & 63, c = lane & 15, hq = lane >> 4;
	const GridRsrc t_grid = make_grid_rsrc(M.grid, M.grid_bytes), t_xgrid = make_grid_rsrc(M.xgrid, M.xgrid_bytes);
	const uint32_t wave = (blockIdx.x * BLOCK + threadIdx.x) >> 6;
	for (int p = 0; p < 4; ++p) {
		const uint32_t s = wave * 64u + 16u * p + c;
		const uint32_t sc = s < n ? s : n - 1;
		EncodeInFlight e;
		encode_issue(t_grid, t_xgrid, s_lv, hq, pos01[3 * sc], pos01[3 * sc + 1], pos01[3 * sc + 2], e);
		const half8 enc = encode_finish(e);
		DensityGrad dg = density_gradient_pass(s_w, M.wfrags, lane, e, enc, s_lv[hq].scale, s_lv[hq + 4].scale, M.density_linear != 0);
#pragma unroll
		for (int k = 0; k < 3; ++k) {
			dg.g[k] += __shfl_xor(dg.g[k], 16, 64);
			dg.g[k] += __shfl_xor(dg.g[k], 32, 64);
		}
		if (s < n && lane < 16) {
#pragma unroll
			for (int k = 0; k < 3; ++k) out[(size_t)s * 3 + k] = dg.g[k] * (1.0f / 128.0f);
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
// Density-grid refresh: generate_grid_samples_nerf_nonuniform + NerfNetwork::density +
// splat_grid_samples_nerf_max_nearest_neighbor in one kernel (src/testbed_nerf.cu:185-232, 2812-2852). A wave owns
// 64 samples: every lane draws its sample (cell, position), then four 16-sample passes run the hash-grid encode and
// the density head on MFMA exactly like the render kernel, and lanes 0..15 splat their pass's results.
template <bool DLIN> // DLIN: a density head without a hidden layer (configs/nerf/linear.json)
__global__ __launch_bounds__(BLOCK) void density_grid_samples_kernel(const ModelParams M, uint32_t n_samples, Pcg32 rng, uint32_t step, uint32_t n_cascades,
                                                                     float thresh, const float* __restrict__ grid_in, float* __restrict__ grid_tmp) {
	__shared__ uint4 s_w[N_FRAGS * 64];
	__shared__ LevelInfo s_lv[N_LEVELS];
	for (int i = threadIdx.x; i < N_FRAGS * 64; i += BLOCK) s_w[i] = M.wfrags[i];
	if (threadIdx.x < N_LEVELS) s_lv[threadIdx.x] = M.levels[threadIdx.x];
	__syncthreads();
	const int lane = threadIdx.x & 63, c = lane & 15;
	const GridRsrc t_grid = make_grid_rsrc(M.grid, M.grid_bytes), t_xgrid = make_grid_rsrc(M.xgrid, M.xgrid_bytes);
	// Sample i visits cell ((i + step n) A + C) mod 2^21 first -- a bijection of i's low 21 bits. The reference lets thread
	// i take sample i, which sends neighbouring lanes to unrelated cells: every hash-grid gather of the wave misses. The
	// set of samples is what matters (the splat is an atomic max), so thread t takes the sample whose first cell is t
	// in Morton order: a wave works on one 4x4x4 block of cells and its gathers share lines.
	const uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
	constexpr uint32_t A = 56924617u, C = 96925573u;
	constexpr uint32_t AINV = []() { uint32_t x = A; for (int k = 0; k < 5; ++k) x *= 2u - A * x; return x; }(); // A^-1 mod 2^32
	static_assert(A * AINV == 1u, "modular inverse");
	// (whole replicas of 2^21 samples only; the samples of a partial replica keep the reference's order)
	const uint32_t n_whole = n_samples & ~(NERF_GRID_N_CELLS - 1u);
	const uint32_t i = t < n_whole ? ((t & ~(NERF_GRID_N_CELLS - 1u)) | ((((t & (NERF_GRID_N_CELLS - 1u)) - C) * AINV - step * n_samples) & (NERF_GRID_N_CELLS - 1u))) : t;
	const bool valid = i < n_samples;
	// 1 random number to select the level, 3 to select the position
	rng.advance((uint64_t)i * 4u);
	const uint32_t level = (uint32_t)(rng.next_float() * (float)n_cascades) % n_cascades;
	uint32_t idx = 0;
	for (uint32_t j = 0; j < 10; ++j) { // a grid cell that has density
		idx = ((i + step * n_samples) * 56924617u + j * 19349663u + 96925573u) % NERF_GRID_N_CELLS;
		idx += level * NERF_GRID_N_CELLS;
		if (!valid || grid_in[idx] > thresh) break;
	}
	const uint32_t pos_idx = idx % NERF_GRID_N_CELLS;
	const float x = (float)morton3D_invert(pos_idx >> 0), y = (float)morton3D_invert(pos_idx >> 1), z = (float)morton3D_invert(pos_idx >> 2);
	const float rx = rng.next_float(), ry = rng.next_float(), rz = rng.next_float();
	const float scale = __builtin_ldexpf(1.0f, (int)level);
	f3 pos = mk3(((x + rx) / (float)NERF_GRIDSIZE - 0.5f) * scale + 0.5f, ((y + ry) / (float)NERF_GRIDSIZE - 0.5f) * scale + 0.5f,
	             ((z + rz) / (float)NERF_GRIDSIZE - 0.5f) * scale + 0.5f);
	f3 w = div3(sub3(pos, mk3(M.aabb_min[0], M.aabb_min[1], M.aabb_min[2])), mk3(M.aabb_diag[0], M.aabb_diag[1], M.aabb_diag[2])); // warp_position
	for (int p = 0; p < 4; ++p) {
		const int src = 16 * p + c;
		const float sx = __shfl(w.x, src, 64), sy = __shfl(w.y, src, 64), sz = __shfl(w.z, src, 64);
		const uint32_t s_idx = (uint32_t)__shfl((int)idx, src, 64);
		const int s_valid = __shfl(valid ? 1 : 0, src, 64);
		half8 enc = encode_level_pair(t_grid, t_xgrid, s_lv, lane >> 4, sx, sy, sz);
		half_t logit = density_pass<DLIN>(s_w, lane, enc);
		if (lane < 16 && s_valid) {
			// optical thickness of the smallest step (level 0, :218); positive floats order like their bit patterns
			float thickness = network_to_density((float)logit, M.density_act) * stepsize();
			atomicMax((unsigned int*)&grid_tmp[s_idx], __float_as_uint(thickness));
		}
	}
}
// The same refresh for a network without a hash grid (configs/nerf/frequency.json): update_density_grid_nerf works for any NerfNetwork
// (src/testbed_nerf.cu:2772-2861). Three launches -- sample cells and positions, the density through the wide-MLP kernel of
// wide_kernels.hip (its rgb head runs along: the entry point is NerfNetwork::inference; 2 M samples cost a few ms), splat.
__global__ void density_grid_positions_kernel(const ModelParams M, uint32_t n_samples, Pcg32 rng, uint32_t step, uint32_t n_cascades, float thresh,
                                              const float* __restrict__ grid_in, float* __restrict__ pos01, uint32_t* __restrict__ cell) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_samples) return;
	rng.advance((uint64_t)i * 4u); // 1 random number to select the level, 3 to select the position
	const uint32_t level = (uint32_t)(rng.next_float() * (float)n_cascades) % n_cascades;
	uint32_t idx = 0;
	for (uint32_t j = 0; j < 10; ++j) { // a grid cell that has density
		idx = ((i + step * n_samples) * 56924617u + j * 19349663u + 96925573u) % NERF_GRID_N_CELLS;
		idx += level * NERF_GRID_N_CELLS;
		if (grid_in[idx] > thresh) break;
	}
	const uint32_t pos_idx = idx % NERF_GRID_N_CELLS;
	const float x = (float)morton3D_invert(pos_idx >> 0), y = (float)morton3D_invert(pos_idx >> 1), z = (float)morton3D_invert(pos_idx >> 2);
	const float rx = rng.next_float(), ry = rng.next_float(), rz = rng.next_float();
	const float scale = __builtin_ldexpf(1.0f, (int)level);
	const f3 pos = mk3(((x + rx) / (float)NERF_GRIDSIZE - 0.5f) * scale + 0.5f, ((y + ry) / (float)NERF_GRIDSIZE - 0.5f) * scale + 0.5f,
	                   ((z + rz) / (float)NERF_GRIDSIZE - 0.5f) * scale + 0.5f);
	const f3 w = div3(sub3(pos, mk3(M.aabb_min[0], M.aabb_min[1], M.aabb_min[2])), mk3(M.aabb_diag[0], M.aabb_diag[1], M.aabb_diag[2])); // warp_position
	pos01[3 * (size_t)i] = w.x; pos01[3 * (size_t)i + 1] = w.y; pos01[3 * (size_t)i + 2] = w.z;
	cell[i] = idx;
}
__global__ void density_grid_splat_kernel(uint32_t n_samples, uint32_t density_act, const uint32_t* __restrict__ cell, const uint16_t* __restrict__ net_out, float* __restrict__ grid_tmp) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_samples) return;
	union { uint16_t u; half_t h; } cv;
	cv.u = net_out[(size_t)i * 4 + 3]; // the density logit
	const float thickness = network_to_density((float)cv.h, density_act) * stepsize(); // optical thickness of the smallest step (:218)
	atomicMax((unsigned int*)&grid_tmp[cell[i]], __float_as_uint(thickness)); // positive floats order like their bit patterns
}

// ema_grid_samples_nerf (:253-276): a decayed maximum, cells marked negative stay
__global__ void density_grid_ema_kernel(uint32_t n_elements, float decay, float* __restrict__ grid, const float* __restrict__ grid_tmp) {
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_elements) return;
	float prev = grid[i];
	grid[i] = prev < 0.f ? prev : fmaxf(prev * decay, grid_tmp[i]);
}

__global__ void init_rays_kernel(const ModelParams M, const CameraParams C, NerfPayload* __restrict__ payloads) {
	uint32_t x = threadIdx.x + blockDim.x * blockIdx.x;
	uint32_t y = threadIdx.y + blockDim.y * blockIdx.y;
	if (x >= (uint32_t)C.width || y >= (uint32_t)C.height) return;
	RayState r;
	init_ray(M, C, x, y, r);
	advance_pos(M, C, r);
	NerfPayload p;
	p.origin[0] = r.o.x; p.origin[1] = r.o.y; p.origin[2] = r.o.z;
	p.dir[0] = r.d.x; p.dir[1] = r.d.y; p.dir[2] = r.d.z;
	p.t = r.t;
	p.max_weight = 0.f;
	p.idx = r.alive || (r.d.x != 0.f || r.d.y != 0.f || r.d.z != 0.f) ? r.idx : 0u;
	p.n_steps = 0;
	p.alive = r.alive ? 1 : 0;
	p.pad = 0;
	payloads[x + (uint32_t)C.width * y] = p;
}

// ---------------------------------------------------------------------------------------------------------
// K8/K9: update_density_grid_mean_and_bitfield (src/testbed_nerf.cu:284-331, 2863-2877)
__global__ void half_to_float_kernel(uint32_t n, const uint16_t* __restrict__ in, float* __restrict__ out) {
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	union { uint16_t u; half_t h; } cv;
	cv.u = in[i];
	out[i] = (float)cv.h;
}

// mean of fmaxf(v,0)/n over level 0; summed in double so that the (unspecified) reduction order cannot change the
// rounded fp32 result.
__global__ void density_mean_kernel(uint32_t n, const float* __restrict__ grid, double* __restrict__ partial) {
	__shared__ double s[256];
	double acc = 0.0;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += (double)(fmaxf(grid[i], 0.0f) / (float)n);
	s[threadIdx.x] = acc;
	__syncthreads();
	for (int off = 128; off > 0; off >>= 1) {
		if ((int)threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
		__syncthreads();
	}
	if (threadIdx.x == 0) partial[blockIdx.x] = s[0];
}

__global__ void grid_to_bitfield_kernel(uint32_t n_elements, uint32_t n_nonzero_elements, const float* __restrict__ grid, uint8_t* __restrict__ bitfield, float mean) {
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_elements) return;
	if (i >= n_nonzero_elements) { bitfield[i] = 0; return; }
	float thresh = fminf(0.01f, mean); // NERF_MIN_OPTICAL_THICKNESS
	uint8_t bits = 0;
#pragma unroll
	for (int j = 0; j < 8; ++j) bits |= grid[(size_t)i * 8 + j] > thresh ? (uint8_t)(1u << j) : 0;
	bitfield[i] = bits;
}

// empty-space summary: one bit per 4x4x4 block (8 consecutive Morton-ordered bytes) of every mip
__global__ void coarse_occupancy_kernel(const uint8_t* __restrict__ bitfield, uint32_t* __restrict__ coarse) {
	uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; // output word: 32 blocks = 256 bytes of bitfield
	if (w >= NERF_CASCADES * COARSE_WORDS_PER_MIP) return;
	const uint2* src = (const uint2*)(bitfield + (size_t)w * 256);
	uint32_t bits = 0;
	for (int b = 0; b < 32; ++b) {
		uint2 v = src[b];
		bits |= ((v.x | v.y) != 0u ? 1u : 0u) << b;
	}
	coarse[w] = bits;
}
// second level: one bit per 16x16x16 block = 64 consecutive 4x4x4 blocks = 2 words of the first level
__global__ void coarse16_occupancy_kernel(uint32_t* __restrict__ coarse) {
	uint32_t w = threadIdx.x; // NERF_CASCADES * 16 output words
	if (w >= NERF_CASCADES * 16) return;
	uint32_t bits = 0;
	for (int b = 0; b < 32; ++b) {
		uint32_t a0 = coarse[(size_t)(w * 32 + b) * 2], a1 = coarse[(size_t)(w * 32 + b) * 2 + 1];
		bits |= ((a0 | a1) != 0u ? 1u : 0u) << b;
	}
	coarse[NERF_CASCADES * COARSE_WORDS_PER_MIP + w] = bits;
}

__global__ void bitfield_max_pool_kernel(uint32_t n_elements, const uint8_t* __restrict__ prev_level, uint8_t* __restrict__ next_level) {
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_elements) return;
	uint8_t bits = 0;
#pragma unroll
	for (int j = 0; j < 8; ++j) bits |= prev_level[(size_t)i * 8 + j] > 0 ? (uint8_t)(1u << j) : 0;
	uint32_t x = morton3D_invert(i >> 0) + NERF_GRIDSIZE / 8;
	uint32_t y = morton3D_invert(i >> 1) + NERF_GRIDSIZE / 8;
	uint32_t z = morton3D_invert(i >> 2) + NERF_GRIDSIZE / 8;
	next_level[morton3D(x, y, z)] |= bits; // each thread owns a distinct output byte
}

// ---------------------------------------------------------------------------------------------------------
// P1: accumulate_kernel + tonemap_kernel (src/render_buffer.cu:228-262, 529-561), fused: colour space Linear,
// tonemap curve Identity, no DLSS. rgba_out may alias nothing else.
__global__ void accumulate_tonemap_kernel(uint32_t n_pixels, const float4* __restrict__ frame_buffer, float4* __restrict__ accumulate_buffer,
                                          float sample_count, float4 background, float exposure_scale, int to_srgb, int color_space, float4* __restrict__ rgba_out) {
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_pixels) return;
	float4 color = frame_buffer[i];
	if (color_space == 1) { // EColorSpace::SRGB (:245)
		color.x = linear_to_srgb(color.x);
		color.y = linear_to_srgb(color.y);
		color.z = linear_to_srgb(color.z);
	}
	float4 tmp = sample_count > 0.f ? accumulate_buffer[i] : make_float4(0.f, 0.f, 0.f, 0.f);
	tmp.x = (tmp.x * sample_count + color.x) / (sample_count + 1.0f);
	tmp.y = (tmp.y * sample_count + color.y) / (sample_count + 1.0f);
	tmp.z = (tmp.z * sample_count + color.z) / (sample_count + 1.0f);
	tmp.w = (tmp.w * sample_count + color.w) / (sample_count + 1.0f);
	accumulate_buffer[i] = tmp;
	if (!rgba_out) return;
	float bgr = background.x, bgg = background.y, bgb = background.z;
	if (color_space != 1) { // the background colour is sRGB (:537-541)
		bgr = srgb_to_linear(bgr);
		bgg = srgb_to_linear(bgg);
		bgb = srgb_to_linear(bgb);
	}
	float weight = (1.0f - tmp.w) * background.w;
	tmp.x += bgr * weight;
	tmp.y += bgg * weight;
	tmp.z += bgb * weight;
	tmp.w += weight;
	if (color_space == 1) { // :326-328
		tmp.x = srgb_to_linear(tmp.x);
		tmp.y = srgb_to_linear(tmp.y);
		tmp.z = srgb_to_linear(tmp.z);
	}
	tmp.x *= exposure_scale;
	tmp.y *= exposure_scale;
	tmp.z *= exposure_scale;
	if (to_srgb) {
		tmp.x = linear_to_srgb(tmp.x);
		tmp.y = linear_to_srgb(tmp.y);
		tmp.z = linear_to_srgb(tmp.z);
	}
	rgba_out[i] = tmp;
}

// ---------------------------------------------------------------------------------------------------------
// launchers (called from ngp_api.cpp)
// Persistent grids are sized to what is resident at once (workgroups per CU from the occupancy query): a workgroup
// that only starts when another one has drained would begin its rays late and stretch the frame by a ray lifetime.
template <typename K>
static int resident_blocks_per_cu(K kernel, int threads = BLOCK) {
	int n = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, 0) != hipSuccess || n < 1) n = 1;
	return n;
}
// the persistent grid of a frame: what is resident, but no more waves than tiles
static int grid_blocks(const FrameParams& F, int resident, int threads = BLOCK) {
	const int waves = threads / 64, needed = (int)((F.n_local_tiles + waves - 1) / waves); // one tile per wave at least
	return resident > needed ? (needed > 0 ? needed : 1) : resident;
}
static FrameParams with_grid(const FrameParams& F, int n_blocks, int threads = BLOCK) {
	FrameParams G = F;
	G.n_waves = (uint32_t)n_blocks * (threads / 64);
	return G;
}
void launch_render_nerf_wide(const ModelParams& M, const CameraParams& C, const FrameParams& F, int n_cus, hipStream_t stream);
void launch_trace_probe_wide(const ModelParams& M, const FrameParams& F, const ProbeParams& P, int n_cus, hipStream_t stream);
void launch_render_nerf(const ModelParams& M, const CameraParams& C, const FrameParams& F, int n_cus, hipStream_t stream) {
	if (M.wide.width) return launch_render_nerf_wide(M, C, F, n_cus, stream); // configs/nerf/frequency.json: wide_kernels.hip
	const bool unit = M.max_cascade == 0 && M.cone_angle <= 1e-5f;
	const bool c5 = !unit && M.max_cascade < 5 && !F.outside_possible;
	if (F.render_mode == 7) { // ERenderMode::Normals: the density head only, whatever the rgb head
		static const int per_cu_normals = resident_blocks_per_cu(render_nerf_fused_normals);
		const int nb = grid_blocks(F, n_cus * per_cu_normals);
		const FrameParams G = with_grid(F, nb);
		hipLaunchKernelGGL(render_nerf_fused_normals, dim3(nb), dim3(BLOCK), 0, stream, M, C, G);
		return;
	}
	if (M.rgb_mid != 1) { // the base_0layer / base_1layer / base_3layer / linear heads: one general kernel each
		static const int per_cu_mid0 = resident_blocks_per_cu(render_nerf_fused_mid0), per_cu_mid2 = resident_blocks_per_cu(render_nerf_fused_mid2),
		                 per_cu_lin_rgb = resident_blocks_per_cu(render_nerf_fused_lin_rgb), per_cu_lin = resident_blocks_per_cu(render_nerf_fused_lin);
		const int nb = grid_blocks(F, n_cus * (M.rgb_mid == 0 ? per_cu_mid0 : M.rgb_mid == 2 ? per_cu_mid2 : M.density_linear ? per_cu_lin : per_cu_lin_rgb));
		const FrameParams G = with_grid(F, nb);
		if (M.rgb_mid == 0) hipLaunchKernelGGL(render_nerf_fused_mid0, dim3(nb), dim3(BLOCK), 0, stream, M, C, G);
		else if (M.rgb_mid < 0 && M.density_linear) hipLaunchKernelGGL(render_nerf_fused_lin, dim3(nb), dim3(BLOCK), 0, stream, M, C, G);
		else if (M.rgb_mid < 0) hipLaunchKernelGGL(render_nerf_fused_lin_rgb, dim3(nb), dim3(BLOCK), 0, stream, M, C, G);
		else hipLaunchKernelGGL(render_nerf_fused_mid2, dim3(nb), dim3(BLOCK), 0, stream, M, C, G);
		return;
	}
	static const int per_cu_generic = resident_blocks_per_cu(render_nerf_fused), per_cu_unit = resident_blocks_per_cu(render_nerf_fused_unit),
	                 per_cu_prof = resident_blocks_per_cu(render_nerf_fused_prof), per_cu_c5 = resident_blocks_per_cu(render_nerf_fused_c5),
	                 per_cu_unit_plain_prof = resident_blocks_per_cu(render_nerf_fused_unit_plain_prof), per_cu_unit_plain_prof2 = resident_blocks_per_cu(render_nerf_fused_unit_plain_prof2),
	                 per_cu_unit_plain = resident_blocks_per_cu(render_nerf_fused_unit_plain, FB_UNIT_PLAIN), per_cu_c5_plain = resident_blocks_per_cu(render_nerf_fused_c5_plain);
	const bool plain = C.lens_mode == 0 && C.aperture_size == 0.0f && !C.moving && !F.envmap;
	int per_cu = F.prof ? (unit && plain ? (F.prof_level >= 2 ? per_cu_unit_plain_prof2 : per_cu_unit_plain_prof) : per_cu_prof) : unit ? (plain ? per_cu_unit_plain : per_cu_unit) : c5 ? (plain ? per_cu_c5_plain : per_cu_c5) : per_cu_generic;
	// a rank of a sharded frame leaves a third of every CU to the collective's kernels and to the next frame's launch
	// (measured on one GPU with two frames in flight: 2 per CU is as fast as 3 from N = 2 on, tools/shard_probe.py)
	static const int shard_per_cu = []() { const char* e = getenv("NGP_SHARD_BLOCKS_PER_CU"); int v = e ? atoi(e) : 2; return v >= 1 && v <= 8 ? v : 2; }(); // experiments: tools/shard_probe.py
	if (F.shard_count > 1 && per_cu > shard_per_cu) per_cu = shard_per_cu;
	// a frame of fewer than ~2 tiles per resident wave (below ~800 x 450) is faster on two workgroups per CU as well, alone (512 x 288: 0.81 -> 0.72 ms,
	// 800 x 450: 0.89 -> 0.84) and with frames in flight (-1..6 %); from 960 x 540 on the third workgroup pays (profiles/r3_small_frame.txt)
	if (F.shard_count <= 1 && F.n_local_tiles <= 6144u && per_cu > 2) per_cu = 2;
	if (const char* e = getenv("NGP_BLOCKS_PER_CU")) { int v = atoi(e); if (v > 0 && v < per_cu) per_cu = v; } // experiments only
	const int threads = (!F.prof && unit && plain) ? FB_UNIT_PLAIN : BLOCK;
	if (F.shard_count > 1 && threads != BLOCK) per_cu = per_cu * threads > 2 * BLOCK ? ((2 * BLOCK) / threads > 0 ? (2 * BLOCK) / threads : 1) : per_cu; // (a rank's share: two thirds of the CU, as above)
	const int n_blocks = grid_blocks(F, n_cus * per_cu, threads);
	const FrameParams G = with_grid(F, n_blocks, threads);
	if (F.prof && unit && plain && F.prof_level >= 2) hipLaunchKernelGGL(render_nerf_fused_unit_plain_prof2, dim3(n_blocks), dim3(BLOCK), 0, stream, M, C, G);
	else if (F.prof && unit && plain) hipLaunchKernelGGL(render_nerf_fused_unit_plain_prof, dim3(n_blocks), dim3(BLOCK), 0, stream, M, C, G);
	else if (F.prof) hipLaunchKernelGGL(render_nerf_fused_prof, dim3(n_blocks), dim3(BLOCK), 0, stream, M, C, G);
	else if (unit && plain) hipLaunchKernelGGL(render_nerf_fused_unit_plain, dim3(n_blocks), dim3(FB_UNIT_PLAIN), 0, stream, M, C, G);
	else if (unit) hipLaunchKernelGGL(render_nerf_fused_unit, dim3(n_blocks), dim3(BLOCK), 0, stream, M, C, G);
	else if (c5 && plain) hipLaunchKernelGGL(render_nerf_fused_c5_plain, dim3(n_blocks), dim3(BLOCK), 0, stream, M, C, G);
	else if (c5) hipLaunchKernelGGL(render_nerf_fused_c5, dim3(n_blocks), dim3(BLOCK), 0, stream, M, C, G);
	else hipLaunchKernelGGL(render_nerf_fused, dim3(n_blocks), dim3(BLOCK), 0, stream, M, C, G);
}
void launch_trace_probe(const ModelParams& M, const FrameParams& F, const ProbeParams& P, int n_cus, hipStream_t stream) {
	if (M.wide.width) return launch_trace_probe_wide(M, F, P, n_cus, stream);
	static const int per_cu = resident_blocks_per_cu(trace_probe_fused);
	const int n_blocks = grid_blocks(F, n_cus * per_cu);
	const FrameParams G = with_grid(F, n_blocks);
	hipLaunchKernelGGL(trace_probe_fused, dim3(n_blocks), dim3(BLOCK), 0, stream, M, G, P);
}
void launch_probe_reduce(const ProbeParams& P, float4* envmap, hipStream_t stream) {
	uint32_t n = P.n_theta * P.n_phi * (P.mode == 3 ? P.grid_x * P.grid_y : 1u);
	hipLaunchKernelGGL(probe_reduce_kernel, dim3((n + 127) / 128), dim3(128), 0, stream, P, envmap);
}
// E(n) of the probe texture(s) described by P (modes and shell positions as traced): normals == nullptr tabulates every
// probe at its texel directions (n = probes * texels)
void launch_irradiance(const ProbeParams& P, const float4* envmap, uint32_t n, const float* normals, float4* out, hipStream_t stream) {
	IrradianceQuery Q{};
	Q.n_theta = P.n_theta; Q.n_phi = P.n_phi;
	Q.envmap = envmap;
	Q.n = n;
	Q.normals = normals;
	Q.outward = P.mode == 1 ? 1 : (P.mode == 3 ? 3 : 0);
	for (int i = 0; i < 3; ++i) { Q.origin[i] = P.origin[i]; Q.center[i] = P.center[i]; }
	Q.grid_x = P.grid_x; Q.grid_y = P.grid_y; Q.shell_radius = P.shell_radius;
	Q.out = out;
	if (n) hipLaunchKernelGGL(irradiance_kernel, dim3(n), dim3(256), 0, stream, Q);
}
void launch_grid_encode(const ModelParams& M, uint32_t n, const float* pos01, uint16_t* out, hipStream_t stream) {
	uint32_t n_waves = (n + 63) / 64;
	hipLaunchKernelGGL(grid_encode_kernel, dim3((n_waves + 3) / 4), dim3(BLOCK), 0, stream, M, n, pos01, out);
}
void launch_network_inference(const ModelParams& M, uint32_t n, const float* pos01, const float* dir01, uint16_t* out, hipStream_t stream) {
	uint32_t n_waves = (n + 63) / 64;
	if (M.rgb_mid < 0 && M.density_linear) hipLaunchKernelGGL(network_inference_kernel<-2>, dim3((n_waves + 3) / 4), dim3(BLOCK), 0, stream, M, n, pos01, dir01, out);
	else if (M.rgb_mid < 0) hipLaunchKernelGGL(network_inference_kernel<-1>, dim3((n_waves + 3) / 4), dim3(BLOCK), 0, stream, M, n, pos01, dir01, out);
	else if (M.rgb_mid == 0) hipLaunchKernelGGL(network_inference_kernel<0>, dim3((n_waves + 3) / 4), dim3(BLOCK), 0, stream, M, n, pos01, dir01, out);
	else if (M.rgb_mid == 2) hipLaunchKernelGGL(network_inference_kernel<2>, dim3((n_waves + 3) / 4), dim3(BLOCK), 0, stream, M, n, pos01, dir01, out);
	else hipLaunchKernelGGL(network_inference_kernel<1>, dim3((n_waves + 3) / 4), dim3(BLOCK), 0, stream, M, n, pos01, dir01, out);
}
void launch_build_normals_fragments(uint4* wfrags, hipStream_t stream) {
	hipLaunchKernelGGL(build_normals_fragments_kernel, dim3(1), dim3(256), 0, stream, (uint16_t*)wfrags);
}
void launch_density_gradient(const ModelParams& M, uint32_t n, const float* pos01, float* out, hipStream_t stream) {
	if (n == 0) return;
	hipLaunchKernelGGL(density_gradient_kernel, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, stream, M, n, pos01, out);
}
void launch_density_grid_update(const ModelParams& M, uint32_t n_samples, const Pcg32& rng, uint32_t step, uint32_t n_cascades, float thresh, const float* grid,
                                float* grid_tmp, hipStream_t stream) {
	if (!n_samples) return;
	if (M.density_linear) hipLaunchKernelGGL(density_grid_samples_kernel<true>, dim3((n_samples + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, stream, M, n_samples, rng, step, n_cascades, thresh, grid, grid_tmp);
	else hipLaunchKernelGGL(density_grid_samples_kernel<false>, dim3((n_samples + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, stream, M, n_samples, rng, step, n_cascades, thresh, grid, grid_tmp);
}
void launch_network_inference_wide(const ModelParams& M, uint32_t n, const float* pos01, const float* dir01, uint16_t* out, int n_cus, hipStream_t stream);
// scratch: n_samples x (3 floats + 1 cell index + 4 fp16 outputs), owned by the caller
void launch_density_grid_update_wide(const ModelParams& M, uint32_t n_samples, const Pcg32& rng, uint32_t step, uint32_t n_cascades, float thresh, const float* grid,
                                     float* grid_tmp, float* d_pos01, uint32_t* d_cell, uint16_t* d_out, int n_cus, hipStream_t stream) {
	if (!n_samples) return;
	hipLaunchKernelGGL(density_grid_positions_kernel, dim3((n_samples + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, stream, M, n_samples, rng, step, n_cascades, thresh, grid, d_pos01, d_cell);
	launch_network_inference_wide(M, n_samples, d_pos01, d_pos01, d_out, n_cus, stream); // (the density does not depend on the direction)
	hipLaunchKernelGGL(density_grid_splat_kernel, dim3((n_samples + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, stream, n_samples, M.density_act, d_cell, d_out, grid_tmp);
}
void launch_density_grid_ema(uint32_t n_elements, float decay, float* grid, const float* grid_tmp, hipStream_t stream) {
	hipLaunchKernelGGL(density_grid_ema_kernel, dim3((n_elements + 255) / 256), dim3(256), 0, stream, n_elements, decay, grid, grid_tmp);
}
void launch_init_rays(const ModelParams& M, const CameraParams& C, NerfPayload* payloads, hipStream_t stream) {
	dim3 threads(16, 8, 1);
	dim3 blocks((C.width + 15) / 16, (C.height + 7) / 8, 1);
	hipLaunchKernelGGL(init_rays_kernel, blocks, threads, 0, stream, M, C, payloads);
}
void launch_density_grid_to_bitfield(const uint16_t* d_grid_fp16, uint32_t n_grid, uint32_t max_cascade, float* d_grid_f32, double* d_partial /*256*/,
                                     uint8_t* d_bitfield, float* out_mean, hipStream_t stream) {
	if (n_grid) hipLaunchKernelGGL(half_to_float_kernel, dim3((n_grid + 255) / 256), dim3(256), 0, stream, n_grid, d_grid_fp16, d_grid_f32);
	hipLaunchKernelGGL(density_mean_kernel, dim3(256), dim3(256), 0, stream, NERF_GRID_N_CELLS, d_grid_f32, d_partial);
	double partial[256];
	(void)hipMemcpyAsync(partial, d_partial, sizeof(partial), hipMemcpyDeviceToHost, stream);
	(void)hipStreamSynchronize(stream);
	double sum = 0.0;
	for (int i = 0; i < 256; ++i) sum += partial[i];
	float mean = (float)sum;
	*out_mean = mean;
	const uint32_t n_elements = NERF_GRID_N_CELLS;
	uint32_t n_bytes = n_elements / 8 * NERF_CASCADES;
	hipLaunchKernelGGL(grid_to_bitfield_kernel, dim3((n_bytes + 255) / 256), dim3(256), 0, stream, n_bytes, n_elements / 8 * (max_cascade + 1), d_grid_f32, d_bitfield, mean);
	for (uint32_t level = 1; level < NERF_CASCADES; ++level) {
		hipLaunchKernelGGL(bitfield_max_pool_kernel, dim3((n_elements / 64 + 255) / 256), dim3(256), 0, stream, n_elements / 64,
		                   d_bitfield + (size_t)(level - 1) * (n_elements / 8), d_bitfield + (size_t)level * (n_elements / 8));
	}
}
void launch_coarse_occupancy(const uint8_t* bitfield, uint32_t* coarse, hipStream_t stream) {
	uint32_t n = NERF_CASCADES * COARSE_WORDS_PER_MIP;
	hipLaunchKernelGGL(coarse_occupancy_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, bitfield, coarse);
	hipLaunchKernelGGL(coarse16_occupancy_kernel, dim3(1), dim3(128), 0, stream, coarse);
}
// tile gather at the primary device of a multi-device context: pixel (x, y) lives in device (tile % n) 's block at
// (tile / n) * 64 + slot -- the layout FrameParams::packed writes
__global__ void unpack_tiles_kernel(const float4* __restrict__ g_rgba, const float* __restrict__ g_depth, uint32_t n_devices, uint32_t n_slots, int width, int height,
                                    float4* __restrict__ rgba, float* __restrict__ depth) {
	const uint32_t x = threadIdx.x + blockDim.x * blockIdx.x, y = threadIdx.y + blockDim.y * blockIdx.y;
	if (x >= (uint32_t)width || y >= (uint32_t)height) return;
	const uint32_t tile = (y >> 3) * (((uint32_t)width + 7u) >> 3) + (x >> 3);
	const size_t src = ((size_t)(tile % n_devices) * n_slots + tile / n_devices) * 64u + (x & 7u) + 8u * (y & 7u);
	rgba[x + (size_t)width * y] = g_rgba[src];
	depth[x + (size_t)width * y] = g_depth[src];
}
void launch_unpack_tiles(const float4* gathered_rgba, const float* gathered_depth, uint32_t n_devices, uint32_t n_slots, int width, int height, float4* rgba, float* depth,
                         hipStream_t stream) {
	hipLaunchKernelGGL(unpack_tiles_kernel, dim3((width + 15) / 16, (height + 7) / 8), dim3(16, 8), 0, stream, gathered_rgba, gathered_depth, n_devices, n_slots, width, height,
	                   rgba, depth);
}
void launch_accumulate_tonemap(uint32_t n_pixels, const float4* frame_buffer, float4* accumulate_buffer, float sample_count, const float* background,
                               float exposure, int to_srgb, int color_space, float4* rgba_out, hipStream_t stream) {
	float4 bg = make_float4(background[0], background[1], background[2], background[3]);
	hipLaunchKernelGGL(accumulate_tonemap_kernel, dim3((n_pixels + 255) / 256), dim3(256), 0, stream, n_pixels, frame_buffer, accumulate_buffer, sample_count,
	                   bg, powf(2.0f, exposure), to_srgb, color_space, rgba_out);
}

} // namespace ngp
