// Per-ray pieces shared by the persistent render kernels (nerf_kernels.hip: the fused base.json kernel; wide_kernels.hip: the
// Frequency-encoding / 128-256 wide MLP kernel): shading of a finished ray, probe ray setup, device clocks and the launch epilogue.
#pragma once

#include "nerf_device.h"

namespace ngp {

struct Accum {
	float r, g, b, a;
	float depth;
	float max_weight;
};

// shade_kernel_nerf (src/testbed_nerf.cu:1361-1401, Shade mode) / shade_kernel_nerf_geometry depth test
// (src/testbed_geometry_training.cu:1843-1846) for one finished ray. compact_kernel_nerf (:1420) only forwards
// rays with alpha > 0.001.
// accumulate_kernel with sample_count 0 (the mean of one sample is the sample) + tonemap_kernel, colour space Linear,
// tonemap curve Identity: background blend, exposure, optional sRGB (src/render_buffer.cu:228-262, 529-561)
NGP_DEV float4 tonemap_pixel(const FrameParams& F, f3 bg_linear, float r, float g, float b, float a) {
	if (F.color_space == 1) { // EColorSpace::SRGB: the sample is averaged as an sRGB value (:245)
		r = linear_to_srgb(r);
		g = linear_to_srgb(g);
		b = linear_to_srgb(b);
	}
	float4 tmp = make_float4(r / 1.0f, g / 1.0f, b / 1.0f, a / 1.0f);
	float weight = (1.0f - tmp.w) * F.background[3];
	tmp.x += bg_linear.x * weight;
	tmp.y += bg_linear.y * weight;
	tmp.z += bg_linear.z * weight;
	tmp.w += weight;
	if (F.color_space == 1) { // back to linear before exposure (:326-328)
		tmp.x = srgb_to_linear(tmp.x);
		tmp.y = srgb_to_linear(tmp.y);
		tmp.z = srgb_to_linear(tmp.z);
	}
	tmp.x *= F.exposure_scale;
	tmp.y *= F.exposure_scale;
	tmp.z *= F.exposure_scale;
	if (F.to_srgb) {
		tmp.x = linear_to_srgb(tmp.x);
		tmp.y = linear_to_srgb(tmp.y);
		tmp.z = linear_to_srgb(tmp.z);
	}
	return tmp;
}

template <bool PROBE, bool PLAIN = false, bool NORMALS = false>
NGP_DEV bool shade_ray(const FrameParams& F, const ProbeParams& P, f3 bg_linear, uint32_t idx, const Accum& acc, uint32_t n_steps, f3 dir) {
	if (!(acc.a > 0.001f)) return false;
	if (!PROBE && F.depth_test && acc.depth > F.depth_buffer[idx]) return true;
	float r = acc.r, g = acc.g, b = acc.b, a = acc.a;
	if (NORMALS) { // ERenderMode::Normals (:1379-1381): the composited normal, normalised again, as a premultiplied colour
		const f3 n = normalize3(mk3(r, g, b));
		r = (0.5f * n.x + 0.5f) * a;
		g = (0.5f * n.y + 0.5f) * a;
		b = (0.5f * n.z + 0.5f) * a;
	}
	if (!PROBE && F.render_mode == 5) { // ERenderMode::Cost: the ray's sample count as a grey level, opaque (:1382-1384)
		r = g = b = (float)n_steps / 128.0f;
		a = 1.0f;
	}
	if (!F.linear_colors && (PROBE || F.render_mode == 0)) { // only ERenderMode::Shade accumulates in linear colours (:1393) -- ShadeEnvMap / ShadeGridEnvMap, the fork's additions, do not
		r = srgb_to_linear(r);
		g = srgb_to_linear(g);
		b = srgb_to_linear(b);
	}
	if (PROBE) {
		P.ray_rgba[idx] = make_float4(r, g, b, a);
		return true;
	}
	if (F.direct) { // the frame buffer would hold zeros (tmp + 0 * (1 - a) == tmp) or the environment map's value for this ray
		const bool deep = a > 0.2f;
		if (!PLAIN && F.envmap) {
			float d3[3] = {dir.x, dir.y, dir.z}, e[4];
			read_envmap(F.envmap, F.env_w, F.env_h, d3, e);
			const float k = 1.0f - a;
			r = r + e[0] * k; g = g + e[1] * k; b = b + e[2] * k; a = a + e[3] * k;
		}
		F.frame_buffer[idx] = tonemap_pixel(F, bg_linear, r, g, b, a);
		if (deep) F.depth_buffer[idx] = acc.depth;
		return true;
	}
	float4 fb = F.frame_buffer[idx];
	float k = 1.0f - a;
	fb.x = r + fb.x * k;
	fb.y = g + fb.y * k;
	fb.z = b + fb.z * k;
	fb.w = a + fb.w * k;
	F.frame_buffer[idx] = fb;
	if (a > 0.2f) F.depth_buffer[idx] = acc.depth;
	return true;
}

// K10 / K11 / K12: init_rays_from_{center, center_outward, multiple_center}_with_payload_kernel_nerf
// (src/testbed_nerf.cu:1559-1773) for probe ray q (= the reference's payload index `mulidx`)
NGP_DEV float halton(uint32_t base, uint32_t idx) { // random_val.cuh:338-350
	float f = 1, result = 0;
	while (idx > 0) {
		f /= (float)base;
		result += f * (float)(idx % base);
		idx /= base;
	}
	return result;
}
NGP_DEV f3 cylindrical_to_dir_nerf(float px, float py) { // src/testbed_nerf.cu:1546-1557
	const float cos_theta = -2.0f * px + 1.0f;
	const float phi = 2.0f * 3.14159265358979323846f * (py - 0.5f);
	const float sin_theta = __builtin_sqrtf(fmaxf(1.0f - cos_theta * cos_theta, 0.0f));
	return mk3(sin_theta * cosf(phi), sin_theta * sinf(phi), cos_theta);
}
// compute_local_frame (random_val.cuh:167-186), column-major: columns (localX, localY, localZ = n)
NGP_DEV void local_frame(f3 n, float* frame) {
	float sz = (n.z >= 0) ? 1.0f : -1.0f;
	float a = 1 / (sz + n.z);
	float ya = n.y * a;
	float b = n.x * ya;
	float c = n.x * sz;
	frame[0] = c * n.x * a - 1; frame[1] = sz * b; frame[2] = c;
	frame[3] = b; frame[4] = n.y * ya - sz; frame[5] = n.y;
	frame[6] = n.x; frame[7] = n.y; frame[8] = n.z;
}
// shell position of probe g of the grid (Testbed::computeEnvmapGrid; definition: include/ngp_hip.h, ngp_compute_envmap_grid)
NGP_DEV f3 probe_grid_origin(const float* center, uint32_t grid_x, uint32_t grid_y, float shell_radius, uint32_t g) {
	const uint32_t i = g % grid_x, j = g / grid_x;
	f3 dir = cylindrical_to_dir_nerf(((float)i + 0.5f) / (float)grid_x, ((float)j + 0.5f) / (float)grid_y);
	return add3(mk3(center[0], center[1], center[2]), scale3(dir, shell_radius));
}
NGP_DEV void init_probe_ray(const ProbeParams& P, uint32_t q, RayState& r) {
	const uint32_t no = P.mode == 2 ? P.n_origin : 1u;
	const uint32_t w = P.n_theta * no;
	const uint32_t per_probe = P.n_theta * P.n_phi * no * no;
	const uint32_t g = P.mode == 3 ? q / per_probe : 0u, ql = P.mode == 3 ? q % per_probe : q;
	uint32_t tm = ql % w, pm = ql / w;
	uint32_t theta_mul = tm / no, theta_rem = tm % no, phi_mul = pm / no, phi_rem = pm % no;
	f3 local = cylindrical_to_dir_nerf((float)theta_mul / (float)P.n_theta, (float)phi_mul / (float)P.n_phi);
	f3 origin = mk3(P.center[0], P.center[1], P.center[2]);
	f3 dir = local;
	const bool outward = P.mode == 1 || P.mode == 3;
	if (outward) {
		origin = P.mode == 3 ? probe_grid_origin(P.center, P.grid_x, P.grid_y, P.shell_radius, g) : mk3(P.origin[0], P.origin[1], P.origin[2]);
		float frame[9];
		local_frame(normalize3(origin), frame);
		dir = m3_mulv(frame, local);
	} else if (P.mode == 2) {
		uint32_t hi = theta_rem * no + phi_rem;
		origin = add3(origin, mk3(halton(2, hi) - 0.5f, halton(3, hi) - 0.5f, halton(5, hi) - 0.5f));
	}
	dir = normalize3(dir);
	if (outward) dir = scale3(dir, -1.0f);
	r.o = origin;
	r.d = dir;
	r.t = 0.0f;
	r.idx = q;
	r.out = q;
	r.alive = true;
}

// s_memtime stamp for the diagnostic section profile (cdna_hip_programming.md "In-kernel stamps"); never executed by
// the production instantiation
// chip-wide 100 MHz counter (the same on every CU, unlike s_memtime): wave timelines of the diagnostic build
NGP_DEV unsigned long long realtime() {
	unsigned long long t;
	asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
	return t;
}
NGP_DEV unsigned long long stamp() {
	unsigned long long t;
	__builtin_amdgcn_sched_barrier(0);
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
	__builtin_amdgcn_sched_barrier(0);
	return t;
}

// ---- counters (one atomic per wave and counter). The last wave to leave moves the launch's totals to the slot the host
// reads and hands the accumulators, the tile queue and the exit count back as zeros: the slot's next launch needs no
// memset (a dependent dispatch per frame behind a persistent kernel). Everything goes through device-scope atomics,
// which execute at the memory side -- no cache holds a stale copy; the release orders this wave's adds before its exit.
NGP_DEV void finish_launch(const FrameParams& F, int lane, uint32_t n_alive_init, uint32_t n_hit, uint32_t n_samples) {
	if (lane == 0) {
		atomicAdd(&F.counters[0], (unsigned long long)n_alive_init);
		atomicAdd(&F.counters[1], (unsigned long long)n_hit);
		atomicAdd(&F.counters[2], (unsigned long long)n_samples);
		const uint32_t left = __hip_atomic_fetch_add(F.done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
		if (left + 1u == F.n_waves) {
#pragma unroll
			for (int k = 0; k < 3; ++k) {
				const unsigned long long total = atomicExch(&F.counters[k], 0ull);
				F.results[k] = F.add_results ? F.results[k] + total : total;
			}
			// the launch on the chip's 100 MHz clock: first wave in (min over the waves' start stamps) to last wave out
			const unsigned long long t_start = ~atomicExch(&F.results[4], 0ull), ticks = realtime() - t_start;
			F.results[3] = F.add_results ? F.results[3] + ticks : ticks;
			atomicExch(F.queue, 0u);
			if (F.xqueue) {
#pragma unroll
				for (int k = 0; k < 8; ++k) atomicExch(F.xqueue + k, 0u);
			}
			atomicExch(F.done, 0u);
		}
	}
}

} // namespace ngp
