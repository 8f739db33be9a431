/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_common.h). PARITY UNPINNED.
 *
 * Irradiance probes. The reference declares computeEnvmap / computeEnvmapMultiple /
 * computeEnvmapGrid (testbed.h:709-743) but ships no body; what it does ship are the ray generators
 * (src/testbed_nerf.cu:1546-1773), the capped tracer they feed (trace_mesh :2146-2262) and the helpers
 * (random_val.cuh:45-72,130-192,338-354). Those are restated here; the texture / irradiance definitions follow the
 * SURVEY: texel (i,j) = mean over its rays of the shaded RGBA, and
 *     E(n) = sum_texels L(w_ij) * max(0, n.w_ij) * dOmega,   dOmega = 4 pi / (n_theta n_phi)
 * (cos(theta)-uniform x phi-uniform parameterisation of cylindrical_to_dir_nerf is equal-area).
 */
#include "oracle.h"
#include "orc_common.h"
#include "orc_probe.h"

static const float PI_F = 3.14159265358979323846f;

static float halton(uint32_t base, size_t idx) { /* random_val.cuh:338-350 */
	float f = 1, result = 0;
	while (idx > 0) {
		f /= (float)base;
		result += f * (float)(idx % base);
		idx /= base;
	}
	return result;
}

static v3 cylindrical_to_dir_nerf(float px, float py) { /* src/testbed_nerf.cu:1546-1557 */
	const float cos_theta = -2.0f * px + 1.0f;
	const float phi = 2.0f * PI_F * (py - 0.5f);
	const float sin_theta = sqrtf(fmaxf(1.0f - cos_theta * cos_theta, 0.0f));
	float sin_phi = sinf(phi), cos_phi = cosf(phi);
	return v3_make(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta);
}

/* compute_local_frame, random_val.cuh:167-186: columns (localX, localY, localZ) */
static void compute_local_frame(v3 z_, float* m9) {
	float x = z_.x, y = z_.y, z = z_.z;
	float sz = (z >= 0) ? 1.0f : -1.0f;
	float a = 1 / (sz + z);
	float ya = y * a;
	float b = x * ya;
	float c = x * sz;
	m9[0] = c * x * a - 1; m9[1] = sz * b; m9[2] = c;
	m9[3] = b; m9[4] = y * ya - sz; m9[5] = y;
	m9[6] = x; m9[7] = y; m9[8] = z;
}

uint32_t orc_probe_n_rays(const orc_probe_desc* d) {
	uint32_t no = d->mode == ORC_PROBE_MULTI_CENTER ? d->n_origin * d->n_origin : 1u;
	return d->n_theta * d->n_phi * no;
}

/* K10 / K11 / K12: init_rays_from_{center, center_outward, multiple_center}_with_payload_kernel_nerf (:1559-1773) */
void orc_probe_payloads(const orc_nerf_model* m, const orc_probe_desc* d, orc_payload* out) {
	v3 center = v3_scale(v3_add(v3_make(m->render_aabb_max[0], m->render_aabb_max[1], m->render_aabb_max[2]),
	                            v3_make(m->render_aabb_min[0], m->render_aabb_min[1], m->render_aabb_min[2])), 0.5f);
	const uint32_t no = d->mode == ORC_PROBE_MULTI_CENTER ? d->n_origin : 1u;
	const uint32_t w = d->n_theta * no, h = d->n_phi * no;
	for (uint32_t pm = 0; pm < h; ++pm) {
		for (uint32_t tm = 0; tm < w; ++tm) {
			uint32_t theta_mul = tm / no, theta_rem = tm % no, phi_mul = pm / no, phi_rem = pm % no;
			uint32_t mulidx = tm + d->n_theta * no * pm;
			uint32_t idx = theta_mul + d->n_theta * phi_mul;
			float cos_theta = (float)theta_mul / (float)d->n_theta;
			float phi = (float)phi_mul / (float)d->n_phi;
			v3 local = cylindrical_to_dir_nerf(cos_theta, phi);
			v3 origin = center, dir = local;
			if (d->mode == ORC_PROBE_CENTER_OUTWARD) {
				origin = v3_make(d->origin[0], d->origin[1], d->origin[2]);
				float frame[9];
				compute_local_frame(v3_normalize(origin), frame);
				dir = m3_mulv(frame, local);
			} else if (d->mode == ORC_PROBE_MULTI_CENTER) {
				uint32_t hi = theta_rem * no + phi_rem;
				origin = v3_add(origin, v3_make(halton(2, hi) - 0.5f, halton(3, hi) - 0.5f, halton(5, hi) - 0.5f));
			}
			dir = v3_normalize(dir);
			if (d->mode == ORC_PROBE_CENTER_OUTWARD) dir = v3_scale(dir, -1.0f);
			orc_payload* p = &out[mulidx];
			memset(p, 0, sizeof(*p));
			p->origin[0] = origin.x; p->origin[1] = origin.y; p->origin[2] = origin.z;
			p->dir[0] = dir.x; p->dir[1] = dir.y; p->dir[2] = dir.z;
			p->t = 0.0f;
			p->max_weight = 0.0f;
			p->idx = idx;
			p->n_steps = 0;
			p->alive = 1;
		}
	}
}

void orc_compute_envmap(const orc_nerf_model* m, const orc_probe_desc* d, const orc_render_opts* o, float* envmap, orc_render_stats* stats) {
	const uint32_t n = orc_probe_n_rays(d);
	const uint32_t no = d->mode == ORC_PROBE_MULTI_CENTER ? d->n_origin : 1u;
	orc_payload* pl = (orc_payload*)malloc(sizeof(orc_payload) * n);
	float* rgba = (float*)calloc((size_t)n * 4, sizeof(float));
	float* depth = (float*)calloc(n, sizeof(float));
	orc_probe_payloads(m, d, pl);
	orc_render_opts ro = *o;
	ro.capped_skip = 1; /* trace_mesh uses generate_next_nerf_network_inputs_geometry (:2208) */
	float cam[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0}; /* depth is not used by the probe */
	orc_trace_payloads(m, cam, &ro, n, pl, rgba, depth, stats);
	memset(envmap, 0, sizeof(float) * 4 * d->n_theta * d->n_phi);
	/* shade (sRGB -> linear unless trained in linear colours; alpha > 0.001 filter), mean over the rays of a texel
	 * in increasing ray index */
	for (uint32_t i = 0; i < n; ++i) {
		if (pl[i].alive || !(rgba[4 * i + 3] > 0.001f)) continue;
		float c[4] = {rgba[4 * i], rgba[4 * i + 1], rgba[4 * i + 2], rgba[4 * i + 3]};
		if (!o->train_in_linear_colors) { c[0] = orc_srgb_to_linear(c[0]); c[1] = orc_srgb_to_linear(c[1]); c[2] = orc_srgb_to_linear(c[2]); }
		for (int k = 0; k < 4; ++k) envmap[4 * pl[i].idx + k] += c[k];
	}
	float inv = 1.0f / (float)(no * no);
	for (uint32_t i = 0; i < 4 * d->n_theta * d->n_phi; ++i) envmap[i] *= inv;
	free(pl); free(rgba); free(depth);
}

void orc_texel_direction(uint32_t n_theta, uint32_t n_phi, uint32_t i, uint32_t j, float* out3) {
	v3 d = cylindrical_to_dir_nerf((float)i / (float)n_theta, (float)j / (float)n_phi);
	out3[0] = d.x; out3[1] = d.y; out3[2] = d.z;
}

void orc_irradiance(uint32_t n_theta, uint32_t n_phi, const float* envmap, uint32_t n, const float* normals, float* out_rgb) {
	const double d_omega = 4.0 * 3.14159265358979323846 / ((double)n_theta * (double)n_phi);
#pragma omp parallel for schedule(static)
	for (int64_t q = 0; q < (int64_t)n; ++q) {
		double acc[3] = {0, 0, 0};
		v3 nrm = v3_make(normals[3 * q], normals[3 * q + 1], normals[3 * q + 2]);
		for (uint32_t j = 0; j < n_phi; ++j) {
			for (uint32_t i = 0; i < n_theta; ++i) {
				v3 w = cylindrical_to_dir_nerf((float)i / (float)n_theta, (float)j / (float)n_phi);
				float c = v3_dot(nrm, w);
				if (!(c > 0.0f)) continue;
				const float* L = envmap + 4 * ((size_t)i + (size_t)n_theta * j);
				acc[0] += (double)(L[0] * c); acc[1] += (double)(L[1] * c); acc[2] += (double)(L[2] * c);
			}
		}
		out_rgb[3 * q] = (float)(acc[0] * d_omega);
		out_rgb[3 * q + 1] = (float)(acc[1] * d_omega);
		out_rgb[3 * q + 2] = (float)(acc[2] * d_omega);
	}
}
