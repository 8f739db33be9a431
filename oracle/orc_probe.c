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

void orc_irradiance_from(uint32_t n_theta, uint32_t n_phi, const float* envmap, const float* origin3, uint32_t n, const float* normals, float* out_rgb) {
	const double d_omega = 4.0 * 3.14159265358979323846 / ((double)n_theta * (double)n_phi);
	float frame[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
	if (origin3) compute_local_frame(v3_normalize(v3_make(origin3[0], origin3[1], origin3[2])), frame);
#pragma omp parallel for schedule(static)
	for (int64_t q = 0; q < (int64_t)n; ++q) {
		double acc[3] = {0, 0, 0};
		v3 nrm = v3_make(normals[3 * q], normals[3 * q + 1], normals[3 * q + 2]);
		for (uint32_t j = 0; j < n_phi; ++j) {
			for (uint32_t i = 0; i < n_theta; ++i) {
				v3 w = cylindrical_to_dir_nerf((float)i / (float)n_theta, (float)j / (float)n_phi);
				if (origin3) w = v3_scale(v3_normalize(m3_mulv(frame, w)), -1.0f); /* the K11 ray of this texel */
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
void orc_irradiance(uint32_t n_theta, uint32_t n_phi, const float* envmap, uint32_t n, const float* normals, float* out_rgb) {
	orc_irradiance_from(n_theta, n_phi, envmap, NULL, n, normals, out_rgb);
}

/* ---------------------------------------------------------------------------------------------- the grid of probes */
void orc_probe_grid_origin(const orc_probe_grid_desc* d, uint32_t g, float* out3) {
	v3 center = v3_make(d->center[0], d->center[1], d->center[2]);
	const uint32_t i = g % d->grid_x, j = g / d->grid_x;
	v3 dir = cylindrical_to_dir_nerf(((float)i + 0.5f) / (float)d->grid_x, ((float)j + 0.5f) / (float)d->grid_y);
	v3 p = v3_add(center, v3_scale(dir, d->shell_radius));
	out3[0] = p.x; out3[1] = p.y; out3[2] = p.z;
}

void orc_compute_envmap_grid(const orc_nerf_model* m, const orc_probe_grid_desc* d, const orc_render_opts* o, float* envmaps, orc_render_stats* stats) {
	const uint32_t n_probes = d->grid_x * d->grid_y;
	const size_t texels = (size_t)d->n_theta * d->n_phi;
	orc_render_stats total = {0, 0, 0, 0};
	for (uint32_t g = 0; g < n_probes; ++g) {
		orc_probe_desc pd;
		pd.mode = ORC_PROBE_CENTER_OUTWARD;
		pd.n_theta = d->n_theta; pd.n_phi = d->n_phi; pd.n_origin = 1;
		orc_probe_grid_origin(d, g, pd.origin);
		orc_render_stats st;
		orc_compute_envmap(m, &pd, o, envmaps + 4 * texels * g, &st);
		total.n_rays += st.n_rays; total.n_rays_alive_after_init += st.n_rays_alive_after_init;
		total.n_rays_hit += st.n_rays_hit; total.n_samples += st.n_samples;
	}
	if (stats) *stats = total;
}

void orc_irradiance_grid_tabulate(const orc_probe_grid_desc* d, const float* envmaps, float* tables) {
	const uint32_t n_probes = d->grid_x * d->grid_y, texels = d->n_theta * d->n_phi;
	float* normals = (float*)malloc(sizeof(float) * 3 * texels);
	float* rgb = (float*)malloc(sizeof(float) * 3 * texels);
	for (uint32_t t = 0; t < texels; ++t) orc_texel_direction(d->n_theta, d->n_phi, t % d->n_theta, t / d->n_theta, normals + 3 * t);
	for (uint32_t g = 0; g < n_probes; ++g) {
		float origin[3];
		orc_probe_grid_origin(d, g, origin);
		orc_irradiance_from(d->n_theta, d->n_phi, envmaps + 4 * (size_t)texels * g, origin, texels, normals, rgb);
		float* out = tables + 4 * (size_t)texels * g;
		for (uint32_t t = 0; t < texels; ++t) { out[4 * t] = rgb[3 * t]; out[4 * t + 1] = rgb[3 * t + 1]; out[4 * t + 2] = rgb[3 * t + 2]; out[4 * t + 3] = 0.f; }
	}
	free(normals); free(rgb);
}

/* inverse of cylindrical_to_dir_nerf: px = (1 - z) / 2, py = atan2(y, x) / (2 pi) + 0.5 */
static void dir_to_cylindrical(v3 n, float* px, float* py) {
	*px = (1.0f - n.z) * 0.5f;
	*py = atan2f(n.y, n.x) / (2.0f * PI_F) + 0.5f;
}
/* cell + weight of a coordinate on an axis of `n` samples sitting at (k + offset) / n: clamped (theta) or periodic (phi) */
static void axis_cell(float coord01, uint32_t n, float offset, int periodic, uint32_t* k0, uint32_t* k1, float* w) {
	float f = coord01 * (float)n - offset;
	float fl = floorf(f);
	int i0 = (int)fl, i1 = i0 + 1;
	*w = f - fl;
	if (periodic) {
		i0 = ((i0 % (int)n) + (int)n) % (int)n;
		i1 = ((i1 % (int)n) + (int)n) % (int)n;
	} else {
		if (i0 < 0) { i0 = 0; *w = 0.0f; }
		if (i1 > (int)n - 1) i1 = (int)n - 1;
		if (i0 > (int)n - 1) i0 = (int)n - 1;
	}
	*k0 = (uint32_t)i0; *k1 = (uint32_t)i1;
}

void orc_irradiance_read(uint32_t n_theta, uint32_t n_phi, const float* table, const float* n3, float* out_rgb) {
	float px, py, wa, wb;
	uint32_t a0, a1, b0, b1;
	dir_to_cylindrical(v3_make(n3[0], n3[1], n3[2]), &px, &py);
	axis_cell(px, n_theta, 0.0f, 0, &a0, &a1, &wa); /* texel a sits at a / n_theta */
	axis_cell(py, n_phi, 0.0f, 1, &b0, &b1, &wb);
	const float* t00 = table + 4 * ((size_t)a0 + (size_t)n_theta * b0);
	const float* t10 = table + 4 * ((size_t)a1 + (size_t)n_theta * b0);
	const float* t01 = table + 4 * ((size_t)a0 + (size_t)n_theta * b1);
	const float* t11 = table + 4 * ((size_t)a1 + (size_t)n_theta * b1);
	for (int k = 0; k < 3; ++k)
		out_rgb[k] = (((1.0f - wa) * (1.0f - wb)) * t00[k] + (wa * (1.0f - wb)) * t10[k]) + (((1.0f - wa) * wb) * t01[k] + (wa * wb) * t11[k]);
}

void orc_irradiance_grid_lookup(const orc_probe_grid_desc* d, const float* tables, uint32_t n, const float* positions, const float* normals, float* out_rgb) {
	const size_t texels = (size_t)d->n_theta * d->n_phi;
	for (uint32_t q = 0; q < n; ++q) {
		v3 rel = v3_sub(v3_make(positions[3 * q], positions[3 * q + 1], positions[3 * q + 2]), v3_make(d->center[0], d->center[1], d->center[2]));
		float len = v3_length(rel);
		v3 dir = len > 0.0f ? v3_divs(rel, len) : v3_make(0.f, 0.f, 1.f);
		float px, py, wi, wj;
		uint32_t i0, i1, j0, j1;
		dir_to_cylindrical(dir, &px, &py);
		axis_cell(px, d->grid_x, 0.5f, 0, &i0, &i1, &wi); /* probe (i, j) sits at ((i + 0.5) / grid_x, (j + 0.5) / grid_y) */
		axis_cell(py, d->grid_y, 0.5f, 1, &j0, &j1, &wj);
		float e00[3], e10[3], e01[3], e11[3];
		orc_irradiance_read(d->n_theta, d->n_phi, tables + 4 * texels * (i0 + (size_t)d->grid_x * j0), normals + 3 * q, e00);
		orc_irradiance_read(d->n_theta, d->n_phi, tables + 4 * texels * (i1 + (size_t)d->grid_x * j0), normals + 3 * q, e10);
		orc_irradiance_read(d->n_theta, d->n_phi, tables + 4 * texels * (i0 + (size_t)d->grid_x * j1), normals + 3 * q, e01);
		orc_irradiance_read(d->n_theta, d->n_phi, tables + 4 * texels * (i1 + (size_t)d->grid_x * j1), normals + 3 * q, e11);
		for (int k = 0; k < 3; ++k)
			out_rgb[3 * q + k] = (((1.0f - wi) * (1.0f - wj)) * e00[k] + (wi * (1.0f - wj)) * e10[k]) + (((1.0f - wi) * wj) * e01[k] + (wi * wj) * e11[k]);
	}
}
