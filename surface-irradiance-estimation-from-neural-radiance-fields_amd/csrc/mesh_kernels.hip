// Geometry mode, mesh side, for MI355X (gfx950): one fused kernel per frame does what the reference spreads over
//   init_rays_with_payload_kernel_mesh_geometry -> mesh_raytrace_kernel -> prepare_shadow_rays_geometry ->
//   mesh_raytrace_kernel -> write_shadow_ray_result_geometry -> shade_kernel_mesh_geometry
// (src/testbed_geometry_training.cu:2202-2320, src/geometry_bvh.cu:646-676) with seven ray-state arrays in DRAM.
// Primary hit, sun shadow ray and BRDF stay in registers; the only traffic is BVH nodes / triangles (cache resident)
// and one frame/depth write per pixel. Software BVH4 traversal replaces OptiX (north_star: no OptiX).
#include "nerf_device.h"

namespace ngp {

constexpr float MAX_DIST = 100.0f; // geometry_bvh.cu:23

NGP_DEV f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
NGP_DEV f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
NGP_DEV bool box_contains(const float* bmin, const float* bmax, f3 p) {
	return p.x >= bmin[0] && p.x <= bmax[0] && p.y >= bmin[1] && p.y <= bmax[1] && p.z >= bmin[2] && p.z <= bmax[2];
}

// Triangle::ray_intersect, triangle.cuh:44-58
NGP_DEV float tri_ray_intersect(const Triangle& tri, f3 ro, f3 rd) {
	f3 a = ld3(tri.a);
	f3 v1v0 = sub3(ld3(tri.b), a);
	f3 v2v0 = sub3(ld3(tri.c), a);
	f3 rov0 = sub3(ro, a);
	f3 n = cross3(v1v0, v2v0);
	f3 q = cross3(rov0, rd);
	float d = 1.0f / dot3(rd, n);
	float u = d * -dot3(q, v2v0);
	float v = d * dot3(q, v1v0);
	float t = d * -dot3(n, rov0);
	if (u < 0.0f || u > 1.0f || v < 0.0f || (u + v) > 1.0f || t < 0.0f) t = 3.402823466e+38f;
	return t;
}

struct DistIdx {
	float dist;
	uint32_t idx;
};
NGP_DEV void cas(DistIdx& a, DistIdx& b) { // compare_and_swap with "<": sorts descending, triangle_bvh.cuh:161-176
	if (a.dist < b.dist) {
		DistIdx t = a;
		a = b;
		b = t;
	}
}

// GeometryBvh4::ray_intersect_triangle (geometry_bvh.cu:61-109) == TriangleBvh4::ray_intersect (triangle_bvh.cu:150-193)
NGP_DEV void bvh4_ray_intersect(const TriangleBvhNode* __restrict__ nodes, const Triangle* __restrict__ tris, f3 ro, f3 rd, int& out_idx, float& out_t) {
	int stack[32];
	int sp = 0;
	stack[sp++] = 0;
	float mint = MAX_DIST;
	int shortest = -1;
	while (sp > 0) {
		int idx = stack[--sp];
		const int left = nodes[idx].left_idx, right = nodes[idx].right_idx;
		if (left < 0) {
			int end = -right - 1;
			for (int i = -left - 1; i < end; ++i) {
				float t = tri_ray_intersect(tris[i], ro, rd);
				if (t < mint) {
					mint = t;
					shortest = i;
				}
			}
		} else {
			DistIdx ch[4];
#pragma unroll
			for (uint32_t i = 0; i < 4; ++i) {
				const TriangleBvhNode& c = nodes[left + (int)i];
				ch[i].dist = aabb_ray_entry(c.bmin, c.bmax, ro, rd);
				ch[i].idx = (uint32_t)left + i;
			}
			cas(ch[0], ch[2]); cas(ch[1], ch[3]); cas(ch[0], ch[1]); cas(ch[2], ch[3]); cas(ch[1], ch[2]);
#pragma unroll
			for (uint32_t i = 0; i < 4; ++i) {
				if (ch[i].dist < mint && sp < 32) stack[sp++] = (int)ch[i].idx;
			}
		}
	}
	out_idx = shortest;
	out_t = mint;
}

// mesh_raytrace_kernel body (geometry_bvh.cu:646-676) + GeometryBvh4::ray_intersect leaf scan (:166-200)
NGP_DEV void trace_mesh(const MeshSceneParams& S, f3& pos, f3& dir) {
	float mint = MAX_DIST;
	int mesh_idx = -1;
	for (uint32_t m = 0; m < S.n_meshes; ++m) {
		float t = aabb_ray_entry(S.meshes[m].bmin, S.meshes[m].bmax, pos, dir);
		if (t < mint && t > -3.402823466e+38f) {
			mint = t;
			mesh_idx = (int)m;
		}
	}
	if (mesh_idx < 0) return;
	const MeshRef& M = S.meshes[mesh_idx];
	int idx;
	float t;
	bvh4_ray_intersect(M.nodes, M.tris, pos, dir, idx, t);
	pos = add3(pos, scale3(dir, t));
	if (idx > -1) {
		const Triangle& tri = M.tris[idx];
		f3 a = ld3(tri.a);
		dir = normalize3(cross3(sub3(ld3(tri.b), a), sub3(ld3(tri.c), a)));
	}
}

// ---- BRDF: testbed_geometry_training.cu:46-144 (double-typed literals are kept, they promote like the reference)
NGP_DEV float square(float x) { return x * x; }
NGP_DEV float mixf(float a, float b, float t) { return a + (b - a) * t; }
NGP_DEV f3 mix3(f3 a, f3 b, float t) { return add3(a, scale3(sub3(b, a), t)); }
NGP_DEV float saturate(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }
NGP_DEV float SchlickFresnel(float u) {
	float m = saturate((float)(1.0 - u));
	return square(square(m)) * m;
}
constexpr float PI_F = 3.14159265358979323846f;
NGP_DEV float G1(float NdotH, float a) {
	if (a >= 1.0) return (float)(1.0 / PI_F);
	float a2 = square(a);
	float t = (float)(1.0 + (a2 - 1.0) * NdotH * NdotH);
	return (float)((a2 - 1.0) / (PI_F * logf(a2) * t));
}
NGP_DEV float G2(float NdotH, float a) {
	float a2 = square(a);
	float t = (float)(1.0 + (a2 - 1.0) * NdotH * NdotH);
	return a2 / (PI_F * t * t);
}
NGP_DEV float SmithG_GGX(float NdotV, float alphaG) {
	float a = alphaG * alphaG;
	float b = NdotV * NdotV;
	return (float)(1.0 / (NdotV + __builtin_sqrtf(a + b - a * b)));
}
NGP_DEV f3 evaluate_shading(f3 base_color, f3 ambient_color, f3 light_color, float metallic, float subsurface, float specular, float roughness,
                            float specular_tint, float sheen, float sheen_tint, float clearcoat, float clearcoat_gloss, f3 L, f3 V, f3 N) {
	float NdotL = dot3(N, L);
	float NdotV = dot3(N, V);
	f3 H = normalize3(add3(L, V));
	float NdotH = dot3(N, H);
	float LdotH = dot3(L, H);
	float FL = SchlickFresnel(NdotL), FV = SchlickFresnel(NdotV);
	f3 amb = scale3(ambient_color, mixf(0.2f, FV, metallic));
	amb = mul3(amb, base_color);
	if (NdotL < 0.f || NdotV < 0.f) return amb;
	float luminance = dot3(base_color, mk3(0.3f, 0.6f, 0.1f));
	f3 Ctint = scale3(base_color, 1.f / (luminance + 0.00001f));
	const f3 one = mk3(1.0f, 1.0f, 1.0f);
	f3 Cspec0 = mix3(scale3(scale3(mix3(one, Ctint, specular_tint), specular), 0.08f), base_color, metallic);
	f3 Csheen = mix3(one, Ctint, sheen_tint);
	float Fd90 = 0.5f + 2.0f * LdotH * LdotH * roughness;
	float Fd = mixf(1, Fd90, FL) * mixf(1.f, Fd90, FV);
	float Fss90 = LdotH * LdotH * roughness;
	float Fss = mixf(1.0f, Fss90, FL) * mixf(1.0f, Fss90, FV);
	float ss = 1.25f * (Fss * (1.f / (NdotL + NdotV) - 0.5f) + 0.5f);
	float a = fmaxf(0.001f, square(roughness));
	float Ds = G2(NdotH, a);
	float FH = SchlickFresnel(LdotH);
	f3 Fs = mix3(Cspec0, one, FH);
	float Gs = SmithG_GGX(NdotL, a) * SmithG_GGX(NdotV, a);
	f3 Fsheen = scale3(Csheen, FH * sheen);
	float Dr = G1(NdotH, mixf(0.1f, 0.001f, clearcoat_gloss));
	float Fr = mixf(0.04f, 1.0f, FH);
	float Gr = SmithG_GGX(NdotL, 0.25f) * SmithG_GGX(NdotV, 0.25f);
	float CCs = 0.25f * clearcoat * Gr * Fr * Dr;
	f3 diffuse = add3(scale3(base_color, (float)(1.0f / PI_F) * mixf(Fd, ss, subsurface)), Fsheen);
	f3 brdf = add3(add3(scale3(diffuse, 1.0f - metallic), scale3(Fs, Gs * Ds)), mk3(CCs, CCs, CCs));
	return add3(scale3(mul3(brdf, light_color), NdotL), amb);
}

// ---- NeRF-derived irradiance (definitions: include/ngp_hip.h, ngp_compute_envmap_grid). The lookups keep one fixed
// expression order.
// cell + weight of a coordinate on an axis of n samples sitting at (k + offset) / n: clamped (theta) or periodic (phi)
NGP_DEV void axis_cell(float coord01, uint32_t n, float offset, bool periodic, uint32_t& k0, uint32_t& k1, float& w) {
	float f = coord01 * (float)n - offset;
	float fl = __builtin_floorf(f);
	int i0 = (int)fl, i1 = i0 + 1;
	w = f - fl;
	if (periodic) {
		i0 = ((i0 % (int)n) + (int)n) % (int)n;
		i1 = ((i1 % (int)n) + (int)n) % (int)n;
	} else {
		if (i0 < 0) { i0 = 0; w = 0.0f; }
		if (i1 > (int)n - 1) i1 = (int)n - 1;
		if (i0 > (int)n - 1) i0 = (int)n - 1;
	}
	k0 = (uint32_t)i0; k1 = (uint32_t)i1;
}
// inverse of cylindrical_to_dir_nerf (src/testbed_nerf.cu:1546-1557): px = (1 - z) / 2, py = atan2(y, x) / (2 pi) + 0.5
NGP_DEV void dir_to_cylindrical(f3 n, float& px, float& py) {
	px = (1.0f - n.z) * 0.5f;
	py = atan2f(n.y, n.x) / (2.0f * PI_F) + 0.5f;
}
// bilinear read of one tabulated irradiance map at direction n, in the manner of read_envmap (envmap.cuh:24-50): the
// four texels around the direction, theta clamped, phi periodic
NGP_DEV f3 irradiance_read(const float4* __restrict__ table, uint32_t n_theta, uint32_t n_phi, f3 n) {
	float px, py, wa, wb;
	uint32_t a0, a1, b0, b1;
	dir_to_cylindrical(n, px, py);
	axis_cell(px, n_theta, 0.0f, false, a0, a1, wa);
	axis_cell(py, n_phi, 0.0f, true, b0, b1, wb);
	const float4 t00 = table[(size_t)a0 + (size_t)n_theta * b0], t10 = table[(size_t)a1 + (size_t)n_theta * b0];
	const float4 t01 = table[(size_t)a0 + (size_t)n_theta * b1], t11 = table[(size_t)a1 + (size_t)n_theta * b1];
	const float w00 = (1.0f - wa) * (1.0f - wb), w10 = wa * (1.0f - wb), w01 = (1.0f - wa) * wb, w11 = wa * wb;
	return mk3((w00 * t00.x + w10 * t10.x) + (w01 * t01.x + w11 * t11.x), (w00 * t00.y + w10 * t10.y) + (w01 * t01.y + w11 * t11.y),
	           (w00 * t00.z + w10 * t10.z) + (w01 * t01.z + w11 * t11.z));
}
// one probe (ShadeEnvMap), or the four probes of the grid around the direction of pos - center (ShadeGridEnvMap)
NGP_DEV f3 irradiance_lookup(const IrradianceMap& I, f3 pos, f3 N) {
	if (I.grid_x == 0u) return irradiance_read(I.irradiance, I.n_theta, I.n_phi, N);
	const size_t texels = (size_t)I.n_theta * I.n_phi;
	f3 rel = sub3(pos, mk3(I.center[0], I.center[1], I.center[2]));
	float len = __builtin_sqrtf(dot3(rel, rel));
	f3 dir = len > 0.0f ? mk3(rel.x / len, rel.y / len, rel.z / len) : mk3(0.f, 0.f, 1.f);
	float px, py, wi, wj;
	uint32_t i0, i1, j0, j1;
	dir_to_cylindrical(dir, px, py);
	axis_cell(px, I.grid_x, 0.5f, false, i0, i1, wi);
	axis_cell(py, I.grid_y, 0.5f, true, j0, j1, wj);
	const f3 e00 = irradiance_read(I.irradiance + texels * (i0 + (size_t)I.grid_x * j0), I.n_theta, I.n_phi, N);
	const f3 e10 = irradiance_read(I.irradiance + texels * (i1 + (size_t)I.grid_x * j0), I.n_theta, I.n_phi, N);
	const f3 e01 = irradiance_read(I.irradiance + texels * (i0 + (size_t)I.grid_x * j1), I.n_theta, I.n_phi, N);
	const f3 e11 = irradiance_read(I.irradiance + texels * (i1 + (size_t)I.grid_x * j1), I.n_theta, I.n_phi, N);
	const float w00 = (1.0f - wi) * (1.0f - wj), w10 = wi * (1.0f - wj), w01 = (1.0f - wi) * wj, w11 = wi * wj;
	return mk3((w00 * e00.x + w10 * e10.x) + (w01 * e01.x + w11 * e11.x), (w00 * e00.y + w10 * e10.y) + (w01 * e01.y + w11 * e11.y),
	           (w00 * e00.z + w10 * e10.z) + (w01 * e01.z + w11 * e11.z));
}

// render_geometry_mesh (src/testbed_geometry_training.cu:2202-2320), Shade mode, floor disabled, one thread per pixel
__global__ void render_mesh_fused(const MeshSceneParams S, const MeshShadeParams P, const IrradianceMap I, const CameraParams C, float4* __restrict__ frame_buffer,
                                  float* __restrict__ depth_buffer, uint32_t shard_index, uint32_t shard_count, int packed) {
	uint32_t x = threadIdx.x + blockDim.x * blockIdx.x;
	uint32_t y = threadIdx.y + blockDim.y * blockIdx.y;
	if (x >= (uint32_t)C.width || y >= (uint32_t)C.height) return;
	// camera-tile sharding: same 8x8 tile -> rank mapping as the NeRF pass
	uint32_t tile = (y >> 3) * (((uint32_t)C.width + 7u) >> 3) + (x >> 3);
	if (tile % shard_count != shard_index) return;
	// tile-packed layout: local tile q = tile / shard_count, slot = (x & 7) + 8 * (y & 7)
	const uint32_t idx = packed ? (tile / shard_count) * 64u + (x & 7u) + 8u * (y & 7u) : x + (uint32_t)C.width * y;
	if (C.moving) return; // (a moving camera is a NeRF-mode feature here; the host refuses the combination)
	const f3 cam_fwd = mk3(C.m[6], C.m[7], C.m[8]);
	const f3 cam_pos = mk3(C.m[9], C.m[10], C.m[11]);
	// M1: init_rays_with_payload_kernel_mesh_geometry (:488-579)
	float u = ((float)x + C.pixel_offset[0]) / (float)C.width;
	float v = ((float)y + C.pixel_offset[1]) / (float)C.height;
	f3 dir;
	lens_direction(C, u, v, dir);
	dir = m3_mulv(C.m, dir);
	f3 origin = cam_pos;
	if (C.aperture_size != 0.0f) {
		float o3[3] = {origin.x, origin.y, origin.z}, d3[3] = {dir.x, dir.y, dir.z}, cm[6];
		for (int i = 0; i < 6; ++i) cm[i] = C.m[i];
		apply_aperture(cm, C.aperture_size, C.focus_z, C.spp, (uint32_t)(int)(u * (float)C.width) * 19349663u + (uint32_t)(int)(v * (float)C.height) * 96925573u, o3, d3);
		origin = mk3(o3[0], o3[1], o3[2]);
		dir = mk3(d3[0], d3[1], d3[2]);
	}
	origin = add3(origin, scale3(dir, C.near_distance));
	depth_buffer[idx] = MAX_DEPTH;
	if (dir.x == 0.0f && dir.y == 0.0f && dir.z == 0.0f) return;
	dir = normalize3(dir);
	float t = fmaxf(aabb_ray_entry(S.scene_min, S.scene_max, origin, dir), 0.0f);
	f3 pos = add3(origin, scale3(dir, t + 1e-6f));
	const f3 primary_dir = dir;
	// M2 on every ray; the normal buffer starts out holding the ray direction (trace_mesh_bvh :2140-2155)
	f3 normal = dir;
	trace_mesh(S, pos, normal);
	// M3: prepare_shadow_rays_geometry (:222-271)
	const f3 sun = normalize3(ld3(P.sun_dir));
	float shadow;
	{
		float nd = dot3(normal, primary_dir);
		f3 ff = nd < 0.0f ? normal : scale3(normal, -1.0f); // faceforward(n, dir, n)
		f3 view_pos = add3(pos, scale3(normalize3(ff), 1e-3f));
		f3 sdir = normalize3(sun);
		float st = fmaxf(aabb_ray_entry(S.scene_min, S.scene_max, view_pos, sdir) + 1e-6f, 0.0f);
		view_pos = add3(view_pos, scale3(sdir, st));
		f3 spos = view_pos;
		f3 sn = box_contains(S.scene_min, S.scene_max, view_pos) ? sdir : primary_dir; // dead shadow rays keep the copied payload.dir
		trace_mesh(S, spos, sn);
		// M4: write_shadow_ray_result_geometry (:273-278), min_visibility == 1
		shadow = box_contains(S.scene_min, S.scene_max, spos) ? 0.0f : 1.0f;
	}
	// M5: shade_kernel_mesh_geometry (:280-355)
	if (!box_contains(S.scene_min, S.scene_max, pos)) return;
	f3 N = normalize3(normal);
	f3 up = normalize3(ld3(P.up_dir));
	float skyam = -dot3(N, up) * 0.5f + 0.5f;
	f3 suncol = scale3(scale3(mk3(255.f / 255.0f, 225.f / 255.0f, 195.f / 255.0f), 4.f), shadow);
	f3 skycol = scale3(scale3(mk3(195.f / 255.0f, 215.f / 255.0f, 255.f / 255.0f), 4.f), skyam);
	f3 base = ld3(P.basecolor);
	f3 ambc = mul3(ld3(P.ambientcolor), skycol);
	if (I.irradiance) { // ShadeEnvMap / ShadeGridEnvMap: ambient light = E(N)/pi from the NeRF-derived irradiance table(s)
		f3 E = irradiance_lookup(I, pos, N);
		ambc = mk3(E.x / PI_F, E.y / PI_F, E.z / PI_F);
	}
	f3 color = evaluate_shading(mul3(base, base), ambc, suncol, P.metallic, P.subsurface, P.specular, P.roughness, 0.f, P.sheen,
	                            0.f, P.clearcoat, P.clearcoat_gloss, sun, scale3(normalize3(primary_dir), -1.0f), N);
	frame_buffer[idx] = make_float4(color.x, color.y, color.z, 1.0f);
	depth_buffer[idx] = dot3(cam_fwd, sub3(pos, cam_pos));
}

// stage kernel: M2 alone (mesh_raytrace_kernel)
__global__ void trace_mesh_rays_kernel(const MeshSceneParams S, uint32_t n, float* __restrict__ positions, float* __restrict__ directions) {
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	f3 p = ld3(positions + 3 * (size_t)i), d = ld3(directions + 3 * (size_t)i);
	trace_mesh(S, p, d);
	positions[3 * (size_t)i] = p.x; positions[3 * (size_t)i + 1] = p.y; positions[3 * (size_t)i + 2] = p.z;
	directions[3 * (size_t)i] = d.x; directions[3 * (size_t)i + 1] = d.y; directions[3 * (size_t)i + 2] = d.z;
}

void launch_render_mesh(const MeshSceneParams& S, const MeshShadeParams& P, const IrradianceMap& I, const CameraParams& C, float4* frame_buffer, float* depth_buffer,
                        uint32_t shard_index, uint32_t shard_count, int packed, hipStream_t stream) {
	dim3 threads(16, 8, 1);
	dim3 blocks((C.width + 15) / 16, (C.height + 7) / 8, 1);
	hipLaunchKernelGGL(render_mesh_fused, blocks, threads, 0, stream, S, P, I, C, frame_buffer, depth_buffer, shard_index, shard_count, packed);
}
// stage kernel: the irradiance lookup at explicit surface points (ngp_irradiance_at)
__global__ void irradiance_lookup_kernel(const IrradianceMap I, uint32_t n, const float* __restrict__ positions, const float* __restrict__ normals, float4* __restrict__ out) {
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	f3 E = irradiance_lookup(I, ld3(positions + 3 * (size_t)i), ld3(normals + 3 * (size_t)i));
	out[i] = make_float4(E.x, E.y, E.z, 0.f);
}
void launch_irradiance_lookup(const IrradianceMap& I, uint32_t n, const float* positions, const float* normals, float4* out, hipStream_t stream) {
	if (n) hipLaunchKernelGGL(irradiance_lookup_kernel, dim3((n + 127) / 128), dim3(128), 0, stream, I, n, positions, normals, out);
}
void launch_trace_mesh_rays(const MeshSceneParams& S, uint32_t n, float* positions, float* directions, hipStream_t stream) {
	hipLaunchKernelGGL(trace_mesh_rays_kernel, dim3((n + 127) / 128), dim3(128), 0, stream, S, n, positions, directions);
}

} // namespace ngp
