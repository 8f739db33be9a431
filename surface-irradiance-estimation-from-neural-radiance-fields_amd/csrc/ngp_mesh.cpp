// Geometry mode, host side of the C ABI: Testbed::load_scene / load_mesh (reference
// src/testbed_geometry_training.cu:2751-2866, 3101-3210), the BVH4 build (src/triangle_bvh.cu:425-508) and the
// .obj reader that replaces the vendored tinyobjloader wrapper (src/tinyobj_loader_wrapper.cu).
#include "ngp_host.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <stack>

using namespace ngp;

namespace {

struct V3 {
	float x, y, z;
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3 vmin(V3 a, V3 b) { return {std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return {std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)}; }
inline V3 ld(const float* p) { return {p[0], p[1], p[2]}; }
inline void st(float* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

inline V3 centroid(const Triangle& t) { return (ld(t.a) + ld(t.b) + ld(t.c)) / 3.0f; }           // triangle.cuh:149-151
inline float centroid(const Triangle& t, int axis) { return (t.a[axis] + t.b[axis] + t.c[axis]) / 3; } // triangle.cuh:153-155

void bounds(const Triangle* begin, const Triangle* end, float* bmin, float* bmax) { // BoundingBox(Triangle*, Triangle*)
	V3 lo = ld(begin->a), hi = lo;
	for (const Triangle* it = begin; it != end; ++it) {
		lo = vmin(lo, vmin(ld(it->a), vmin(ld(it->b), ld(it->c))));
		hi = vmax(hi, vmax(ld(it->a), vmax(ld(it->b), ld(it->c))));
	}
	st(bmin, lo);
	st(bmax, hi);
}

// TriangleBvhWithBranchingFactor<4>::build (src/triangle_bvh.cu:425-508)
void build_bvh4(std::vector<Triangle>& triangles, uint32_t n_primitives_per_leaf, std::vector<TriangleBvhNode>& nodes) {
	constexpr int BF = 4;
	nodes.clear();
	nodes.emplace_back();
	bounds(triangles.data(), triangles.data() + triangles.size(), nodes.front().bmin, nodes.front().bmax);
	nodes.front().left_idx = nodes.front().right_idx = 0;
	struct BuildNode {
		int node_idx;
		std::vector<Triangle>::iterator begin, end;
	};
	std::stack<BuildNode> build_stack;
	build_stack.push({0, triangles.begin(), triangles.end()});
	while (!build_stack.empty()) {
		BuildNode curr = build_stack.top();
		build_stack.pop();
		std::array<BuildNode, BF> children;
		children[0].begin = curr.begin;
		children[0].end = curr.end;
		int n_children = 1;
		while (n_children < BF) {
			for (int i = n_children - 1; i >= 0; --i) {
				BuildNode child = children[i];
				const float count = (float)std::distance(child.begin, child.end);
				V3 mean{0.f, 0.f, 0.f};
				for (auto it = child.begin; it != child.end; ++it) mean = mean + centroid(*it);
				mean = mean / count;
				V3 var{0.f, 0.f, 0.f};
				for (auto it = child.begin; it != child.end; ++it) {
					V3 diff = centroid(*it) - mean;
					var = var + diff * diff;
				}
				var = var / count;
				float max_val = std::max(std::max(var.x, var.y), var.z);
				int axis = var.x == max_val ? 0 : (var.y == max_val ? 1 : 2);
				auto m = child.begin + std::distance(child.begin, child.end) / 2;
				std::nth_element(child.begin, m, child.end, [&](const Triangle& t1, const Triangle& t2) { return centroid(t1, axis) < centroid(t2, axis); });
				children[i * 2].begin = child.begin;
				children[i * 2 + 1].end = child.end;
				children[i * 2].end = children[i * 2 + 1].begin = m;
			}
			n_children *= 2;
		}
		nodes[curr.node_idx].left_idx = (int)nodes.size();
		for (int i = 0; i < BF; ++i) {
			BuildNode& child = children[i];
			child.node_idx = (int)nodes.size();
			nodes.emplace_back();
			TriangleBvhNode& nd = nodes.back();
			if (child.begin != child.end) {
				bounds(&*child.begin, &*child.begin + std::distance(child.begin, child.end), nd.bmin, nd.bmax);
			} else { // the reference asserts this away; an empty child is an empty leaf
				for (int k = 0; k < 3; ++k) { nd.bmin[k] = std::numeric_limits<float>::infinity(); nd.bmax[k] = -std::numeric_limits<float>::infinity(); }
			}
			if (std::distance(child.begin, child.end) <= (std::ptrdiff_t)n_primitives_per_leaf) {
				nd.left_idx = -(int)std::distance(triangles.begin(), child.begin) - 1;
				nd.right_idx = -(int)std::distance(triangles.begin(), child.end) - 1;
			} else {
				nd.left_idx = nd.right_idx = 0;
				build_stack.push(child);
			}
		}
		nodes[curr.node_idx].right_idx = (int)nodes.size();
	}
}

void free_mesh_device(HostMesh& m) {
	if (m.d_tris) (void)hipFree(m.d_tris);
	if (m.d_nodes) (void)hipFree(m.d_nodes);
	m.d_tris = nullptr;
	m.d_nodes = nullptr;
}

// after any change of the mesh list: scene AABB (load_scene :3183-3189) and the device-side MeshRef table
void rebuild_scene(ngp_ctx* ctx) {
	++ctx->mesh_generation;
	if (ctx->d_meshrefs) {
		(void)hipFree(ctx->d_meshrefs);
		ctx->d_meshrefs = nullptr;
	}
	ctx->mesh_scene = MeshSceneParams{};
	if (ctx->meshes.empty()) return;
	V3 lo = ld(ctx->meshes[0].bmin), hi = ld(ctx->meshes[0].bmax);
	for (auto& m : ctx->meshes) {
		lo = vmin(lo, ld(m.bmin));
		hi = vmax(hi, ld(m.bmax));
	}
	st(ctx->mesh_scene.scene_min, {lo.x - 4.0f, lo.y - 4.0f, lo.z - 4.0f});
	st(ctx->mesh_scene.scene_max, {hi.x + 4.0f, hi.y + 4.0f, hi.z + 4.0f});
	ctx->mesh_scene.n_meshes = (uint32_t)ctx->meshes.size();
	if (ctx->device < 0) return;
	std::vector<MeshRef> refs(ctx->meshes.size());
	for (size_t i = 0; i < refs.size(); ++i) {
		HostMesh& m = ctx->meshes[i];
		if (!m.d_tris) {
			NGP_HIP_CHECK(hipMalloc((void**)&m.d_tris, m.tris.size() * sizeof(Triangle)));
			NGP_HIP_CHECK(hipMemcpy(m.d_tris, m.tris.data(), m.tris.size() * sizeof(Triangle), hipMemcpyHostToDevice));
			NGP_HIP_CHECK(hipMalloc((void**)&m.d_nodes, m.nodes.size() * sizeof(TriangleBvhNode)));
			NGP_HIP_CHECK(hipMemcpy(m.d_nodes, m.nodes.data(), m.nodes.size() * sizeof(TriangleBvhNode), hipMemcpyHostToDevice));
		}
		refs[i].nodes = m.d_nodes;
		refs[i].tris = m.d_tris;
		memcpy(refs[i].bmin, m.bmin, sizeof(m.bmin));
		memcpy(refs[i].bmax, m.bmax, sizeof(m.bmax));
		refs[i].n_tris = (uint32_t)m.tris.size();
		refs[i].n_nodes = (uint32_t)m.nodes.size();
	}
	NGP_HIP_CHECK(hipMalloc((void**)&ctx->d_meshrefs, refs.size() * sizeof(MeshRef)));
	NGP_HIP_CHECK(hipMemcpy(ctx->d_meshrefs, refs.data(), refs.size() * sizeof(MeshRef), hipMemcpyHostToDevice));
	ctx->mesh_scene.meshes = ctx->d_meshrefs;
}

// Testbed::load_mesh (:2786-2866): normalise into the unit cube around `center`, build the BVH
void add_mesh_impl(ngp_ctx* ctx, const float* vertices, uint32_t n_tris, const float* center3) {
	if (!vertices || n_tris == 0) throw std::runtime_error("mesh has no triangles");
	const size_t n_vertices = (size_t)n_tris * 3;
	const float inf = std::numeric_limits<float>::infinity();
	V3 lo{inf, inf, inf}, hi{-inf, -inf, -inf};
	for (size_t i = 0; i < n_vertices; ++i) {
		lo = vmin(lo, ld(vertices + 3 * i));
		hi = vmax(hi, ld(vertices + 3 * i));
	}
	const float inflation = 0.005f;
	V3 d0 = hi - lo;
	float amount = std::sqrt((d0.x * d0.x + d0.y * d0.y) + d0.z * d0.z) * inflation;
	lo = {lo.x - amount, lo.y - amount, lo.z - amount};
	hi = {hi.x + amount, hi.y + amount, hi.z + amount};
	V3 diag = hi - lo;
	float mesh_scale = std::max(std::max(diag.x, diag.y), diag.z);
	V3 center = center3 ? ld(center3) : V3{0.f, 0.f, 0.f};
	HostMesh mesh;
	mesh.tris.resize(n_tris);
	st(mesh.center, center);
	for (size_t i = 0; i < n_vertices; ++i) {
		V3 p = ld(vertices + 3 * i);
		V3 q = (p - lo - V3{diag.x * 0.5f, diag.y * 0.5f, diag.z * 0.5f}) / mesh_scale;
		q = {q.x + 0.5f, q.y + 0.5f, q.z + 0.5f};
		q = q + center;
		Triangle& t = mesh.tris[i / 3];
		st(i % 3 == 0 ? t.a : (i % 3 == 1 ? t.b : t.c), q);
	}
	build_bvh4(mesh.tris, 8, mesh.nodes);
	bounds(mesh.tris.data(), mesh.tris.data() + mesh.tris.size(), mesh.bmin, mesh.bmax); // BoundingBox(MeshData*), geometry_bvh.cu:14-34
	ctx->meshes.push_back(std::move(mesh));
	rebuild_scene(ctx);
}

// ascii .obj: 'v' and 'f' records; polygons are fan-triangulated (tinyobj triangulate=true), non-triangle faces kept
std::vector<float> load_obj(const std::string& path) {
	std::string text = read_file(path);
	std::vector<float> verts, out;
	size_t pos = 0;
	while (pos < text.size()) {
		size_t eol = text.find('\n', pos);
		if (eol == std::string::npos) eol = text.size();
		const char* p = text.data() + pos;
		const char* end = text.data() + eol;
		while (p < end && (*p == ' ' || *p == '\t')) ++p;
		if (end - p > 2 && p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
			char* q = nullptr;
			p += 2;
			for (int k = 0; k < 3; ++k) {
				verts.push_back(strtof(p, &q));
				p = q;
			}
		} else if (end - p > 2 && p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
			p += 2;
			std::vector<long> idx;
			while (p < end) {
				while (p < end && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
				if (p >= end) break;
				char* q = nullptr;
				long i = strtol(p, &q, 10);
				if (q == p) break;
				long nv = (long)(verts.size() / 3);
				idx.push_back(i > 0 ? i - 1 : nv + i);
				p = q;
				while (p < end && *p != ' ' && *p != '\t') ++p; // skip /vt/vn
			}
			for (size_t k = 1; k + 1 < idx.size(); ++k) {
				for (long vi : {idx[0], idx[k], idx[k + 1]}) {
					if (vi < 0 || (size_t)vi * 3 + 2 >= verts.size()) throw std::runtime_error("Error loading '" + path + "': face index out of range");
					out.insert(out.end(), verts.begin() + vi * 3, verts.begin() + vi * 3 + 3);
				}
			}
		}
		pos = eol + 1;
	}
	return out;
}

// geometry_load_stl (:2751-2784): binary STL
std::vector<float> load_stl(const std::string& path) {
	std::string data = read_file(path);
	if (data.size() < 84) throw std::runtime_error("Mesh file '" + path + "' too small for STL header");
	uint32_t nfaces;
	memcpy(&nfaces, data.data() + 80, 4);
	if (memcmp(data.data(), "solid", 5) == 0 || nfaces == 0) throw std::runtime_error("ASCII STL file '" + path + "' not supported");
	std::vector<float> out;
	for (uint32_t i = 0; i < nfaces; ++i) {
		size_t off = 84 + (size_t)i * 50;
		if (off + 50 > data.size()) break;
		float v[9];
		memcpy(v, data.data() + off + 12, 36);
		out.insert(out.end(), v, v + 9);
	}
	return out;
}

void load_mesh_file_impl(ngp_ctx* ctx, const std::string& path, const float* center3) {
	std::vector<float> v;
	if (ends_with_ci(path, ".obj")) v = load_obj(path);
	else if (ends_with_ci(path, ".stl")) v = load_stl(path);
	else throw std::runtime_error("mesh data path must be a mesh in ascii .obj or binary .stl format.");
	add_mesh_impl(ctx, v.data(), (uint32_t)(v.size() / 9), center3);
}

} // namespace

extern "C" {

int ngp_clear_meshes(ngp_ctx* ctx) {
	return guarded(ctx, [&] {
		for (auto& m : ctx->meshes) free_mesh_device(m);
		ctx->meshes.clear();
		rebuild_scene(ctx);
	});
}

int ngp_add_mesh(ngp_ctx* ctx, const float* vertices, uint32_t n_tris, const float* center3) {
	return guarded(ctx, [&] { add_mesh_impl(ctx, vertices, n_tris, center3); });
}

int ngp_load_mesh_file(ngp_ctx* ctx, const char* path, const float* center3) {
	return guarded(ctx, [&] {
		if (!path) throw std::runtime_error("null path");
		load_mesh_file_impl(ctx, path, center3);
	});
}

int ngp_load_scene(ngp_ctx* ctx, const char* json_path) {
	return guarded(ctx, [&] {
		if (!json_path) throw std::runtime_error("null path");
		mj::Value json = mj::parse_json(read_file(json_path));
		if (!json.is_object() || json.size() == 0) throw std::runtime_error("Geometry file must contain an array of geometry metadata.");
		const mj::Value& geometries = json.at("geometry");
		const std::string base = parent_dir(json_path);
		for (auto& m : ctx->meshes) free_mesh_device(m);
		ctx->meshes.clear();
		for (const mj::Value& g : geometries.arr) {
			std::string path = g.at("path").str();
			if (!path.empty() && path[0] != '/') path = base + "/" + path;
			const std::string& type = g.at("type").str();
			float center[3];
			for (int i = 0; i < 3; ++i) center[i] = (float)g.at("center").at((size_t)i).num();
			if (type == "Mesh") load_mesh_file_impl(ctx, path, center);
			else if (type == "Nerf") load_snapshot_path(ctx, path);
			else throw std::runtime_error("Geometry type must be either 'Mesh' or 'Nerf'.");
		}
		rebuild_scene(ctx);
	});
}

int ngp_n_meshes(const ngp_ctx* ctx) { return ctx ? (int)ctx->meshes.size() : -1; }

int ngp_get_mesh_info(const ngp_ctx* ctx, int mesh, uint32_t* n_tris, uint32_t* n_nodes, float* aabb6) {
	if (!ctx || ctx->meshes.empty()) return -1;
	if (mesh == -1) {
		if (aabb6) { memcpy(aabb6, ctx->mesh_scene.scene_min, 12); memcpy(aabb6 + 3, ctx->mesh_scene.scene_max, 12); }
		if (n_tris) { *n_tris = 0; for (auto& m : ctx->meshes) *n_tris += (uint32_t)m.tris.size(); }
		if (n_nodes) { *n_nodes = 0; for (auto& m : ctx->meshes) *n_nodes += (uint32_t)m.nodes.size(); }
		return 0;
	}
	if (mesh < 0 || mesh >= (int)ctx->meshes.size()) return -1;
	const HostMesh& m = ctx->meshes[(size_t)mesh];
	if (n_tris) *n_tris = (uint32_t)m.tris.size();
	if (n_nodes) *n_nodes = (uint32_t)m.nodes.size();
	if (aabb6) { memcpy(aabb6, m.bmin, 12); memcpy(aabb6 + 3, m.bmax, 12); }
	return 0;
}

int ngp_get_mesh_bvh(const ngp_ctx* ctx, int mesh, void* nodes_out, void* triangles_out) {
	if (!ctx || mesh < 0 || mesh >= (int)ctx->meshes.size()) return -1;
	const HostMesh& m = ctx->meshes[(size_t)mesh];
	if (nodes_out) memcpy(nodes_out, m.nodes.data(), m.nodes.size() * sizeof(TriangleBvhNode));
	if (triangles_out) memcpy(triangles_out, m.tris.data(), m.tris.size() * sizeof(Triangle));
	return 0;
}

int ngp_set_geometry_opts(ngp_ctx* ctx, const ngp_geometry_opts* o) {
	if (!ctx || !o) return -1;
	static_assert(sizeof(ngp_geometry_opts) == sizeof(MeshShadeParams), "geometry opts layout");
	memcpy(&ctx->shade, o, sizeof(MeshShadeParams));
	return 0;
}

int ngp_trace_mesh_rays(ngp_ctx* ctx, uint32_t n, float* positions, float* directions) {
	return guarded(ctx, [&] {
		if (ctx->device < 0) throw std::runtime_error("this context has no HIP device (host-only); there is no CPU fallback");
		if (ctx->meshes.empty()) throw std::runtime_error("no meshes loaded");
		if (n == 0) return;
		if (!positions || !directions) throw std::runtime_error("null argument");
		float *d_p = nullptr, *d_d = nullptr;
		const size_t bytes = (size_t)n * 3 * sizeof(float);
		NGP_HIP_CHECK(hipMalloc((void**)&d_p, bytes));
		NGP_HIP_CHECK(hipMalloc((void**)&d_d, bytes));
		NGP_HIP_CHECK(hipMemcpy(d_p, positions, bytes, hipMemcpyHostToDevice));
		NGP_HIP_CHECK(hipMemcpy(d_d, directions, bytes, hipMemcpyHostToDevice));
		launch_trace_mesh_rays(ctx->mesh_scene, n, d_p, d_d, ctx->stream);
		NGP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
		NGP_HIP_CHECK(hipMemcpy(positions, d_p, bytes, hipMemcpyDeviceToHost));
		NGP_HIP_CHECK(hipMemcpy(directions, d_d, bytes, hipMemcpyDeviceToHost));
		(void)hipFree(d_p);
		(void)hipFree(d_d);
		NGP_HIP_CHECK(hipGetLastError());
	});
}


// ------------------------------------------------------------------------------------------------ irradiance probes
namespace {
// trace the fan(s) described by P in ONE persistent launch, reduce to the probe texture(s), tabulate E(n) at the texel directions
void compute_probes(ngp_ctx* ctx, ngp::ProbeParams P, float min_transmittance) {
	using namespace ngp;
	if (ctx->device < 0) throw std::runtime_error("this context has no HIP device (host-only); there is no CPU fallback");
	if (!ctx->model_loaded) throw std::runtime_error("No network available.");
	ngp::sync_inference_model(ctx);
	if (ctx->M.rgb_mid != 1 && !ctx->M.wide.width) throw std::runtime_error("irradiance probes are built for the configs/nerf/base.json rgb head (2 hidden layers)");
	ensure_sync_buffers(ctx);
	for (int i = 0; i < 3; ++i) P.center[i] = 0.5f * (ctx->M.raabb_max[i] + ctx->M.raabb_min[i]); // render_aabb.center()
	const uint32_t no = P.mode == NGP_PROBE_MULTI_CENTER ? P.n_origin : 1u;
	const uint32_t n_probes = P.mode == 3 ? P.grid_x * P.grid_y : 1u;
	const uint64_t n_rays64 = (uint64_t)P.n_theta * P.n_phi * no * no * n_probes;
	if (n_rays64 > (1ull << 28)) throw std::runtime_error("probe too large");
	P.n_rays = (uint32_t)n_rays64;
	const uint32_t n_texels = P.n_theta * P.n_phi * n_probes;
	NGP_HIP_CHECK(hipMalloc((void**)&P.ray_rgba, (size_t)P.n_rays * sizeof(float4)));
	if (ctx->d_envmap) (void)hipFree(ctx->d_envmap);
	if (ctx->d_irradiance) (void)hipFree(ctx->d_irradiance);
	ctx->d_envmap = ctx->d_irradiance = nullptr;
	NGP_HIP_CHECK(hipMalloc((void**)&ctx->d_envmap, (size_t)n_texels * sizeof(float4)));
	NGP_HIP_CHECK(hipMalloc((void**)&ctx->d_irradiance, (size_t)n_texels * sizeof(float4)));
	hipStream_t stream = ctx->stream;
	if (ctx->last_stream && ctx->last_stream != stream) NGP_HIP_CHECK(hipStreamSynchronize(ctx->last_stream));
	const int slot = (int)(ctx->n_calls % ngp_ctx::HISTORY);
	if (ctx->n_calls >= (uint64_t)ngp_ctx::HISTORY) NGP_HIP_CHECK(hipStreamWaitEvent(stream, ctx->ev_frame1[slot], 0)); // the slot's previous launch (render_frames, ngp_api.cpp)
	FrameParams F{};
	ctx->bind_slot(F, slot);
	F.n_local_tiles = (P.n_rays + 63) / 64;
	F.shard_index = 0;
	F.shard_count = 1;
	F.min_transmittance = min_transmittance > 0.f ? min_transmittance : 0.01f;
	F.linear_colors = ctx->desc.linear_colors;
	memcpy(F.tune, ctx->tune, sizeof(F.tune));
	NGP_HIP_CHECK(hipEventRecord(ctx->ev_frame0[slot], stream));
	NGP_HIP_CHECK(hipMemsetAsync(P.ray_rgba, 0, (size_t)P.n_rays * sizeof(float4), stream));
	NGP_HIP_CHECK(hipEventRecord(ctx->ev_kern0[slot], stream));
	ModelParams M = ctx->M;
	if (!ctx->meshes.empty()) { // Geometry mode: load_scene made the inflated mesh box the render box (testbed_geometry_training.cu:3185-3189); the shell positions lie inside it
		for (int i = 0; i < 3; ++i) { M.raabb_min[i] = ctx->mesh_scene.scene_min[i]; M.raabb_max[i] = ctx->mesh_scene.scene_max[i]; }
		const float ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
		memcpy(M.r2l, ident, sizeof(ident));
		M.r2l_identity = 1u;
	}
	launch_trace_probe(M, F, P, ctx->n_cus, stream);
	NGP_HIP_CHECK(hipEventRecord(ctx->ev_kern1[slot], stream));
	launch_probe_reduce(P, ctx->d_envmap, stream);
	launch_irradiance(P, ctx->d_envmap, n_texels, nullptr, ctx->d_irradiance, stream);
	NGP_HIP_CHECK(hipEventRecord(ctx->ev_frame1[slot], stream));
	ctx->hist_n_rays[slot] = P.n_rays;
	ctx->last_stream = stream;
	++ctx->n_calls;
	NGP_HIP_CHECK(hipStreamSynchronize(stream));
	NGP_HIP_CHECK(hipGetLastError());
	(void)hipFree(P.ray_rgba);
	P.ray_rgba = nullptr;
	++ctx->probe_generation;
	ctx->env_probe = P;
	ctx->env_n_theta = P.n_theta;
	ctx->env_n_phi = P.n_phi;
}
size_t env_texels(const ngp_ctx* ctx) {
	const ngp::ProbeParams& P = ctx->env_probe;
	return (size_t)P.n_theta * P.n_phi * (P.mode == 3 ? P.grid_x * P.grid_y : 1u);
}
} // namespace

int ngp_compute_envmap(ngp_ctx* ctx, const ngp_probe_desc* d, float* rgba_out) {
	return guarded(ctx, [&] {
		if (!d || d->n_theta == 0 || d->n_phi == 0 || d->mode < 0 || d->mode > 2) throw std::runtime_error("invalid probe descriptor");
		if (d->mode == NGP_PROBE_MULTI_CENTER && d->n_origin == 0) throw std::runtime_error("invalid probe descriptor: n_origin");
		ngp::ProbeParams P{};
		P.mode = d->mode;
		P.n_theta = d->n_theta;
		P.n_phi = d->n_phi;
		P.n_origin = d->mode == NGP_PROBE_MULTI_CENTER ? d->n_origin : 1u;
		for (int i = 0; i < 3; ++i) P.origin[i] = d->origin[i];
		compute_probes(ctx, P, d->min_transmittance);
		if (rgba_out) NGP_HIP_CHECK(hipMemcpy(rgba_out, ctx->d_envmap, env_texels(ctx) * sizeof(float4), hipMemcpyDeviceToHost));
	});
}

int ngp_compute_envmap_grid(ngp_ctx* ctx, const ngp_probe_grid_desc* d, float* rgba_out) {
	return guarded(ctx, [&] {
		if (!d || d->n_theta == 0 || d->n_phi == 0 || d->grid_x == 0 || d->grid_y == 0 || !(d->shell_radius > 0.f)) throw std::runtime_error("invalid probe grid descriptor");
		if ((uint64_t)d->grid_x * d->grid_y > 65536ull) throw std::runtime_error("probe grid too large");
		ngp::ProbeParams P{};
		P.mode = 3;
		P.n_theta = d->n_theta;
		P.n_phi = d->n_phi;
		P.n_origin = 1;
		P.grid_x = d->grid_x;
		P.grid_y = d->grid_y;
		P.shell_radius = d->shell_radius;
		compute_probes(ctx, P, d->min_transmittance);
		if (rgba_out) NGP_HIP_CHECK(hipMemcpy(rgba_out, ctx->d_envmap, env_texels(ctx) * sizeof(float4), hipMemcpyDeviceToHost));
	});
}

int ngp_get_envmap(ngp_ctx* ctx, uint32_t* n_theta, uint32_t* n_phi, float* rgba_out, float* irradiance_rgba_out) {
	return guarded(ctx, [&] {
		if (!ctx->d_envmap) throw std::runtime_error("no probe texture: call ngp_compute_envmap first");
		if (n_theta) *n_theta = ctx->env_n_theta;
		if (n_phi) *n_phi = ctx->env_n_phi;
		const size_t bytes = env_texels(ctx) * sizeof(float4); // a grid returns grid_x * grid_y textures back to back (ngp_get_envmap_grid tells how many)
		if (rgba_out) NGP_HIP_CHECK(hipMemcpy(rgba_out, ctx->d_envmap, bytes, hipMemcpyDeviceToHost));
		if (irradiance_rgba_out) NGP_HIP_CHECK(hipMemcpy(irradiance_rgba_out, ctx->d_irradiance, bytes, hipMemcpyDeviceToHost));
	});
}

int ngp_get_envmap_grid(ngp_ctx* ctx, ngp_probe_grid_desc* desc_out, float* origins_out) {
	return guarded(ctx, [&] {
		if (!ctx->d_envmap || ctx->env_probe.mode != 3) throw std::runtime_error("no probe grid: call ngp_compute_envmap_grid first");
		const ngp::ProbeParams& P = ctx->env_probe;
		if (desc_out) {
			desc_out->grid_x = P.grid_x; desc_out->grid_y = P.grid_y; desc_out->n_theta = P.n_theta; desc_out->n_phi = P.n_phi;
			desc_out->shell_radius = P.shell_radius;
			desc_out->min_transmittance = 0.f;
		}
		if (origins_out) { // shell positions, for callers that place things: the same arithmetic as the kernel's probe_grid_origin
			const float PI = 3.14159265358979323846f;
			for (uint32_t g = 0; g < P.grid_x * P.grid_y; ++g) {
				const uint32_t i = g % P.grid_x, j = g / P.grid_x;
				const float px = ((float)i + 0.5f) / (float)P.grid_x, py = ((float)j + 0.5f) / (float)P.grid_y;
				const float cos_theta = -2.0f * px + 1.0f, phi = 2.0f * PI * (py - 0.5f);
				const float sin_theta = sqrtf(fmaxf(1.0f - cos_theta * cos_theta, 0.0f));
				origins_out[3 * g] = P.center[0] + sin_theta * cosf(phi) * P.shell_radius;
				origins_out[3 * g + 1] = P.center[1] + sin_theta * sinf(phi) * P.shell_radius;
				origins_out[3 * g + 2] = P.center[2] + cos_theta * P.shell_radius;
			}
		}
	});
}

namespace {
void download_rgb(ngp_ctx* ctx, const float4* d_o, uint32_t n, float* rgb_out) {
	NGP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	std::vector<float4> tmp(n);
	NGP_HIP_CHECK(hipMemcpy(tmp.data(), d_o, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost));
	for (uint32_t i = 0; i < n; ++i) { rgb_out[3 * i] = tmp[i].x; rgb_out[3 * i + 1] = tmp[i].y; rgb_out[3 * i + 2] = tmp[i].z; }
}
} // namespace

int ngp_irradiance(ngp_ctx* ctx, uint32_t n, const float* normals, float* rgb_out) {
	return guarded(ctx, [&] {
		if (!ctx->d_envmap) throw std::runtime_error("no probe texture: call ngp_compute_envmap first");
		if (ctx->env_probe.mode == 3) throw std::runtime_error("the probe texture is a grid: use ngp_irradiance_at (position + normal)");
		if (n == 0) return;
		if (!normals || !rgb_out) throw std::runtime_error("null argument");
		float* d_n = nullptr;
		float4* d_o = nullptr;
		NGP_HIP_CHECK(hipMalloc((void**)&d_n, (size_t)n * 3 * sizeof(float)));
		NGP_HIP_CHECK(hipMalloc((void**)&d_o, (size_t)n * sizeof(float4)));
		NGP_HIP_CHECK(hipMemcpy(d_n, normals, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
		launch_irradiance(ctx->env_probe, ctx->d_envmap, n, d_n, d_o, ctx->stream);
		download_rgb(ctx, d_o, n, rgb_out);
		(void)hipFree(d_n);
		(void)hipFree(d_o);
		NGP_HIP_CHECK(hipGetLastError());
	});
}

int ngp_irradiance_at(ngp_ctx* ctx, uint32_t n, const float* positions, const float* normals, float* rgb_out) {
	return guarded(ctx, [&] {
		if (!ctx->d_irradiance) throw std::runtime_error("no probe texture: call ngp_compute_envmap / ngp_compute_envmap_grid first");
		if (n == 0) return;
		if (!positions || !normals || !rgb_out) throw std::runtime_error("null argument");
		float *d_p = nullptr, *d_n = nullptr;
		float4* d_o = nullptr;
		NGP_HIP_CHECK(hipMalloc((void**)&d_p, (size_t)n * 3 * sizeof(float)));
		NGP_HIP_CHECK(hipMalloc((void**)&d_n, (size_t)n * 3 * sizeof(float)));
		NGP_HIP_CHECK(hipMalloc((void**)&d_o, (size_t)n * sizeof(float4)));
		NGP_HIP_CHECK(hipMemcpy(d_p, positions, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
		NGP_HIP_CHECK(hipMemcpy(d_n, normals, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
		launch_irradiance_lookup(ngp::irradiance_map_of(ctx), n, d_p, d_n, d_o, ctx->stream);
		download_rgb(ctx, d_o, n, rgb_out);
		(void)hipFree(d_p);
		(void)hipFree(d_n);
		(void)hipFree(d_o);
		NGP_HIP_CHECK(hipGetLastError());
	});
}

} // extern "C"

// Geometry mode on a multi-device context (the reference's render_frame serves every mode on every device, src/testbed.cu:4833-4889,
// 5575-5616): an auxiliary device gets the primary's meshes exactly as built (same triangle order, same BVH4 nodes -- rebuilt nowhere),
// the BRDF / sun parameters and, when probes were computed, the tabulated irradiance E(n). Generation counters: only what changed moves.
namespace ngp {
void sync_peer_geometry(ngp_ctx* primary, ngp_ctx* peer) {
	peer->shade = primary->shade;
	if (peer->synced_mesh_generation != primary->mesh_generation) {
		NGP_HIP_CHECK(hipSetDevice(peer->device));
		NGP_HIP_CHECK(hipStreamSynchronize(peer->stream)); // (a rare event: frames on the peer still trace the old BVHs)
		for (auto& m : peer->meshes) free_mesh_device(m);
		peer->meshes.clear();
		for (const HostMesh& m : primary->meshes) {
			HostMesh c;
			c.tris = m.tris;
			c.nodes = m.nodes;
			memcpy(c.bmin, m.bmin, sizeof(c.bmin));
			memcpy(c.bmax, m.bmax, sizeof(c.bmax));
			memcpy(c.center, m.center, sizeof(c.center));
			peer->meshes.push_back(std::move(c));
		}
		rebuild_scene(peer);
		peer->synced_mesh_generation = primary->mesh_generation;
		NGP_HIP_CHECK(hipSetDevice(primary->device));
	}
	if (peer->synced_probe_generation != primary->probe_generation && primary->d_irradiance) {
		const size_t texels = (size_t)primary->env_n_theta * primary->env_n_phi * (primary->env_probe.mode == 3 ? (size_t)primary->env_probe.grid_x * primary->env_probe.grid_y : 1u);
		NGP_HIP_CHECK(hipSetDevice(peer->device));
		NGP_HIP_CHECK(hipStreamSynchronize(peer->stream));
		if (peer->d_envmap) (void)hipFree(peer->d_envmap);
		if (peer->d_irradiance) (void)hipFree(peer->d_irradiance);
		peer->d_envmap = peer->d_irradiance = nullptr;
		NGP_HIP_CHECK(hipMalloc((void**)&peer->d_envmap, texels * sizeof(float4)));
		NGP_HIP_CHECK(hipMalloc((void**)&peer->d_irradiance, texels * sizeof(float4)));
		NGP_HIP_CHECK(hipSetDevice(primary->device));
		NGP_HIP_CHECK(hipStreamSynchronize(primary->stream)); // compute_probes ends synchronised; a no-op in practice
		NGP_HIP_CHECK(hipMemcpyPeer(peer->d_envmap, peer->device, primary->d_envmap, primary->device, texels * sizeof(float4)));
		NGP_HIP_CHECK(hipMemcpyPeer(peer->d_irradiance, peer->device, primary->d_irradiance, primary->device, texels * sizeof(float4)));
		peer->env_probe = primary->env_probe;
		peer->env_n_theta = primary->env_n_theta;
		peer->env_n_phi = primary->env_n_phi;
		peer->synced_probe_generation = primary->probe_generation;
	}
}
} // namespace ngp
