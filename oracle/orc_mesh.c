/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_common.h). PARITY UNPINNED.
 *
 * Geometry mode, mesh side: mesh normalisation (load_mesh), BVH4 build, two-level trace, sun shadow rays and the
 * Disney-style shade. Restates src/testbed_geometry_training.cu, src/geometry_bvh.cu, src/triangle_bvh.cu,
 * include/neural-graphics-primitives/triangle.cuh; each function cites its lines.
 */
#include "oracle.h"
#include "orc_common.h"
#include "orc_mesh.h"
#include "orc_probe.h"

#include <stdio.h>

#define MAX_DIST 100.0f /* geometry_bvh.cu:23, triangle_bvh.cu */
#define BRANCHING 4
static const float PI_F = 3.14159265358979323846f;

/* ------------------------------------------------------------------ triangle (triangle.cuh:25-63) */
static v3 tri_normal(const orc_triangle* t) { return v3_normalize(v3_cross(v3_sub(t->b, t->a), v3_sub(t->c, t->a))); }
static v3 tri_centroid(const orc_triangle* t) { return v3_divs(v3_add(v3_add(t->a, t->b), t->c), 3.0f); }
static float tri_centroid_axis(const orc_triangle* t, int axis) { return (v3_get(t->a, axis) + v3_get(t->b, axis) + v3_get(t->c, axis)) / 3; }

static float tri_ray_intersect(const orc_triangle* tri, v3 ro, v3 rd) {
	v3 v1v0 = v3_sub(tri->b, tri->a);
	v3 v2v0 = v3_sub(tri->c, tri->a);
	v3 rov0 = v3_sub(ro, tri->a);
	v3 n = v3_cross(v1v0, v2v0);
	v3 q = v3_cross(rov0, rd);
	float d = 1.0f / v3_dot(rd, n);
	float u = d * -v3_dot(q, v2v0);
	float v = d * v3_dot(q, v1v0);
	float t = d * -v3_dot(n, rov0);
	if (u < 0.0f || u > 1.0f || v < 0.0f || (u + v) > 1.0f || t < 0.0f) t = FLT_MAX;
	return t;
}

/* ------------------------------------------------------------------ BVH4 build (triangle_bvh.cu:425-508)
 * Median split on the axis of largest centroid variance. The reference partitions with std::nth_element, whose
 * order inside the two halves is implementation-defined; any median partition yields the same hit distances,
 * so a plain quickselect is used here. */
static void swap_tri(orc_triangle* a, orc_triangle* b) { orc_triangle t = *a; *a = *b; *b = t; }

static void nth_element_axis(orc_triangle* tris, int64_t lo, int64_t hi /*exclusive*/, int64_t nth, int axis) {
	while (hi - lo > 1) {
		float pivot = tri_centroid_axis(&tris[lo + (hi - lo) / 2], axis);
		int64_t i = lo, j = hi - 1;
		while (i <= j) {
			while (tri_centroid_axis(&tris[i], axis) < pivot) ++i;
			while (tri_centroid_axis(&tris[j], axis) > pivot) --j;
			if (i <= j) { swap_tri(&tris[i], &tris[j]); ++i; --j; }
		}
		if (nth <= j) hi = j + 1;
		else if (nth >= i) lo = i;
		else return;
	}
}

static aabb_t aabb_of_tris(const orc_triangle* begin, const orc_triangle* end) { /* bounding_box.cuh:56-61 */
	aabb_t b;
	b.min = b.max = begin->a;
	for (const orc_triangle* it = begin; it != end; ++it) {
		b.min = v3_min(b.min, v3_min(it->a, v3_min(it->b, it->c)));
		b.max = v3_max(b.max, v3_max(it->a, v3_max(it->b, it->c)));
	}
	return b;
}

typedef struct { int node_idx; int64_t begin, end; } build_node;

int orc_bvh_build(orc_triangle* tris, uint32_t n_tris, uint32_t n_primitives_per_leaf, orc_bvh_node** out_nodes, uint32_t* out_n_nodes) {
	if (n_tris == 0) return -1;
	size_t cap = 16 + 4 * (size_t)n_tris;
	orc_bvh_node* nodes = (orc_bvh_node*)malloc(sizeof(orc_bvh_node) * cap);
	build_node* stack = (build_node*)malloc(sizeof(build_node) * (64 + (size_t)n_tris));
	uint32_t n_nodes = 1;
	int sp = 0;
	nodes[0].bb = aabb_of_tris(tris, tris + n_tris);
	nodes[0].left_idx = nodes[0].right_idx = 0;
	stack[sp++] = (build_node){0, 0, (int64_t)n_tris};
	while (sp > 0) {
		build_node curr = stack[--sp];
		build_node children[BRANCHING];
		children[0].begin = curr.begin;
		children[0].end = curr.end;
		int n_children = 1;
		while (n_children < BRANCHING) {
			for (int i = n_children - 1; i >= 0; --i) {
				build_node child = children[i];
				int64_t count = child.end - child.begin;
				v3 mean = v3_make(0, 0, 0);
				for (int64_t k = child.begin; k < child.end; ++k) mean = v3_add(mean, tri_centroid(&tris[k]));
				mean = v3_divs(mean, (float)count);
				v3 var = v3_make(0, 0, 0);
				for (int64_t k = child.begin; k < child.end; ++k) {
					v3 diff = v3_sub(tri_centroid(&tris[k]), mean);
					var = v3_add(var, v3_mul(diff, diff));
				}
				var = v3_divs(var, (float)count);
				float max_val = fmaxf(fmaxf(var.x, var.y), var.z);
				int axis = var.x == max_val ? 0 : (var.y == max_val ? 1 : 2);
				int64_t m = child.begin + count / 2;
				nth_element_axis(tris, child.begin, child.end, m, axis);
				children[i * 2].begin = child.begin;
				children[i * 2 + 1].end = child.end;
				children[i * 2].end = children[i * 2 + 1].begin = m;
			}
			n_children *= 2;
		}
		nodes[curr.node_idx].left_idx = (int)n_nodes;
		for (int i = 0; i < BRANCHING; ++i) {
			build_node* child = &children[i];
			child->node_idx = (int)n_nodes;
			if (n_nodes + 1 > cap) { free(nodes); free(stack); return -2; }
			orc_bvh_node* nd = &nodes[n_nodes++];
			if (child->end > child->begin) nd->bb = aabb_of_tris(tris + child->begin, tris + child->end);
			else { nd->bb.min = nd->bb.max = tris[child->begin < (int64_t)n_tris ? child->begin : 0].a; }
			if (child->end - child->begin <= (int64_t)n_primitives_per_leaf) {
				nd->left_idx = -(int)child->begin - 1;
				nd->right_idx = -(int)child->end - 1;
			} else {
				nd->left_idx = nd->right_idx = 0;
				stack[sp++] = *child;
			}
		}
		nodes[curr.node_idx].right_idx = (int)n_nodes;
	}
	free(stack);
	*out_nodes = nodes;
	*out_n_nodes = n_nodes;
	return 0;
}

/* ------------------------------------------------------------------ BVH4 traversal (geometry_bvh.cu:61-109, triangle_bvh.cu:150-193) */
typedef struct { float dist; uint32_t idx; } dist_idx;
static void cas(dist_idx* a, dist_idx* b) { if (a->dist < b->dist) { dist_idx t = *a; *a = *b; *b = t; } } /* sorts descending */

void orc_bvh_ray_intersect(const orc_bvh_node* nodes, const orc_triangle* tris, const float* ro3, const float* rd3, int* out_idx, float* out_t) {
	v3 ro = v3_make(ro3[0], ro3[1], ro3[2]), rd = v3_make(rd3[0], rd3[1], rd3[2]);
	int stack[64];
	int sp = 0;
	stack[sp++] = 0;
	float mint = MAX_DIST;
	int shortest = -1;
	while (sp > 0) {
		int idx = stack[--sp];
		const orc_bvh_node* node = &nodes[idx];
		if (node->left_idx < 0) {
			int end = -node->right_idx - 1;
			for (int i = -node->left_idx - 1; i < end; ++i) {
				float t = tri_ray_intersect(&tris[i], ro, rd);
				if (t < mint) { mint = t; shortest = i; }
			}
		} else {
			dist_idx ch[BRANCHING];
			uint32_t first = (uint32_t)node->left_idx;
			for (uint32_t i = 0; i < BRANCHING; ++i) {
				float tmin, tmax;
				aabb_ray_intersect(&nodes[i + first].bb, ro, rd, &tmin, &tmax);
				ch[i].dist = tmin;
				ch[i].idx = i + first;
			}
			/* sorting network N == 4, triangle_bvh.cuh:48-53 */
			cas(&ch[0], &ch[2]); cas(&ch[1], &ch[3]); cas(&ch[0], &ch[1]); cas(&ch[2], &ch[3]); cas(&ch[1], &ch[2]);
			for (uint32_t i = 0; i < BRANCHING; ++i) {
				if (ch[i].dist < mint && sp < 64) stack[sp++] = (int)ch[i].idx;
			}
		}
	}
	*out_idx = shortest;
	*out_t = mint;
}

/* ------------------------------------------------------------------ scene (load_mesh :2786-2866, load_scene :3101-3210) */
struct orc_mesh_scene {
	uint32_t n_meshes;
	orc_triangle** tris;
	uint32_t* n_tris;
	orc_bvh_node** nodes;
	uint32_t* n_nodes;
	aabb_t* mesh_bb;
	aabb_t scene_bb; /* root bb inflated by 4 */
};

static void aabb_inflate(aabb_t* b, float a) { b->min = v3_adds(b->min, -a); b->max = v3_adds(b->max, a); }

orc_mesh_scene* orc_mesh_scene_create(uint32_t n_meshes, const float* const* vertices, const uint32_t* n_tris, const float* centers) {
	orc_mesh_scene* s = (orc_mesh_scene*)calloc(1, sizeof(*s));
	s->n_meshes = n_meshes;
	s->tris = (orc_triangle**)calloc(n_meshes, sizeof(void*));
	s->nodes = (orc_bvh_node**)calloc(n_meshes, sizeof(void*));
	s->n_tris = (uint32_t*)calloc(n_meshes, sizeof(uint32_t));
	s->n_nodes = (uint32_t*)calloc(n_meshes, sizeof(uint32_t));
	s->mesh_bb = (aabb_t*)calloc(n_meshes, sizeof(aabb_t));
	for (uint32_t m = 0; m < n_meshes; ++m) {
		uint32_t nv = n_tris[m] * 3;
		const float* src = vertices[m];
		v3 center = v3_make(centers[3 * m], centers[3 * m + 1], centers[3 * m + 2]);
		aabb_t aabb;
		aabb.min = v3_make(INFINITY, INFINITY, INFINITY);
		aabb.max = v3_make(-INFINITY, -INFINITY, -INFINITY);
		for (uint32_t i = 0; i < nv; ++i) {
			v3 p = v3_make(src[3 * i], src[3 * i + 1], src[3 * i + 2]);
			aabb.min = v3_min(aabb.min, p);
			aabb.max = v3_max(aabb.max, p);
		}
		const float inflation = 0.005f;
		aabb_inflate(&aabb, v3_length(v3_sub(aabb.max, aabb.min)) * inflation);
		v3 diag = v3_sub(aabb.max, aabb.min);
		float mesh_scale = fmaxf(fmaxf(diag.x, diag.y), diag.z);
		orc_triangle* tris = (orc_triangle*)malloc(sizeof(orc_triangle) * n_tris[m]);
		for (uint32_t i = 0; i < nv; ++i) {
			v3 p = v3_make(src[3 * i], src[3 * i + 1], src[3 * i + 2]);
			p = v3_adds(v3_divs(v3_sub(v3_sub(p, aabb.min), v3_scale(diag, 0.5f)), mesh_scale), 0.5f);
			p = v3_add(p, center);
			v3* dst = (i % 3 == 0) ? &tris[i / 3].a : ((i % 3 == 1) ? &tris[i / 3].b : &tris[i / 3].c);
			*dst = p;
		}
		s->tris[m] = tris;
		s->n_tris[m] = n_tris[m];
		orc_bvh_build(tris, n_tris[m], 8, &s->nodes[m], &s->n_nodes[m]);
		s->mesh_bb[m] = aabb_of_tris(tris, tris + n_tris[m]); /* BoundingBox(MeshData*), geometry_bvh.cu:14-34 */
	}
	aabb_t root = s->mesh_bb[0];
	for (uint32_t m = 1; m < n_meshes; ++m) { root.min = v3_min(root.min, s->mesh_bb[m].min); root.max = v3_max(root.max, s->mesh_bb[m].max); }
	aabb_inflate(&root, 4.0f);
	s->scene_bb = root;
	return s;
}

void orc_mesh_scene_destroy(orc_mesh_scene* s) {
	if (!s) return;
	for (uint32_t m = 0; m < s->n_meshes; ++m) { free(s->tris[m]); free(s->nodes[m]); }
	free(s->tris); free(s->nodes); free(s->n_tris); free(s->n_nodes); free(s->mesh_bb); free(s);
}

void orc_mesh_scene_aabb(const orc_mesh_scene* s, float* out6) {
	out6[0] = s->scene_bb.min.x; out6[1] = s->scene_bb.min.y; out6[2] = s->scene_bb.min.z;
	out6[3] = s->scene_bb.max.x; out6[4] = s->scene_bb.max.y; out6[5] = s->scene_bb.max.z;
}
uint32_t orc_mesh_scene_n_nodes(const orc_mesh_scene* s, uint32_t mesh) { return s->n_nodes[mesh]; }
const orc_triangle* orc_mesh_scene_triangles(const orc_mesh_scene* s, uint32_t mesh) { return s->tris[mesh]; }
const orc_bvh_node* orc_mesh_scene_nodes(const orc_mesh_scene* s, uint32_t mesh) { return s->nodes[mesh]; }

/* mesh_raytrace_kernel (geometry_bvh.cu:646-676) with GeometryBvh4::ray_intersect's leaf scan (:166-200): the mesh
 * whose AABB has the smallest entry distance (< MAX_DIST, > -FLT_MAX) is traversed; a miss of every mesh AABB leaves
 * the ray untouched. Leaves are scanned in mesh order (the reference's node order is a median partition by
 * center.x+center.y+center.z and only matters for exact ties). */
static void trace_one(const orc_mesh_scene* s, v3* pos, v3* dir) {
	float mint = MAX_DIST;
	int mesh_idx = -1;
	for (uint32_t m = 0; m < s->n_meshes; ++m) {
		float tmin, tmax;
		aabb_ray_intersect(&s->mesh_bb[m], *pos, *dir, &tmin, &tmax);
		if (tmin < mint && tmin > -FLT_MAX) { mint = tmin; mesh_idx = (int)m; }
	}
	if (mesh_idx < 0) return;
	int idx;
	float t;
	float ro[3] = {pos->x, pos->y, pos->z}, rd[3] = {dir->x, dir->y, dir->z};
	orc_bvh_ray_intersect(s->nodes[mesh_idx], s->tris[mesh_idx], ro, rd, &idx, &t);
	*pos = v3_add(*pos, v3_scale(*dir, t));
	if (idx > -1) *dir = tri_normal(&s->tris[mesh_idx][idx]);
}

void orc_trace_mesh(const orc_mesh_scene* s, uint32_t n, float* positions, float* directions) {
#pragma omp parallel for schedule(dynamic, 256)
	for (int64_t i = 0; i < (int64_t)n; ++i) {
		v3 p = v3_make(positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]);
		v3 d = v3_make(directions[3 * i], directions[3 * i + 1], directions[3 * i + 2]);
		trace_one(s, &p, &d);
		positions[3 * i] = p.x; positions[3 * i + 1] = p.y; positions[3 * i + 2] = p.z;
		directions[3 * i] = d.x; directions[3 * i + 1] = d.y; directions[3 * i + 2] = d.z;
	}
}

/* ------------------------------------------------------------------ BRDF (testbed_geometry_training.cu:46-144) */
static float square_(float x) { return x * x; }
static float mixf(float a, float b, float t) { return a + (b - a) * t; }
static v3 mix3(v3 a, v3 b, float t) { return v3_add(a, v3_scale(v3_sub(b, a), t)); }
static float saturatef_(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }
static float SchlickFresnel(float u) {
	float m = saturatef_((float)(1.0 - u));
	return square_(square_(m)) * m;
}
static float G1(float NdotH, float a) {
	if (a >= 1.0) return (float)(1.0 / PI_F);
	float a2 = square_(a);
	float t = (float)(1.0 + (a2 - 1.0) * NdotH * NdotH);
	return (float)((a2 - 1.0) / (PI_F * logf(a2) * t));
}
static float G2(float NdotH, float a) {
	float a2 = square_(a);
	float t = (float)(1.0 + (a2 - 1.0) * NdotH * NdotH);
	return a2 / (PI_F * t * t);
}
static float SmithG_GGX(float NdotV, float alphaG) {
	float a = alphaG * alphaG;
	float b = NdotV * NdotV;
	return (float)(1.0 / (NdotV + sqrtf(a + b - a * b)));
}

static v3 evaluate_shading(v3 base_color, v3 ambient_color, v3 light_color, float metallic, float subsurface, float specular, float roughness,
                           float specular_tint, float sheen, float sheen_tint, float clearcoat, float clearcoat_gloss, v3 L, v3 V, v3 N) {
	float NdotL = v3_dot(N, L);
	float NdotV = v3_dot(N, V);
	v3 H = v3_normalize(v3_add(L, V));
	float NdotH = v3_dot(N, H);
	float LdotH = v3_dot(L, H);
	float FL = SchlickFresnel(NdotL), FV = SchlickFresnel(NdotV);
	v3 amb = v3_scale(ambient_color, mixf(0.2f, FV, metallic));
	amb = v3_mul(amb, base_color);
	if (NdotL < 0.f || NdotV < 0.f) return amb;
	float luminance = v3_dot(base_color, v3_make(0.3f, 0.6f, 0.1f));
	v3 Ctint = v3_scale(base_color, 1.f / (luminance + 0.00001f));
	v3 one = v3_make(1.0f, 1.0f, 1.0f);
	v3 Cspec0 = mix3(v3_scale(v3_scale(mix3(one, Ctint, specular_tint), specular), 0.08f), base_color, metallic);
	v3 Csheen = mix3(one, Ctint, sheen_tint);
	float Fd90 = 0.5f + 2.0f * LdotH * LdotH * roughness;
	float Fd = mixf(1, Fd90, FL) * mixf(1.f, Fd90, FV);
	float Fss90 = LdotH * LdotH * roughness;
	float Fss = mixf(1.0f, Fss90, FL) * mixf(1.0f, Fss90, FV);
	float ss = 1.25f * (Fss * (1.f / (NdotL + NdotV) - 0.5f) + 0.5f);
	float a = fmaxf(0.001f, square_(roughness));
	float Ds = G2(NdotH, a);
	float FH = SchlickFresnel(LdotH);
	v3 Fs = mix3(Cspec0, one, FH);
	float Gs = SmithG_GGX(NdotL, a) * SmithG_GGX(NdotV, a);
	v3 Fsheen = v3_scale(Csheen, FH * sheen);
	float Dr = G1(NdotH, mixf(0.1f, 0.001f, clearcoat_gloss));
	float Fr = mixf(0.04f, 1.0f, FH);
	float Gr = SmithG_GGX(NdotL, 0.25f) * SmithG_GGX(NdotV, 0.25f);
	float CCs = 0.25f * clearcoat * Gr * Fr * Dr;
	v3 diffuse = v3_add(v3_scale(base_color, (float)(1.0f / PI_F) * mixf(Fd, ss, subsurface)), Fsheen);
	v3 brdf = v3_add(v3_add(v3_scale(diffuse, 1.0f - metallic), v3_scale(Fs, Gs * Ds)), v3_make(CCs, CCs, CCs));
	return v3_add(v3_scale(v3_mul(brdf, light_color), NdotL), amb);
}

/* ------------------------------------------------------------------ render_geometry_mesh (:2202-2320), Shade mode, floor disabled */
void orc_render_mesh(const orc_mesh_scene* s, const orc_camera* cam, const orc_mesh_opts* o, float* frame_buffer, float* depth_buffer) {
	const int64_t n = (int64_t)cam->width * cam->height;
	const aabb_t bb = s->scene_bb; /* inflated by sdf.zero_offset = 0 */
	v3 sun = v3_normalize(v3_make(o->sun_dir[0], o->sun_dir[1], o->sun_dir[2]));
	v3 up = v3_normalize(v3_make(o->up_dir[0], o->up_dir[1], o->up_dir[2]));
	v3 cam_fwd = v3_make(cam->matrix[6], cam->matrix[7], cam->matrix[8]);
	v3 cam_pos = v3_make(cam->matrix[9], cam->matrix[10], cam->matrix[11]);
	float off[2];
	orc_ld_random_pixel_offset(cam->snap_to_pixel_centers ? 0u : cam->spp_index, off);
#pragma omp parallel for schedule(dynamic, 256)
	for (int64_t i = 0; i < n; ++i) {
		uint32_t x = (uint32_t)(i % cam->width), y = (uint32_t)(i / cam->width);
		/* M1: init_rays_with_payload_kernel_mesh_geometry (:488-579) */
		float u = ((float)x + off[0]) / (float)cam->width;
		float v = ((float)y + off[1]) / (float)cam->height;
		float d3[3];
		orc_lens_direction(cam, u, v, d3);
		v3 dir = v3_make(d3[0], d3[1], d3[2]);
		dir = m3_mulv(cam->matrix, dir);
		v3 origin = cam_pos;
		{
			float o3[3] = {origin.x, origin.y, origin.z}, dd[3] = {dir.x, dir.y, dir.z};
			orc_apply_aperture(cam, u, v, o3, dd);
			origin = v3_make(o3[0], o3[1], o3[2]);
			dir = v3_make(dd[0], dd[1], dd[2]);
		}
		origin = v3_add(origin, v3_scale(dir, cam->near_distance));
		depth_buffer[i] = ORC_MAX_DEPTH;
		if (dir.x == 0.0f && dir.y == 0.0f && dir.z == 0.0f) continue;
		dir = v3_normalize(dir);
		float tmin, tmax;
		aabb_ray_intersect(&bb, origin, dir, &tmin, &tmax);
		float t = fmaxf(tmin, 0.0f);
		v3 pos = v3_add(origin, v3_scale(dir, t + 1e-6f));
		v3 primary_dir = dir;
		/* M2 on every ray (trace_mesh_bvh :2140-2155); the normal buffer starts out holding the ray direction */
		v3 normal = dir;
		trace_one(s, &pos, &normal);
		/* M3: prepare_shadow_rays_geometry (:222-271) */
		float shadow = 1.0f;
		{
			float nd = v3_dot(normal, primary_dir);
			v3 ff = nd < 0.0f ? normal : v3_scale(normal, -1.0f); /* faceforward(n, dir, n) */
			v3 view_pos = v3_add(pos, v3_scale(v3_normalize(ff), 1e-3f));
			v3 sdir = v3_normalize(sun);
			float stmin, stmax;
			aabb_ray_intersect(&bb, view_pos, sdir, &stmin, &stmax);
			float st = fmaxf(stmin + 1e-6f, 0.0f);
			view_pos = v3_add(view_pos, v3_scale(sdir, st));
			v3 spos = view_pos;
			v3 strace_dir = primary_dir; /* a shadow ray that starts outside keeps the copied primary payload.dir */
			if (aabb_contains(&bb, view_pos)) strace_dir = sdir;
			v3 snormal = strace_dir;
			trace_one(s, &spos, &snormal);
			/* M4: write_shadow_ray_result_geometry (:273-278), min_visibility == 1 */
			shadow = aabb_contains(&bb, spos) ? 0.0f : 1.0f;
		}
		/* M5: shade_kernel_mesh_geometry (:280-355) */
		if (!aabb_contains(&bb, pos)) continue;
		v3 N = v3_normalize(normal);
		float skyam = -v3_dot(N, up) * 0.5f + 0.5f;
		v3 suncol = v3_scale(v3_scale(v3_make(255.f / 255.0f, 225.f / 255.0f, 195.f / 255.0f), 4.f), shadow);
		v3 skycol = v3_scale(v3_scale(v3_make(195.f / 255.0f, 215.f / 255.0f, 255.f / 255.0f), 4.f), skyam);
		v3 base = v3_make(o->basecolor[0], o->basecolor[1], o->basecolor[2]);
		v3 ambc = v3_mul(v3_make(o->ambientcolor[0], o->ambientcolor[1], o->ambientcolor[2]), skycol);
		if (o->irradiance) {
			float E[3], n3[3] = {N.x, N.y, N.z};
			if (o->grid_x) {
				orc_probe_grid_desc gd = {o->grid_x, o->grid_y, o->n_theta, o->n_phi, 0.0f, {o->probe_center[0], o->probe_center[1], o->probe_center[2]}};
				float p3[3] = {pos.x, pos.y, pos.z};
				orc_irradiance_grid_lookup(&gd, o->irradiance, 1, p3, n3, E);
			} else {
				orc_irradiance_read(o->n_theta, o->n_phi, o->irradiance, n3, E);
			}
			ambc = v3_make(E[0] / PI_F, E[1] / PI_F, E[2] / PI_F);
		}
		v3 color = evaluate_shading(v3_mul(base, base), ambc, suncol, o->metallic, o->subsurface, o->specular, o->roughness, 0.f,
		                            o->sheen, 0.f, o->clearcoat, o->clearcoat_gloss, sun, v3_scale(v3_normalize(primary_dir), -1.0f), N);
		frame_buffer[4 * i + 0] = color.x;
		frame_buffer[4 * i + 1] = color.y;
		frame_buffer[4 * i + 2] = color.z;
		frame_buffer[4 * i + 3] = 1.0f;
		depth_buffer[i] = v3_dot(cam_fwd, v3_sub(pos, cam_pos));
	}
}
