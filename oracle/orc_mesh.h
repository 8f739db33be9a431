/* ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_common.h). PARITY UNPINNED. Mesh / geometry-mode interface. */
#ifndef ORC_MESH_H
#define ORC_MESH_H
#include "oracle.h"
#include "orc_common.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { v3 a, b, c; } orc_triangle;                          /* triangle.cuh:163, 36 B */
typedef struct { aabb_t bb; int left_idx, right_idx; } orc_bvh_node;  /* triangle_bvh.cuh:28-32, 32 B */
typedef struct orc_mesh_scene orc_mesh_scene;

typedef struct orc_mesh_opts { /* BRDFParams (common.h:167-177) + m_sun_dir, m_up_dir (testbed.h:875-876) */
	float sun_dir[3], up_dir[3];
	float metallic, subsurface, specular, roughness, sheen, clearcoat, clearcoat_gloss;
	float basecolor[3], ambientcolor[3];
	/* ShadeEnvMap: ambient light = E(N)/pi read bilinearly (orc_irradiance_read) from an irradiance map tabulated at the
	 * probe texture's texel directions (n_theta*n_phi*4 floats); NULL = Shade mode (sky ambient).
	 * ShadeGridEnvMap (grid_x > 0): `irradiance` holds grid_x*grid_y such maps and the light at a surface point comes
	 * from the probes around the direction of (point - probe_center): orc_irradiance_grid_lookup */
	const float* irradiance;
	uint32_t n_theta, n_phi;
	uint32_t grid_x, grid_y;
	float probe_center[3];
} orc_mesh_opts;

/* triangles are reordered in place */
int orc_bvh_build(orc_triangle* tris, uint32_t n_tris, uint32_t n_primitives_per_leaf, orc_bvh_node** out_nodes, uint32_t* out_n_nodes);
void orc_bvh_ray_intersect(const orc_bvh_node* nodes, const orc_triangle* tris, const float* ro3, const float* rd3, int* out_idx, float* out_t);

/* vertices[m]: n_tris[m]*9 floats in file space; centers: n_meshes*3 */
orc_mesh_scene* orc_mesh_scene_create(uint32_t n_meshes, const float* const* vertices, const uint32_t* n_tris, const float* centers);
void orc_mesh_scene_destroy(orc_mesh_scene* s);
void orc_mesh_scene_aabb(const orc_mesh_scene* s, float* out6);
uint32_t orc_mesh_scene_n_nodes(const orc_mesh_scene* s, uint32_t mesh);
const orc_triangle* orc_mesh_scene_triangles(const orc_mesh_scene* s, uint32_t mesh);
const orc_bvh_node* orc_mesh_scene_nodes(const orc_mesh_scene* s, uint32_t mesh);
/* M2: positions/directions n x 3 in place (direction <- face normal on a hit) */
void orc_trace_mesh(const orc_mesh_scene* s, uint32_t n, float* positions, float* directions);
/* M1-M5: writes frame_buffer (W*H*4) / depth_buffer (W*H) exactly like render_geometry_mesh */
void orc_render_mesh(const orc_mesh_scene* s, const orc_camera* cam, const orc_mesh_opts* o, float* frame_buffer, float* depth_buffer);

#ifdef __cplusplus
}
#endif
#endif
