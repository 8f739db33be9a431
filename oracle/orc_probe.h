/* ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_common.h). PARITY UNPINNED. Irradiance probes. */
#ifndef ORC_PROBE_H
#define ORC_PROBE_H
#include "oracle.h"
#ifdef __cplusplus
extern "C" {
#endif
enum { ORC_PROBE_CENTER = 0, ORC_PROBE_CENTER_OUTWARD = 1, ORC_PROBE_MULTI_CENTER = 2 };
typedef struct orc_probe_desc {
	int32_t mode;
	uint32_t n_theta, n_phi, n_origin;
	float origin[3]; /* ORC_PROBE_CENTER_OUTWARD: the shell position */
} orc_probe_desc;
uint32_t orc_probe_n_rays(const orc_probe_desc* d);
void orc_probe_payloads(const orc_nerf_model* m, const orc_probe_desc* d, orc_payload* out);
/* envmap: n_theta*n_phi*4 floats, texel idx = i_theta + n_theta*j_phi */
void orc_compute_envmap(const orc_nerf_model* m, const orc_probe_desc* d, const orc_render_opts* o, float* envmap, orc_render_stats* stats);
void orc_texel_direction(uint32_t n_theta, uint32_t n_phi, uint32_t i, uint32_t j, float* out3);
/* E(n) = sum_texels L(w) max(0, n.w) dOmega with w = the direction the texel's ray travelled: the texel direction itself for
 * the centre fans (origin3 == NULL), -frame(normalize(origin)) * texel direction for an outward probe at origin3 (K11) */
void orc_irradiance(uint32_t n_theta, uint32_t n_phi, const float* envmap, uint32_t n, const float* normals, float* out_rgb);
void orc_irradiance_from(uint32_t n_theta, uint32_t n_phi, const float* envmap, const float* origin3, uint32_t n, const float* normals, float* out_rgb);

/* ---- the grid of probes: Testbed::computeEnvmapGrid (declared testbed.h:743, called src/main.cu:187-188, no body in the
 * reference), gridSize + m_envmap_tex (testbed.h:949-950), render mode ShadeGridEnvMap (common.h:63). DEFINITION FOR THE
 * BUILD (the reference ships the ray generator K11 and the tracer, nothing else):
 *   shell positions  p(i,j) = c + R * cylindrical_to_dir_nerf((i + 0.5) / grid_x, (j + 0.5) / grid_y), c = the NeRF's render_aabb.center()
 *   probe (i,j)      = the K11 fan from p(i,j) (init_rays_from_center_outward_with_payload_kernel_nerf, :1611-1673), traced
 *                      with the capped tracer like every probe; texture g = i + grid_x * j, n_theta x n_phi texels
 *   E_g(n)           tabulated at the texel directions n_ab = cylindrical_to_dir_nerf(a / n_theta, b / n_phi) (world space)
 *   lookup(x, N)     = bilinear over the four probes around the direction of x - c (theta clamped, phi periodic) of the
 *                      bilinear read of each probe's table at N (theta clamped, phi periodic; the scheme of read_envmap,
 *                      envmap.cuh:24-50) */
typedef struct orc_probe_grid_desc {
	uint32_t grid_x, grid_y, n_theta, n_phi;
	float shell_radius;
	float center[3]; /* c: the centre of the NeRF's own render box (in Geometry mode the box the rays are traced in is the inflated scene box, load_scene) */
} orc_probe_grid_desc;
void orc_probe_grid_origin(const orc_probe_grid_desc* d, uint32_t g, float* out3);
/* envmaps: grid_x*grid_y probe textures of n_theta*n_phi*4 floats */
void orc_compute_envmap_grid(const orc_nerf_model* m, const orc_probe_grid_desc* d, const orc_render_opts* o, float* envmaps, orc_render_stats* stats);
/* tables: the same shape, E_g at the texel directions */
void orc_irradiance_grid_tabulate(const orc_probe_grid_desc* d, const float* envmaps, float* tables);
/* bilinear read of one tabulated map (n_theta*n_phi*4) at direction n */
void orc_irradiance_read(uint32_t n_theta, uint32_t n_phi, const float* table, const float* n3, float* out_rgb);
/* the full lookup at surface points: positions / normals n x 3 -> rgb n x 3 */
void orc_irradiance_grid_lookup(const orc_probe_grid_desc* d, const float* tables, uint32_t n, const float* positions, const float* normals, float* out_rgb);
#ifdef __cplusplus
}
#endif
#endif
