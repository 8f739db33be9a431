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
void orc_irradiance(uint32_t n_theta, uint32_t n_phi, const float* envmap, uint32_t n, const float* normals, float* out_rgb);
#ifdef __cplusplus
}
#endif
#endif
