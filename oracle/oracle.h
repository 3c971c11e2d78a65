/*
 * oracle.h -- entry points of the CPU checker (liboracle.so).
 * TEST INFRASTRUCTURE ONLY: see the header of oracle.cpp.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include "../include/rtgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Work counted under the reference's own semantics: every BVH node whose box
 * is tested (mesh.rs:326-345), every triangle of every reached leaf
 * (mesh.rs:365-371), every primitive of the linear object scan
 * (scene.rs:61-66), texels fetched (material.rs:126, skybox.rs:68-75). */
typedef struct orc_counts {
    uint64_t node_tests, tri_tests, sphere_tests, plane_tests, box_tests;
    uint64_t texel_image, texel_cubemap;
    uint64_t would_panic;   /* places where the reference would panic */
    /* the same tests split by the cast they belong to: [0] closest-hit casts of trace_ray (scene.rs:97),
     * [1] shadow casts (sun.rs:89, point.rs:91), [2] the same-object helper intersects of the refraction
     * branch (scene.rs:143-147); columns: node, triangle, sphere, plane, box */
    uint64_t by_ctx[3][5];
} orc_counts;

int rt_cpu_render(const rt_scene_desc* d, int n_threads, uint8_t* rgb_out, uint32_t* prim_id_out,
                  rt_stats* stats_out, orc_counts* counts_out);
int rt_cpu_render_pixels(const rt_scene_desc* d, int n_threads, uint64_t n, const uint32_t* pixels,
                         uint8_t* rgb_out, uint32_t* prim_id_out, rt_stats* stats_out, orc_counts* counts_out);
int rt_cpu_cast_rays(const rt_scene_desc* d, uint32_t n, const double* origins, const double* dirs, rt_ray_hit* out);

int orc_sphere_intersect(const rt_sphere* s, const double* o, const double* d, rt_ray_hit* out);
int orc_plane_intersect(const rt_plane* p, const double* o, const double* d, rt_ray_hit* out);
int orc_box_intersect(const rt_box* b, const double* o, const double* d, rt_ray_hit* out);
int orc_aabb_bool(const double* mn, const double* mx, const double* o, const double* d);
int orc_triangle_intersect(const double* v0, const double* v1, const double* v2, const double* o, const double* d, double* out);
int orc_refraction_vec(const double* d, const double* n, double from_ior, double to_ior, double* out);
void orc_color_newf(double r, double g, double b, uint8_t* out);
void orc_camera_direction_at(const rt_camera* cam, double x, double y, double* out);
double orc_camera_chf(double fov);
void orc_texture_at(const rt_texture* t, float u, float v, uint8_t* out);
void orc_skybox_ray_color(const rt_skybox* sb, const double* d, uint8_t* out);
double orc_powi(double a, int b);
/* area.rs:121 draws from an unseeded generator; the checker's stand-in is seeded, and this sets the seed of later renders */
void orc_set_area_seed(uint64_t seed);
/* negative controls of the tests: bit 0 nearest-texel cubemap sampling, bit 1 two cubemap face-table entries exchanged;
 * 0 (the default) = the reference's behaviour */
void orc_set_mistake(uint32_t flags);

#ifdef __cplusplus
}
#endif
#endif
