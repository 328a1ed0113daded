/*
 * rt_host.h -- C view of the host-side mirror of the reference's plugin interface
 * (ray-tracing_amd/host/ray_tracing.hpp), so tests and bench drivers in any language can
 * drive it: the World the plugin reads, prepare_meshinfo (src/ray_tracing.rs:98-169), the
 * camera's View, the default scene of src/main.rs:51-117 and the synthetic scenes of
 * SURVEY.md section 8d, and RayTracingPlugin / RayTracingPassNode's per-frame lifecycle.
 * The GPU work itself goes through include/rt_amd.h.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include <stddef.h>
#include <stdint.h>
#include "rt_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rth_app rth_app;   /* App + World + RayTracingPlugin state */

/* App::new().add_plugins(RayTracingPlugin): Plugin::build (src/ray_tracing.rs:29-42).
 * No GPU is touched until rth_app_finish. */
rth_app *rth_app_create(void);
void rth_app_destroy(rth_app *app);
/* Plugin::finish (src/ray_tracing.rs:44-48): creates the RayTracingPipeline on a device.
 * Returns the rt_create status. */
int rth_app_finish(rth_app *app, int device_ordinal);
/* One frame: ExtractSchedule (prepare_meshinfo) then RayTracingPassNode::run (src/node.rs:29-79).
 * RT_NOT_READY = frame skipped silently, as the reference does. */
int rth_app_update(rth_app *app, const rt_render_params *params);
rt_ctx *rth_app_ctx(rth_app *app);
const char *rth_last_error(rth_app *app);

/* ---- world building (what main.rs `setup` does through Commands/Assets) ---------------- */
int rth_add_mesh(rth_app *app, const float *positions, const float *normals, uint32_t n_vertices,
                 const uint32_t *indices, uint32_t n_indices);   /* NULL positions/normals/indices = missing */
int rth_add_cuboid(rth_app *app, float sx, float sy, float sz);
int rth_add_plane(rth_app *app, float half_x, float half_z);
int rth_add_icosphere(rth_app *app, float radius, uint32_t subdivisions);
int rth_add_material(rth_app *app, const float base_color_linear[4]);
/* scale[3], rotation quaternion xyzw[4], translation[3]; Aabb computed from the mesh */
int rth_spawn(rth_app *app, int mesh, int material, const float scale[3], const float rotation[4],
              const float translation[3]);
int rth_set_transform(rth_app *app, int entity, const float scale[3], const float rotation[4],
                      const float translation[3]);
int rth_set_aabb(rth_app *app, int entity, const float center[3], const float half_extents[3]);
void rth_clear_world(rth_app *app);

void rth_spawn_default_scene(rth_app *app);                                         /* src/main.rs:51-117 */
void rth_spawn_sphere_field(rth_app *app, uint32_t n, uint64_t seed, uint32_t subdivisions);   /* config 2 */
void rth_spawn_icosphere_soup(rth_app *app, uint32_t n, uint64_t seed, uint32_t subdivisions); /* config 3/5 */
void rth_rotate_tagged(rth_app *app, float dt);                                     /* src/main.rs:122-126 */

/* ---- camera ---------------------------------------------------------------------------------- */
void rth_camera_look_at(rth_app *app, const float eye[3], const float target[3], const float up[3]);
void rth_camera_frame_scene(rth_app *app);                 /* eye at 2.5 bounding radii */
void rth_camera_frame_scene_at(rth_app *app, float radii); /* same, at a chosen distance */
void rth_camera_set_target(rth_app *app, uint32_t width, uint32_t height, const uint32_t viewport[4] /* or NULL */);
void rth_camera_set_projection(rth_app *app, float fov_y, float near_plane);
void rth_camera_set_graph(rth_app *app, int ray_tracing_graph);   /* Tab toggle, src/main.rs:128-147 */
/* fly_cam restatement (src/fly_cam.rs:98-176): mouse delta in pixels, keys as a bit mask */
#define RTH_KEY_FORWARD 1u
#define RTH_KEY_BACKWARD 2u
#define RTH_KEY_LEFT 4u
#define RTH_KEY_RIGHT 8u
#define RTH_KEY_ASCEND 16u
#define RTH_KEY_DESCEND 32u
void rth_fly_cam_step(rth_app *app, float mouse_dx, float mouse_dy, uint32_t keys, float dt);
/* The same driven from text, one step per line: "dt mouse_dx mouse_dy [keys]", keys from
 * W A S D (src/fly_cam.rs:47-50), ^ = Space / ascend (:51), v = ShiftLeft / descend (:52), - = none; # comments.
 * Returns the number of steps run, or -(line number) of the first malformed line. */
long rth_fly_cam_run_script(rth_app *app, const char *text);
/* View uniform pieces: world_from_clip[16] column-major, world_position[3] */
void rth_camera_view(rth_app *app, float world_from_clip[16], float world_position[3]);

/* ---- extraction result: the RayTracingInfo resource --------------------------------------------- */
int rth_prepare_meshinfo(rth_app *app);   /* 0 ok, -1 = the reference would panic (rth_last_error) */
void rth_info_counts(rth_app *app, size_t *n_tri, size_t *n_mesh, size_t *n_vtx, size_t *n_mat);
void rth_info_copy(rth_app *app, rt_triangle *tris, rt_mesh_info *meshes, rt_vertex *verts, rt_material *mats);
const char *rth_log(rth_app *app);
/* How RayTracingPassNode::run applied vertex-only changes so far: rt_update_vertex_range (one run of moved meshes,
 * SURVEY 8f item 1) or rt_update_vertices (everything). */
void rth_update_counts(rth_app *app, uint32_t *range_updates, uint32_t *full_updates);

/* ---- presentation to files: the BGRA8 sRGB frame of rt_read_bgra8_srgb, i.e. what the
 * reference's Bgra8UnormSrgb target holds (src/pipeline.rs:56-60), as binary PPM (RGB) or PNG (RGBA, sRGB chunk,
 * stored deflate).  0 ok, -1 bad argument, -2 I/O error.  Host only. */
int rth_write_ppm(const char *path, const uint8_t *bgra, uint32_t width, uint32_t height);
int rth_write_png(const char *path, const uint8_t *bgra, uint32_t width, uint32_t height);

/* ---- entity introspection (to feed the same entities to an independent checker) ---------------- */
size_t rth_entity_count(rth_app *app);
typedef struct {
    const float *positions; const float *normals; uint32_t n_vertices;
    const uint32_t *indices; uint32_t n_indices;
    float base_color_linear[4];
    float scale[3]; float rotation[4]; float translation[3];
    float aabb_center[3]; float aabb_half_extents[3];
} rth_entity_view;
int rth_entity_get(rth_app *app, size_t i, rth_entity_view *out);

#ifdef __cplusplus
}
#endif
#endif
