// host_api.cpp -- C view (include/rt_host.h) of the host-side mirror in ray_tracing.hpp.
#include <cmath>
#include <cstring>
#include <exception>
#include <string>

#include "../../include/rt_host.h"
#include "ray_tracing.hpp"

using namespace ray_tracing;

struct rth_app {
    App app;
    RayTracingPlugin plugin;
    std::string err;
};

namespace {
Vec3 V3(const float *p) { Vec3 v; v.x = p[0]; v.y = p[1]; v.z = p[2]; return v; }
GlobalTransform make_transform(const float scale[3], const float rot[4], const float trans[3]) {
    GlobalTransform t;
    if (scale) t.scale = V3(scale);
    if (rot) { t.rotation.x = rot[0]; t.rotation.y = rot[1]; t.rotation.z = rot[2]; t.rotation.w = rot[3]; }
    if (trans) t.translation = V3(trans);
    return t;
}
int push_mesh(rth_app *a, Mesh m) { a->app.world.meshes.push_back(std::move(m)); a->app.world.asset_generation++; return (int)a->app.world.meshes.size() - 1; }
}  // namespace

extern "C" {

rth_app *rth_app_create(void) {
    rth_app *a = new (std::nothrow) rth_app();
    if (!a) return nullptr;
    a->plugin.build(a->app);
    return a;
}
void rth_app_destroy(rth_app *a) {
    if (!a) return;
    a->app.pipeline.destroy();
    delete a;
}
int rth_app_finish(rth_app *a, int device_ordinal) {
    if (!a) return RT_ERR_ARG;
    a->app.device_ordinal = device_ordinal;
    try {
        a->plugin.finish(a->app);
    } catch (const std::exception &e) {
        a->err = e.what();
        return RT_ERR_ARG;
    }
    if (a->app.pipeline.status != RT_OK) a->err = rt_last_error(nullptr);
    return a->app.pipeline.status;
}
int rth_app_update(rth_app *a, const rt_render_params *params) {
    if (!a) return RT_ERR_ARG;
    try {
        int rc = a->app.update(params);
        if (rc < 0 && a->app.pipeline.ctx) a->err = rt_last_error(a->app.pipeline.ctx);
        return rc;
    } catch (const std::exception &e) {
        a->err = e.what();
        return RT_ERR_SCENE;
    }
}
rt_ctx *rth_app_ctx(rth_app *a) { return a ? a->app.pipeline.ctx : nullptr; }
const char *rth_last_error(rth_app *a) { return a ? a->err.c_str() : ""; }

int rth_add_mesh(rth_app *a, const float *positions, const float *normals, uint32_t n_vertices,
                 const uint32_t *indices, uint32_t n_indices) {
    if (!a) return -1;
    Mesh m;
    m.has_positions = positions != nullptr;
    m.has_normals = normals != nullptr;
    m.has_indices = indices != nullptr;
    if (positions) for (uint32_t i = 0; i < n_vertices; ++i) m.positions.push_back(V3(positions + 3 * i));
    if (normals) for (uint32_t i = 0; i < n_vertices; ++i) m.normals.push_back(V3(normals + 3 * i));
    if (indices) m.indices.assign(indices, indices + n_indices);
    return push_mesh(a, std::move(m));
}
int rth_add_cuboid(rth_app *a, float sx, float sy, float sz) { return a ? push_mesh(a, cuboid_mesh(sx, sy, sz)) : -1; }
int rth_add_plane(rth_app *a, float hx, float hz) { return a ? push_mesh(a, plane_mesh(hx, hz)) : -1; }
int rth_add_icosphere(rth_app *a, float r, uint32_t s) { return a ? push_mesh(a, icosphere_mesh(r, s)) : -1; }
int rth_add_material(rth_app *a, const float c[4]) {
    if (!a || !c) return -1;
    StandardMaterial m;
    std::memcpy(m.base_color_linear, c, 16);
    a->app.world.materials.push_back(m);
    a->app.world.asset_generation++;
    return (int)a->app.world.materials.size() - 1;
}
int rth_spawn(rth_app *a, int mesh, int material, const float scale[3], const float rot[4], const float trans[3]) {
    if (!a) return -1;
    Entity e;
    e.mesh = mesh; e.material = material;
    e.transform = make_transform(scale, rot, trans);
    if (mesh >= 0 && (size_t)mesh < a->app.world.meshes.size()) e.aabb = compute_aabb(a->app.world.meshes[mesh]);
    a->app.world.entities.push_back(e);
    return (int)a->app.world.entities.size() - 1;
}
int rth_set_transform(rth_app *a, int entity, const float scale[3], const float rot[4], const float trans[3]) {
    if (!a || entity < 0 || (size_t)entity >= a->app.world.entities.size()) return -1;
    a->app.world.entities[entity].transform = make_transform(scale, rot, trans);
    return 0;
}
int rth_set_aabb(rth_app *a, int entity, const float center[3], const float half[3]) {
    if (!a || entity < 0 || (size_t)entity >= a->app.world.entities.size() || !center || !half) return -1;
    a->app.world.entities[entity].aabb.center = V3(center);
    a->app.world.entities[entity].aabb.half_extents = V3(half);
    return 0;
}
void rth_clear_world(rth_app *a) {
    if (!a) return;
    a->app.world.meshes.clear(); a->app.world.materials.clear(); a->app.world.entities.clear();
    a->app.world.log.clear();
    a->app.world.asset_generation++;          // whatever comes next is not what prepare_meshinfo baked
}

void rth_spawn_default_scene(rth_app *a) { if (a) spawn_default_scene(a->app.world); }
void rth_spawn_sphere_field(rth_app *a, uint32_t n, uint64_t seed, uint32_t s) { if (a) spawn_sphere_field(a->app.world, n, seed, s); }
void rth_spawn_icosphere_soup(rth_app *a, uint32_t n, uint64_t seed, uint32_t s) { if (a) spawn_icosphere_soup(a->app.world, n, seed, s); }
void rth_rotate_tagged(rth_app *a, float dt) { if (a) rotate_tagged(a->app.world, dt); }

void rth_camera_look_at(rth_app *a, const float eye[3], const float target[3], const float up[3]) {
    if (a) a->app.world.camera.transform = looking_at(V3(eye), V3(target), V3(up));
}
void rth_camera_frame_scene(rth_app *a) { if (a) frame_synthetic_camera(a->app.world); }
void rth_camera_frame_scene_at(rth_app *a, float radii) { if (a) frame_synthetic_camera(a->app.world, radii); }
void rth_camera_set_target(rth_app *a, uint32_t w, uint32_t h, const uint32_t vp[4]) {
    if (!a) return;
    Camera &c = a->app.world.camera;
    c.width = w; c.height = h;
    c.has_viewport = vp != nullptr;
    if (vp) { c.viewport.x = vp[0]; c.viewport.y = vp[1]; c.viewport.w = vp[2]; c.viewport.h = vp[3]; }
}
void rth_camera_set_projection(rth_app *a, float fov_y, float near_plane) {
    if (!a) return;
    a->app.world.camera.fov_y = fov_y; a->app.world.camera.near = near_plane;
}
void rth_camera_set_graph(rth_app *a, int on) { if (a) a->app.world.camera.ray_tracing_graph = on != 0; }

// src/fly_cam.rs:98-176.  MovementSettings defaults (:23-30): sensitivity 0.00012, speed 12.
void rth_fly_cam_step(rth_app *a, float mouse_dx, float mouse_dy, uint32_t keys, float dt) {
    if (!a) return;
    Camera &cam = a->app.world.camera;
    const float sensitivity = 0.00012f, speed = 12.0f;
    // player_look (:145-176): yaw/pitch from the rotation (EulerRot::YXZ), clamp, recompose
    if (mouse_dx != 0.0f || mouse_dy != 0.0f) {
        Vec3 z; z.x = 0; z.y = 0; z.z = 1;
        Vec3 lz = quat_rotate(cam.transform.rotation, z);   // third column of the rotation matrix
        float yaw = std::atan2(lz.x, lz.z);
        float pitch = std::asin(std::fmax(-1.0f, std::fmin(1.0f, -lz.y)));
        float window_scale = (float)(cam.height < cam.width ? cam.height : cam.width);
        const float deg = 3.14159265358979323846f / 180.0f;
        pitch -= (sensitivity * mouse_dy * window_scale) * deg;
        yaw -= (sensitivity * mouse_dx * window_scale) * deg;
        pitch = std::fmax(-1.54f, std::fmin(1.54f, pitch));
        Vec3 Y; Y.x = 0; Y.y = 1; Y.z = 0;
        Vec3 X; X.x = 1; X.y = 0; X.z = 0;
        cam.transform.rotation = quat_mul(quat_from_axis_angle(Y, yaw), quat_from_axis_angle(X, pitch));
    }
    // player_move (:98-142): velocity is renormalised and applied once per pressed key
    Vec3 z; z.x = 0; z.y = 0; z.z = 1;
    Vec3 lz = quat_rotate(cam.transform.rotation, z);
    Vec3 forward; forward.x = -lz.x; forward.y = 0; forward.z = -lz.z;
    Vec3 right; right.x = lz.z; right.y = 0; right.z = -lz.x;
    Vec3 vel;
    const uint32_t order[6] = {RTH_KEY_FORWARD, RTH_KEY_BACKWARD, RTH_KEY_LEFT, RTH_KEY_RIGHT, RTH_KEY_ASCEND, RTH_KEY_DESCEND};
    for (uint32_t k : order) {
        if (!(keys & k)) continue;
        float s = 1.0f; Vec3 d;
        switch (k) {
            case RTH_KEY_FORWARD: d = forward; break;
            case RTH_KEY_BACKWARD: d = forward; s = -1.0f; break;
            case RTH_KEY_LEFT: d = right; s = -1.0f; break;
            case RTH_KEY_RIGHT: d = right; break;
            case RTH_KEY_ASCEND: d.y = 1.0f; break;
            default: d.y = 1.0f; s = -1.0f; break;
        }
        vel.x += s * d.x; vel.y += s * d.y; vel.z += s * d.z;
        float len = std::sqrt(vel.x * vel.x + vel.y * vel.y + vel.z * vel.z);
        if (len > 0.0f && std::isfinite(1.0f / len)) { vel.x /= len; vel.y /= len; vel.z /= len; }
        else { vel.x = vel.y = vel.z = 0.0f; }   // normalize_or_zero
        cam.transform.translation.x += vel.x * dt * speed;
        cam.transform.translation.y += vel.y * dt * speed;
        cam.transform.translation.z += vel.z * dt * speed;
    }
}

void rth_camera_view(rth_app *a, float m[16], float pos[3]) {
    if (!a) return;
    View v = extract_view(a->app.world.camera);
    if (m) std::memcpy(m, v.world_from_clip, 64);
    if (pos) std::memcpy(pos, v.world_position, 12);
}

int rth_prepare_meshinfo(rth_app *a) {
    if (!a) return -1;
    try {
        prepare_meshinfo(a->app.world);
    } catch (const std::exception &e) {
        a->err = e.what();
        return -1;
    }
    return 0;
}
void rth_info_counts(rth_app *a, size_t *nt, size_t *nm, size_t *nv, size_t *nc) {
    if (!a) return;
    const RayTracingInfo &i = a->app.world.ray_tracing_info;
    if (nt) *nt = i.triangles.size();
    if (nm) *nm = i.meshes.size();
    if (nv) *nv = i.vertices.size();
    if (nc) *nc = i.materials.size();
}
void rth_info_copy(rth_app *a, rt_triangle *t, rt_mesh_info *m, rt_vertex *v, rt_material *c) {
    if (!a) return;
    const RayTracingInfo &i = a->app.world.ray_tracing_info;
    if (t && !i.triangles.empty()) std::memcpy(t, i.triangles.data(), i.triangles.size() * sizeof(rt_triangle));
    if (m && !i.meshes.empty()) std::memcpy(m, i.meshes.data(), i.meshes.size() * sizeof(rt_mesh_info));
    if (v && !i.vertices.empty()) std::memcpy(v, i.vertices.data(), i.vertices.size() * sizeof(rt_vertex));
    if (c && !i.materials.empty()) std::memcpy(c, i.materials.data(), i.materials.size() * sizeof(rt_material));
}
const char *rth_log(rth_app *a) { return a ? a->app.world.log.c_str() : ""; }
void rth_update_counts(rth_app *a, uint32_t *range_updates, uint32_t *full_updates) {
    if (range_updates) *range_updates = a ? a->app.node.range_updates : 0;
    if (full_updates) *full_updates = a ? a->app.node.full_updates : 0;
}

size_t rth_entity_count(rth_app *a) { return a ? a->app.world.entities.size() : 0; }
int rth_entity_get(rth_app *a, size_t i, rth_entity_view *o) {
    if (!a || !o || i >= a->app.world.entities.size()) return -1;
    const World &w = a->app.world;
    const Entity &e = w.entities[i];
    std::memset(o, 0, sizeof *o);
    if (e.mesh >= 0 && (size_t)e.mesh < w.meshes.size()) {
        const Mesh &m = w.meshes[e.mesh];
        o->positions = m.has_positions && !m.positions.empty() ? &m.positions[0].x : nullptr;
        o->normals = m.has_normals && !m.normals.empty() ? &m.normals[0].x : nullptr;
        o->n_vertices = (uint32_t)(m.positions.size() < m.normals.size() || !m.has_normals ? m.positions.size() : m.normals.size());
        o->indices = m.has_indices && !m.indices.empty() ? m.indices.data() : nullptr;
        o->n_indices = (uint32_t)m.indices.size();
    }
    if (e.material >= 0 && (size_t)e.material < w.materials.size())
        std::memcpy(o->base_color_linear, w.materials[e.material].base_color_linear, 16);
    o->scale[0] = e.transform.scale.x; o->scale[1] = e.transform.scale.y; o->scale[2] = e.transform.scale.z;
    o->rotation[0] = e.transform.rotation.x; o->rotation[1] = e.transform.rotation.y;
    o->rotation[2] = e.transform.rotation.z; o->rotation[3] = e.transform.rotation.w;
    o->translation[0] = e.transform.translation.x; o->translation[1] = e.transform.translation.y; o->translation[2] = e.transform.translation.z;
    o->aabb_center[0] = e.aabb.center.x; o->aabb_center[1] = e.aabb.center.y; o->aabb_center[2] = e.aabb.center.z;
    o->aabb_half_extents[0] = e.aabb.half_extents.x; o->aabb_half_extents[1] = e.aabb.half_extents.y; o->aabb_half_extents[2] = e.aabb.half_extents.z;
    return 0;
}

}  // extern "C"
