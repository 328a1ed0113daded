// ray_tracing.hpp -- host-side mirror of the reference's plugin interface for the hot path,
// in C++ because the reference is compiled code and no Rust toolchain exists in this image.
// Same names, argument meaning and error behaviour as the Rust items they stand for:
//   Triangle / MeshInfo / SimpleMaterial / RayTracingInfo   src/ray_tracing.rs:51-96
//   prepare_meshinfo                                         src/ray_tracing.rs:98-169
//   RayTracingPlugin::build / finish, RayTracingGraph        src/ray_tracing.rs:23-49
//   RayTracingPipeline::from_world                           src/pipeline.rs:19-72
//   RayTracingPassNode::run                                  src/node.rs:18-80
// The ECS, asset server and render graph of Bevy are NOT rebuilt: World/App below carry just
// the data those items read (mesh + material assets, entities with transform and Aabb, one
// camera), nothing more.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"

namespace ray_tracing {

struct Vec3 { float x = 0, y = 0, z = 0; };
struct Quat { float x = 0, y = 0, z = 0, w = 1; };

// bevy Mesh, reduced to what prepare_meshinfo reads (positions, normals, indices)
struct Mesh {
    std::vector<Vec3> positions;
    std::vector<Vec3> normals;
    std::vector<uint32_t> indices;
    bool has_positions = true, has_normals = true, has_indices = true;
};
// bevy render::primitives::Aabb
struct Aabb { Vec3 center, half_extents; };
Aabb compute_aabb(const Mesh &m);

struct StandardMaterial { float base_color_linear[4] = {1, 1, 1, 1}; };
struct GlobalTransform { Vec3 scale{1, 1, 1}; Quat rotation; Vec3 translation; };

struct Entity {
    int mesh = -1, material = -1;   // handles into World::meshes / World::materials
    GlobalTransform transform;
    Aabb aabb;
    bool rotate_tag = false;        // the `Rotate` marker of src/main.rs:119-126
};

struct Viewport { uint32_t x = 0, y = 0, w = 0, h = 0; };
struct Camera {
    GlobalTransform transform;      // only rotation + translation are used
    float fov_y = 0.78539816339744830962f;   // PerspectiveProjection default pi/4
    float near = 0.1f;
    uint32_t width = 1280, height = 720;
    bool has_viewport = false;
    Viewport viewport;
    bool ray_tracing_graph = true;  // CameraRenderGraph::new(RayTracingGraph), src/main.rs:60
};

// src/ray_tracing.rs:51-96.  Layouts are the C ABI's (byte-identical to the WGSL bindings).
using Triangle = rt_triangle;
using MeshInfo = rt_mesh_info;
using Vertex = rt_vertex;
using SimpleMaterial = rt_material;
struct RayTracingInfo {
    uint32_t count = 0;
    std::vector<Triangle> triangles;
    std::vector<MeshInfo> meshes;
    std::vector<Vertex> vertices;
    std::vector<SimpleMaterial> materials;
};

struct World {
    std::vector<Mesh> meshes;
    std::vector<StandardMaterial> materials;
    std::vector<Entity> entities;     // query order = spawn order (src/ray_tracing.rs:119)
    Camera camera;
    RayTracingInfo ray_tracing_info;  // the resource prepare_meshinfo inserts
    std::string log;                  // "Mesh missing attribute" lines (src/ray_tracing.rs:127)

    // f1 ("re-bake only moved meshes"), host side.  prepare_meshinfo remembers what it baked per
    // entity; while the entity list, its asset handles and the assets themselves stay the same it re-bakes only the
    // entities whose transform or Aabb changed, in place -- the resource it leaves is the one a full pass would leave.
    // Code that edits `meshes` / `materials` in place must bump `asset_generation` (the host API does).
    struct Baked {
        int mesh, material;
        GlobalTransform transform;
        Aabb aabb;
        uint32_t first_vertex, n_vertices;
        int mesh_info;                // index of its MeshInfo, -1 if the entity was skipped (missing attribute)
    };
    std::vector<Baked> baked;
    uint64_t asset_generation = 0, baked_asset_generation = ~0ull;
    size_t baked_meshes = 0, baked_materials = 0;
    uint64_t info_topology_generation = 0;   // +1 with every full rebuild of ray_tracing_info
    uint64_t info_generation = 0;            // +1 with every change of ray_tracing_info (0: never written by prepare_meshinfo)
    std::vector<uint32_t> moved_meshes;      // MeshInfo indices re-baked in place since the pass node last looked
};

// src/ray_tracing.rs:98-169.  Throws std::runtime_error where the reference panics
// (`mesh.indices().unwrap()` on an unindexed mesh, :145; a dangling handle, :120-121).
void prepare_meshinfo(World &world);

// View uniform pieces the shader reads (bevy_render View: world_from_clip, world_position).
struct View {
    float world_from_clip[16];
    float world_position[3];
    uint32_t width, height;
    bool has_viewport;
    Viewport viewport;
};
View extract_view(const Camera &cam);

// camera helpers (glam restatements; un-vendored dependency, parity unpinned)
Quat quat_from_axis_angle(Vec3 axis, float angle);
Quat quat_mul(Quat a, Quat b);
Vec3 quat_rotate(Quat q, Vec3 v);
GlobalTransform looking_at(Vec3 eye, Vec3 target, Vec3 up);

// mesh generators standing in for bevy's Cuboid / Plane3d / Sphere::mesh() (ico, 5 subdivisions)
Mesh cuboid_mesh(float sx, float sy, float sz);
Mesh plane_mesh(float half_x, float half_z);
Mesh icosphere_mesh(float radius, uint32_t subdivisions);

// scenes
void spawn_default_scene(World &world);                                  // src/main.rs:51-117
void spawn_sphere_field(World &world, uint32_t n, uint64_t seed, uint32_t subdivisions);
void spawn_icosphere_soup(World &world, uint32_t n, uint64_t seed, uint32_t subdivisions);
void frame_synthetic_camera(World &world, float radii = 2.5f);   // eye at `radii` bounding radii, looking at the centroid
void rotate_tagged(World &world, float dt);  // `rotate` system, src/main.rs:122-126

// src/pipeline.rs:19-72: one-time GPU-side init (here: the rt_ctx).
struct RayTracingPipeline {
    rt_ctx *ctx = nullptr;
    int status = RT_NOT_READY;
    static RayTracingPipeline from_world(World &world, int device_ordinal);
    void destroy();
};

// src/node.rs:18-80
struct RayTracingPassNode {
    uint64_t uploaded_generation = 0, uploaded_topology_generation = 0;   // World::info_*_generation the device holds
    uint64_t uploaded_signature = 0;   // scene re-upload only when the arrays change
    uint64_t topology_signature = 0;   // triangles + mesh ranges + materials: unchanged => refit instead of rebuild
    // what the device holds, to find the meshes that moved since (dirty-mesh update, rt_update_vertex_range)
    std::vector<Vertex> uploaded_vertices;
    std::vector<MeshInfo> uploaded_meshes;
    std::vector<uint32_t> mesh_vlo, mesh_vhi;   // per mesh: span of vertex indices its triangles use
    uint32_t range_updates = 0, full_updates = 0;   // how the vertex-only changes were applied (for tests)
    // Returns RT_OK, or RT_NOT_READY when the pipeline is not ready (silent skip, :52-56).
    int run(const View &view, World &world, RayTracingPipeline &pipeline, const rt_render_params *params);
};

struct RayTracingGraph {};   // RenderSubGraph label a camera opts into (src/ray_tracing.rs:23-24)

struct App {
    World world;
    bool has_render_app = true;
    bool extract_registered = false, graph_registered = false;
    RayTracingPipeline pipeline;
    RayTracingPassNode node;
    int device_ordinal = 0;
    // one frame of the RayTracingGraph for the camera: extract, then the pass node
    int update(const rt_render_params *params);
};

// src/ray_tracing.rs:26-49
struct RayTracingPlugin {
    void build(App &app) const;    // registers the resource, the extract system, the sub-graph
    void finish(App &app) const;   // init_resource::<RayTracingPipeline>()
};

}  // namespace ray_tracing
