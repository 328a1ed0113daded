// ray_tracing.cpp -- host-side mirror of the reference's plugin interface (see ray_tracing.hpp).
// Built with -ffp-contract=off: the arithmetic here is part of the parity contract with
// oracle/rt_oracle.c (prepare_meshinfo restatement).
#include "ray_tracing.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <stdexcept>
#include <thread>
#include <tuple>

namespace ray_tracing {

namespace {
inline Vec3 v3(float x, float y, float z) { Vec3 r; r.x = x; r.y = y; r.z = z; return r; }
inline Vec3 add(Vec3 a, Vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 sub(Vec3 a, Vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 scl(Vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline Vec3 normalize(Vec3 a) { float l = std::sqrt(dot(a, a)); return v3(a.x / l, a.y / l, a.z / l); }

// glam Mat3::from_quat (glam 0.27.0, Cargo.lock:2129-2130; un-vendored, restated)
void mat3_from_quat(Quat q, Vec3 &c0, Vec3 &c1, Vec3 &c2) {
    float x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
    float xx = q.x * x2, xy = q.x * y2, xz = q.x * z2, yy = q.y * y2, yz = q.y * z2, zz = q.z * z2;
    float wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
    c0 = v3(1.0f - (yy + zz), xy + wz, xz - wy);
    c1 = v3(xy - wz, 1.0f - (xx + zz), yz + wx);
    c2 = v3(xz + wy, yz - wx, 1.0f - (xx + yy));
}
inline Vec3 mat3_mul(Vec3 c0, Vec3 c1, Vec3 c2, Vec3 p) { return add(add(scl(c0, p.x), scl(c1, p.y)), scl(c2, p.z)); }
inline Vec3 vabs(Vec3 a) { return v3(std::fabs(a.x), std::fabs(a.y), std::fabs(a.z)); }

struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    float uniform(float lo, float hi) {   // 24-bit mantissa, [lo, hi)
        float u = (float)(next() >> 40) * (1.0f / 16777216.0f);
        return lo + (hi - lo) * u;
    }
};
}  // namespace

// ---- glam restatements -----------------------------------------------------------------------
Quat quat_from_axis_angle(Vec3 axis, float angle) {
    float s = std::sin(angle * 0.5f), c = std::cos(angle * 0.5f);
    Quat q; q.x = axis.x * s; q.y = axis.y * s; q.z = axis.z * s; q.w = c;
    return q;
}
Quat quat_mul(Quat a, Quat b) {
    Quat q;
    q.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    q.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    q.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    q.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    return q;
}
Vec3 quat_rotate(Quat q, Vec3 r) {   // glam Quat * Vec3 (scalar path)
    Vec3 b = v3(q.x, q.y, q.z);
    float b2 = dot(b, b);
    return add(add(scl(r, q.w * q.w - b2), scl(b, dot(r, b) * 2.0f)), scl(cross(b, r), q.w * 2.0f));
}
static Quat quat_from_axes(Vec3 x, Vec3 y, Vec3 z) {   // glam Quat::from_rotation_axes
    float m00 = x.x, m01 = x.y, m02 = x.z, m10 = y.x, m11 = y.y, m12 = y.z, m20 = z.x, m21 = z.y, m22 = z.z;
    Quat q;
    if (m22 <= 0.0f) {
        float dif10 = m11 - m00, omm22 = 1.0f - m22;
        if (dif10 <= 0.0f) {
            float four_xsq = omm22 - dif10, inv4x = 0.5f / std::sqrt(four_xsq);
            q.x = four_xsq * inv4x; q.y = (m01 + m10) * inv4x; q.z = (m02 + m20) * inv4x; q.w = (m12 - m21) * inv4x;
        } else {
            float four_ysq = omm22 + dif10, inv4y = 0.5f / std::sqrt(four_ysq);
            q.x = (m01 + m10) * inv4y; q.y = four_ysq * inv4y; q.z = (m12 + m21) * inv4y; q.w = (m20 - m02) * inv4y;
        }
    } else {
        float sum10 = m11 + m00, opm22 = 1.0f + m22;
        if (sum10 <= 0.0f) {
            float four_zsq = opm22 - sum10, inv4z = 0.5f / std::sqrt(four_zsq);
            q.x = (m02 + m20) * inv4z; q.y = (m12 + m21) * inv4z; q.z = four_zsq * inv4z; q.w = (m01 - m10) * inv4z;
        } else {
            float four_wsq = opm22 + sum10, inv4w = 0.5f / std::sqrt(four_wsq);
            q.x = (m12 - m21) * inv4w; q.y = (m20 - m02) * inv4w; q.z = (m01 - m10) * inv4w; q.w = four_wsq * inv4w;
        }
    }
    return q;
}
GlobalTransform looking_at(Vec3 eye, Vec3 target, Vec3 up) {   // Transform::looking_at, src/main.rs:61
    Vec3 back = normalize(sub(eye, target));     // camera looks down -Z
    Vec3 right = normalize(cross(up, back));
    Vec3 upv = cross(back, right);
    GlobalTransform t;
    t.translation = eye;
    t.rotation = quat_from_axes(right, upv, back);
    return t;
}

// bevy View: world_from_clip = world_from_view * inverse(clip_from_view), with
// clip_from_view = Mat4::perspective_infinite_reverse_rh(fov_y, aspect, near).  Its inverse maps
// clip (x, y, z, w) to view (x*aspect/f, y/f, -w, z/near), f = 1/tan(fov_y/2).
View extract_view(const Camera &cam) {
    View v{};
    uint32_t w = cam.has_viewport ? cam.viewport.w : cam.width;
    uint32_t h = cam.has_viewport ? cam.viewport.h : cam.height;
    float aspect = (float)w / (float)h;
    float f = 1.0f / std::tan(0.5f * cam.fov_y);
    Vec3 c0, c1, c2;
    mat3_from_quat(cam.transform.rotation, c0, c1, c2);
    Vec3 t = cam.transform.translation;
    // inverse projection columns (view-space 4-vectors): col0=(a/f,0,0,0) col1=(0,1/f,0,0)
    // col2=(0,0,0,1/near) col3=(0,0,-1,0); world_from_view = [c0 c1 c2 t; 0 0 0 1]
    float ix = aspect / f, iy = 1.0f / f, iw = 1.0f / cam.near;
    float *m = v.world_from_clip;
    m[0] = c0.x * ix; m[1] = c0.y * ix; m[2] = c0.z * ix; m[3] = 0.0f;
    m[4] = c1.x * iy; m[5] = c1.y * iy; m[6] = c1.z * iy; m[7] = 0.0f;
    m[8] = t.x * iw;  m[9] = t.y * iw;  m[10] = t.z * iw; m[11] = iw;
    m[12] = -c2.x;    m[13] = -c2.y;    m[14] = -c2.z;    m[15] = 0.0f;
    v.world_position[0] = t.x; v.world_position[1] = t.y; v.world_position[2] = t.z;
    v.width = cam.width; v.height = cam.height;
    v.has_viewport = cam.has_viewport; v.viewport = cam.viewport;
    return v;
}

// ---- meshes ---------------------------------------------------------------------------------------
Aabb compute_aabb(const Mesh &m) {   // bevy Mesh::compute_aabb: centre/half-extents of the positions
    Aabb a;
    if (m.positions.empty()) return a;
    Vec3 lo = m.positions[0], hi = m.positions[0];
    for (const Vec3 &p : m.positions) {
        lo = v3(std::min(lo.x, p.x), std::min(lo.y, p.y), std::min(lo.z, p.z));
        hi = v3(std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z));
    }
    a.center = scl(add(lo, hi), 0.5f);
    a.half_extents = scl(sub(hi, lo), 0.5f);
    return a;
}

Mesh cuboid_mesh(float sx, float sy, float sz) {   // stands in for bevy Cuboid::new(..).mesh(): 24 verts, 12 tris
    Mesh m;
    float hx = sx * 0.5f, hy = sy * 0.5f, hz = sz * 0.5f;
    struct Face { Vec3 n, u, v; };
    const Face faces[6] = {
        {v3(0, 0, 1), v3(1, 0, 0), v3(0, 1, 0)},  {v3(0, 0, -1), v3(-1, 0, 0), v3(0, 1, 0)},
        {v3(1, 0, 0), v3(0, 0, -1), v3(0, 1, 0)}, {v3(-1, 0, 0), v3(0, 0, 1), v3(0, 1, 0)},
        {v3(0, 1, 0), v3(1, 0, 0), v3(0, 0, -1)}, {v3(0, -1, 0), v3(1, 0, 0), v3(0, 0, 1)},
    };
    Vec3 h = v3(hx, hy, hz);
    for (const Face &f : faces) {
        uint32_t base = (uint32_t)m.positions.size();
        const float su[4] = {-1, 1, 1, -1}, sv[4] = {-1, -1, 1, 1};
        for (int k = 0; k < 4; ++k) {
            Vec3 d = add(add(f.n, scl(f.u, su[k])), scl(f.v, sv[k]));
            m.positions.push_back(v3(d.x * h.x, d.y * h.y, d.z * h.z));
            m.normals.push_back(f.n);
        }
        const uint32_t idx[6] = {0, 1, 2, 2, 3, 0};   // CCW seen from outside (u x v = n)
        for (uint32_t i : idx) m.indices.push_back(base + i);
    }
    return m;
}

Mesh plane_mesh(float half_x, float half_z) {   // Plane3d::new(Vec3::Y, half_size): 4 verts, 2 tris, facing +Y
    Mesh m;
    m.positions = {v3(-half_x, 0, -half_z), v3(-half_x, 0, half_z), v3(half_x, 0, half_z), v3(half_x, 0, -half_z)};
    m.normals.assign(4, v3(0, 1, 0));
    m.indices = {0, 1, 2, 2, 3, 0};
    return m;
}

Mesh icosphere_mesh(float radius, uint32_t subdivisions) {
    // Each icosahedron face is cut into (s+1)^2 triangles, s = subdivisions, and the grid points
    // are pushed onto the sphere: 20(s+1)^2 triangles, 10(s+1)^2+2 vertices (s=5: 720 / 362, the
    // counts of bevy's Sphere::mesh() default).  Vertex order is this generator's own.
    Mesh m;
    const float t = (1.0f + std::sqrt(5.0f)) * 0.5f;
    Vec3 base[12] = {v3(-1, t, 0), v3(1, t, 0), v3(-1, -t, 0), v3(1, -t, 0), v3(0, -1, t), v3(0, 1, t),
                     v3(0, -1, -t), v3(0, 1, -t), v3(t, 0, -1), v3(t, 0, 1), v3(-t, 0, -1), v3(-t, 0, 1)};
    for (Vec3 &b : base) b = normalize(b);
    const int F[20][3] = {{0, 11, 5}, {0, 5, 1}, {0, 1, 7}, {0, 7, 10}, {0, 10, 11}, {1, 5, 9}, {5, 11, 4},
                          {11, 10, 2}, {10, 7, 6}, {7, 1, 8}, {3, 9, 4}, {3, 4, 2}, {3, 2, 6}, {3, 6, 8},
                          {3, 8, 9}, {4, 9, 5}, {2, 4, 11}, {6, 2, 10}, {8, 6, 7}, {9, 8, 1}};
    const uint32_t N = subdivisions + 1;
    std::map<std::tuple<int, int, int, int>, uint32_t> shared;   // corner / edge points
    auto emit = [&](Vec3 dir) {
        Vec3 n = normalize(dir);
        m.normals.push_back(n);
        m.positions.push_back(scl(n, radius));
        return (uint32_t)(m.positions.size() - 1);
    };
    for (int f = 0; f < 20; ++f) {
        int A = F[f][0], B = F[f][1], C = F[f][2];
        if (dot(cross(sub(base[B], base[A]), sub(base[C], base[A])), add(add(base[A], base[B]), base[C])) < 0.0f)
            std::swap(B, C);
        std::vector<uint32_t> grid((N + 1) * (N + 1), 0xFFFFFFFFu);
        for (uint32_t i = 0; i <= N; ++i) {        // weight of B
            for (uint32_t j = 0; i + j <= N; ++j) {  // weight of C
                uint32_t k = N - i - j;              // weight of A
                std::tuple<int, int, int, int> key;
                bool is_shared = true;
                if (k == N) key = std::make_tuple(A, A, 0, 0);
                else if (i == N) key = std::make_tuple(B, B, 0, 0);
                else if (j == N) key = std::make_tuple(C, C, 0, 0);
                else if (j == 0) key = A < B ? std::make_tuple(A, B, (int)i, 1) : std::make_tuple(B, A, (int)k, 1);
                else if (i == 0) key = A < C ? std::make_tuple(A, C, (int)j, 1) : std::make_tuple(C, A, (int)k, 1);
                else if (k == 0) key = B < C ? std::make_tuple(B, C, (int)j, 1) : std::make_tuple(C, B, (int)i, 1);
                else is_shared = false;
                uint32_t idx;
                if (is_shared) {
                    auto it = shared.find(key);
                    if (it != shared.end()) idx = it->second;
                    else {
                        // point on the edge P->Q at step s/N from P, computed from the edge's
                        // endpoints only so both faces sharing the edge get the same vertex
                        int P = std::get<0>(key), Q = std::get<1>(key), s = std::get<2>(key);
                        Vec3 dir = P == Q ? base[P]
                                          : add(scl(base[P], (float)(N - (uint32_t)s) / (float)N), scl(base[Q], (float)s / (float)N));
                        idx = emit(dir);
                        shared[key] = idx;
                    }
                } else {
                    Vec3 dir = add(add(scl(base[A], (float)k / (float)N), scl(base[B], (float)i / (float)N)),
                                   scl(base[C], (float)j / (float)N));
                    idx = emit(dir);
                }
                grid[i * (N + 1) + j] = idx;
            }
        }
        for (uint32_t i = 0; i < N; ++i) {
            for (uint32_t j = 0; i + j < N; ++j) {
                uint32_t p00 = grid[i * (N + 1) + j], p10 = grid[(i + 1) * (N + 1) + j], p01 = grid[i * (N + 1) + j + 1];
                m.indices.push_back(p00); m.indices.push_back(p10); m.indices.push_back(p01);
                if (i + j + 1 < N) {
                    uint32_t p11 = grid[(i + 1) * (N + 1) + j + 1];
                    m.indices.push_back(p10); m.indices.push_back(p11); m.indices.push_back(p01);
                }
            }
        }
    }
    return m;
}

// ---- P1 prepare_meshinfo: src/ray_tracing.rs:98-169, quirks kept ---------------------------------------
// vertices and box of one entity, as the loop body of src/ray_tracing.rs:131-149 computes them
static void bake_entity(const Entity &e, const Mesh &mesh, Vertex *out, size_t nv, float aabb_min[3], float aabb_max[3]) {
    const GlobalTransform &T = e.transform;
    Vec3 c0, c1, c2;
    mat3_from_quat(T.rotation, c0, c1, c2);
    Vec3 m0 = scl(c0, T.scale.x), m1 = scl(c1, T.scale.y), m2 = scl(c2, T.scale.z);
    for (size_t i = 0; i < nv; ++i) {
        Vertex v;
        std::memset(&v, 0, sizeof v);
        Vec3 p = add(mat3_mul(m0, m1, m2, mesh.positions[i]), T.translation);   // transform_point (:137)
        Vec3 n = quat_rotate(T.rotation, mesh.normals[i]);                      // rotate only (:139-141)
        v.pos[0] = p.x; v.pos[1] = p.y; v.pos[2] = p.z;
        v.norm[0] = n.x; v.norm[1] = n.y; v.norm[2] = n.z;
        out[i] = v;
    }
    // Cuboid::from_size(half_extents * 2).aabb_3d(translate, rotate) (:148-149): the Aabb's
    // centre and the transform's scale are ignored
    Vec3 he = v3((e.aabb.half_extents.x * 2.0f) / 2.0f, (e.aabb.half_extents.y * 2.0f) / 2.0f,
                 (e.aabb.half_extents.z * 2.0f) / 2.0f);
    Vec3 half = mat3_mul(vabs(c0), vabs(c1), vabs(c2), he);
    Vec3 lo = sub(T.translation, half), hi = add(T.translation, half);
    aabb_min[0] = lo.x; aabb_min[1] = lo.y; aabb_min[2] = lo.z;
    aabb_max[0] = hi.x; aabb_max[1] = hi.y; aabb_max[2] = hi.z;
}

void prepare_meshinfo(World &world) {
    // f1, host side: same entities, same handles, same assets as last time => only the entities that moved are baked
    // again, in place (hpp: World::Baked).  The resource left behind is byte for byte the one the full pass below leaves.
    bool same = world.info_generation != 0 && world.baked.size() == world.entities.size() &&
                world.baked_asset_generation == world.asset_generation && world.baked_meshes == world.meshes.size() &&
                world.baked_materials == world.materials.size();
    for (size_t i = 0; same && i < world.entities.size(); ++i)
        same = world.entities[i].mesh == world.baked[i].mesh && world.entities[i].material == world.baked[i].material;
    if (same) {
        std::vector<uint32_t> moved;
        size_t moved_vertices = 0;
        for (size_t i = 0; i < world.entities.size(); ++i) {
            const Entity &e = world.entities[i];
            const World::Baked &b = world.baked[i];
            if (b.mesh_info < 0) { world.log += "Mesh missing attribute\n"; continue; }
            if (std::memcmp(&e.transform, &b.transform, sizeof e.transform) == 0 &&
                std::memcmp(&e.aabb.half_extents, &b.aabb.half_extents, sizeof e.aabb.half_extents) == 0) continue;
            moved.push_back((uint32_t)i);
            moved_vertices += b.n_vertices;
        }
        auto rebake = [&](size_t mb, size_t me) {          // entities own disjoint runs of the arrays
            for (size_t q = mb; q < me; ++q) {
                const Entity &e = world.entities[moved[q]];
                World::Baked &b = world.baked[moved[q]];
                MeshInfo &mi = world.ray_tracing_info.meshes[b.mesh_info];
                bake_entity(e, world.meshes[e.mesh], world.ray_tracing_info.vertices.data() + b.first_vertex, b.n_vertices, mi.aabb_min, mi.aabb_max);
                b.transform = e.transform; b.aabb = e.aabb;
            }
        };
        int threads = (int)std::min<size_t>(std::min<size_t>(moved.size(), moved_vertices >> 15), std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())));
        if (threads <= 1) rebake(0, moved.size());
        else {                                             // many entities moved at once: their runs are baked side by side
            std::vector<std::thread> th;
            for (int t = 0; t < threads; ++t) th.emplace_back(rebake, moved.size() * (size_t)t / threads, moved.size() * (size_t)(t + 1) / threads);
            for (auto &t : th) t.join();
        }
        for (uint32_t i : moved) world.moved_meshes.push_back((uint32_t)world.baked[i].mesh_info);
        if (!moved.empty()) world.info_generation++;
        return;
    }
    RayTracingInfo info = world.ray_tracing_info;   // `ray_tracing_info.clone()` keeps `count` (:112)
    info.vertices.clear(); info.triangles.clear(); info.meshes.clear(); info.materials.clear();
    world.baked.clear(); world.moved_meshes.clear();
    uint32_t material_index = 0;
    for (const Entity &e : world.entities) {
        if (e.mesh < 0 || (size_t)e.mesh >= world.meshes.size() || e.material < 0 ||
            (size_t)e.material >= world.materials.size())
            throw std::runtime_error("prepare_meshinfo: dangling asset handle (reference: unwrap() panic, src/ray_tracing.rs:120-121)");
        const Mesh &mesh = world.meshes[e.mesh];
        const StandardMaterial &mat = world.materials[e.material];
        SimpleMaterial sm;
        std::memcpy(sm.color, mat.base_color_linear, 16);
        info.materials.push_back(sm);                 // pushed BEFORE the attribute check (:122)
        World::Baked bk;
        bk.mesh = e.mesh; bk.material = e.material; bk.transform = e.transform; bk.aabb = e.aabb;
        bk.first_vertex = 0; bk.n_vertices = 0; bk.mesh_info = -1;
        if (!mesh.has_positions || !mesh.has_normals) {
            world.log += "Mesh missing attribute\n";  // (:127), then `continue`: no MeshInfo, no index bump
            world.baked.push_back(bk);
            continue;
        }
        size_t vertices_len = info.vertices.size();
        size_t nv = std::min(mesh.positions.size(), mesh.normals.size());   // zip() stops at the shorter
        info.vertices.resize(vertices_len + nv);
        if (!mesh.has_indices)
            throw std::runtime_error("prepare_meshinfo: mesh has no indices (reference: unwrap() panic, src/ray_tracing.rs:145)");
        size_t len = mesh.indices.size();
        MeshInfo mi;
        std::memset(&mi, 0, sizeof mi);
        bake_entity(e, mesh, info.vertices.data() + vertices_len, nv, mi.aabb_min, mi.aabb_max);
        mi.first_tri = (uint32_t)info.triangles.size();
        mi.tri_count = (uint32_t)len / 3u;
        mi.material = material_index;
        bk.first_vertex = (uint32_t)vertices_len; bk.n_vertices = (uint32_t)nv; bk.mesh_info = (int)info.meshes.size();
        world.baked.push_back(bk);
        info.meshes.push_back(mi);
        material_index += 1;
        for (size_t i = 0; i + 2 < len; i += 3) {     // tuples() drops a trailing partial triple (:161)
            Triangle t;
            t.indices[0] = (uint32_t)(mesh.indices[i + 0] + vertices_len);
            t.indices[1] = (uint32_t)(mesh.indices[i + 1] + vertices_len);
            t.indices[2] = (uint32_t)(mesh.indices[i + 2] + vertices_len);
            info.triangles.push_back(t);
        }
    }
    world.ray_tracing_info = std::move(info);         // commands.insert_resource (:168)
    world.baked_asset_generation = world.asset_generation;
    world.baked_meshes = world.meshes.size(); world.baked_materials = world.materials.size();
    world.info_topology_generation++;
    world.info_generation++;
}

// ---- scenes ---------------------------------------------------------------------------------------------
static int add_mesh(World &w, Mesh m) { w.meshes.push_back(std::move(m)); return (int)w.meshes.size() - 1; }
static int add_material(World &w, float r, float g, float b, float a) {
    StandardMaterial m; m.base_color_linear[0] = r; m.base_color_linear[1] = g; m.base_color_linear[2] = b; m.base_color_linear[3] = a;
    w.materials.push_back(m);
    return (int)w.materials.size() - 1;
}
static Entity &spawn(World &w, int mesh, int material, GlobalTransform t) {
    Entity e;
    e.mesh = mesh; e.material = material; e.transform = t;
    e.aabb = compute_aabb(w.meshes[mesh]);
    w.entities.push_back(e);
    return w.entities.back();
}

void spawn_default_scene(World &world) {   // src/main.rs:51-117
    int blue = add_material(world, 0.0f, 0.0f, 1.0f, 1.0f);   // LinearRgba::BLUE (:56)
    int red = add_material(world, 1.0f, 0.0f, 0.0f, 1.0f);    // Color::srgb(1,0,0) -> linear (1,0,0) (:57)
    world.camera.transform = looking_at(v3(0.0f, 3.0f, 5.0f), v3(0, 0, 0), v3(0, 1, 0));   // (:61)
    world.camera.ray_tracing_graph = true;                                                  // (:60)
    GlobalTransform t;
    t.rotation = quat_from_axis_angle(v3(1, 0, 0), 45.0f * 3.14159265358979323846f / 180.0f);   // (:71-74)
    spawn(world, add_mesh(world, cuboid_mesh(1.0f, 1.0f, 1.0f)), blue, t).rotate_tag = true;      // (:67-78)
    spawn(world, add_mesh(world, plane_mesh(5.0f, 5.0f)), blue, GlobalTransform());               // (:79-83)
    const float xs[4] = {1.0f, -1.0f, 1.0f, -1.0f}, zs[4] = {1.0f, 1.0f, -1.0f, -1.0f};          // (:84-107)
    for (int i = 0; i < 4; ++i) {
        GlobalTransform s;
        s.translation = v3(xs[i], 0.5f, zs[i]);
        spawn(world, add_mesh(world, icosphere_mesh(0.5f, 5)), red, s);   // each spawn adds its own mesh asset
    }
    // the PointLight of :108-116 is not consumed by the ray tracer
}

void spawn_sphere_field(World &world, uint32_t n, uint64_t seed, uint32_t subdivisions) {
    SplitMix64 rng(seed);
    int ground = add_material(world, 0.5f, 0.5f, 0.5f, 1.0f);
    spawn(world, add_mesh(world, plane_mesh(60.0f, 60.0f)), ground, GlobalTransform());
    for (uint32_t i = 0; i < n; ++i) {
        float cx = rng.uniform(-50.0f, 50.0f), cy = rng.uniform(0.0f, 20.0f), cz = rng.uniform(-50.0f, 50.0f);
        float r = rng.uniform(0.2f, 0.8f);
        float cr = rng.uniform(0.1f, 1.0f), cg = rng.uniform(0.1f, 1.0f), cb = rng.uniform(0.1f, 1.0f);
        int mat = add_material(world, cr, cg, cb, 1.0f);
        GlobalTransform t;
        t.translation = v3(cx, cy, cz);
        // radius baked into the mesh, translation-only transform (as src/main.rs:84-88 does):
        // prepare_meshinfo's AABB ignores scale, so a scaled unit sphere would be mis-culled
        spawn(world, add_mesh(world, icosphere_mesh(r, subdivisions)), mat, t);
    }
}

void spawn_icosphere_soup(World &world, uint32_t n, uint64_t seed, uint32_t subdivisions) {
    SplitMix64 rng(seed);
    for (uint32_t i = 0; i < n; ++i) {
        float cx = rng.uniform(-50.0f, 50.0f), cy = rng.uniform(-50.0f, 50.0f), cz = rng.uniform(-50.0f, 50.0f);
        float r = rng.uniform(0.5f, 2.0f);
        float cr = rng.uniform(0.1f, 1.0f), cg = rng.uniform(0.1f, 1.0f), cb = rng.uniform(0.1f, 1.0f);
        int mat = add_material(world, cr, cg, cb, 1.0f);
        GlobalTransform t;
        t.translation = v3(cx, cy, cz);
        spawn(world, add_mesh(world, icosphere_mesh(r, subdivisions)), mat, t);
    }
}

void frame_synthetic_camera(World &world, float radii) {
    // eye at `radii` bounding-sphere radii from the centroid (2.5), looking at the centroid
    if (world.entities.empty()) return;
    Vec3 lo = world.entities[0].transform.translation, hi = lo;
    for (const Entity &e : world.entities) {
        Vec3 p = e.transform.translation, h = e.aabb.half_extents;
        lo = v3(std::min(lo.x, p.x - h.x), std::min(lo.y, p.y - h.y), std::min(lo.z, p.z - h.z));
        hi = v3(std::max(hi.x, p.x + h.x), std::max(hi.y, p.y + h.y), std::max(hi.z, p.z + h.z));
    }
    Vec3 c = scl(add(lo, hi), 0.5f);
    Vec3 d = sub(hi, c);
    float radius = std::sqrt(dot(d, d));
    Vec3 dir = normalize(v3(0.35f, 0.45f, 1.0f));
    world.camera.transform = looking_at(add(c, scl(dir, radii * radius)), c, v3(0, 1, 0));
}

void rotate_tagged(World &world, float dt) {   // src/main.rs:122-126: ele.rotate_y(1.0 * dt)
    Quat r = quat_from_axis_angle(v3(0, 1, 0), 1.0f * dt);
    for (Entity &e : world.entities)
        if (e.rotate_tag) e.transform.rotation = quat_mul(r, e.transform.rotation);
}

// ---- L2 pipeline init / L1 launch / plugin ------------------------------------------------------------
RayTracingPipeline RayTracingPipeline::from_world(World &, int device_ordinal) {   // src/pipeline.rs:25-72
    RayTracingPipeline p;
    p.status = rt_create(device_ordinal, &p.ctx);
    return p;
}
void RayTracingPipeline::destroy() {
    if (ctx) rt_destroy(ctx);
    ctx = nullptr; status = RT_NOT_READY;
}

static uint64_t fnv(const void *data, size_t n, uint64_t h) {
    const unsigned char *p = (const unsigned char *)data;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001B3ull; }
    return h;
}

static int set_view_and_render(const View &view, const RayTracingInfo &info, RayTracingPipeline &pipeline, const rt_render_params *params) {
    uint32_t vp[4] = {view.viewport.x, view.viewport.y, view.viewport.w, view.viewport.h};
    int rc = rt_set_view(pipeline.ctx, view.world_from_clip, view.world_position, view.width, view.height,
                         view.has_viewport ? vp : nullptr);                 // bind group 0 + viewport (:58-62,71-76)
    if (rc != RT_OK) return rc;
    rt_render_params p;
    std::memset(&p, 0, sizeof p);
    if (params) p = *params;
    p.struct_size = sizeof p;
    p.frame_count = info.count;
    return rt_render(pipeline.ctx, &p);                                      // draw(0..3, 0..1) (:77)
}

int RayTracingPassNode::run(const View &view, World &world, RayTracingPipeline &pipeline,
                            const rt_render_params *params) {   // src/node.rs:29-79
    if (!pipeline.ctx || pipeline.status != RT_OK) return RT_NOT_READY;   // pipeline not ready: skip (:52-56)
    const RayTracingInfo &info = world.ray_tracing_info;
    // The reference re-creates and re-uploads all five bindings every frame (:42-50).  Here the
    // arrays go to the GPU (and the BVH is rebuilt) only when their contents change.
    if (world.info_generation != 0) {
        // the resource was written by prepare_meshinfo, which says what it changed: no need to look at the arrays
        int rc = RT_OK;
        const size_t nm = info.meshes.size();
        if (uploaded_generation != 0 && world.info_topology_generation == uploaded_topology_generation) {
            if (world.info_generation != uploaded_generation) {
                // vertices / mesh boxes of some entities moved (main.rs rotates the cube every frame, src/main.rs:122-126):
                // the run of MeshInfo entries from the first to the last moved one, if that is a small part of the scene
                rc = RT_ERR_SCENE;
                uint32_t m0 = 0xFFFFFFFFu, m1 = 0;
                for (uint32_t m : world.moved_meshes) { m0 = std::min(m0, m); m1 = std::max(m1, m + 1); }
                if (m0 < m1 && m1 <= nm && (size_t)(m1 - m0) * 2 <= nm && mesh_vlo.size() == nm) {
                    size_t v0 = info.vertices.size(), v1 = 0;
                    for (size_t m = m0; m < m1; ++m)
                        if (info.meshes[m].tri_count) { v0 = std::min<size_t>(v0, mesh_vlo[m]); v1 = std::max<size_t>(v1, (size_t)mesh_vhi[m] + 1); }
                    if (v0 < v1) {
                        rc = rt_update_vertex_range(pipeline.ctx, info.vertices.data() + v0, v0, v1 - v0, info.meshes.data() + m0, m0, m1 - m0);
                        if (rc == RT_OK) ++range_updates;
                        else if (rc != RT_ERR_SCENE) return rc;
                    }
                }
                if (rc != RT_OK) {
                    rc = rt_update_vertices(pipeline.ctx, info.vertices.data(), info.vertices.size(), info.meshes.data(), info.meshes.size());
                    if (rc == RT_OK) ++full_updates;
                    else if (rc != RT_ERR_SCENE) return rc;
                }
            }
        } else rc = RT_ERR_SCENE;
        if (rc != RT_OK) {        // new topology (or an update the library refused): upload and rebuild
            rc = rt_upload_scene(pipeline.ctx, info.triangles.data(), info.triangles.size(), info.meshes.data(),
                                 info.meshes.size(), info.vertices.data(), info.vertices.size(),
                                 info.materials.data(), info.materials.size());
            if (rc != RT_OK) return rc;
            mesh_vlo.assign(nm, 0xFFFFFFFFu);
            mesh_vhi.assign(nm, 0u);
            for (size_t m = 0; m < nm; ++m)
                for (uint32_t t = info.meshes[m].first_tri; t < info.meshes[m].first_tri + info.meshes[m].tri_count; ++t)
                    for (int k = 0; k < 3; ++k) {
                        mesh_vlo[m] = std::min(mesh_vlo[m], info.triangles[t].indices[k]);
                        mesh_vhi[m] = std::max(mesh_vhi[m], info.triangles[t].indices[k]);
                    }
        }
        uploaded_generation = world.info_generation;
        uploaded_topology_generation = world.info_topology_generation;
        world.moved_meshes.clear();
        uploaded_signature = 0;       // the hash path below starts over if the resource is ever filled by hand
        return set_view_and_render(view, info, pipeline, params);
    }
    // a resource filled by other code than prepare_meshinfo: compare by content
    uint64_t sig = 0xCBF29CE484222325ull;
    sig = fnv(info.triangles.data(), info.triangles.size() * sizeof(Triangle), sig);
    sig = fnv(info.meshes.data(), info.meshes.size() * sizeof(MeshInfo), sig);
    sig = fnv(info.vertices.data(), info.vertices.size() * sizeof(Vertex), sig);
    sig = fnv(info.materials.data(), info.materials.size() * sizeof(SimpleMaterial), sig);
    sig ^= info.triangles.size() * 0x9E3779B97F4A7C15ull;
    if (sig == 0) sig = 1;
    uint64_t topo = 0xCBF29CE484222325ull;
    topo = fnv(info.triangles.data(), info.triangles.size() * sizeof(Triangle), topo);
    topo = fnv(info.materials.data(), info.materials.size() * sizeof(SimpleMaterial), topo);
    for (const MeshInfo &mi : info.meshes) topo = fnv(&mi, 12, topo);   // first_tri, tri_count, material
    topo ^= info.vertices.size() * 0x9E3779B97F4A7C15ull;
    if (topo == 0) topo = 1;
    if (sig != uploaded_signature && topo == topology_signature && uploaded_signature != 0) {
        // only vertices / mesh boxes moved (main.rs rotates the cube every frame,
        // src/main.rs:122-126): re-bake records and refit the BVH instead of rebuilding it
        int rc = RT_ERR_SCENE;
        // dirty-mesh form first: the meshes whose T2 entry or vertices changed, if they are one run of mesh indices
        // with one run of vertices (an entity that moves, like main.rs's cube); the library checks the ranges
        const size_t nv = info.vertices.size(), nm = info.meshes.size();
        if (uploaded_vertices.size() == nv && uploaded_meshes.size() == nm && mesh_vlo.size() == nm && nv && nm) {
            size_t lo = 0, hi = nv;
            while (lo < nv && std::memcmp(&info.vertices[lo], &uploaded_vertices[lo], sizeof(Vertex)) == 0) ++lo;
            while (hi > lo && std::memcmp(&info.vertices[hi - 1], &uploaded_vertices[hi - 1], sizeof(Vertex)) == 0) --hi;
            size_t m0 = nm, m1 = 0;
            for (size_t m = 0; m < nm; ++m) {
                bool dirty = std::memcmp(&info.meshes[m], &uploaded_meshes[m], sizeof(MeshInfo)) != 0;
                if (!dirty && hi > lo && info.meshes[m].tri_count) dirty = mesh_vhi[m] >= lo && mesh_vlo[m] < hi;
                if (dirty) { m0 = std::min(m0, m); m1 = std::max(m1, m + 1); }
            }
            if (m0 < m1 && (m1 - m0) * 2 <= nm) {          // a small part of the scene: worth the ranged path
                size_t v0 = nv, v1 = 0;
                for (size_t m = m0; m < m1; ++m)
                    if (info.meshes[m].tri_count) { v0 = std::min<size_t>(v0, mesh_vlo[m]); v1 = std::max<size_t>(v1, (size_t)mesh_vhi[m] + 1); }
                if (v0 < v1 && v0 <= lo && hi <= v1) {
                    rc = rt_update_vertex_range(pipeline.ctx, info.vertices.data() + v0, v0, v1 - v0, info.meshes.data() + m0, m0, m1 - m0);
                    if (rc == RT_OK) ++range_updates;
                    else if (rc != RT_ERR_SCENE) return rc;
                }
            }
        }
        if (rc != RT_OK) {
            rc = rt_update_vertices(pipeline.ctx, info.vertices.data(), info.vertices.size(), info.meshes.data(), info.meshes.size());
            if (rc == RT_OK) ++full_updates;
        }
        if (rc == RT_OK) {
            uploaded_signature = sig;
            uploaded_vertices = info.vertices;
            uploaded_meshes = info.meshes;
        } else if (rc != RT_ERR_SCENE) return rc;
    }
    if (sig != uploaded_signature) {
        int rc = rt_upload_scene(pipeline.ctx, info.triangles.data(), info.triangles.size(), info.meshes.data(),
                                 info.meshes.size(), info.vertices.data(), info.vertices.size(),
                                 info.materials.data(), info.materials.size());
        if (rc != RT_OK) return rc;
        uploaded_signature = sig;
        topology_signature = topo;
        uploaded_vertices = info.vertices;
        uploaded_meshes = info.meshes;
        mesh_vlo.assign(info.meshes.size(), 0xFFFFFFFFu);
        mesh_vhi.assign(info.meshes.size(), 0u);
        for (size_t m = 0; m < info.meshes.size(); ++m)
            for (uint32_t t = info.meshes[m].first_tri; t < info.meshes[m].first_tri + info.meshes[m].tri_count; ++t)
                for (int k = 0; k < 3; ++k) {
                    mesh_vlo[m] = std::min(mesh_vlo[m], info.triangles[t].indices[k]);
                    mesh_vhi[m] = std::max(mesh_vhi[m], info.triangles[t].indices[k]);
                }
    }
    return set_view_and_render(view, info, pipeline, params);
}

void RayTracingPlugin::build(App &app) const {   // src/ray_tracing.rs:29-42
    app.world.ray_tracing_info = RayTracingInfo();          // insert_resource(RayTracingInfo::default())
    if (app.has_render_app) {
        app.extract_registered = true;                      // add_systems(ExtractSchedule, prepare_meshinfo)
        app.graph_registered = true;                        // add_render_sub_graph + the pass node
    }
}
void RayTracingPlugin::finish(App &app) const {  // src/ray_tracing.rs:44-48
    if (!app.has_render_app)
        throw std::runtime_error("RayTracingPlugin::finish: no RenderApp (reference: unwrap() panic, src/ray_tracing.rs:46)");
    app.pipeline.destroy();                                 // a second finish(): the old context goes, and with it
    app.node = RayTracingPassNode();                        // everything the node believes the device holds
    app.pipeline = RayTracingPipeline::from_world(app.world, app.device_ordinal);
}

int App::update(const rt_render_params *params) {
    if (extract_registered) prepare_meshinfo(world);
    if (!graph_registered || !world.camera.ray_tracing_graph) return RT_NOT_READY;   // camera on another graph
    View v = extract_view(world.camera);
    return node.run(v, world, pipeline, params);
}

}  // namespace ray_tracing
