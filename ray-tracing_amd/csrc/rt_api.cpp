// rt_api.cpp -- the C ABI of include/rt_amd.h: context, device memory, streams, launches.
// Replaces what src/pipeline.rs (init) and src/node.rs (per-frame launch) do through wgpu.
// There is NO CPU fallback: without a gfx950 device rt_create fails with RT_ERR_NO_GPU.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_amd.h"
#include "bvh_build.h"
#include "rt_kernels.h"
#include "rt_layout.h"

namespace {
thread_local std::string g_create_error;

struct QueueMem {
    float *f = nullptr;       // 9 float planes
    uint32_t *u = nullptr;    // 2 u32 planes
};
}  // namespace

// Frames in flight (rt_set_frames_in_flight): every set is one frame's outputs, its counters / work-queue heads and the
// stream it is rendered on.  rt_render takes the sets in turn; the context's d_rgba / d_prim / d_t / d_cnt / h_cnt / stream
// members are ALIASES of the set rendered last, so every reader sees the latest frame.
#define RT_MAX_IN_FLIGHT 4
struct FrameSet {
    float *d_rgba = nullptr, *d_t = nullptr;
    uint32_t *d_prim = nullptr;
    unsigned long long *d_cnt = nullptr, *h_cnt = nullptr;
    hipStream_t own_stream = nullptr;      // set 0 only: the context's ordinary stream (one frame at a time)
    hipStream_t flight_stream = nullptr;   // with frames in flight: a stream with a hardware queue of its own (rt_set_frames_in_flight)
    uint32_t coverage[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // what the buffers were last cleared for, compared field by field
    bool coverage_valid = false;
    // path tracing: queues (2, ping-pong) and the per-sample contribution planes of a batch of samples.  Sets 0 and
    // 1 only (RT_PATH_SETS): a path-traced frame's tracing overlaps with the frame before it, its additions to the ONE
    // accumulation buffer wait for that frame's (rt_render)
    QueueMem q[2];
    size_t q_capacity = 0;
    float *d_contrib = nullptr;
    uint32_t contrib_planes = 0;
    uint8_t *d_bgra = nullptr;            // rt_present_async: this set's encoded frame
    hipEvent_t ev_present = nullptr;      // ... and the arrival of its copy on the host
    bool present_pending = false;
};
#define RT_PATH_SETS 2u

struct rt_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipStream_t ext_stream = nullptr;     // rt_set_stream: every set runs on the caller's stream (frames then serialise)
    bool use_null_stream = false;
    FrameSet fs[RT_MAX_IN_FLIGHT];
    uint32_t n_sets = 1, cur_set = 0;
    std::string err;

    // scene
    bool has_scene = false;
    RtSceneDev S{};
    void *d_nodes = nullptr, *d_recs = nullptr, *d_tris = nullptr, *d_meshes = nullptr, *d_verts = nullptr,
         *d_mats = nullptr, *d_spheres = nullptr, *d_sphere_mats = nullptr, *d_wide = nullptr, *d_wide_src = nullptr;
    rt_bvh_info info{};
    RtBvhBuildResult bvh_host;            // host copy of the tree (RT_TUNE_HOST_REFIT path)
    bool bvh_host_stale = false;          // the device tree has been refitted since
    std::vector<rt_triangle> h_tris;      // topology of the uploaded scene
    std::vector<rt_mesh_info> h_meshes;
    std::vector<uint32_t> h_mesh_vlo, h_mesh_vhi;   // per mesh: lowest / highest vertex index its triangles use
    std::vector<uint8_t> h_vert_used;     // per vertex: some triangle of the scene uses it
    void *d_dirty = nullptr;              // one byte per binary node (rt_update_vertex_range)
    // rt_update_vertex_range, short-list form (topology only: valid through refits): parent and depth of every binary
    // node, the wide node its two boxes are slots of, per mesh the nodes that hold one of its records in a leaf child
    // (CSR), a visit stamp per node; and a device buffer for the list of nodes to redo
    std::vector<uint32_t> h_parent, h_depth, h_stamp, h_wide_of_bin, h_mesh_node_off, h_mesh_nodes;
    uint32_t stamp = 0;
    void *d_list = nullptr; size_t list_capacity = 0;
    // the list on the device is kept with the mesh range it was made for: an animation moves the same meshes every frame
    // (the reference's rotating cube, src/main.rs:122-126), and the tree's topology only changes with rt_upload_scene
    struct { bool valid = false, level_lists = false; size_t first_mesh = 0, n_meshes = 0; std::vector<uint32_t> level_end; uint32_t n_wide = 0; } list_cache;
    size_t last_range_first = (size_t)-1, last_range_n = 0;   // the mesh range of the call before this one
    uint32_t tune_refit_list_max = 16384;  // RT_TUNE_REFIT_LIST_MAX: lists up to this many nodes are redone by ONE launch of one workgroup (0 = never); tools/refit_time.py
    uint32_t tune_refit_levels_max = 262144;   // RT_TUNE_REFIT_LEVELS_MAX: longer lists, up to this, by one launch per tree level over the LISTED nodes; beyond: the whole-tree level launches
    void *d_level_nodes = nullptr;        // node indices grouped by tree level, deepest level first (GPU refit)
    std::vector<uint32_t> level_count;    // nodes per group, in that order

    // view + framebuffers
    bool has_view = false;
    RtViewDev V{};
    size_t n_px = 0;
    float *d_rgba = nullptr, *d_accum = nullptr, *d_t = nullptr;
    uint32_t *d_prim = nullptr;
    uint8_t *d_bgra = nullptr;
    hipEvent_t ev_update[2] = {nullptr, nullptr};   // timing of rt_update_vertices / rt_update_vertex_range (created once)
    uint32_t accum_samples = 0;
    uint32_t accum_cov[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // the coverage the running sums were started under
    bool accum_dirty = false;        // camera moved: the next path-traced frame restarts accumulation

    hipEvent_t ev_accum = nullptr;        // recorded at the end of every path-traced frame: the next one's additions wait for it
    bool accum_pending = false;

    // counters
    unsigned long long *d_cnt = nullptr;
    unsigned long long *h_cnt = nullptr;   // pinned
    rt_counters counters{};
    uint64_t primary_total = 0;

    RtTune tune{};
    void *static_scratch = nullptr;       // device address of rt_kernels.hip's static scratch array on THIS device (rt_render_static)
    uint32_t static_accum_samples = 0;    // samples in the caller's accumulation buffer (rt_render_static)
    uint32_t n_cus = 0;                   // compute units of the device (CU mask of the frames-in-flight streams)
    uint32_t tune_flight_prio = 0;        // RT_TUNE_FLIGHT_PRIO=1: priority streams instead of CU-mask streams (A/B)
    uint32_t tune_wide = 0;               // build and walk the 4-wide quantised tree
    uint32_t tune_sample_batch = 16;      // path tracing: most samples of a pixel traced per launch (16, 4 or 1 are used)
    RtWideTree wide_host;

    // kernel timing
    std::vector<hipEvent_t> ev;            // pairs: start, stop
    size_t ev_used = 0;
};

namespace {

int fail(rt_ctx *c, int code, const char *what, hipError_t e = hipSuccess) {
    char buf[512];
    if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof buf, "%s", what);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(c, call)                                                        \
    do {                                                                       \
        hipError_t e_ = (call);                                                \
        if (e_ != hipSuccess) return fail((c), e_ == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP, #call, e_); \
    } while (0)

void free_dev(void *&p) { if (p) { (void)hipFree(p); p = nullptr; } }

hipStream_t set_stream_of(const rt_ctx *c, uint32_t k) {
    if (c->use_null_stream) return nullptr;
    if (c->ext_stream) return c->ext_stream;
    return c->n_sets > 1 && c->fs[k].flight_stream ? c->fs[k].flight_stream : c->fs[0].own_stream;
}
// the context's aliases point at set k
void select_set(rt_ctx *c, uint32_t k) {
    c->cur_set = k;
    c->d_rgba = c->fs[k].d_rgba; c->d_t = c->fs[k].d_t; c->d_prim = c->fs[k].d_prim;
    c->d_cnt = c->fs[k].d_cnt; c->h_cnt = c->fs[k].h_cnt;
    c->stream = set_stream_of(c, k);
}
// every stream any frame of this context may still be running on
hipError_t sync_all(rt_ctx *c) {
    hipError_t first = hipSuccess;
    for (uint32_t k = 0; k < c->n_sets; ++k) {
        hipError_t e = hipStreamSynchronize(set_stream_of(c, k));
        if (e != hipSuccess && first == hipSuccess) first = e;
    }
    return first;
}
// The accumulation buffer belongs to the path-traced frames, which run on sets 0 and 1 (rt_render): a reader that arrives
// after a later primary frame on another set has to wait for those sets' streams, not only for the frame rendered last.
hipError_t wait_path_frames(rt_ctx *c) {
    hipError_t first = hipSuccess;
    for (uint32_t k = 0; k < c->n_sets && k < RT_PATH_SETS && c->n_sets > 1; ++k) {
        hipError_t e = hipStreamSynchronize(set_stream_of(c, k));
        if (e != hipSuccess && first == hipSuccess) first = e;
    }
    return first;
}
void invalidate_coverage(rt_ctx *c) { for (FrameSet &f : c->fs) f.coverage_valid = false; }

void free_scene(rt_ctx *c) {
    free_dev(c->d_nodes); free_dev(c->d_recs); free_dev(c->d_tris);
    free_dev(c->d_meshes); free_dev(c->d_verts); free_dev(c->d_mats);
    free_dev(c->d_spheres); free_dev(c->d_sphere_mats); free_dev(c->d_level_nodes);
    free_dev(c->d_wide); free_dev(c->d_wide_src); free_dev(c->d_dirty);
    free_dev(c->d_list); c->list_capacity = 0;
    c->list_cache.valid = false; c->last_range_first = (size_t)-1; c->last_range_n = 0;
    c->level_count.clear();
    c->has_scene = false;
    std::memset(&c->S, 0, sizeof c->S);
}

void free_frame(rt_ctx *c) {
    void *p;
    for (FrameSet &f : c->fs) {
        p = f.d_rgba; free_dev(p); f.d_rgba = nullptr;
        p = f.d_t; free_dev(p); f.d_t = nullptr;
        p = f.d_prim; free_dev(p); f.d_prim = nullptr;
        f.coverage_valid = false;
    }
    c->d_rgba = nullptr; c->d_t = nullptr; c->d_prim = nullptr;
    p = c->d_accum; free_dev(p); c->d_accum = nullptr;
    p = c->d_bgra; free_dev(p); c->d_bgra = nullptr;
    for (FrameSet &f : c->fs) {
        for (int i = 0; i < 2; ++i) {
            p = f.q[i].f; free_dev(p); f.q[i].f = nullptr;
            p = f.q[i].u; free_dev(p); f.q[i].u = nullptr;
        }
        f.q_capacity = 0;
        p = f.d_contrib; free_dev(p); f.d_contrib = nullptr; f.contrib_planes = 0;
        p = f.d_bgra; free_dev(p); f.d_bgra = nullptr;
        f.present_pending = false;
    }
    c->n_px = 0;
}

template <class T>
int upload(rt_ctx *c, void *&dst, const T *src, size_t n) {
    free_dev(dst);
    // +64 B tail: the wave-coherent kernels fetch 48-B records through a 64-B window
    size_t bytes = (n ? n : 1) * sizeof(T) + 64;
    HIPCHK(c, hipMalloc(&dst, bytes));
    HIPCHK(c, hipMemsetAsync((char *)dst + bytes - 64 - (n ? 0 : sizeof(T)), 0, 64 + (n ? 0 : sizeof(T)), c->stream));
    if (n) HIPCHK(c, hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return RT_OK;
}

// One pass over a vertex array on up to 8 host threads: the largest |coordinate| (NaN never raises it) and whether a vertex
// that some triangle uses (used == nullptr: any vertex) has a non-finite position.  3.6 M vertices: two single-threaded
// passes took 5 ms of rt_update_vertices' 13.
void scan_vertices(const rt_vertex *verts, size_t n, const uint8_t *used, float *amax_out, bool *bad_out) {
    unsigned nt = n < ((size_t)1 << 16) ? 1u : std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    std::vector<float> amax(nt, 0.0f);
    std::vector<char> bad(nt, 0);
    auto part = [&](unsigned t) {
        float a = 0.0f; bool b = false;
        for (size_t i = n * t / nt, e = n * (t + 1) / nt; i < e; ++i) {
            const float x = std::fabs(verts[i].pos[0]), y = std::fabs(verts[i].pos[1]), z = std::fabs(verts[i].pos[2]);
            if (x > a) a = x;
            if (y > a) a = y;
            if (z > a) a = z;
            if (!(x < INFINITY && y < INFINITY && z < INFINITY) && (!used || used[i])) b = true;     // inf or NaN
        }
        amax[t] = a; bad[t] = b;
    };
    if (nt == 1) part(0);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t) th.emplace_back(part, t);
        for (auto &x : th) x.join();
    }
    *amax_out = 0.0f; *bad_out = false;
    for (unsigned t = 0; t < nt; ++t) { if (amax[t] > *amax_out) *amax_out = amax[t]; *bad_out = *bad_out || bad[t]; }
}

RtPathQueue queue_view(const QueueMem &m, size_t cap) {
    RtPathQueue q;
    q.ox = m.f + 0 * cap; q.oy = m.f + 1 * cap; q.oz = m.f + 2 * cap;
    q.dx = m.f + 3 * cap; q.dy = m.f + 4 * cap; q.dz = m.f + 5 * cap;
    q.cr = m.f + 6 * cap; q.cg = m.f + 7 * cap; q.cb = m.f + 8 * cap;
    q.rng = m.u; q.pixel = m.u + cap;
    return q;
}

// path queues for `batch` samples of every pixel in flight, and the batch's per-sample contribution planes
int ensure_queues(rt_ctx *c, FrameSet &f, uint32_t batch) {
    const size_t need = c->n_px * batch;
    if (f.q_capacity < need || !f.q[0].f) {
        for (int i = 0; i < 2; ++i) {
            void *p = f.q[i].f; free_dev(p); f.q[i].f = nullptr;
            p = f.q[i].u; free_dev(p); f.q[i].u = nullptr;
        }
        f.q_capacity = 0;
        for (int i = 0; i < 2; ++i) {
            HIPCHK(c, hipMalloc((void **)&f.q[i].f, need * 9 * sizeof(float)));
            HIPCHK(c, hipMalloc((void **)&f.q[i].u, need * 2 * sizeof(uint32_t)));
        }
        f.q_capacity = need;
    }
    if (f.contrib_planes < batch || !f.d_contrib) {
        void *p = f.d_contrib; free_dev(p); f.d_contrib = nullptr; f.contrib_planes = 0;
        HIPCHK(c, hipMalloc((void **)&f.d_contrib, need * 16));
        f.contrib_planes = batch;
    }
    return RT_OK;
}

}  // namespace

extern "C" {

const char *rt_version(void) { return "ray-tracing_amd 0.1 (gfx950)"; }

const char *rt_status_string(int s) {
    switch (s) {
        case RT_OK: return "ok";
        case RT_NOT_READY: return "not ready (no scene or view yet; frame skipped)";
        case RT_ERR_ARG: return "invalid argument";
        case RT_ERR_HIP: return "HIP runtime error";
        case RT_ERR_NO_GPU: return "no usable gfx950 GPU (this library has no CPU fallback)";
        case RT_ERR_SCENE: return "inconsistent scene arrays";
        case RT_ERR_OOM: return "out of device memory";
        default: return "unknown status";
    }
}

const char *rt_last_error(rt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int rt_create(int device_ordinal, rt_ctx **out) {
    if (!out) return fail(nullptr, RT_ERR_ARG, "rt_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(nullptr, RT_ERR_NO_GPU, "rt_create: no HIP device", e);
    if (device_ordinal < 0 || device_ordinal >= n) return fail(nullptr, RT_ERR_ARG, "rt_create: device ordinal out of range");
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_ordinal);
    if (e != hipSuccess) return fail(nullptr, RT_ERR_HIP, "hipGetDeviceProperties", e);
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::string msg = std::string("rt_create: device is ") + prop.gcnArchName + ", kernels are built for gfx950 only";
        return fail(nullptr, RT_ERR_NO_GPU, msg.c_str());
    }
    rt_ctx *c = new (std::nothrow) rt_ctx();
    if (!c) return fail(nullptr, RT_ERR_OOM, "rt_create: host allocation failed");
    c->device = device_ordinal;
    // persistent-kernel geometry: 8 blocks of 256 threads per CU (16 KiB LDS stack each, <= 64 VGPRs);
    // the RT_TUNE_* variables are for experiments (tools/ab_traverse.py)
    {
        auto env = [](const char *name, uint32_t dflt, uint32_t lo, uint32_t hi) {
            const char *e = std::getenv(name);
            if (!e) return dflt;
            long v = std::atol(e);
            return (v < (long)lo || v > (long)hi) ? dflt : (uint32_t)v;
        };
        c->tune.cus = (uint32_t)prop.multiProcessorCount;
        c->tune.persist_blocks = c->tune.cus * (uint32_t)env("RT_TUNE_PERSIST_PER_CU", 7, 1, 8);   // what a CU holds of the 7-wave kernels
        c->tune.blocks_per_cu = env("RT_TUNE_BLOCKS_PER_CU", 0, 0, 8);
        c->tune.refill_min = env("RT_TUNE_REFILL_MIN", 64, 1, 64);
        c->tune.finish_min = env("RT_TUNE_FINISH_MIN", 64, 1, 64);
        c->tune_refit_list_max = (uint32_t)env("RT_TUNE_REFIT_LIST_MAX", 16384, 0, 1 << 24);
        c->tune_refit_levels_max = (uint32_t)env("RT_TUNE_REFIT_LEVELS_MAX", 262144, 0, 1 << 26);
        c->tune.vote_node = env("RT_TUNE_VOTE_NODE", 3, 1, 64);
        c->tune.vote_tri = env("RT_TUNE_VOTE_TRI", 1, 1, 64);
        c->tune.uni_min = env("RT_TUNE_UNI_MIN", 16, 1, 65);
        c->tune.wide_vote_node = env("RT_TUNE_WIDE_VOTE_NODE", 3, 1, 64);
        c->tune.wide_vote_tri = env("RT_TUNE_WIDE_VOTE_TRI", 1, 1, 64);
        c->tune.stream_waves = env("RT_TUNE_STREAM_WAVES", 0, 0, 8);
        c->tune_wide = env("RT_TUNE_WIDE", 1, 0, 1);
        c->tune_flight_prio = env("RT_TUNE_FLIGHT_PRIO", 0, 0, 1);
        c->tune_sample_batch = env("RT_TUNE_SAMPLE_BATCH", 16, 1, 16);
    }
    int rc = RT_OK;
    do {
        if ((e = hipSetDevice(device_ordinal)) != hipSuccess) { rc = fail(nullptr, RT_ERR_HIP, "hipSetDevice", e); break; }
        // set 0's ordinary stream; the streams of frames in flight are made by rt_set_frames_in_flight
        if ((e = hipStreamCreateWithFlags(&c->fs[0].own_stream, hipStreamNonBlocking)) != hipSuccess) { rc = fail(nullptr, RT_ERR_HIP, "hipStreamCreate", e); break; }
        c->n_cus = (uint32_t)prop.multiProcessorCount;
        c->own_stream = c->fs[0].own_stream;
        if ((e = hipMalloc((void **)&c->fs[0].d_cnt, RT_CNT_N * sizeof(unsigned long long))) != hipSuccess) { rc = fail(nullptr, RT_ERR_HIP, "hipMalloc counters", e); break; }
        if ((e = hipHostMalloc((void **)&c->fs[0].h_cnt, RT_CNT_N * sizeof(unsigned long long), hipHostMallocDefault)) != hipSuccess) { rc = fail(nullptr, RT_ERR_HIP, "hipHostMalloc counters", e); break; }
        std::memset(c->fs[0].h_cnt, 0, RT_CNT_N * sizeof(unsigned long long));
        if ((e = hipEventCreateWithFlags(&c->ev_accum, hipEventDisableTiming)) != hipSuccess) { rc = fail(nullptr, RT_ERR_HIP, "hipEventCreate", e); break; }
        select_set(c, 0);
        // the static scratch array of rt_render_static: every device holds its own copy at its own address, so the
        // address is asked for here, once per context, with this context's device current
        if ((e = rtk_static_scratch_address(&c->static_scratch)) != hipSuccess) { rc = fail(nullptr, RT_ERR_HIP, "hipGetSymbolAddress(rt_static_scratch)", e); break; }
    } while (0);
    if (rc != RT_OK) { rt_destroy(c); return rc; }
    *out = c;
    return RT_OK;
}

int rt_destroy(rt_ctx *c) {
    if (!c) return RT_OK;
    (void)hipSetDevice(c->device);
    (void)sync_all(c);
    free_scene(c);
    free_frame(c);
    for (FrameSet &f : c->fs) {
        if (f.d_cnt) (void)hipFree(f.d_cnt);
        if (f.h_cnt) (void)hipHostFree(f.h_cnt);
    }
    for (hipEvent_t ev : c->ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : c->ev_update) if (ev) (void)hipEventDestroy(ev);
    if (c->ev_accum) (void)hipEventDestroy(c->ev_accum);
    for (FrameSet &f : c->fs) {
        if (f.own_stream) (void)hipStreamDestroy(f.own_stream);
        if (f.flight_stream) (void)hipStreamDestroy(f.flight_stream);
        if (f.ev_present) (void)hipEventDestroy(f.ev_present);
    }
    delete c;
    return RT_OK;
}

int rt_set_stream(rt_ctx *c, void *hip_stream) {
    if (!c) return RT_ERR_ARG;
    (void)hipSetDevice(c->device);
    (void)sync_all(c);      // best effort: the stream being replaced may be one its owner has already destroyed
    c->use_null_stream = hip_stream == RT_STREAM_NULL;          // HIP's null stream, asked for explicitly
    c->ext_stream = c->use_null_stream ? nullptr : (hipStream_t)hip_stream;   // NULL: back on the sets' own streams
    select_set(c, c->cur_set);
    return RT_OK;
}

int rt_set_frames_in_flight(rt_ctx *c, uint32_t n) {
    if (!c) return RT_ERR_ARG;
    if (n == 0 || n > RT_MAX_IN_FLIGHT) return fail(c, RT_ERR_ARG, "rt_set_frames_in_flight: 1..4 frames");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_all(c));
    // Frames overlap only if their streams sit on different hardware queues, and HIP maps ordinary streams onto few of
    // them (4 per process on this pool) as it sees fit: next to RCCL's streams two sets shared one queue and their frames
    // ran strictly one after the other (kernel trace).  A stream created with a CU mask gets a
    // hardware queue of its own, so with n > 1 EVERY set (set 0 too) renders on such a stream, mask = all CUs.  These
    // streams are of HIP's default kind: they also order against the null stream.  Should the call fail, stream
    // priorities are the fallback (one queue pool per priority: normal, high, low keep three sets apart).
    for (uint32_t k = 0; k < n && n > 1; ++k) {
        FrameSet &f = c->fs[k];
        if (f.flight_stream) continue;
        uint32_t mask[32];
        for (uint32_t &w : mask) w = 0xFFFFFFFFu;
        const uint32_t words = std::min<uint32_t>(32u, (c->n_cus + 31u) / 32u);
        if (c->tune_flight_prio || hipExtStreamCreateWithCUMask(&f.flight_stream, words, mask) != hipSuccess) {
            (void)hipGetLastError();
            int least = 0, greatest = 0;
            HIPCHK(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
            const int prio[RT_MAX_IN_FLIGHT] = {0, greatest, least, 0};
            HIPCHK(c, hipStreamCreateWithPriority(&f.flight_stream, hipStreamNonBlocking, prio[k]));
        }
        // first use now (a stream takes its queue at its first submission), not inside the first timed frame
        HIPCHK(c, hipMemsetAsync(c->fs[0].d_cnt, 0, sizeof(unsigned long long), f.flight_stream));
        HIPCHK(c, hipStreamSynchronize(f.flight_stream));
    }
    for (uint32_t k = 1; k < n; ++k) {      // sets beyond the first get their counters and (if a view exists) frame buffers
        FrameSet &f = c->fs[k];
        if (!f.d_cnt) HIPCHK(c, hipMalloc((void **)&f.d_cnt, RT_CNT_N * sizeof(unsigned long long)));
        if (!f.h_cnt) {
            HIPCHK(c, hipHostMalloc((void **)&f.h_cnt, RT_CNT_N * sizeof(unsigned long long), hipHostMallocDefault));
            std::memset(f.h_cnt, 0, RT_CNT_N * sizeof(unsigned long long));
        }
        if (c->n_px) {                      // each buffer on its own: an earlier call may have failed half way
            if (!f.d_rgba) HIPCHK(c, hipMalloc((void **)&f.d_rgba, c->n_px * 16));
            if (!f.d_t) HIPCHK(c, hipMalloc((void **)&f.d_t, c->n_px * 4));
            if (!f.d_prim) HIPCHK(c, hipMalloc((void **)&f.d_prim, c->n_px * 4));
            f.coverage_valid = false;
        }
    }
    c->n_sets = n;
    select_set(c, c->cur_set >= n ? 0 : c->cur_set);     // the aliases, the stream among them, follow the new configuration
    return RT_OK;
}

int rt_get_stream(rt_ctx *c, void **hip_stream) {
    if (!c || !hip_stream) return RT_ERR_ARG;
    *hip_stream = (void *)c->stream;          // the stream of the frame set rendered last (NULL: HIP's null stream)
    return RT_OK;
}

int rt_synchronize(rt_ctx *c) {
    if (!c) return RT_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_all(c));
    return RT_OK;
}

static int validate_scene(rt_ctx *c, const rt_triangle *tris, size_t n_tris, const rt_mesh_info *meshes, size_t n_meshes,
                          const rt_vertex *verts, size_t n_verts) {
    if ((n_tris && !tris) || (n_meshes && !meshes) || (n_verts && !verts))
        return fail(c, RT_ERR_ARG, "scene: NULL array with non-zero count");
    if (n_tris >= (1u << 28) || n_verts > 0xFFFFFFFFull || n_meshes > 0xFFFFFFFFull)
        return fail(c, RT_ERR_ARG, "scene: too large (2^28 triangles max)");
    uint64_t loop_items = 0;
    for (size_t m = 0; m < n_meshes; ++m) {
        uint64_t end = (uint64_t)meshes[m].first_tri + meshes[m].tri_count;
        if (end > n_tris) return fail(c, RT_ERR_SCENE, "scene: mesh triangle range exceeds the triangle array");
        loop_items += meshes[m].tri_count;
    }
    if (loop_items >= (1u << 28)) return fail(c, RT_ERR_ARG, "scene: too many (mesh, triangle) items");
    for (size_t t = 0; t < n_tris; ++t)
        for (int k = 0; k < 3; ++k)
            if (tris[t].indices[k] >= n_verts) return fail(c, RT_ERR_SCENE, "scene: vertex index out of range");
    return RT_OK;
}

struct rt_host_bvh { RtBvhBuildResult r; };

static void fill_info(rt_bvh_info *info, const RtBvhBuildResult &bvh, size_t n_meshes, size_t n_verts) {
    info->n_nodes = bvh.nodes.size();
    info->n_triangles = bvh.recs.size();
    info->n_meshes = n_meshes;
    info->n_vertices = n_verts;
    info->node_bytes = sizeof(RtBvhNode);
    info->tri_bytes = sizeof(RtTriRecord);
    info->max_depth = bvh.max_depth;
    info->max_leaf = bvh.max_leaf;
    info->build_ms = bvh.build_ms;
    info->sah_cost = bvh.sah_cost;
    info->n_wide_nodes = 0; info->wide_max_depth = 0; info->wide_stack_need = 0;
}

int rt_host_bvh_build(const rt_triangle *tris, size_t n_tris, const rt_mesh_info *meshes, size_t n_meshes,
                      const rt_vertex *verts, size_t n_verts, int n_threads, rt_host_bvh **out, rt_bvh_info *info) {
    if (!out) return RT_ERR_ARG;
    *out = nullptr;
    int rc = validate_scene(nullptr, tris, n_tris, meshes, n_meshes, verts, n_verts);
    if (rc != RT_OK) return rc;
    rt_host_bvh *h = new (std::nothrow) rt_host_bvh();
    if (!h) return RT_ERR_OOM;
    try {
        rt_bvh_build(tris, n_tris, meshes, n_meshes, verts, n_verts, n_threads, &h->r);
    } catch (const std::bad_alloc &) {
        delete h;
        return fail(nullptr, RT_ERR_OOM, "rt_host_bvh_build: host allocation failed");
    }
    if (info) fill_info(info, h->r, n_meshes, n_verts);
    *out = h;
    return RT_OK;
}
int rt_host_bvh_copy(const rt_host_bvh *h, void *nodes, size_t nb, void *tris, size_t tb) {
    if (!h) return RT_ERR_ARG;
    if (nb != h->r.nodes.size() * sizeof(RtBvhNode) || tb != h->r.recs.size() * sizeof(RtTriRecord)) return RT_ERR_ARG;
    if (nodes && nb) std::memcpy(nodes, h->r.nodes.data(), nb);
    if (tris && tb) std::memcpy(tris, h->r.recs.data(), tb);
    return RT_OK;
}
int rt_host_bvh_refit(rt_host_bvh *h, const rt_triangle *tris, size_t n_tris, const rt_vertex *verts, size_t n_verts,
                      int n_threads) {
    if (!h || (n_tris && !tris) || (n_verts && !verts)) return RT_ERR_ARG;
    for (const RtTriRecord &R : h->r.recs) {
        if (R.tri >= n_tris) return RT_ERR_SCENE;
        for (int k = 0; k < 3; ++k) if (tris[R.tri].indices[k] >= n_verts) return RT_ERR_SCENE;
    }
    return rt_bvh_refit(&h->r, tris, verts, n_threads) ? RT_OK : RT_ERR_SCENE;
}
int rt_host_bvh_wide(const rt_host_bvh *h, void *wide_nodes, void *wide_src, size_t capacity_nodes, size_t *n_wide,
                     uint32_t *stack_need) {
    if (!h || !n_wide) return RT_ERR_ARG;
    RtWideTree w;
    try {
        rt_bvh_widen(h->r, &w);
    } catch (const std::bad_alloc &) {
        return fail(nullptr, RT_ERR_OOM, "rt_host_bvh_wide: host allocation failed");
    }
    *n_wide = w.nodes.size();
    if (stack_need) *stack_need = w.stack_need;
    if (wide_nodes || wide_src) {
        if (capacity_nodes < w.nodes.size()) return RT_ERR_ARG;
        if (wide_nodes && !w.nodes.empty()) std::memcpy(wide_nodes, w.nodes.data(), w.nodes.size() * sizeof(RtWideNode));
        if (wide_src && !w.src.empty()) std::memcpy(wide_src, w.src.data(), w.src.size() * sizeof(RtWideSrc));
    }
    return RT_OK;
}
void rt_host_bvh_free(rt_host_bvh *h) { delete h; }

int rt_upload_scene(rt_ctx *c, const rt_triangle *tris, size_t n_tris, const rt_mesh_info *meshes, size_t n_meshes,
                    const rt_vertex *verts, size_t n_verts, const rt_material *mats, size_t n_mats) {
    return rt_upload_scene_ex(c, tris, n_tris, meshes, n_meshes, verts, n_verts, mats, n_mats, nullptr, nullptr, 0);
}

int rt_upload_scene_ex(rt_ctx *c, const rt_triangle *tris, size_t n_tris, const rt_mesh_info *meshes, size_t n_meshes,
                       const rt_vertex *verts, size_t n_verts, const rt_material *mats, size_t n_mats,
                       const rt_sphere *spheres, const uint32_t *sphere_mats, size_t n_spheres) {
    if (!c) return RT_ERR_ARG;
    if (n_mats && !mats) return fail(c, RT_ERR_ARG, "rt_upload_scene: NULL materials with non-zero count");
    if (n_spheres && (!spheres || !sphere_mats)) return fail(c, RT_ERR_ARG, "rt_upload_scene_ex: NULL sphere arrays with non-zero count");
    if (n_spheres >= (1u << 28) || n_tris + n_spheres >= 0xFFFFFFFFull)
        return fail(c, RT_ERR_ARG, "rt_upload_scene_ex: too many spheres");
    {
        uint64_t items = n_spheres;
        for (size_t m = 0; m < n_meshes && meshes; ++m) items += meshes[m].tri_count;
        if (items >= (1u << 28)) return fail(c, RT_ERR_ARG, "rt_upload_scene_ex: too many leaf records (2^28 max)");
    }
    // validate before touching the GPU: a kernel must never see an out-of-range index
    {
        int vrc = validate_scene(c, tris, n_tris, meshes, n_meshes, verts, n_verts);
        if (vrc != RT_OK) return vrc;
    }

    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_all(c));
    free_scene(c);

    RtBvhBuildResult bvh;
    try {
        rt_bvh_build(tris, n_tris, meshes, n_meshes, verts, n_verts, 0, &bvh, spheres, n_spheres);
    } catch (const std::bad_alloc &) {
        return fail(c, RT_ERR_OOM, "rt_upload_scene: host allocation failed in the BVH build");
    }
    if (bvh.max_depth >= RT_MAX_DEPTH) return fail(c, RT_ERR_SCENE, "rt_upload_scene: BVH depth guard violated");

    int rc;
    if ((rc = upload(c, c->d_nodes, bvh.nodes.data(), bvh.nodes.size())) != RT_OK) return rc;
    if ((rc = upload(c, c->d_recs, bvh.recs.data(), bvh.recs.size())) != RT_OK) return rc;
    if ((rc = upload(c, c->d_tris, tris, n_tris)) != RT_OK) return rc;
    if ((rc = upload(c, c->d_meshes, meshes, n_meshes)) != RT_OK) return rc;
    if ((rc = upload(c, c->d_verts, verts, n_verts)) != RT_OK) return rc;
    if ((rc = upload(c, c->d_mats, mats, n_mats)) != RT_OK) return rc;
    if ((rc = upload(c, c->d_spheres, spheres, n_spheres)) != RT_OK) return rc;
    if ((rc = upload(c, c->d_sphere_mats, sphere_mats, n_spheres)) != RT_OK) return rc;
    {   // tree levels for the GPU refit: depth of every node (children are stored after their parent)
        const size_t nn = bvh.nodes.size();
        std::vector<uint32_t> depth(nn, 0), order(nn);
        uint32_t maxd = 0;
        for (size_t i = 0; i < nn; ++i) {
            const RtBvhNode &N = bvh.nodes[i];
            if (N.child0 >= 0) depth[N.child0] = depth[i] + 1;
            if (N.child1 >= 0) depth[N.child1] = depth[i] + 1;
            if (depth[i] > maxd) maxd = depth[i];
        }
        std::vector<uint32_t> count(nn ? maxd + 1 : 0, 0), start(nn ? maxd + 2 : 1, 0);
        for (size_t i = 0; i < nn; ++i) count[maxd - depth[i]]++;              // group 0 = deepest level
        for (size_t g = 0; g < count.size(); ++g) start[g + 1] = start[g] + count[g];
        std::vector<uint32_t> fill(start.begin(), start.end());
        for (size_t i = 0; i < nn; ++i) order[fill[maxd - depth[i]]++] = (uint32_t)i;
        if ((rc = upload(c, c->d_level_nodes, order.data(), nn)) != RT_OK) return rc;
        c->level_count = count;
    }
    c->wide_host = RtWideTree();
    bool wide_ok = c->tune_wide && !bvh.nodes.empty();
    if (wide_ok) {      // the root's two boxes bound every coordinate in the tree
        const RtBvhNode &R = bvh.nodes[0];
        for (int k = 0; k < 3; ++k) {
            float ext[4] = {R.lo0[k], R.hi0[k], R.lo1[k], R.hi1[k]};
            for (float v : ext) if (std::isfinite(v) ? std::fabs(v) >= RT_WIDE_COORD_MAX : !std::isnan(v)) wide_ok = false;
        }
    }
    if (wide_ok) {
        try {
            rt_bvh_widen(bvh, &c->wide_host);
        } catch (const std::bad_alloc &) {
            return fail(c, RT_ERR_OOM, "rt_upload_scene: host allocation failed in the wide-tree build");
        }
        // a tree whose worst-case walk does not fit the per-lane stack keeps the binary tree (RT_MAX_DEPTH bounds that one)
        if (c->wide_host.stack_need + 1 > RT_STACK_CAP) c->wide_host = RtWideTree();
        else {
            if ((rc = upload(c, c->d_wide, c->wide_host.nodes.data(), c->wide_host.nodes.size())) != RT_OK) return rc;
            if ((rc = upload(c, c->d_wide_src, c->wide_host.src.data(), c->wide_host.src.size())) != RT_OK) return rc;
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));   // host arrays are borrowed for this call only

    c->S.wide = (const RtWideNode *)c->d_wide;
    c->S.n_wide = (uint32_t)c->wide_host.nodes.size();
    c->S.nodes = (const RtBvhNode *)c->d_nodes;
    c->S.recs = (const RtTriRecord *)c->d_recs;
    c->S.tris = (const rt_triangle *)c->d_tris;
    c->S.meshes = (const rt_mesh_info *)c->d_meshes;
    c->S.verts = (const rt_vertex *)c->d_verts;
    c->S.mats = (const rt_material *)c->d_mats;
    c->S.spheres = (const rt_sphere *)c->d_spheres;
    c->S.sphere_mats = (const uint32_t *)c->d_sphere_mats;
    c->S.n_spheres = (uint32_t)n_spheres;
    c->S.n_nodes = (uint32_t)bvh.nodes.size();
    c->S.n_recs = (uint32_t)bvh.recs.size();
    c->S.n_tris = (uint32_t)n_tris;
    c->S.n_meshes = (uint32_t)n_meshes;
    c->S.n_verts = (uint32_t)n_verts;
    c->S.n_mats = (uint32_t)n_mats;
    uint32_t nb = c->info.n_builds + 1, nr = c->info.n_refits;
    fill_info(&c->info, bvh, n_meshes, n_verts);
    c->info.n_builds = nb; c->info.n_refits = nr;
    c->info.n_wide_nodes = c->wide_host.nodes.size();
    c->info.wide_max_depth = c->wide_host.max_depth; c->info.wide_stack_need = c->wide_host.stack_need;
    c->h_tris.assign(tris, tris + n_tris);
    c->h_meshes.assign(meshes, meshes + n_meshes);
    c->h_mesh_vlo.assign(n_meshes, 0xFFFFFFFFu); c->h_mesh_vhi.assign(n_meshes, 0u);
    c->h_vert_used.assign(n_verts, 0);
    for (size_t t = 0; t < n_tris; ++t) for (int k = 0; k < 3; ++k) c->h_vert_used[tris[t].indices[k]] = 1;
    for (size_t m = 0; m < n_meshes; ++m)
        for (uint32_t t = meshes[m].first_tri; t < meshes[m].first_tri + meshes[m].tri_count; ++t)
            for (int k = 0; k < 3; ++k) {
                c->h_mesh_vlo[m] = std::min(c->h_mesh_vlo[m], tris[t].indices[k]);
                c->h_mesh_vhi[m] = std::max(c->h_mesh_vhi[m], tris[t].indices[k]);
            }
    {   // parent / depth of every binary node (children are stored after their parent), the nodes above each mesh's
        // records, and the wide node of every binary node
        const size_t nn = bvh.nodes.size();
        c->h_parent.assign(nn, 0xFFFFFFFFu); c->h_depth.assign(nn, 0u); c->h_stamp.assign(nn, 0u); c->stamp = 0;
        c->h_mesh_node_off.assign(n_meshes + 1, 0u);
        auto each_leaf_mesh = [&](size_t i, auto &&fn) {      // fn(mesh) for the distinct meshes in node i's leaf children
            const RtBvhNode &N = bvh.nodes[i];
            uint32_t seen[16]; int ns = 0;
            for (int32_t code : {N.child0, N.child1}) {
                if (code >= 0 || code == RT_CHILD_NONE) continue;
                uint32_t v = (uint32_t)(~code), first = v >> 3, cnt = (v & 7u) + 1u;
                for (uint32_t k = first; k < first + cnt; ++k) {
                    const uint32_t m = bvh.recs[k].mesh;
                    if (m == RT_REC_SPHERE || m >= n_meshes) continue;
                    bool dup = false;
                    for (int q = 0; q < ns; ++q) dup = dup || seen[q] == m;
                    if (!dup) { seen[ns++] = m; fn(m); }
                }
            }
        };
        for (size_t i = 0; i < nn; ++i) {
            const RtBvhNode &N = bvh.nodes[i];
            if (N.child0 >= 0) { c->h_parent[N.child0] = (uint32_t)i; c->h_depth[N.child0] = c->h_depth[i] + 1; }
            if (N.child1 >= 0) { c->h_parent[N.child1] = (uint32_t)i; c->h_depth[N.child1] = c->h_depth[i] + 1; }
            each_leaf_mesh(i, [&](uint32_t m) { c->h_mesh_node_off[m + 1]++; });
        }
        for (size_t m = 0; m < n_meshes; ++m) c->h_mesh_node_off[m + 1] += c->h_mesh_node_off[m];
        c->h_mesh_nodes.assign(c->h_mesh_node_off[n_meshes], 0u);
        std::vector<uint32_t> fill(c->h_mesh_node_off.begin(), c->h_mesh_node_off.end() - 1);
        for (size_t i = 0; i < nn; ++i) each_leaf_mesh(i, [&](uint32_t m) { c->h_mesh_nodes[fill[m]++] = (uint32_t)i; });
        c->h_wide_of_bin.assign(nn, 0xFFFFFFFFu);
        for (size_t w = 0; w < c->wide_host.src.size(); ++w)
            for (int k = 0; k < 4; ++k)
                if (c->wide_host.src[w].slot[k] != RT_WIDE_EMPTY) c->h_wide_of_bin[c->wide_host.src[w].slot[k] >> 1] = (uint32_t)w;
    }
    c->bvh_host = std::move(bvh);
    c->bvh_host_stale = false;
    c->has_scene = true;
    c->accum_samples = 0;
    c->accum_dirty = true;           // the running sums belong to the scene before this one: the next path-traced frame restarts them
    invalidate_coverage(c);
    return RT_OK;
}

int rt_update_vertices(rt_ctx *c, const rt_vertex *verts, size_t n_verts, const rt_mesh_info *meshes, size_t n_meshes) {
    if (!c) return RT_ERR_ARG;
    if (!c->has_scene) return RT_NOT_READY;
    if ((n_verts && !verts) || (n_meshes && !meshes)) return fail(c, RT_ERR_ARG, "rt_update_vertices: NULL array");
    if (n_verts != c->S.n_verts || n_meshes != c->h_meshes.size())
        return fail(c, RT_ERR_SCENE, "rt_update_vertices: vertex or mesh count differs from the uploaded scene");
    for (size_t m = 0; m < n_meshes; ++m)
        if (meshes[m].first_tri != c->h_meshes[m].first_tri || meshes[m].tri_count != c->h_meshes[m].tri_count)
            return fail(c, RT_ERR_SCENE, "rt_update_vertices: mesh triangle ranges differ from the uploaded scene");
    if (c->bvh_host.dropped)   // triangles left out of the tree at build time could have become valid
        return fail(c, RT_ERR_SCENE, "rt_update_vertices: the built tree dropped non-finite triangles, rebuild with rt_upload_scene");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_all(c));          // frames in flight on other sets still walk the tree this call rewrites
    double refit_ms = 0.0;
    float upd_amax = 0.0f; bool upd_bad = false;
    scan_vertices(verts, n_verts, c->h_vert_used.size() == n_verts ? c->h_vert_used.data() : nullptr, &upd_amax, &upd_bad);
    if (std::getenv("RT_TUNE_HOST_REFIT")) {
        // the round's first version, kept for A/B and as the reference the GPU refit is tested against: refit on
        // the host, then re-upload nodes and records (0.6 GB over PCIe for the 7.2 M-triangle bench scene)
        if (c->bvh_host_stale) return fail(c, RT_ERR_SCENE, "rt_update_vertices: host refit after a GPU refit needs a rebuild");
        if (!rt_bvh_refit(&c->bvh_host, c->h_tris.data(), verts, 0))
            return fail(c, RT_ERR_SCENE, "rt_update_vertices: non-finite vertex, rebuild with rt_upload_scene");
        const RtBvhBuildResult &b = c->bvh_host;
        if (!b.nodes.empty()) HIPCHK(c, hipMemcpyAsync(c->d_nodes, b.nodes.data(), b.nodes.size() * sizeof(RtBvhNode), hipMemcpyHostToDevice, c->stream));
        if (!b.recs.empty()) HIPCHK(c, hipMemcpyAsync(c->d_recs, b.recs.data(), b.recs.size() * sizeof(RtTriRecord), hipMemcpyHostToDevice, c->stream));
        if (!c->wide_host.nodes.empty()) {
            rt_wide_requantise(b, &c->wide_host);
            HIPCHK(c, hipMemcpyAsync(c->d_wide, c->wide_host.nodes.data(), c->wide_host.nodes.size() * sizeof(RtWideNode), hipMemcpyHostToDevice, c->stream));
        }
        if (n_verts) HIPCHK(c, hipMemcpyAsync(c->d_verts, verts, n_verts * sizeof(rt_vertex), hipMemcpyHostToDevice, c->stream));
        if (n_meshes) HIPCHK(c, hipMemcpyAsync(c->d_meshes, meshes, n_meshes * sizeof(rt_mesh_info), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));   // host arrays are borrowed for this call only
        refit_ms = b.build_ms;
    } else {
        // GPU refit: only the moved data (T3 vertices, T2 mesh boxes) crosses PCIe; records are re-baked and the
        // boxes refitted on the device, one launch per tree level, deepest first.  A non-finite vertex is refused
        // up front (the tree would have to drop that triangle), before anything on the device changes.
        if (upd_bad)     // only vertices some triangle of the scene uses matter (as in the host refit)
            return fail(c, RT_ERR_SCENE, "rt_update_vertices: non-finite vertex, rebuild with rt_upload_scene");
        for (hipEvent_t &ev : c->ev_update) if (!ev) HIPCHK(c, hipEventCreate(&ev));
        const hipEvent_t e0 = c->ev_update[0], e1 = c->ev_update[1];
        if (n_verts) HIPCHK(c, hipMemcpyAsync(c->d_verts, verts, n_verts * sizeof(rt_vertex), hipMemcpyHostToDevice, c->stream));
        if (n_meshes) HIPCHK(c, hipMemcpyAsync(c->d_meshes, meshes, n_meshes * sizeof(rt_mesh_info), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipEventRecord(e0, c->stream));
        size_t off = 0;
        for (uint32_t cnt : c->level_count) {
            HIPCHK(c, rtk_launch_refit_level(c->stream, (RtBvhNode *)c->d_nodes, (RtTriRecord *)c->d_recs, (const rt_triangle *)c->d_tris,
                                             (const rt_vertex *)c->d_verts, (const uint32_t *)c->d_level_nodes + off, cnt));
            off += cnt;
        }
        // the wide tree's boxes are re-quantised from the refitted binary boxes (its topology does not change)
        HIPCHK(c, rtk_launch_wide_requant(c->stream, (const RtBvhNode *)c->d_nodes, (const RtWideSrc *)c->d_wide_src,
                                          (RtWideNode *)c->d_wide, (uint32_t)c->wide_host.nodes.size()));
        HIPCHK(c, hipEventRecord(e1, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));   // host arrays are borrowed for this call only
        float ms = 0.0f;
        HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
        refit_ms = ms;
        c->bvh_host_stale = true;
    }
    // moved vertices may have left the coordinate range the wide step is good for (rt_layout.h): then the binary tree
    // is walked until they are back (spheres do not move; unused vertices count too, which only errs on the safe side)
    c->S.n_wide = upd_amax * 1.0001f < RT_WIDE_COORD_MAX ? (uint32_t)c->wide_host.nodes.size() : 0u;     // an infinite coordinate (unused vertex) gives inf: binary tree
    c->h_meshes.assign(meshes, meshes + n_meshes);
    c->info.build_ms = refit_ms;
    c->info.n_refits += 1;
    c->accum_dirty = true;                        // geometry moved: accumulation restarts
    return RT_OK;
}

int rt_update_vertex_range(rt_ctx *c, const rt_vertex *verts, size_t first_vertex, size_t n_verts,
                           const rt_mesh_info *meshes, size_t first_mesh, size_t n_meshes) {
    if (!c) return RT_ERR_ARG;
    if (!c->has_scene) return RT_NOT_READY;
    if ((n_verts && !verts) || (n_meshes && !meshes)) return fail(c, RT_ERR_ARG, "rt_update_vertex_range: NULL array");
    if (first_vertex + n_verts > c->S.n_verts || first_vertex + n_verts < first_vertex)
        return fail(c, RT_ERR_SCENE, "rt_update_vertex_range: vertex range outside the uploaded scene");
    if (first_mesh + n_meshes > c->h_meshes.size() || first_mesh + n_meshes < first_mesh)
        return fail(c, RT_ERR_SCENE, "rt_update_vertex_range: mesh range outside the uploaded scene");
    for (size_t m = 0; m < n_meshes; ++m)
        if (meshes[m].first_tri != c->h_meshes[first_mesh + m].first_tri || meshes[m].tri_count != c->h_meshes[first_mesh + m].tri_count)
            return fail(c, RT_ERR_SCENE, "rt_update_vertex_range: mesh triangle ranges differ from the uploaded scene");
    if (c->bvh_host.dropped)
        return fail(c, RT_ERR_SCENE, "rt_update_vertex_range: the built tree dropped non-finite triangles, rebuild with rt_upload_scene");
    // the vertex range must hold every vertex the moved meshes use and none that another mesh uses (a leaf that is not
    // re-baked would keep stale values): checked on the per-mesh vertex index spans recorded at upload
    for (size_t m = 0; m < c->h_meshes.size(); ++m) {
        if (c->h_meshes[m].tri_count == 0) continue;
        const bool moved = m >= first_mesh && m < first_mesh + n_meshes;
        const size_t lo = c->h_mesh_vlo[m], hi = c->h_mesh_vhi[m];      // inclusive span
        if (moved && (lo < first_vertex || hi >= first_vertex + n_verts))
            return fail(c, RT_ERR_SCENE, "rt_update_vertex_range: a moved mesh uses a vertex outside the range");
        if (!moved && n_verts && hi >= first_vertex && lo < first_vertex + n_verts)
            return fail(c, RT_ERR_SCENE, "rt_update_vertex_range: the range overlaps the vertices of a mesh that is not listed as moved");
    }
    float amax = 0.0f;
    for (size_t i = 0; i < n_verts; ++i)
        for (int k = 0; k < 3; ++k) {
            if (!std::isfinite(verts[i].pos[k])) return fail(c, RT_ERR_SCENE, "rt_update_vertex_range: non-finite vertex, rebuild with rt_upload_scene");
            amax = std::fmax(amax, std::fabs(verts[i].pos[k]));
        }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_all(c));          // frames in flight on other sets still walk the tree this call rewrites
    const size_t nn = c->S.n_nodes;
    // Which nodes have a moved mesh below them: walked on the host's copy of the topology (mesh-id spans per node), a few
    // thousand nodes for a mesh of a few hundred triangles.  A short list is redone by ONE launch of one workgroup
    // (k_refit_list); a medium one -- tens to hundreds of meshes -- by one launch per tree level over the LISTED nodes of that
    // level (by_level_lists); a long one by the per-level launches over the whole tree with a dirty byte per node.
    // Making a long list costs the host more than the whole-tree launches cost the device (climbing 55,000 nodes for 50
    // spheres: 1 ms of cache misses), so a long list is made only when the call before this one named the same meshes -- an
    // animation -- and then kept on the device: every further frame of it pays the launches alone.
    std::vector<uint32_t> list, level_end;
    uint32_t n_wide_listed = 0;
    bool by_list = false, by_level_lists = false, cached = false;
    const size_t list_limit = std::max(c->tune_refit_list_max, c->tune_refit_levels_max);
    const bool repeated = c->last_range_first == first_mesh && c->last_range_n == n_meshes;
    c->last_range_first = first_mesh; c->last_range_n = n_meshes;
    // leaf-holding nodes of the moved meshes (CSR offsets): the list is about three times as long
    const size_t est = nn && n_meshes && c->h_mesh_node_off.size() > first_mesh + n_meshes ?
                       3 * (size_t)(c->h_mesh_node_off[first_mesh + n_meshes] - c->h_mesh_node_off[first_mesh]) : 0;
    if (c->list_cache.valid && c->list_cache.first_mesh == first_mesh && c->list_cache.n_meshes == n_meshes) {
        cached = by_list = true;
        by_level_lists = c->list_cache.level_lists;
        level_end = c->list_cache.level_end;
        n_wide_listed = c->list_cache.n_wide;
    } else if (nn && n_meshes && list_limit && c->h_parent.size() == nn && (est <= c->tune_refit_list_max || repeated)) {
        std::vector<std::vector<uint32_t>> per_level;
        size_t total = 0;
        by_list = true;
        if (++c->stamp == 0u) { std::fill(c->h_stamp.begin(), c->h_stamp.end(), 0u); c->stamp = 1u; }
        for (size_t m = first_mesh; m < first_mesh + n_meshes && by_list; ++m)
            for (uint32_t q = c->h_mesh_node_off[m]; q < c->h_mesh_node_off[m + 1] && by_list; ++q)
                for (uint32_t n = c->h_mesh_nodes[q]; n != 0xFFFFFFFFu && c->h_stamp[n] != c->stamp; n = c->h_parent[n]) {   // up to the first node already listed
                    c->h_stamp[n] = c->stamp;
                    const uint32_t d = c->h_depth[n];
                    if (++total > list_limit || d + 1 >= RT_REFIT_LIST_MAX_LEVELS) { by_list = false; break; }
                    if (per_level.size() <= d) per_level.resize(d + 1);
                    per_level[d].push_back(n);
                }
        by_level_lists = by_list && total > c->tune_refit_list_max;
        if (by_list) {
            for (size_t d = per_level.size(); d-- > 0;) {           // deepest level first
                list.insert(list.end(), per_level[d].begin(), per_level[d].end());
                level_end.push_back((uint32_t)list.size());
            }
            const size_t n_bin = list.size();
            if (!c->wide_host.nodes.empty()) {
                std::vector<uint32_t> w;
                w.reserve(n_bin);
                for (size_t i = 0; i < n_bin; ++i) if (c->h_wide_of_bin[list[i]] != 0xFFFFFFFFu) w.push_back(c->h_wide_of_bin[list[i]]);
                std::sort(w.begin(), w.end());
                w.erase(std::unique(w.begin(), w.end()), w.end());
                n_wide_listed = (uint32_t)w.size();
                list.insert(list.end(), w.begin(), w.end());
            }
            c->list_cache.valid = false;                            // d_list is about to hold another list
            if (list.size() > c->list_capacity) {
                HIPCHK(c, hipStreamSynchronize(c->stream));
                free_dev(c->d_list); c->list_capacity = 0;
                const size_t cap = std::max<size_t>(list.size() * 2, 4096);
                HIPCHK(c, hipMalloc(&c->d_list, cap * sizeof(uint32_t)));
                c->list_capacity = cap;
            }
        }
    }
    if (!by_list && !c->d_dirty && nn) HIPCHK(c, hipMalloc(&c->d_dirty, nn));
    for (hipEvent_t &ev : c->ev_update) if (!ev) HIPCHK(c, hipEventCreate(&ev));
    const hipEvent_t e0 = c->ev_update[0], e1 = c->ev_update[1];
    if (n_verts) HIPCHK(c, hipMemcpyAsync((rt_vertex *)c->d_verts + first_vertex, verts, n_verts * sizeof(rt_vertex), hipMemcpyHostToDevice, c->stream));
    if (n_meshes) HIPCHK(c, hipMemcpyAsync((rt_mesh_info *)c->d_meshes + first_mesh, meshes, n_meshes * sizeof(rt_mesh_info), hipMemcpyHostToDevice, c->stream));
    if (by_list && !cached && !list.empty()) HIPCHK(c, hipMemcpyAsync(c->d_list, list.data(), list.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipEventRecord(e0, c->stream));
    if (by_level_lists) {
        uint32_t begin = 0;
        for (uint32_t end : level_end) {                            // a launch per level, over that level's listed nodes only
            HIPCHK(c, rtk_launch_refit_level(c->stream, (RtBvhNode *)c->d_nodes, (RtTriRecord *)c->d_recs, (const rt_triangle *)c->d_tris,
                                             (const rt_vertex *)c->d_verts, (const uint32_t *)c->d_list + begin, end - begin));
            begin = end;
        }
        HIPCHK(c, rtk_launch_wide_requant_list(c->stream, (const RtBvhNode *)c->d_nodes, (const RtWideSrc *)c->d_wide_src, (RtWideNode *)c->d_wide,
                                               (const uint32_t *)c->d_list + begin, n_wide_listed));
    } else if (by_list) {
        if (!level_end.empty())
            HIPCHK(c, rtk_launch_refit_list(c->stream, (RtBvhNode *)c->d_nodes, (RtTriRecord *)c->d_recs, (const rt_triangle *)c->d_tris,
                                            (const rt_vertex *)c->d_verts, (const uint32_t *)c->d_list, level_end.data(), (uint32_t)level_end.size(),
                                            (const RtWideSrc *)c->d_wide_src, (RtWideNode *)c->d_wide, n_wide_listed));
    } else {
        if (nn) HIPCHK(c, hipMemsetAsync(c->d_dirty, 0, nn, c->stream));
        size_t off = 0;
        for (uint32_t cnt : c->level_count) {
            HIPCHK(c, rtk_launch_refit_level_dirty(c->stream, (RtBvhNode *)c->d_nodes, (RtTriRecord *)c->d_recs, (const rt_triangle *)c->d_tris,
                                                   (const rt_vertex *)c->d_verts, (const uint32_t *)c->d_level_nodes + off, cnt,
                                                   (uint32_t)first_mesh, (uint32_t)(first_mesh + n_meshes), (uint8_t *)c->d_dirty));
            off += cnt;
        }
        HIPCHK(c, rtk_launch_wide_requant_dirty(c->stream, (const RtBvhNode *)c->d_nodes, (const RtWideSrc *)c->d_wide_src,
                                                (RtWideNode *)c->d_wide, (uint32_t)c->wide_host.nodes.size(), (const uint8_t *)c->d_dirty));
    }
    HIPCHK(c, hipEventRecord(e1, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));   // host arrays are borrowed for this call only
    float ms = 0.0f;
    HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
    c->bvh_host_stale = true;
    if (by_list && !cached && !list.empty()) {       // the list has arrived (the stream was synchronised): good for the next call with this range
        c->list_cache.valid = true; c->list_cache.level_lists = by_level_lists;
        c->list_cache.first_mesh = first_mesh; c->list_cache.n_meshes = n_meshes;
        c->list_cache.level_end = level_end; c->list_cache.n_wide = n_wide_listed;
    }
    // out of the wide step's coordinate range: binary tree until a full update or upload shows everything is back in
    if (!(amax * 1.0001f < RT_WIDE_COORD_MAX)) c->S.n_wide = 0;
    for (size_t m = 0; m < n_meshes; ++m) c->h_meshes[first_mesh + m] = meshes[m];
    c->info.build_ms = ms;
    c->info.n_refits += 1;
    c->accum_dirty = true;
    return RT_OK;
}

int rt_set_view(rt_ctx *c, const float m[16], const float pos[3], uint32_t width, uint32_t height,
                const uint32_t viewport[4]) {
    if (!c || !m || !pos) return RT_ERR_ARG;
    if (width == 0 || height == 0 || (uint64_t)width * height > (1ull << 31))
        return fail(c, RT_ERR_ARG, "rt_set_view: bad target size");
    uint32_t vx = 0, vy = 0, vw = width, vh = height;
    if (viewport) {
        vx = viewport[0]; vy = viewport[1]; vw = viewport[2]; vh = viewport[3];
        if (vw == 0 || vh == 0 || (uint64_t)vx + vw > width || (uint64_t)vy + vh > height)
            return fail(c, RT_ERR_ARG, "rt_set_view: viewport outside the target");
    }
    // a NaN or infinite matrix / position makes every ray of the frame NaN: refused here, not traced
    for (int k = 0; k < 16; ++k) if (!std::isfinite(m[k])) return fail(c, RT_ERR_ARG, "rt_set_view: world_from_clip is not finite");
    for (int k = 0; k < 3; ++k) if (!std::isfinite(pos[k])) return fail(c, RT_ERR_ARG, "rt_set_view: world_position is not finite");
    HIPCHK(c, hipSetDevice(c->device));
    size_t n_px = (size_t)width * height;
    if (n_px != c->n_px || c->V.width != width) {
        HIPCHK(c, sync_all(c));
        // until every buffer of the new size exists there is no view: a failed allocation leaves rt_render at
        // RT_NOT_READY instead of launching on freed or NULL buffers
        const bool had_view = c->has_view;
        c->has_view = false;
        invalidate_coverage(c);
        free_frame(c);
        for (uint32_t k = 0; k < c->n_sets; ++k) {
            HIPCHK(c, hipMalloc((void **)&c->fs[k].d_rgba, n_px * 16));
            HIPCHK(c, hipMalloc((void **)&c->fs[k].d_t, n_px * 4));
            HIPCHK(c, hipMalloc((void **)&c->fs[k].d_prim, n_px * 4));
        }
        HIPCHK(c, hipMalloc((void **)&c->d_accum, n_px * 16));
        select_set(c, c->cur_set);
        c->n_px = n_px;
        c->accum_samples = 0;
        c->accum_dirty = true;       // the new accumulation buffer is uninitialised: restart, whatever the camera and coverage are
        c->has_view = had_view;
    }
    if (c->has_view && (std::memcmp(c->V.m, m, 64) != 0 || std::memcmp(c->V.pos, pos, 12) != 0))
        c->accum_dirty = true;       // accumulation resets on camera movement (intent of src/scene.rs:94-96)
    std::memcpy(c->V.m, m, 64);
    std::memcpy(c->V.pos, pos, 12);
    c->V.width = width; c->V.height = height;
    c->V.vx = vx; c->V.vy = vy; c->V.vw = vw; c->V.vh = vh;
    c->has_view = true;
    return RT_OK;
}

// Where one frame's results and scratch live: the context's own buffers (rt_render) or the caller's outputs plus
// the static scratch array (rt_render_static).
struct RenderTargets {
    float *rgba, *accum, *t;
    uint32_t *prim;
    unsigned long long *cnt;       // counters + work-queue heads (RT_CNT_N words)
    RtPathQueue qa, qb;            // path queues, capacity q_cap paths each
    float *contrib;                // per-sample contribution planes: max_batch planes of n_px float4
    size_t q_cap;
    uint32_t max_batch;            // most samples of a pixel traced per launch (16, 4 or 1)
};

static int clear_outputs(rt_ctx *c, const RenderTargets &T) {
    HIPCHK(c, hipMemsetAsync(T.rgba, 0, c->n_px * 16, c->stream));
    HIPCHK(c, hipMemsetAsync(T.t, 0, c->n_px * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(T.prim, 0xFF, c->n_px * 4, c->stream));
    return RT_OK;
}

// rt_render's parameter block, checked and defaulted; the shard description for the kernels
static int check_params(rt_ctx *c, const rt_render_params *params, rt_render_params &P, RtShardDev &sh, bool &partial) {
    std::memset(&P, 0, sizeof P);
    if (params) {
        if (params->struct_size != sizeof(rt_render_params)) return fail(c, RT_ERR_ARG, "rt_render: struct_size mismatch");
        P = *params;
    }
    if (!c->has_scene || !c->has_view || !c->n_px || !c->d_rgba) return RT_NOT_READY;   // src/node.rs:52-56: skip silently
    if (P.shard_count == 0) P.shard_count = 1;
    // default tile: 64 px when tiles are what ranks own, 32 px when they are only k_stream's unit of work
    // distribution over the XCD queues (tools/tile_sweep.py: 16 / 32 / 64 / 128 / 256 px: 0.82 / 0.79 / 0.80 / 0.81 / 0.89 ms)
    if (P.tile_size == 0) P.tile_size = P.shard_count > 1 ? 64 : 32;
    if (P.tile_size % RT_WORK_TILE != 0 || P.tile_size > 4096) return fail(c, RT_ERR_ARG, "rt_render: tile_size must be a multiple of 16");
    if (P.shard_rank >= P.shard_count) return fail(c, RT_ERR_ARG, "rt_render: shard_rank >= shard_count");
    sh.rank = P.shard_rank; sh.count = P.shard_count; sh.tile = P.tile_size;
    sh.tiles_x = (c->V.vw + sh.tile - 1) / sh.tile;
    sh.tiles_y = (c->V.vh + sh.tile - 1) / sh.tile;
    uint32_t n_tiles = sh.tiles_x * sh.tiles_y;
    sh.n_owned = n_tiles > sh.rank ? (n_tiles - sh.rank + sh.count - 1) / sh.count : 0;
    partial = sh.count > 1 || c->V.vw != c->V.width || c->V.vh != c->V.height;
    for (int k = 0; k < 3; ++k)      // a camera beyond the wide step's coordinate range: binary tree (rt_layout.h)
        if (!(std::fabs(c->V.pos[k]) < RT_WIDE_COORD_MAX)) P.flags |= RT_FLAG_BINARY_TREE;
    if (P.flags & RT_FLAG_PATH_TRACE) {
        if (P.spp == 0) P.spp = 1;
        if (P.max_segments == 0) P.max_segments = 4;   // reference shape: 4 segments (wgsl:185)
        if (c->n_px > ((size_t)1 << 26)) return fail(c, RT_ERR_ARG, "rt_render: path tracing supports at most 2^26 pixels");
    }
    return RT_OK;
}

// samples of a pixel traced per launch: 16 or 4 where there are that many left and the queues hold them (rt_kernels.hip
// k_path_start), else 1.  Bounded by the queue capacity, RT_TUNE_SAMPLE_BATCH and the magnified grid's 2^31 threads.
static uint32_t batch_of(const rt_ctx *c, const RenderTargets &T, uint32_t left) {
    uint32_t k2 = 1;
    for (uint32_t cand : {4u, 16u})
        if (cand <= left && cand <= T.max_batch && c->n_px * cand <= T.q_cap &&
            ((uint64_t)c->V.width << (cand == 4u ? 1 : 2)) * ((uint64_t)c->V.height << (cand == 4u ? 1 : 2)) <= (1ull << 31)) k2 = cand;
    return k2;
}

// The launches of one frame.  No allocation, no synchronisation: everything is enqueued on the context's stream.
// Path tracing: the tracing launches touch only this frame's own queues and contribution planes; the running sums are ONE
// buffer, so the first thing that touches them -- the restart's clear (restart_accum) or the first batch's additions --
// waits for wait_accum (the end of the path-traced frame before this one, possibly on another set's stream; may be NULL).
static int launch_frame(rt_ctx *c, const rt_render_params &P, const RtShardDev &sh, bool partial, const RenderTargets &T,
                        uint32_t *accum_samples, uint32_t *launches_out, bool restart_accum = false, hipEvent_t wait_accum = nullptr) {
    HIPCHK(c, hipMemsetAsync(T.cnt, 0, RT_CNT_N * sizeof(unsigned long long), c->stream));
    uint32_t launches = 0;
    c->tune.frames_in_flight = (c->ext_stream || c->use_null_stream) ? 1u : c->n_sets;     // on one stream frames do not overlap
    if (!(P.flags & RT_FLAG_PATH_TRACE)) {
        HIPCHK(c, rtk_launch_primary(c->stream, c->S, c->V, sh, P.flags, T.rgba, T.prim, T.t, T.cnt, c->tune));
        launches = 1;
    } else {
        uint32_t cap1 = (uint32_t)((size_t)rtk_primary_blocks(sh) * RT_BLOCK);
        if (cap1 > c->n_px) cap1 = (uint32_t)c->n_px;
        for (uint32_t s = 0, b = 0; s < P.spp; s += b) {
            b = batch_of(c, T, P.spp - s);                 // 16, 4 or 1 samples of every pixel in this round of launches
            const uint32_t kshift = b == 16u ? 2u : (b == 4u ? 1u : 0u);
            const uint32_t cap = cap1 * b;
            // k_path_start writes every covered (pixel, slot) of the planes; only pixels this call does not cover
            // (outside the viewport, other ranks' tiles) have to be cleared for k_accumulate_batch
            if (partial) HIPCHK(c, hipMemsetAsync(T.contrib, 0, c->n_px * 16 * b, c->stream));
            HIPCHK(c, hipMemsetAsync(T.cnt + RT_CNT_QUEUE_A, 0, 2 * sizeof(unsigned long long), c->stream));
            HIPCHK(c, rtk_launch_path_start(c->stream, c->S, c->V, sh, P.flags, P.sample_first + s, P.seed, P.stream_id,
                                            P.max_segments, T.contrib, (uint32_t)c->n_px, kshift, T.qa, T.cnt, RT_CNT_QUEUE_A, c->tune));
            ++launches;
            int in = RT_CNT_QUEUE_A, out = RT_CNT_QUEUE_B;
            for (uint32_t seg = 1; seg < P.max_segments; ++seg) {
                HIPCHK(c, hipMemsetAsync(T.cnt + out, 0, sizeof(unsigned long long), c->stream));
                HIPCHK(c, rtk_launch_path_bounce(c->stream, c->S, P.flags, cap, seg + 1 == P.max_segments, T.contrib, (uint32_t)c->n_px,
                                                 in == RT_CNT_QUEUE_A ? T.qa : T.qb, in == RT_CNT_QUEUE_A ? T.qb : T.qa,
                                                 T.cnt, in, out, c->tune));
                ++launches;
                int tmp = in; in = out; out = tmp;
            }
            if (s == 0) {
                if (wait_accum) HIPCHK(c, hipStreamWaitEvent(c->stream, wait_accum, 0));
                if (restart_accum) HIPCHK(c, hipMemsetAsync(T.accum, 0, c->n_px * 16, c->stream));
            }
            HIPCHK(c, rtk_launch_accumulate_batch(c->stream, T.accum, T.contrib, c->n_px, b));
        }
        *accum_samples += P.spp;
        HIPCHK(c, rtk_launch_resolve(c->stream, T.accum, T.rgba, c->n_px, 0.0f, *accum_samples));
    }
    *launches_out = launches;
    return RT_OK;
}

int rt_render(rt_ctx *c, const rt_render_params *params) {
    if (!c) return RT_ERR_ARG;
    rt_render_params P;
    RtShardDev sh{};
    bool partial = false;
    int rc = check_params(c, params, P, sh, partial);
    if (rc != RT_OK) return rc;
    HIPCHK(c, hipSetDevice(c->device));

    // Frames in flight: frames take the sets in turn (each set has its own stream, so frame k + 1 fills the compute units
    // frame k's last regions leave idle).  Path-traced frames alternate between the first two sets only (their queues are
    // large: 104 B per path; two in flight give what there is to gain, tools/path_two_contexts.py) and share ONE
    // accumulation buffer: a frame's tracing overlaps with the frame before it, its additions wait for that frame's end.
    const bool path_frame = (P.flags & RT_FLAG_PATH_TRACE) != 0;
    if (c->n_sets > 1) select_set(c, (c->cur_set + 1) % (path_frame ? std::min(c->n_sets, RT_PATH_SETS) : c->n_sets));
    FrameSet &F = c->fs[c->cur_set];
    RenderTargets T{};
    bool restart_accum = false;
    T.rgba = c->d_rgba; T.accum = c->d_accum; T.t = c->d_t; T.prim = c->d_prim; T.cnt = c->d_cnt;
    // Pixels this call does not cover (outside the viewport, other ranks' tiles) must read as cleared; clear only
    // when the coverage changes.
    const uint32_t cov[8] = {sh.rank, sh.count, sh.tile, c->V.vx, c->V.vy, c->V.vw, c->V.vh, partial ? 1u : 0u};
    if (!F.coverage_valid || std::memcmp(cov, F.coverage, sizeof cov) != 0) {
        rc = clear_outputs(c, T);
        if (rc != RT_OK) return rc;
        std::memcpy(F.coverage, cov, sizeof cov);
        F.coverage_valid = true;
    }
    if (P.flags & RT_FLAG_PATH_TRACE) {
        T.max_batch = c->tune_sample_batch;
        T.q_cap = (size_t)64 << 20;                      // at most 64 M queued paths
        const uint32_t batch = batch_of(c, T, P.spp);
        rc = ensure_queues(c, F, batch);
        if (rc != RT_OK) return rc;
        T.q_cap = F.q_capacity;
        T.max_batch = std::min(c->tune_sample_batch, F.contrib_planes);
        T.qa = queue_view(F.q[0], F.q_capacity); T.qb = queue_view(F.q[1], F.q_capacity);
        T.contrib = F.d_contrib;
        // the running sums restart unless asked to go on -- and also when the camera or the geometry moved or the coverage
        // is not the one they were started under (intent of src/scene.rs:94-96); the clear is enqueued by launch_frame,
        // behind the frame before this one
        if (!(P.flags & RT_FLAG_ACCUMULATE) || c->accum_dirty || std::memcmp(cov, c->accum_cov, sizeof cov) != 0) {
            restart_accum = true;
            c->accum_samples = 0;
            c->accum_dirty = false;
            std::memcpy(c->accum_cov, cov, sizeof cov);
        }
    }

    const bool timed = (P.flags & RT_FLAG_TIME_KERNELS) != 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (timed) {
        if (c->ev_used + 2 > c->ev.size()) {
            for (int i = 0; i < 2; ++i) {
                hipEvent_t ev;
                HIPCHK(c, hipEventCreate(&ev));
                c->ev.push_back(ev);
            }
        }
        ev0 = c->ev[c->ev_used]; ev1 = c->ev[c->ev_used + 1];
        HIPCHK(c, hipEventRecord(ev0, c->stream));
    }
    uint32_t launches = 0;
    rc = launch_frame(c, P, sh, partial, T, &c->accum_samples, &launches, restart_accum, path_frame && c->accum_pending ? c->ev_accum : nullptr);
    if (rc != RT_OK) return rc;
    if (path_frame) {
        HIPCHK(c, hipEventRecord(c->ev_accum, c->stream));
        c->accum_pending = true;
    }
    if (timed) {
        HIPCHK(c, hipEventRecord(ev1, c->stream));
        c->ev_used += 2; c->counters.kernel_launches += launches;
    }
    HIPCHK(c, hipMemcpyAsync(c->h_cnt, c->d_cnt, RT_CNT_N * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    return RT_OK;
}

// ---- rt_render_static: the same frame, allocating nothing (include/rt_amd.h) -----------------------------------------
// Outputs go to the caller's device buffers; the frame's scratch -- work-queue heads, counters, path queues, contribution
// plane -- lives in the static device array of rt_kernels.hip, whose address on this context's device rt_create looked
// up once (c->static_scratch).  No hipMalloc / hipFree / hipEventCreate on this path, first call included.
namespace {
constexpr size_t kStaticCntBytes = 4096;                                   // RT_CNT_N * 8 = 1152 B, rounded up
constexpr size_t kStaticPathBytesPerPx = 2 * 11 * 4 + 16;                  // two queues of 11 dwords + one float4 contribution plane
constexpr size_t kStaticAlignSlack = 16 * 256;                             // every sub-buffer starts on a 256-B boundary
constexpr size_t kStaticMaxPathPx = (RT_STATIC_SCRATCH_BYTES - kStaticCntBytes - kStaticAlignSlack) / kStaticPathBytesPerPx;
static_assert(kStaticMaxPathPx == RT_STATIC_MAX_PATH_PIXELS, "include/rt_amd.h documents this limit");
}  // namespace
int rt_render_static(rt_ctx *c, const rt_render_params *params, void *d_radiance, void *d_prim_ids, void *d_hit_t,
                     void *d_accum, size_t n_pixels) {
    if (!c) return RT_ERR_ARG;
    rt_render_params P;
    RtShardDev sh{};
    bool partial = false;
    int rc = check_params(c, params, P, sh, partial);
    if (rc != RT_OK) return rc;
    const bool path = (P.flags & RT_FLAG_PATH_TRACE) != 0;
    if (!d_radiance || !d_prim_ids || !d_hit_t || (path && !d_accum))
        return fail(c, RT_ERR_ARG, "rt_render_static: NULL output buffer (radiance, prim ids, hit t; accumulation when path tracing)");
    if (n_pixels != c->n_px) return fail(c, RT_ERR_ARG, "rt_render_static: n_pixels must be width * height of the current view");
    if (P.flags & RT_FLAG_TIME_KERNELS) return fail(c, RT_ERR_ARG, "rt_render_static: RT_FLAG_TIME_KERNELS creates events; time the stream yourself");
    if (!c->static_scratch) return fail(c, RT_ERR_HIP, "rt_render_static: the static scratch array was not found on this device");
    if (path && c->n_px > kStaticMaxPathPx) {
        char msg[200];
        snprintf(msg, sizeof msg, "rt_render_static: path tracing needs %zu B of scratch per pixel; the static array (%u MiB) holds frames of at most %zu pixels, this one has %zu",
                 kStaticPathBytesPerPx, RT_STATIC_SCRATCH_BYTES >> 20, kStaticMaxPathPx, c->n_px);
        return fail(c, RT_ERR_ARG, msg);
    }
    HIPCHK(c, hipSetDevice(c->device));
    // ONE stream for every static call of a context, whatever rt_render did in between: with frames in flight the sets (and
    // their streams) take turns, and two static calls on two streams would race on the one scratch array.
    // Set 0's stream it is; the context's "frame rendered last" aliases then name set 0.
    if (c->n_sets > 1) select_set(c, 0);
    // Under stream capture (a caller recording this call into a HIP graph; the null stream cannot be captured) every
    // argument of every launch is frozen into the graph: see include/rt_amd.h for what a replay may rely on.  What is wrong
    // by construction is refused: the divisor of an accumulating frame's resolve is host state (samples so far).
    {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (c->stream && hipStreamIsCapturing(c->stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
        if (cap == hipStreamCaptureStatusActive && path && (P.flags & RT_FLAG_ACCUMULATE))
            return fail(c, RT_ERR_ARG, "rt_render_static: RT_FLAG_ACCUMULATE cannot be captured into a graph (the sample count that divides the sums is host state)");
        if (cap == hipStreamCaptureStatusInvalidated)
            return fail(c, RT_ERR_HIP, "rt_render_static: the stream's capture has been invalidated");
    }

    RenderTargets T{};
    T.rgba = (float *)d_radiance; T.prim = (uint32_t *)d_prim_ids; T.t = (float *)d_hit_t; T.accum = (float *)d_accum;
    unsigned char *base = (unsigned char *)c->static_scratch;      // the device address, never the symbol
    T.cnt = (unsigned long long *)base;
    if (path) {
        size_t off = kStaticCntBytes;
        auto take = [&](size_t bytes) { void *p = base + off; off = (off + bytes + 255) & ~(size_t)255; return p; };
        QueueMem q[2];
        for (int i = 0; i < 2; ++i) { q[i].f = (float *)take(c->n_px * 9 * 4); q[i].u = (uint32_t *)take(c->n_px * 2 * 4); }
        T.contrib = (float *)take(c->n_px * 16);
        if (off > RT_STATIC_SCRATCH_BYTES) return fail(c, RT_ERR_ARG, "rt_render_static: frame does not fit the static scratch array");
        T.qa = queue_view(q[0], c->n_px); T.qb = queue_view(q[1], c->n_px);
        T.q_cap = c->n_px; T.max_batch = 1;               // one sample per launch: same sums, bit for bit
        if (!(P.flags & RT_FLAG_ACCUMULATE)) {
            HIPCHK(c, hipMemsetAsync(T.accum, 0, c->n_px * 16, c->stream));
            c->static_accum_samples = 0;
        }
    }
    // the caller's buffers carry no history the library knows of: pixels this call does not cover are cleared every time
    if (partial) { rc = clear_outputs(c, T); if (rc != RT_OK) return rc; }
    uint32_t launches = 0;
    rc = launch_frame(c, P, sh, partial, T, &c->static_accum_samples, &launches);
    if (rc != RT_OK) return rc;
    HIPCHK(c, hipMemcpyAsync(c->h_cnt, T.cnt, RT_CNT_N * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    return RT_OK;
}

// ---- multi-GPU tile exchange -----------------------------------------------------------------------------
static int shard_of(rt_ctx *c, uint32_t rank, uint32_t count, uint32_t tile, RtShardDev &sh, const char *who) {
    if (!c->has_view || !c->d_rgba) return RT_NOT_READY;
    if (count == 0) count = 1;
    if (tile == 0) tile = 64;
    if (tile % RT_WORK_TILE != 0 || tile > 4096 || rank >= count) return fail(c, RT_ERR_ARG, who);
    sh.rank = rank; sh.count = count; sh.tile = tile;
    sh.tiles_x = (c->V.vw + tile - 1) / tile;
    sh.tiles_y = (c->V.vh + tile - 1) / tile;
    const uint32_t n_tiles = sh.tiles_x * sh.tiles_y;
    sh.n_owned = n_tiles > rank ? (n_tiles - rank + count - 1) / count : 0;
    return RT_OK;
}
int rt_tile_slot_bytes(rt_ctx *c, uint32_t shard_count, uint32_t tile_size, size_t *bytes) {
    if (!c || !bytes) return RT_ERR_ARG;
    RtShardDev sh{};
    int rc = shard_of(c, 0, shard_count, tile_size, sh, "rt_tile_slot_bytes: bad shard_count / tile_size");
    if (rc != RT_OK) return rc;
    *bytes = (size_t)sh.n_owned * sh.tile * sh.tile * 16;     // rank 0 owns the most tiles
    return RT_OK;
}
int rt_pack_tiles(rt_ctx *c, uint32_t shard_rank, uint32_t shard_count, uint32_t tile_size, void *d_packed, size_t bytes) {
    if (!c || !d_packed) return RT_ERR_ARG;
    RtShardDev sh{};
    int rc = shard_of(c, shard_rank, shard_count, tile_size, sh, "rt_pack_tiles: bad shard_rank / shard_count / tile_size");
    if (rc != RT_OK) return rc;
    if (bytes < (size_t)sh.n_owned * sh.tile * sh.tile * 16) return fail(c, RT_ERR_ARG, "rt_pack_tiles: buffer smaller than this rank's tiles");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, rtk_launch_tile_pack(c->stream, c->d_rgba, (float *)d_packed, c->V, sh));
    return RT_OK;
}
int rt_unpack_tiles(rt_ctx *c, uint32_t first_rank, uint32_t n_ranks, uint32_t shard_count, uint32_t tile_size,
                    const void *d_packed, size_t slot_bytes) {
    if (!c || !d_packed) return RT_ERR_ARG;
    if (n_ranks == 0) return RT_OK;
    RtShardDev sh{};
    int rc = shard_of(c, first_rank, shard_count, tile_size, sh, "rt_unpack_tiles: bad first_rank / shard_count / tile_size");
    if (rc != RT_OK) return rc;
    if ((uint64_t)first_rank + n_ranks > sh.count) return fail(c, RT_ERR_ARG, "rt_unpack_tiles: first_rank + n_ranks > shard_count");
    const uint32_t n_tiles = sh.tiles_x * sh.tiles_y;
    const size_t need = (size_t)((n_tiles + sh.count - 1) / sh.count) * sh.tile * sh.tile * 16;
    if (slot_bytes < need || slot_bytes % 16 != 0) return fail(c, RT_ERR_ARG, "rt_unpack_tiles: slot_bytes smaller than rt_tile_slot_bytes (or not a multiple of 16)");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, rtk_launch_tile_unpack(c->stream, c->d_rgba, (const float *)d_packed, slot_bytes / 16, c->V, sh.tile, sh.tiles_x, n_tiles,
                                     sh.count, first_rank, n_ranks));
    return RT_OK;
}

int rt_get_counters(rt_ctx *c, rt_counters *out) {
    if (!c || !out) return RT_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->ev_used && c->n_sets > 1) HIPCHK(c, sync_all(c));     // timed frames of other sets: their events must have completed
    double ms = 0.0;
    for (size_t i = 0; i + 1 < c->ev_used; i += 2) {
        float t = 0.0f;
        HIPCHK(c, hipEventElapsedTime(&t, c->ev[i], c->ev[i + 1]));
        ms += t;
    }
    c->counters.rays_primary = c->h_cnt[RT_CNT_PRIMARY];
    c->counters.rays_shadow = c->h_cnt[RT_CNT_SHADOW];
    c->counters.rays_bounce = c->h_cnt[RT_CNT_BOUNCE];
    c->counters.node_visits = c->h_cnt[RT_CNT_NODES];
    c->counters.tri_tests = c->h_cnt[RT_CNT_TRIS];
    c->counters.mesh_aabb_tests = c->h_cnt[RT_CNT_MESH];
    c->counters.wave_node_iters = c->h_cnt[RT_CNT_WNODE];
    c->counters.wave_tri_iters = c->h_cnt[RT_CNT_WTRI];
    c->counters.wave_alive_lanes = c->h_cnt[RT_CNT_WALIVE];
    c->counters.kernel_ms = ms;
    *out = c->counters;
    // timing accumulates between two rt_get_counters calls; ray counters are of the last render
    c->ev_used = 0;
    c->counters.kernel_launches = 0;
    return RT_OK;
}

static int read_back(rt_ctx *c, void *dst, const void *src, size_t have, size_t want_bytes, const char *what) {
    if (!c || !dst) return RT_ERR_ARG;
    if (!c->has_view || !src) return RT_NOT_READY;
    if (want_bytes != have) return fail(c, RT_ERR_ARG, what);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(dst, src, have, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_read_radiance(rt_ctx *c, float *rgba, size_t n_floats) {
    return read_back(c, rgba, c ? c->d_rgba : nullptr, c ? c->n_px * 16 : 0, n_floats * 4, "rt_read_radiance: size must be W*H*4 floats");
}
int rt_read_accumulation(rt_ctx *c, float *accum, size_t n_floats) {
    if (c && c->has_view) { HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, wait_path_frames(c)); }
    return read_back(c, accum, c ? c->d_accum : nullptr, c ? c->n_px * 16 : 0, n_floats * 4, "rt_read_accumulation: size must be W*H*4 floats");
}
int rt_read_prim_ids(rt_ctx *c, uint32_t *prim, size_t n) {
    return read_back(c, prim, c ? c->d_prim : nullptr, c ? c->n_px * 4 : 0, n * 4, "rt_read_prim_ids: size must be W*H");
}
int rt_read_hit_t(rt_ctx *c, float *t, size_t n) {
    return read_back(c, t, c ? c->d_t : nullptr, c ? c->n_px * 4 : 0, n * 4, "rt_read_hit_t: size must be W*H");
}
int rt_read_bgra8_srgb(rt_ctx *c, uint8_t *bgra, size_t n_bytes) {
    if (!c || !bgra) return RT_ERR_ARG;
    if (!c->has_view) return RT_NOT_READY;
    if (n_bytes != c->n_px * 4) return fail(c, RT_ERR_ARG, "rt_read_bgra8_srgb: size must be W*H*4 bytes");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->d_bgra) HIPCHK(c, hipMalloc((void **)&c->d_bgra, c->n_px * 4));
    HIPCHK(c, rtk_launch_encode_bgra8(c->stream, c->d_rgba, c->d_bgra, c->n_px));
    return read_back(c, bgra, c->d_bgra, c->n_px * 4, n_bytes, "");
}

int rt_encode_bgra8_srgb(rt_ctx *c, const float *rgba, size_t n_pixels, uint8_t *bgra) {
    if (!c || (n_pixels && (!rgba || !bgra))) return RT_ERR_ARG;
    if (n_pixels == 0) return RT_OK;
    HIPCHK(c, hipSetDevice(c->device));
    void *d_in = nullptr, *d_out = nullptr;
    hipError_t e = hipMalloc(&d_in, n_pixels * 16);
    if (e == hipSuccess) e = hipMalloc(&d_out, n_pixels * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, rgba, n_pixels * 16, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = rtk_launch_encode_bgra8(c->stream, (const float *)d_in, (uint8_t *)d_out, n_pixels);
    if (e == hipSuccess) e = hipMemcpyAsync(bgra, d_out, n_pixels * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    free_dev(d_in); free_dev(d_out);
    if (e != hipSuccess) return fail(c, e == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP, "rt_encode_bgra8_srgb", e);
    return RT_OK;
}

int rt_present_async(rt_ctx *c, uint8_t *host_bgra, size_t n_bytes, uint32_t *ticket) {
    if (!c || !host_bgra || !ticket) return RT_ERR_ARG;
    if (!c->has_view || !c->d_rgba) return RT_NOT_READY;
    if (n_bytes != c->n_px * 4) return fail(c, RT_ERR_ARG, "rt_present_async: size must be W*H*4 bytes");
    HIPCHK(c, hipSetDevice(c->device));
    FrameSet &F = c->fs[c->cur_set];
    if (!F.d_bgra) HIPCHK(c, hipMalloc((void **)&F.d_bgra, c->n_px * 4));
    if (!F.ev_present) HIPCHK(c, hipEventCreateWithFlags(&F.ev_present, hipEventDisableTiming));
    HIPCHK(c, rtk_launch_encode_bgra8(c->stream, c->d_rgba, F.d_bgra, c->n_px));
    HIPCHK(c, hipMemcpyAsync(host_bgra, F.d_bgra, c->n_px * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipEventRecord(F.ev_present, c->stream));
    F.present_pending = true;
    *ticket = c->cur_set;
    return RT_OK;
}

int rt_present_wait(rt_ctx *c, uint32_t ticket) {
    if (!c) return RT_ERR_ARG;
    if (ticket >= RT_MAX_IN_FLIGHT || !c->fs[ticket].present_pending) return fail(c, RT_ERR_ARG, "rt_present_wait: no presentation pending under this ticket");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventSynchronize(c->fs[ticket].ev_present));
    c->fs[ticket].present_pending = false;
    return RT_OK;
}

int rt_device_buffers(rt_ctx *c, void **d_radiance, void **d_prim, void **d_accum) {
    if (!c) return RT_ERR_ARG;
    if (!c->has_view) return RT_NOT_READY;
    if (d_radiance) *d_radiance = c->d_rgba;
    if (d_prim) *d_prim = c->d_prim;
    if (d_accum) *d_accum = c->d_accum;
    return RT_OK;
}

int rt_resolve_accumulation(rt_ctx *c, uint32_t total_samples) {
    if (!c || total_samples == 0) return RT_ERR_ARG;
    if (!c->has_view) return RT_NOT_READY;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, wait_path_frames(c));
    HIPCHK(c, rtk_launch_resolve(c->stream, c->d_accum, c->d_rgba, c->n_px, 0.0f, total_samples));
    return RT_OK;
}

int rt_get_bvh_info(rt_ctx *c, rt_bvh_info *out) {
    if (!c || !out) return RT_ERR_ARG;
    if (!c->has_scene) return RT_NOT_READY;
    *out = c->info;
    return RT_OK;
}

int rt_read_bvh(rt_ctx *c, void *nodes, size_t node_bytes_total, void *tris, size_t tri_bytes_total) {
    if (!c) return RT_ERR_ARG;
    if (!c->has_scene) return RT_NOT_READY;
    if (node_bytes_total != c->info.n_nodes * sizeof(RtBvhNode) || tri_bytes_total != c->info.n_triangles * sizeof(RtTriRecord))
        return fail(c, RT_ERR_ARG, "rt_read_bvh: buffer sizes must match rt_get_bvh_info");
    HIPCHK(c, hipSetDevice(c->device));
    if (nodes && node_bytes_total) HIPCHK(c, hipMemcpyAsync(nodes, c->d_nodes, node_bytes_total, hipMemcpyDeviceToHost, c->stream));
    if (tris && tri_bytes_total) HIPCHK(c, hipMemcpyAsync(tris, c->d_recs, tri_bytes_total, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_read_bvh_wide(rt_ctx *c, void *wide_nodes, size_t bytes_total) {
    if (!c) return RT_ERR_ARG;
    if (!c->has_scene) return RT_NOT_READY;
    if (bytes_total != c->wide_host.nodes.size() * sizeof(RtWideNode))
        return fail(c, RT_ERR_ARG, "rt_read_bvh_wide: buffer size must be n_wide_nodes * 64");
    HIPCHK(c, hipSetDevice(c->device));
    if (wide_nodes && bytes_total) HIPCHK(c, hipMemcpyAsync(wide_nodes, c->d_wide, bytes_total, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

}  // extern "C"
