// rt_kernels.h -- launchers of the HIP kernels (rt_kernels.hip).  Host-callable, asynchronous
// on the given stream, no allocation, no synchronisation.
#pragma once
#include <hip/hip_runtime_api.h>
#include "rt_layout.h"

#ifndef RT_BLOCK
#define RT_BLOCK 256          // 4 waves; each wave owns one 8x8 pixel tile
#define RT_WORK_TILE 16       // a block covers 16x16 pixels (8 for RT_BLOCK 64)
#endif

// Number of blocks a primary/path-start launch needs for this shard description.
uint32_t rtk_primary_blocks(const RtShardDev &sh);

// Live path of ray_tracing.wgsl:173-211 (+ build-defined shadow ray): ray gen, closest hit,
// shade, outputs.  flags: RT_FLAG_* of rt_amd.h (SHADOW, NO_MESH_CULL, BRUTE_FORCE, COUNTERS).
struct RtTune {
    uint32_t persist_blocks;   // grid of the persistent path kernels (CUs x resident blocks per CU)
    uint32_t cus;              // compute units of the device
    uint32_t blocks_per_cu;    // k_stream: resident blocks per CU; 0 = by BVH size and frames in flight (rtk_launch_primary)
    uint32_t frames_in_flight; // sets rt_render rotates over (rt_set_frames_in_flight), 1 on a caller's stream
    uint32_t refill_min;       // idle lanes in a wave that trigger a refill (64 = whole 8x8 blocks only)
    uint32_t finish_min;       // ended rays in a wave that trigger a hand-over
    uint32_t vote_node, vote_tri;   // relative issue cost of a node step / a triangle step (phase vote)
    uint32_t uni_min;          // lanes that must share one node for the scalar node fetch (65 = never)
    uint32_t wide_vote_node, wide_vote_tri;   // the same vote weights when the 4-wide tree is walked
    uint32_t stream_waves;     // k_stream build: 0 = by frames in flight (8 waves per SIMD with three or more, else 7), 7 or 8 = that one
};
hipError_t rtk_launch_primary(hipStream_t st, const RtSceneDev &S, const RtViewDev &V, const RtShardDev &sh,
                              uint32_t flags, float *rgba, uint32_t *prim, float *t,
                              unsigned long long *counters, const RtTune &tune);

// Progressive path tracing (loop shape of ray_tracing.wgsl:180-202) as a wavefront:
// start = ray gen + jitter + first segment; bounce = one further segment for every queued path.
// A launch covers k x k consecutive samples of every pixel (k = 1 << kshift) on the frame magnified k times; sample slot s
// adds its light to contrib[s * n_px + pixel] (zeroed by the caller), rtk_launch_accumulate_batch then adds the slots to
// the running sums in sample order.  The queues must hold k * k * pixels paths.
hipError_t rtk_launch_path_start(hipStream_t st, const RtSceneDev &S, const RtViewDev &V, const RtShardDev &sh,
                                 uint32_t flags, uint32_t sample, uint32_t seed, uint32_t stream_id,
                                 uint32_t max_segments, float *contrib, uint32_t n_px, uint32_t kshift, RtPathQueue qout,
                                 unsigned long long *counters, int qout_counter, const RtTune &tune);
hipError_t rtk_launch_path_bounce(hipStream_t st, const RtSceneDev &S, uint32_t flags, uint32_t capacity,
                                  int is_last, float *contrib, uint32_t n_px, RtPathQueue qin, RtPathQueue qout,
                                  unsigned long long *counters, int qin_counter, int qout_counter, const RtTune &tune);

// f1: refit one tree level (nodes listed in level_nodes) from the vertices now in `verts`; leaf records re-baked.
hipError_t rtk_launch_refit_level(hipStream_t st, RtBvhNode *nodes, RtTriRecord *recs, const rt_triangle *tris,
                                  const rt_vertex *verts, const uint32_t *level_nodes, uint32_t n);

// The same for a dirty range of meshes [mesh_lo, mesh_hi): a leaf is re-baked only if one of its records belongs to
// such a mesh, an inner child re-read only if it was marked dirty by the level below; dirty[node] (one byte per binary
// node, zeroed by the caller before the deepest level) is what the next level and the re-quantisation look at.
// one launch for a short list of nodes (grouped by level, deepest first) followed by the wide nodes to re-quantise
#define RT_REFIT_LIST_MAX_LEVELS 128
hipError_t rtk_launch_refit_list(hipStream_t st, RtBvhNode *nodes, RtTriRecord *recs, const rt_triangle *tris, const rt_vertex *verts,
                                 const uint32_t *list, const uint32_t *level_end, uint32_t n_levels,
                                 const RtWideSrc *src, RtWideNode *wide, uint32_t n_wide_listed);
hipError_t rtk_launch_refit_level_dirty(hipStream_t st, RtBvhNode *nodes, RtTriRecord *recs, const rt_triangle *tris,
                                        const rt_vertex *verts, const uint32_t *level_nodes, uint32_t n,
                                        uint32_t mesh_lo, uint32_t mesh_hi, uint8_t *dirty);
hipError_t rtk_launch_wide_requant_dirty(hipStream_t st, const RtBvhNode *nodes, const RtWideSrc *src, RtWideNode *wide,
                                         uint32_t n, const uint8_t *dirty);

// the wide nodes named in list[0..n) re-quantised (the medium-sized update of rt_update_vertex_range)
hipError_t rtk_launch_wide_requant_list(hipStream_t st, const RtBvhNode *nodes, const RtWideSrc *src, RtWideNode *wide, const uint32_t *list, uint32_t n);

// after a refit: every wide node's child boxes re-quantised from the binary tree's boxes (rt_wide.h)
hipError_t rtk_launch_wide_requant(hipStream_t st, const RtBvhNode *nodes, const RtWideSrc *src, RtWideNode *wide, uint32_t n);

hipError_t rtk_launch_resolve(hipStream_t st, const float *accum, float *rgba, size_t n_px, float inv_unused,
                              uint32_t total_samples);
hipError_t rtk_launch_accumulate_batch(hipStream_t st, float *accum, const float *contrib, size_t n_px, uint32_t batch);
hipError_t rtk_launch_encode_bgra8(hipStream_t st, const float *rgba, uint8_t *bgra, size_t n_px);
// multi-GPU tile exchange: this rank's owned shard tiles packed contiguously (tile*tile float4 per tile, owned-tile
// order), and the scatter of n_src ranks' packed tiles (slot s = rank first_rank + s, slots slot_stride_px float4 apart)
hipError_t rtk_launch_tile_pack(hipStream_t st, const float *frame, float *packed, const RtViewDev &V, const RtShardDev &sh);
hipError_t rtk_launch_tile_unpack(hipStream_t st, float *frame, const float *packed, size_t slot_stride_px, const RtViewDev &V,
                                  uint32_t tile, uint32_t tiles_x, uint32_t n_tiles, uint32_t count, uint32_t first_rank, uint32_t n_src);

// rt_render_static: the address, on the CURRENT device, of the file-scope scratch array (RT_STATIC_SCRATCH_BYTES)
#ifndef RT_STATIC_SCRATCH_BYTES
#define RT_STATIC_SCRATCH_BYTES (32u << 20)      // include/rt_amd.h has the same value (static_asserted in rt_api.cpp)
#endif
hipError_t rtk_static_scratch_address(void **ptr);

hipError_t rtk_launch_fill_u32(hipStream_t st, uint32_t *p, uint32_t value, size_t n);
