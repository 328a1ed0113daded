"""GPU parity cases added in round 2 (VERDICT r1 "close the GPU test gaps"): the shade branches of
wgsl:205-207 on the device, NaN rays, refused views, the multi-GPU tile pack / unpack kernels, the analytic field and
a 16-sample batch at 1080p, and configs[3] sharded four ways at 4K."""
import numpy as np
import pytest

import oracle_py as orc
import ray_tracing_amd as rt
from ray_tracing_amd import multi_gpu
from helpers import bitwise_equal, default_app, gpu_render, oracle_scene, oracle_view, soup_app

pytestmark = pytest.mark.gpu

VARIANTS = [0, rt.FLAG_BINARY_TREE, rt.FLAG_STATIC_GRID, rt.FLAG_WAVE_TRAVERSAL, rt.FLAG_BRUTE_FORCE]


@pytest.mark.parametrize("extra", VARIANTS)
def test_debug_grey_and_out_of_range_materials_on_the_device(extra):
    """A6 (wgsl:205-207): material >= 100 is the debug grey 0.5 + (m - 100) / 6 in all four channels; a material index
    past the materials array (the P1 desync, src/ray_tracing.rs:122-129) reads as a zero colour.  Every kernel variant,
    with the shadow ray (which scales the grey too)."""
    app = default_app(96, 54)
    scene = app.prepare_meshinfo()
    m = scene.meshes.copy()
    m["material"][0] = 103                       # the cube: grey 1.0
    m["material"][1] = 100                       # the plane: grey 0.5
    m["material"][2] = 7                         # a sphere: >= n_mat (6), < 100 -> zero colour
    m["material"][3] = 99                        # another: still < 100, out of range
    m["material"][4] = 0xFFFFFFFF                # >= 100: grey from a huge index
    s2 = rt.Scene(scene.triangles, m, scene.vertices, scene.materials)
    rgba_o, prim_o, t_o, _ = orc.render_primary(oracle_scene(s2), oracle_view(app), orc.FLAG_SHADOW)
    ctx, rgba_g, prim_g, t_g = gpu_render(app, s2, rt.FLAG_SHADOW | extra)
    assert bitwise_equal(prim_g, prim_o) and bitwise_equal(t_g, t_o) and bitwise_equal(rgba_g, rgba_o)
    greys = np.unique(rgba_g[(rgba_g[..., 0] == rgba_g[..., 3]) & (rgba_g[..., 0] > 0)])
    assert 1.0 in greys and 0.5 in greys         # both debug greys are on screen, unshadowed somewhere
    ctx.close()


@pytest.mark.parametrize("extra", [0, rt.FLAG_BINARY_TREE, rt.FLAG_STATIC_GRID])
def test_nan_rays_end_at_once_and_match_the_oracle(extra):
    """zero-length vertex normals make the shadow-ray origin NaN for every pixel that hits them.  The wide step
    drops NaN through min/max and would walk the whole tree for such a ray; it is ended before it starts, as a miss --
    which is what the reference's loop makes of it.  Counted visits must equal the oracle walk's (no visit for a NaN ray)."""
    app = soup_app(40, 0x5EED0003, 320, 180)
    app.camera_frame_scene(1.0)
    scene = app.prepare_meshinfo()
    v = scene.vertices.copy()
    v["norm"][:] = 0                              # every hit pixel gets a NaN normal -> NaN shadow origin
    s2 = rt.Scene(scene.triangles, scene.meshes, v, scene.materials)
    ctx, rgba_g, prim_g, t_g = gpu_render(app, s2, rt.FLAG_SHADOW | rt.FLAG_COUNTERS | extra)
    c = ctx.counters()
    rgba_o, prim_o, t_o, _ = orc.render_primary(oracle_scene(s2), oracle_view(app), orc.FLAG_SHADOW)
    assert bitwise_equal(prim_g, prim_o) and bitwise_equal(t_g, t_o)
    assert np.array_equal(np.isnan(rgba_g), np.isnan(rgba_o)) and np.array_equal(np.nan_to_num(rgba_g), np.nan_to_num(rgba_o))
    hit = prim_g != rt.MISS_PRIM
    assert hit.any() and np.isnan(rgba_g[hit]).all()
    wide = None if extra & rt.FLAG_BINARY_TREE else ctx.read_bvh_wide()
    tree = orc.Tree(*ctx.read_bvh(), wide)
    _, prim_w, _, rays_w, cnt_w = orc.render_primary_tree(oracle_scene(s2), tree, oracle_view(app), orc.FLAG_SHADOW)
    assert bitwise_equal(prim_g, prim_w) and c.rays_primary + c.rays_shadow == rays_w
    assert (c.node_visits, c.tri_tests, c.mesh_aabb_tests) == cnt_w
    # the shadow rays cost nothing: the same frame without them visits exactly as many nodes
    gpu_render(app, s2, rt.FLAG_COUNTERS | extra, ctx=ctx)
    assert ctx.counters().node_visits == c.node_visits
    ctx.close()


def test_non_finite_and_degenerate_views():
    app = default_app(64, 36)
    scene = app.prepare_meshinfo()
    m, p = app.camera_view()
    with rt.Context(0) as ctx:
        ctx.upload_scene(scene)
        for bad in (np.nan, np.inf, -np.inf):
            mb = np.array(m, np.float32); mb[5] = bad
            with pytest.raises(rt.RtError) as e:
                ctx.set_view(mb, p, 64, 36)
            assert e.value.status == rt.RT_ERR_ARG
            with pytest.raises(rt.RtError):
                ctx.set_view(m, (0.0, bad, 0.0), 64, 36)
        assert ctx.render() == rt.RT_NOT_READY            # no view was accepted
        # an all-zero matrix is finite but gives 0/0 directions: every ray is NaN, ends at once, and the frame is a miss
        ctx.set_view(np.zeros(16, np.float32), p, 64, 36)
        for f in (rt.FLAG_SHADOW | rt.FLAG_COUNTERS, rt.FLAG_SHADOW | rt.FLAG_STATIC_GRID | rt.FLAG_COUNTERS, rt.FLAG_PATH_TRACE | rt.FLAG_COUNTERS):
            assert ctx.render(flags=f, spp=1) == rt.RT_OK
            assert np.all(ctx.read_prim_ids() == rt.MISS_PRIM)
            assert ctx.counters().node_visits == 0
        rgba_o, prim_o, _, _ = orc.render_primary(oracle_scene(scene), orc.make_view(np.zeros(16, np.float32), p, 64, 36), orc.FLAG_SHADOW)
        assert np.all(prim_o == rt.MISS_PRIM)


def test_stream_selection_null_stream_sentinel():
    """NULL restores the context's own (non-blocking) stream; the null stream has to be asked for by name."""
    import ctypes as C
    app = default_app(64, 36)
    scene = app.prepare_meshinfo()
    ctx, rgba, prim, _ = gpu_render(app, scene, rt.FLAG_SHADOW)
    ctx.set_stream(C.c_void_p(-1).value)          # RT_STREAM_NULL
    assert ctx.render(flags=rt.FLAG_SHADOW) == rt.RT_OK
    assert bitwise_equal(ctx.read_radiance(), rgba)
    ctx.set_stream(0)                             # back on the context's own stream
    assert ctx.render(flags=rt.FLAG_SHADOW) == rt.RT_OK
    assert bitwise_equal(ctx.read_prim_ids(), prim)
    ctx.close()


@pytest.mark.parametrize("size,vp,count,tile", [((200, 120), None, 3, 16), ((1920, 1080), None, 8, 64), ((330, 200), (9, 14, 300, 170), 4, 32),
                                                ((64, 36), None, 5, 64)])
def test_tile_pack_and_unpack_kernels(size, vp, count, tile):
    """every rank renders its tiles, packs them (k_tile_pack), and rank 0 scatters the peers' packed tiles
    into its own frame (k_tile_unpack, all peers in one launch).  The packed bytes are the layout multi_gpu.pack_tiles_host
    defines (what the gloo rehearsal moves), and the combined frame is the unsharded frame, bit for bit."""
    import torch
    W, H = size
    app = default_app(W, H, viewport=vp)
    scene = app.prepare_meshinfo()
    ctx, full, _, _ = gpu_render(app, scene, rt.FLAG_SHADOW, viewport=vp)
    slot = ctx.tile_slot_bytes(count, tile)
    rect = vp if vp is not None else (0, 0, W, H)
    assert slot == multi_gpu.tile_slot_floats(rect[2], rect[3], tile, count) * 4
    recv = torch.zeros(count * slot // 4, dtype=torch.float32, device="cuda")
    for rank in range(count):
        ctx.render(flags=rt.FLAG_SHADOW, shard_rank=rank, shard_count=count, tile_size=tile)
        ctx.pack_tiles(rank, count, tile, recv.data_ptr() + rank * slot, slot)
        ctx.synchronize()
        part = ctx.read_radiance()
        want = multi_gpu.pack_tiles_host(part, rect, tile, count, rank)
        got = recv[rank * slot // 4: rank * slot // 4 + want.size].cpu().numpy()
        assert bitwise_equal(got, want)
    # rank 0's own frame + everyone else's packed tiles
    ctx.render(flags=rt.FLAG_SHADOW, shard_rank=0, shard_count=count, tile_size=tile)
    if count > 1:
        ctx.unpack_tiles(1, count - 1, count, tile, recv.data_ptr() + slot, slot)
    combined = ctx.read_radiance()
    assert bitwise_equal(combined, full)
    # the host restatement of the scatter (the gloo path) gives the same frame
    host = multi_gpu.unpack_tiles_host(np.zeros_like(full), recv.cpu().numpy().reshape(count, -1), rect, tile, count, range(count))
    assert np.array_equal(host, full)
    with pytest.raises(rt.RtError):
        ctx.pack_tiles(0, count, tile, recv.data_ptr(), 16)             # buffer too small
    with pytest.raises(rt.RtError):
        ctx.unpack_tiles(1, count, count, tile, recv.data_ptr(), slot)   # ranks beyond the shard count
    ctx.close()


def test_analytic_sphere_field_10k_1080p_whole_frame():
    """configs[1] variant (ii) at full size (the bench checks it too; here as a test): 10,000 analytic spheres + ground,
    1920x1080, primary + shadow, the whole frame and the counted visits against the oracle's walk of the device's tree."""
    app = rt.App()
    app.spawn_sphere_field(10000, 0x5EED0002, 0)
    app.camera_frame_scene(1.0)
    app.camera_set_target(1920, 1080)
    scene = rt.analytic_sphere_field(10000, 0x5EED0002)
    ctx, rgba, prim, t = gpu_render(app, scene, rt.FLAG_SHADOW | rt.FLAG_COUNTERS)
    c = ctx.counters()
    osc = orc.Scene(scene.triangles, scene.meshes, scene.vertices, scene.materials, scene.spheres, scene.sphere_materials)
    tree = orc.Tree(*ctx.read_bvh(), ctx.read_bvh_wide())
    rgba_w, prim_w, t_w, rays_w, cnt_w = orc.render_primary_tree(osc, tree, oracle_view(app), orc.FLAG_SHADOW, threads=16)
    assert bitwise_equal(prim, prim_w) and bitwise_equal(t, t_w) and bitwise_equal(rgba, rgba_w)
    assert (c.node_visits, c.tri_tests, c.mesh_aabb_tests) == cnt_w and c.rays_primary + c.rays_shadow == rays_w
    assert (prim >= len(scene.triangles)).mean() > 0.2 and (prim != rt.MISS_PRIM).mean() > 0.5       # spheres are what is seen
    pix = np.random.default_rng(3).choice(1920 * 1080, 3000, replace=False).astype(np.uint32)
    rgba_o, prim_o, _, _ = orc.render_primary(osc, oracle_view(app), orc.FLAG_SHADOW, threads=16, pixel_list=pix)
    assert bitwise_equal(prim.ravel()[pix], prim_o) and bitwise_equal(rgba.reshape(-1, 4)[pix], rgba_o)
    ctx.close()


def test_sixteen_sample_batch_at_1080p_and_config3_four_way_shard_at_4k():
    """(1) 21 = 16 + 4 + 1 samples per pixel at 1920x1080 on the 1 M-triangle soup: the 16-sample batch (magnified 4x:
    7680x4320 threads, 33 M queued paths) on a real frame, running sums bit-identical to the oracle's sequential ones.
    (2) configs[3]: 3840x2160, <= 4 segments, 16 of its 64 samples, rendered as four 64x64-tile shards; the four shards'
    running sums add up to the unsharded frame's, which is the oracle's, bit for bit."""
    app = soup_app(1389, 0x5EED0003, 1920, 1080)
    app.camera_frame_scene(1.0)
    scene = app.prepare_meshinfo()
    osc = oracle_scene(scene)
    with rt.Context(0) as ctx:
        ctx.upload_scene(scene)
        m, p = app.camera_view()
        ctx.set_view(m, p, 1920, 1080)
        tree = orc.Tree(*ctx.read_bvh(), ctx.read_bvh_wide())
        assert ctx.render(flags=rt.FLAG_PATH_TRACE | rt.FLAG_TIME_KERNELS, spp=21, max_segments=4, seed=5) == rt.RT_OK
        c = ctx.counters()
        assert c.kernel_launches == 3 * 4                                         # three batches (16, 4, 1) of 1 + 3 launches
        acc = ctx.read_accumulation()
        acc_w, segs_w, _ = orc.render_path_tree(osc, tree, oracle_view(app), spp=21, max_segments=4, seed=5, threads=16)
        assert bitwise_equal(acc, acc_w) and c.rays_primary + c.rays_bounce == segs_w
        del acc, acc_w
        app.camera_set_target(3840, 2160)
        m, p = app.camera_view()
        ctx.set_view(m, p, 3840, 2160)
        assert ctx.render(flags=rt.FLAG_PATH_TRACE, spp=16, max_segments=4, seed=9) == rt.RT_OK
        full = ctx.read_accumulation()
        acc_w, _, _ = orc.render_path_tree(osc, tree, oracle_view(app), spp=16, max_segments=4, seed=9, threads=16)
        assert bitwise_equal(full, acc_w)
        del acc_w
        total = np.zeros_like(full)
        owner = multi_gpu.tile_owner(3840, 2160, 64, 4)
        for rank in range(4):
            assert ctx.render(flags=rt.FLAG_PATH_TRACE, spp=16, max_segments=4, seed=9, shard_rank=rank, shard_count=4, tile_size=64) == rt.RT_OK
            part = ctx.read_accumulation()
            assert np.all(part[owner != rank] == 0)
            total += part
        assert np.array_equal(total, full)


@pytest.mark.parametrize("n", [2, 3, 4])      # 4: bench.py's default (k_stream8 from n = 3 on)
def test_frames_in_flight_inside_one_context(n):
    """rt_set_frames_in_flight: rt_render takes n sets of frame buffers / streams in turn without waiting for the frame
    before.  Every frame must still be the serial frame, bit for bit, readers must see the frame rendered last, a scene
    update must wait for the frames in flight, and path-traced frames (one shared accumulation buffer) must not overlap."""
    app = soup_app(200, 0x5EED0003, 640, 360)
    app.camera_frame_scene(1.0)
    scene = app.prepare_meshinfo()
    cams = []
    for k in range(7):
        app.camera_look_at((30.0 * np.cos(k), 25.0 + 3 * k, 30.0 * np.sin(k)), (0, 0, 0), (0, 1, 0))
        cams.append(app.camera_view())
    with rt.Context(0) as ctx:
        ctx.upload_scene(scene)
        serial = []
        for m, p in cams:
            ctx.set_view(m, p, 640, 360)
            ctx.render(flags=rt.FLAG_SHADOW | rt.FLAG_COUNTERS)
            c = ctx.counters()
            serial.append((ctx.read_radiance(), ctx.read_prim_ids(), ctx.read_hit_t(), (c.rays_primary, c.rays_shadow, c.node_visits, c.tri_tests)))
        ctx.set_frames_in_flight(n)
        for rep in range(2):
            for k, (m, p) in enumerate(cams):                          # frames queued back to back, read only at the end of a burst
                ctx.set_view(m, p, 640, 360)
                ctx.render(flags=rt.FLAG_SHADOW | rt.FLAG_COUNTERS)
                if k % 3 == 2 or k == len(cams) - 1:
                    c = ctx.counters()
                    assert bitwise_equal(ctx.read_radiance(), serial[k][0]) and bitwise_equal(ctx.read_prim_ids(), serial[k][1])
                    assert bitwise_equal(ctx.read_hit_t(), serial[k][2])
                    assert (c.rays_primary, c.rays_shadow, c.node_visits, c.tri_tests) == serial[k][3]
        # a tile shard on every set: pixels of other ranks' tiles read as cleared whichever set the frame lands on
        m, p = cams[0]
        ctx.set_view(m, p, 640, 360)
        own = multi_gpu.tile_owner(640, 360, 64, 3) == 1
        for _ in range(n + 1):
            ctx.render(flags=rt.FLAG_SHADOW, shard_rank=1, shard_count=3, tile_size=64)
            part = ctx.read_radiance()
            assert bitwise_equal(part[own], serial[0][0][own]) and np.all(part[~own] == 0)
        # path tracing between overlapped frames: running sums as in a serial context
        ctx.render(flags=rt.FLAG_SHADOW)
        ctx.render(flags=rt.FLAG_PATH_TRACE, spp=3, max_segments=4, seed=4)
        ctx.render(flags=rt.FLAG_SHADOW)
        ctx.render(flags=rt.FLAG_PATH_TRACE | rt.FLAG_ACCUMULATE, spp=2, sample_first=3, max_segments=4, seed=4)
        ctx.render(flags=rt.FLAG_SHADOW)           # the frame rendered last is on another set: the sums' reader still waits for set 0
        acc = ctx.read_accumulation()
        acc_o, _ = orc.render_path(oracle_scene(scene), orc.make_view(m, p, 640, 360), 5, 4, seed=4, threads=16)
        assert bitwise_equal(acc, acc_o)
        # path-traced frames back to back: they alternate between two sets, a frame's tracing overlaps with the frame before
        # it and its additions to the one accumulation buffer wait for that frame's -- five one-sample frames give the sums
        # of five samples, and a restart in between (no RT_FLAG_ACCUMULATE) clears behind the frame before it
        ctx.render(flags=rt.FLAG_PATH_TRACE, spp=2, max_segments=4, seed=9)
        for k in range(5):
            ctx.render(flags=rt.FLAG_PATH_TRACE | (rt.FLAG_ACCUMULATE if k else 0), spp=1, sample_first=k, max_segments=4, seed=4)
        assert bitwise_equal(ctx.read_accumulation(), acc_o)
        assert bitwise_equal(ctx.read_radiance(), orc.resolve(acc_o, 5))
        # timed frames on every set, counters asked for without a synchronize in between: all event pairs are complete
        for _ in range(n + 1):
            ctx.render(flags=rt.FLAG_SHADOW | rt.FLAG_TIME_KERNELS)
        c = ctx.counters()
        assert c.kernel_launches == n + 1 and c.kernel_ms > 0.0
        # a vertex update while frames are in flight: the next frame sees the moved scene, the one before it did not
        v2 = scene.vertices.copy()
        v2["pos"][:, 1] += np.float32(0.75)
        moved = rt.Scene(scene.triangles, scene.meshes, v2, scene.materials)
        mm = moved.meshes.copy(); mm["aabb_min"][:, 1] += np.float32(0.75); mm["aabb_max"][:, 1] += np.float32(0.75)
        moved = rt.Scene(scene.triangles, mm, v2, scene.materials)
        ctx.render(flags=rt.FLAG_SHADOW)
        ctx.render(flags=rt.FLAG_SHADOW)
        ctx.update_vertices(moved)
        ctx.render(flags=rt.FLAG_SHADOW)
        after = ctx.read_prim_ids()
        ctx.set_frames_in_flight(1)
        ctx.render(flags=rt.FLAG_SHADOW)
        assert bitwise_equal(ctx.read_prim_ids(), after) and not bitwise_equal(after, serial[0][1])
        with pytest.raises(rt.RtError):
            ctx.set_frames_in_flight(5)


def test_get_stream_and_tile_exchange_with_frames_in_flight():
    """rt_get_stream names the stream of the frame rendered last, so that a consumer (the packed-tile exchange of
    multi_gpu.TileGather.combine_in_flight) can order its work behind THAT frame while the next frame renders on the next
    set's stream.  A second context plays the peer rank: it renders and packs its tiles, rank 0 unpacks them behind an
    event, nothing synchronises with the host inside a burst, and the frame read at the end of each burst is the
    unsharded frame, bit for bit."""
    import ctypes as C
    import torch
    app = soup_app(200, 0x5EED0003, 640, 360)
    app.camera_frame_scene(1.0)
    scene = app.prepare_meshinfo()
    cams = []
    for k in range(6):
        app.camera_look_at((28.0 * np.cos(k), 20.0 + 2 * k, 28.0 * np.sin(k)), (0, 0, 0), (0, 1, 0))
        cams.append(app.camera_view())
    with rt.Context(0) as ctx, rt.Context(0) as peer:
        for c in (ctx, peer):
            c.upload_scene(scene)
        full = []
        for m, p in cams:
            ctx.set_view(m, p, 640, 360)
            ctx.render(flags=rt.FLAG_SHADOW)
            full.append(ctx.read_radiance())
        own = ctx.get_stream()
        assert own != 0
        ctx.set_frames_in_flight(2)
        seen = []
        for _ in range(4):
            ctx.render(flags=rt.FLAG_SHADOW)
            seen.append(ctx.get_stream())
        assert seen[0] != seen[1] and seen[0] == seen[2] and seen[1] == seen[3] and 0 not in seen
        assert own not in seen                                   # with frames in flight every set has a stream with a queue of its own
        side = torch.cuda.Stream(device=0)
        ctx.set_stream(side.cuda_stream)
        ctx.render(flags=rt.FLAG_SHADOW)
        assert ctx.get_stream() == side.cuda_stream               # a caller's stream: every set renders on it
        ctx.set_stream(C.c_void_p(-1).value)
        ctx.render(flags=rt.FLAG_SHADOW)
        assert ctx.get_stream() == 0                              # RT_STREAM_NULL: HIP's null stream
        ctx.set_stream(0)
        slot = ctx.tile_slot_bytes(2, 64)
        msgs = [torch.zeros(slot // 4, dtype=torch.float32, device="cuda") for _ in range(2)]
        peer_stream = torch.cuda.ExternalStream(peer.get_stream())
        k = 0
        for burst in (1, 2, 3):
            for _ in range(burst):
                m, p = cams[k % len(cams)]
                msg = msgs[k & 1]
                peer.set_view(m, p, 640, 360)
                peer.render(flags=rt.FLAG_SHADOW, shard_rank=1, shard_count=2, tile_size=64)
                peer.pack_tiles(1, 2, 64, msg.data_ptr(), slot)
                packed = torch.cuda.Event()
                packed.record(peer_stream)
                ctx.set_view(m, p, 640, 360)
                ctx.render(flags=rt.FLAG_SHADOW, shard_rank=0, shard_count=2, tile_size=64)
                mine = torch.cuda.ExternalStream(ctx.get_stream())
                mine.wait_event(packed)                           # what the collective does between pack and unpack
                ctx.unpack_tiles(1, 1, 2, 64, msg.data_ptr(), slot)
                taken = torch.cuda.Event()
                taken.record(mine)
                peer_stream.wait_event(taken)                     # the message buffer is free again two frames later
                k += 1
            assert bitwise_equal(ctx.read_radiance(), full[(k - 1) % len(cams)])
        ctx.set_frames_in_flight(1)
        assert ctx.get_stream() == own                            # one frame at a time: the context's ordinary stream again


@pytest.mark.parametrize("n", [1, 3, 4])
def test_present_async_delivers_the_bytes_of_each_frame_in_flight(n):
    """rt_present_async / rt_present_wait: the Bgra8UnormSrgb bytes of the frame rendered last, encoded and copied on that
    frame's stream while the next frames render.  Every ticket's bytes must be those rt_read_bgra8_srgb gives for the same
    frame (and so the oracle's, byte for byte), into pinned and into pageable host memory, with up to n tickets open."""
    import torch
    app = default_app(320, 180)
    scene = app.prepare_meshinfo()
    cams = []
    for k in range(5):
        app.camera_look_at((9.0 * np.cos(k), 4.0 + k, 9.0 * np.sin(k)), (0, 1, 0), (0, 1, 0))
        cams.append(app.camera_view())
    with rt.Context(0) as ctx:
        ctx.upload_scene(scene)
        want = []
        for m, p in cams:
            ctx.set_view(m, p, 320, 180)
            ctx.render(flags=rt.FLAG_SHADOW)
            want.append(ctx.read_bgra8_srgb())
            assert np.array_equal(want[-1], orc.encode_bgra8_srgb(ctx.read_radiance()))
            assert want[-1][..., :3].max() > 0
        ctx.set_frames_in_flight(n)
        pinned = [torch.empty(180 * 320 * 4, dtype=torch.uint8).pin_memory().numpy().reshape(180, 320, 4) for _ in range(n)]
        with pytest.raises(rt.RtError):
            ctx.present_wait(0)                                  # nothing pending yet
        with pytest.raises(rt.RtError):
            ctx.present_async(np.empty(16, dtype=np.uint8))      # wrong size
        open_tickets = []                                         # (ticket, buffer, frame index)
        for k in range(3 * len(cams)):
            m, p = cams[k % len(cams)]
            ctx.set_view(m, p, 320, 180)
            ctx.render(flags=rt.FLAG_SHADOW)
            if len(open_tickets) == n:                            # this frame's set is about to be presented again: collect it first
                t, buf, j = open_tickets.pop(0)
                ctx.present_wait(t)
                assert np.array_equal(buf, want[j]), (k, t)
            buf = pinned[k % n] if k % 2 == 0 else np.empty((180, 320, 4), dtype=np.uint8)
            buf[...] = 0x5A
            open_tickets.append((ctx.present_async(buf), buf, k % len(cams)))
        for t, buf, j in open_tickets:
            ctx.present_wait(t)
            assert np.array_equal(buf, want[j])
        assert len({t for t, _, _ in open_tickets}) == n


def test_config4_sample_slices_sum_to_the_unsliced_frame():
    """configs[4] (sample slicing): the 10 M-triangle scene, <= 8 segments, 8 samples per pixel rendered as EIGHT
    slices of one sample each (sample_first = r, one RNG stream: what rank r of 8 renders), their running sums added in rank
    order -- SampleReduce's order -- against ONE context doing all 8 samples, and that against the oracle's walk.  One sample per
    slice keeps the additions in the same order, so here the sum is bit-identical, not merely inside the RMSE gate; slices of
    two samples (4 ranks) change the order: RMSE gate."""
    app = soup_app(13889, 0x5EED0005, 960, 540)
    app.camera_frame_scene(1.0)
    scene = app.prepare_meshinfo()
    assert len(scene.triangles) == 10000080
    with rt.Context(0) as ctx:
        ctx.upload_scene(scene)
        m, p = app.camera_view()
        ctx.set_view(m, p, 960, 540)
        assert ctx.render(flags=rt.FLAG_PATH_TRACE, spp=8, max_segments=8, seed=6) == rt.RT_OK
        full = ctx.read_accumulation()
        tree = orc.Tree(*ctx.read_bvh(), ctx.read_bvh_wide())
        acc_w, _, _ = orc.render_path_tree(oracle_scene(scene), tree, oracle_view(app), spp=8, max_segments=8, seed=6, threads=16)
        assert bitwise_equal(full, acc_w)
        total = np.zeros_like(full)
        for r in range(8):
            assert ctx.render(flags=rt.FLAG_PATH_TRACE, spp=1, sample_first=r, max_segments=8, seed=6, stream_id=0) == rt.RT_OK
            total = total + ctx.read_accumulation()                       # rank order, as SampleReduce adds the slices
        assert bitwise_equal(total, full)
        total2 = np.zeros_like(full)
        for r in range(4):
            assert ctx.render(flags=rt.FLAG_PATH_TRACE, spp=2, sample_first=2 * r, max_segments=8, seed=6, stream_id=0) == rt.RT_OK
            total2 = total2 + ctx.read_accumulation()
        d = (total2.astype(np.float64) - full.astype(np.float64)) / 8.0
        assert float(np.sqrt(np.mean(d * d))) <= 1e-3 and np.all(total2[..., 3] == 8.0)


def _oracle_rects(osc, tree, app, W, H, rects, **kw):
    """The oracle's running sums on a few rectangles of the W x H frame (orc_render_path_tree_crop: a pixel's value does
    not depend on the crop)."""
    m, p = app.camera_view()
    view = orc.make_view(m, p, W, H)
    return [orc.render_path_tree_crop(osc, tree, view, r, threads=16, **kw) for r in rects]


def test_config3_full_size_64spp_eight_way_tile_shard():
    """configs[3] at its own numbers: 3840x2160, 64 samples per pixel, <= 4 segments, 64x64 tiles dealt to EIGHT ranks.  One
    GPU plays the eight ranks in turn: each renders its tiles with all 64 samples and packs its radiance tiles
    (rt_pack_tiles), rank 0 scatters the seven peers' messages into its frame (rt_unpack_tiles) -- the data path of
    bench.py --gpus 8 minus the wire.  The combined frame is the unsharded 64-sample frame bit for bit, and that frame's
    running sums are the oracle's on three viewports (the whole frame at 64 samples is minutes of host time)."""
    import torch
    W, H, SPP, SEGS, RANKS = 3840, 2160, 64, 4, 8
    app = soup_app(1389, 0x5EED0003, W, H)
    app.camera_frame_scene(1.0)
    scene = app.prepare_meshinfo()
    with rt.Context(0) as ctx:
        ctx.upload_scene(scene)
        m, p = app.camera_view()
        ctx.set_view(m, p, W, H)
        assert ctx.render(flags=rt.FLAG_PATH_TRACE, spp=SPP, max_segments=SEGS, seed=11) == rt.RT_OK
        full_acc = ctx.read_accumulation()
        full = ctx.read_radiance()
        assert np.all(full_acc[..., 3] == float(SPP))
        tree = orc.Tree(*ctx.read_bvh(), ctx.read_bvh_wide())
        rects = [(0, 0, 96, 64), (1900, 1050, 128, 48), (W - 80, H - 56, 80, 56)]
        for (x, y, w, h), want in zip(rects, _oracle_rects(oracle_scene(scene), tree, app, W, H, rects, spp=SPP, max_segments=SEGS, seed=11)):
            assert bitwise_equal(full_acc[y:y + h, x:x + w], want)
        slot = ctx.tile_slot_bytes(RANKS, 64)
        recv = torch.zeros(RANKS * slot // 4, dtype=torch.float32, device="cuda")
        for rank in range(RANKS - 1, -1, -1):              # rank 0 last: its frame is the one the peers' tiles go into
            assert ctx.render(flags=rt.FLAG_PATH_TRACE, spp=SPP, max_segments=SEGS, seed=11, stream_id=0,
                              shard_rank=rank, shard_count=RANKS, tile_size=64) == rt.RT_OK
            if rank:
                ctx.pack_tiles(rank, RANKS, 64, recv.data_ptr() + rank * slot, slot)
        ctx.unpack_tiles(1, RANKS - 1, RANKS, 64, recv.data_ptr() + slot, slot)
        assert bitwise_equal(ctx.read_radiance(), full)


def test_config4_full_size_8k_sample_slices():
    """configs[4] at its own size: 7680x4320, the 10 M-triangle scene, <= 8 segments, sample-sliced over EIGHT ranks (one
    sample each, one RNG stream).  One GPU plays the eight ranks in turn; their running sums, added in rank order as
    SampleReduce adds them, are the 8-sample frame of one context bit for bit, and that frame's sums are the oracle's on
    three viewports."""
    W, H, SEGS, RANKS = 7680, 4320, 8, 8
    app = soup_app(13889, 0x5EED0005, W, H)
    app.camera_frame_scene(1.0)
    scene = app.prepare_meshinfo()
    with rt.Context(0) as ctx:
        ctx.upload_scene(scene)
        m, p = app.camera_view()
        ctx.set_view(m, p, W, H)
        assert ctx.render(flags=rt.FLAG_PATH_TRACE, spp=RANKS, max_segments=SEGS, seed=6) == rt.RT_OK
        full = ctx.read_accumulation()
        tree = orc.Tree(*ctx.read_bvh(), ctx.read_bvh_wide())
        rects = [(0, 0, 64, 48), (3800, 2100, 96, 64), (W - 64, H - 40, 64, 40)]
        for (x, y, w, h), want in zip(rects, _oracle_rects(oracle_scene(scene), tree, app, W, H, rects, spp=RANKS, max_segments=SEGS, seed=6)):
            assert bitwise_equal(full[y:y + h, x:x + w], want)
        total = np.zeros_like(full)
        for r in range(RANKS):
            assert ctx.render(flags=rt.FLAG_PATH_TRACE, spp=1, sample_first=r, max_segments=SEGS, seed=6, stream_id=0) == rt.RT_OK
            total += ctx.read_accumulation()                # rank order
        assert bitwise_equal(total, full)
