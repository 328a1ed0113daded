"""rt_render_static: the entry point that allocates nothing (include/rt_amd.h).  Its frame's scratch -- work-queue
heads, counters, path queues, contribution plane -- lives in a static device array instead of hipMalloc'ed memory, the
outputs are the caller's device buffers.  Checked through the C ABI against rt_render, bit for bit, on two calls in a
row, the second with a smaller frame; collected last (test_zz_*) so a failure here hides no other test."""
import numpy as np
import pytest

import ray_tracing_amd as rt
from helpers import bitwise_equal, default_app, soup_app

pytestmark = pytest.mark.gpu


def _buffers(torch, n_px):
    out = _alloc(torch, n_px)
    torch.cuda.synchronize()          # the fills run on torch's stream, the library on its own non-blocking one
    return out


def _alloc(torch, n_px):
    return (torch.full((n_px * 4,), 7.0, dtype=torch.float32, device="cuda"),     # stale contents on purpose
            torch.full((n_px,), 12345, dtype=torch.int32, device="cuda"),
            torch.full((n_px,), 3.0, dtype=torch.float32, device="cuda"),
            torch.full((n_px * 4,), 9.0, dtype=torch.float32, device="cuda"))


def _static_frame(torch, ctx, W, H, **kw):
    rad, prim, t, acc = _buffers(torch, W * H)
    assert ctx.render_static(rad.data_ptr(), prim.data_ptr(), t.data_ptr(), acc.data_ptr(), W * H, **kw) == rt.RT_OK
    ctx.synchronize()
    return (rad.cpu().numpy().reshape(H, W, 4), prim.cpu().numpy().view(np.uint32).reshape(H, W), t.cpu().numpy().reshape(H, W),
            acc.cpu().numpy().reshape(H, W, 4))


def test_render_static_matches_rt_render_on_two_calls_the_second_smaller():
    import torch
    app = soup_app(60, 0x5EED0003, 640, 360)
    app.camera_frame_scene(1.0)
    scene = app.prepare_meshinfo()
    with rt.Context(0) as ctx:
        ctx.upload_scene(scene)
        for (W, H) in ((640, 360), (333, 190)):                       # the second call: a smaller, ragged frame
            app.camera_set_target(W, H)
            m, p = app.camera_view()
            ctx.set_view(m, p, W, H)
            # primary + shadow
            assert ctx.render(flags=rt.FLAG_SHADOW | rt.FLAG_COUNTERS) == rt.RT_OK
            want = (ctx.read_radiance(), ctx.read_prim_ids(), ctx.read_hit_t())
            c0 = ctx.counters()
            got = _static_frame(torch, ctx, W, H, flags=rt.FLAG_SHADOW | rt.FLAG_COUNTERS)
            c1 = ctx.counters()
            assert bitwise_equal(got[0], want[0]) and bitwise_equal(got[1], want[1]) and bitwise_equal(got[2], want[2])
            assert (c1.rays_primary, c1.rays_shadow, c1.node_visits, c1.tri_tests) == (c0.rays_primary, c0.rays_shadow, c0.node_visits, c0.tri_tests)
            # path tracing, 5 samples (rt_render: a batch of 4 + 1; static: one by one -- the same sums), then 2 more accumulated
            assert ctx.render(flags=rt.FLAG_PATH_TRACE, spp=5, max_segments=4, seed=11) == rt.RT_OK
            assert ctx.render(flags=rt.FLAG_PATH_TRACE | rt.FLAG_ACCUMULATE, spp=2, sample_first=5, max_segments=4, seed=11) == rt.RT_OK
            want_acc, want_rad = ctx.read_accumulation(), ctx.read_radiance()
            rad, prim, t, acc = _buffers(torch, W * H)
            args = (rad.data_ptr(), prim.data_ptr(), t.data_ptr(), acc.data_ptr(), W * H)
            assert ctx.render_static(*args, flags=rt.FLAG_PATH_TRACE, spp=5, max_segments=4, seed=11) == rt.RT_OK
            assert ctx.render_static(*args, flags=rt.FLAG_PATH_TRACE | rt.FLAG_ACCUMULATE, spp=2, sample_first=5, max_segments=4, seed=11) == rt.RT_OK
            ctx.synchronize()
            assert bitwise_equal(acc.cpu().numpy().reshape(H, W, 4), want_acc)
            assert bitwise_equal(rad.cpu().numpy().reshape(H, W, 4), want_rad)
            # a tile shard: pixels of other ranks' tiles read as cleared although the caller's buffers held stale values
            assert ctx.render(flags=rt.FLAG_SHADOW, shard_rank=1, shard_count=3, tile_size=32) == rt.RT_OK
            want = (ctx.read_radiance(), ctx.read_prim_ids(), ctx.read_hit_t())
            got = _static_frame(torch, ctx, W, H, flags=rt.FLAG_SHADOW, shard_rank=1, shard_count=3, tile_size=32)
            assert bitwise_equal(got[0], want[0]) and bitwise_equal(got[1], want[1]) and bitwise_equal(got[2], want[2])


def test_render_static_rejects_what_does_not_fit_and_bad_arguments():
    import torch
    app = default_app(1920, 1080)
    scene = app.prepare_meshinfo()
    with rt.Context(0) as ctx:
        rad, prim, t, acc = _buffers(torch, 1920 * 1080)
        args = (rad.data_ptr(), prim.data_ptr(), t.data_ptr(), acc.data_ptr())
        assert ctx.render_static(*args, 1920 * 1080) == rt.RT_NOT_READY        # no scene, no view: the silent skip
        ctx.upload_scene(scene)
        m, p = app.camera_view()
        ctx.set_view(m, p, 1920, 1080)
        # primary + shadow needs only the fixed part of the scratch: any frame size
        assert ctx.render(flags=rt.FLAG_SHADOW) == rt.RT_OK
        want = ctx.read_radiance()
        assert ctx.render_static(*args, 1920 * 1080, flags=rt.FLAG_SHADOW) == rt.RT_OK
        ctx.synchronize()
        assert bitwise_equal(rad.cpu().numpy().reshape(1080, 1920, 4), want)
        # path tracing at 1080p needs 216 MB of queues: beyond the 32 MiB array -> a clear error, nothing launched
        with pytest.raises(rt.RtError) as e:
            ctx.render_static(*args, 1920 * 1080, flags=rt.FLAG_PATH_TRACE, spp=1)
        assert e.value.status == rt.RT_ERR_ARG and "at most" in str(e.value)
        with pytest.raises(rt.RtError):
            ctx.render_static(*args, 1920 * 1080 - 1, flags=rt.FLAG_SHADOW)    # n_pixels != width * height
        with pytest.raises(rt.RtError):
            ctx.render_static(0, prim.data_ptr(), t.data_ptr(), acc.data_ptr(), 1920 * 1080)
        with pytest.raises(rt.RtError):
            ctx.render_static(rad.data_ptr(), prim.data_ptr(), t.data_ptr(), 0, 1920 * 1080, flags=rt.FLAG_PATH_TRACE, spp=1)
        with pytest.raises(rt.RtError):
            ctx.render_static(*args, 1920 * 1080, flags=rt.FLAG_SHADOW | rt.FLAG_TIME_KERNELS)
        # the context's own path still works afterwards
        assert ctx.render(flags=rt.FLAG_SHADOW) == rt.RT_OK and bitwise_equal(ctx.read_radiance(), want)


def test_render_static_captured_into_a_graph_replays_the_frame():
    """rt_render_static recorded into a HIP graph and replayed.  What the header promises (include/rt_amd.h,
    "Under stream capture"): the call is recorded on the stream handed to rt_set_stream BEFORE the capture begins, a replay
    redoes exactly the recorded frame -- same scene buffers, same camera, same parameters, into the same caller's buffers --
    bit for bit and as often as wanted while the context, its scene and those buffers live; RT_FLAG_ACCUMULATE is refused under
    capture.  Three replays each of a primary + shadow graph and of a path-tracing graph, stale buffers before every one."""
    import torch
    W, H = 640, 360
    app = soup_app(60, 0x5EED0003, W, H)
    app.camera_frame_scene(1.0)
    scene = app.prepare_meshinfo()
    with rt.Context(0) as ctx:
        ctx.upload_scene(scene)
        m, p = app.camera_view()
        ctx.set_view(m, p, W, H)
        assert ctx.render(flags=rt.FLAG_SHADOW | rt.FLAG_COUNTERS) == rt.RT_OK
        want = (ctx.read_radiance(), ctx.read_prim_ids(), ctx.read_hit_t())
        c0 = ctx.counters()
        assert ctx.render(flags=rt.FLAG_PATH_TRACE, spp=2, max_segments=3, seed=11) == rt.RT_OK
        want_acc, want_rad = ctx.read_accumulation(), ctx.read_radiance()
        ctx.synchronize()
        # the caller's outputs: allocated BEFORE the capture and alive for as long as the graphs are
        rad = torch.full((W * H * 4,), 7.0, dtype=torch.float32, device="cuda")
        prim = torch.full((W * H,), 12345, dtype=torch.int32, device="cuda")
        t = torch.full((W * H,), 3.0, dtype=torch.float32, device="cuda")
        acc = torch.full((W * H * 4,), 9.0, dtype=torch.float32, device="cuda")
        args = (rad.data_ptr(), prim.data_ptr(), t.data_ptr(), acc.data_ptr(), W * H)
        torch.cuda.synchronize()
        side = torch.cuda.Stream(device=0)
        ctx.set_stream(side.cuda_stream)                 # outside any capture (it synchronises)
        # one direct call on that stream first: same launches, not captured
        assert ctx.render_static(*args, flags=rt.FLAG_SHADOW | rt.FLAG_COUNTERS) == rt.RT_OK
        ctx.synchronize()
        assert bitwise_equal(rad.cpu().numpy().reshape(H, W, 4), want[0])
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, stream=side):
            rc1 = ctx.render_static(*args, flags=rt.FLAG_SHADOW | rt.FLAG_COUNTERS)
        assert rc1 == rt.RT_OK
        with torch.cuda.graph(g2, stream=side):
            refused = None
            try:
                ctx.render_static(*args, flags=rt.FLAG_PATH_TRACE | rt.FLAG_ACCUMULATE, spp=2, max_segments=3, seed=11)
            except rt.RtError as e:
                refused = e
            rc2 = ctx.render_static(*args, flags=rt.FLAG_PATH_TRACE, spp=2, max_segments=3, seed=11)
        assert rc2 == rt.RT_OK and refused is not None and refused.status == rt.RT_ERR_ARG
        for rep in range(3):
            with torch.cuda.stream(side):
                rad.fill_(7.0); prim.fill_(12345); t.fill_(3.0); acc.fill_(9.0)
                g1.replay()
            side.synchronize()
            c1 = ctx.counters()
            assert bitwise_equal(rad.cpu().numpy().reshape(H, W, 4), want[0]), rep
            assert bitwise_equal(prim.cpu().numpy().view(np.uint32).reshape(H, W), want[1]), rep
            assert bitwise_equal(t.cpu().numpy().reshape(H, W), want[2]), rep
            assert (c1.rays_primary, c1.rays_shadow, c1.node_visits, c1.tri_tests) == (c0.rays_primary, c0.rays_shadow, c0.node_visits, c0.tri_tests)
            with torch.cuda.stream(side):
                rad.fill_(7.0); acc.fill_(9.0)
                g2.replay()
            side.synchronize()
            assert bitwise_equal(acc.cpu().numpy().reshape(H, W, 4), want_acc), rep
            assert bitwise_equal(rad.cpu().numpy().reshape(H, W, 4), want_rad), rep
        # the graphs go before the context and the buffers do
        del g1, g2
        torch.cuda.synchronize()
        ctx.set_stream(0)
