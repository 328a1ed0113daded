"""Experiment script (GPU): would path-traced frames gain from frames in flight?  Two contexts (own streams, own queues and
accumulation buffers) rendering alternate path-traced frames against one context rendering them in a row.  Not a test."""
import os, sys, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import ray_tracing_amd as rt
for name, n, seed, W, H, spp, segs in (("soup10M_1080p_1spp_8seg", 13889, 0x5EED0005, 1920, 1080, 1, 8), ("soup1M_4K_4spp_4seg", 1389, 0x5EED0003, 3840, 2160, 4, 4),
                                       ("soup1M_1080p_4spp_1seg", 1389, 0x5EED0003, 1920, 1080, 4, 1)):
    app = rt.App(); app.spawn_icosphere_soup(n, seed, 5); app.camera_frame_scene(1.0); app.camera_set_target(W, H)
    sc = app.prepare_meshinfo(); m, p = app.camera_view()
    ctxs = []
    for _ in range(3):
        c = rt.Context(0); c.set_frames_in_flight(2); c.set_frames_in_flight(1)   # no effect on streams; keeps the contexts alike
        c.upload_scene(sc); c.set_view(m, p, W, H); ctxs.append(c)
    def run(k, frames=12):
        for i in range(4): ctxs[i % k].render(flags=rt.FLAG_PATH_TRACE, spp=spp, max_segments=segs, seed=1)
        for c in ctxs: c.synchronize()
        t0 = time.perf_counter()
        for i in range(frames): ctxs[i % k].render(flags=rt.FLAG_PATH_TRACE, spp=spp, max_segments=segs, seed=1)
        for c in ctxs: c.synchronize()
        return (time.perf_counter() - t0) / frames * 1e3
    for k in (1, 2, 3):
        print(name, "contexts", k, "ms/frame %.3f" % run(k), flush=True)
    for c in ctxs: c.close()
