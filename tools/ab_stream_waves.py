"""Experiment script (GPU): the 7- and the 8-wave build of k_stream (RT_TUNE_STREAM_WAVES) by frames in flight, bench scene,
1 M-triangle soup and default scene at 1080p.  Not a test."""
import os, sys, time, zlib
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import ray_tracing_amd as rt
for kind in os.environ.get("AB_KINDS", "field,soup,default").split(","):
    app = rt.App()
    if kind == "field": app.spawn_sphere_field(10000, 0x5EED0002, 5); app.camera_frame_scene(1.0)
    elif kind == "soup": app.spawn_icosphere_soup(1389, 0x5EED0003, 5); app.camera_frame_scene(1.0)
    else: app.spawn_default_scene()
    app.camera_set_target(1920, 1080)
    sc, (m, p) = app.prepare_meshinfo(), app.camera_view()
    for rep in range(2):
        for waves in ("7", "8", "0"):
            os.environ["RT_TUNE_STREAM_WAVES"] = waves
            ctx = rt.Context(0); ctx.upload_scene(sc); ctx.set_view(m, p, 1920, 1080)
            res = []
            for nf in (1, 2, 3, 4):
                ctx.set_frames_in_flight(nf)
                for _ in range(8): ctx.render(flags=rt.FLAG_SHADOW)
                ctx.synchronize()
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    for _ in range(60): ctx.render(flags=rt.FLAG_SHADOW)
                    ctx.synchronize()
                    best = min(best, (time.perf_counter() - t0) / 60 * 1e3)
                res.append(best)
            print("%-7s waves %s: ms per frame with 1 / 2 / 3 / 4 in flight: %s  crc %08x" % (kind, waves if waves != "0" else "auto", " ".join("%.4f" % v for v in res),
                  zlib.crc32(ctx.read_radiance().tobytes())), flush=True)
            ctx.close()
