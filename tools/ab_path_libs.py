"""Experiment script (GPU): builds of the library (comma-separated paths relative to the repo, "default" = the shipped one) on
the path-tracing workloads: 10 M triangles 1080p <= 8 segments, 1 M triangles 4K 4 spp <= 4 segments; ms per frame by HIP events,
one frame at a time, and the accumulation's checksum.  Not a test."""
import os, subprocess, sys, time, zlib
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    for rep in range(2):
        for lib in sys.argv[1].split(","):
            env = dict(os.environ, AB_NAME=lib)
            if lib != "default": env["RT_AMD_LIB"] = os.path.join(root, lib)
            subprocess.run([sys.executable, "-u", os.path.abspath(__file__)], env=env, check=True, timeout=400)
    sys.exit(0)
sys.path[:0] = [root]
import ray_tracing_amd as rt
for name, n, seed, W, H, spp, segs in (("soup10M_1080p", 13889, 0x5EED0005, 1920, 1080, 1, 8), ("soup1M_4K", 1389, 0x5EED0003, 3840, 2160, 4, 4)):
    app = rt.App(); app.spawn_icosphere_soup(n, seed, 5); app.camera_frame_scene(1.0); app.camera_set_target(W, H)
    sc = app.prepare_meshinfo(); m, p = app.camera_view()
    ctx = rt.Context(0); ctx.upload_scene(sc); ctx.set_view(m, p, W, H)
    for _ in range(2): ctx.render(flags=rt.FLAG_PATH_TRACE, spp=spp, max_segments=segs, seed=1)
    ctx.counters()
    for _ in range(6): ctx.render(flags=rt.FLAG_PATH_TRACE | rt.FLAG_TIME_KERNELS, spp=spp, max_segments=segs, seed=1)
    ms = ctx.counters().kernel_ms / 6
    ctx.set_frames_in_flight(2)
    for _ in range(3): ctx.render(flags=rt.FLAG_PATH_TRACE, spp=spp, max_segments=segs, seed=1)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ctx.render(flags=rt.FLAG_PATH_TRACE, spp=spp, max_segments=segs, seed=1)
    ctx.synchronize(); ms2 = (time.perf_counter() - t0) / 10 * 1e3
    print("%-14s %-40s %.3f ms/frame one at a time, %.3f with 2 in flight, accumulation crc %08x" % (name, os.environ.get("AB_NAME", "default"), ms, ms2, zlib.crc32(ctx.read_accumulation().tobytes())), flush=True)
    ctx.close()
