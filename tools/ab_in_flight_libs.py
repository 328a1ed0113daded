"""Experiment script (GPU): builds of the library (RT_AMD_LIB, comma-separated paths relative to the repo, "default" = the
shipped one) on the bench scene and the 1 M-triangle soup, 1 and 4 frames in flight, one process per build.  Not a test."""
import os, subprocess, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    for rep in range(2):
        for lib in sys.argv[1].split(","):
            env = dict(os.environ)
            name = lib
            lib, _, envs = lib.partition("@")                 # "default@RT_TUNE_X=1@RT_TUNE_Y=2": the build plus environment for that run
            for kv in filter(None, envs.split("@")):
                k, _, v = kv.partition("=")
                env[k] = v
            if lib != "default": env["RT_AMD_LIB"] = os.path.join(root, lib)
            env["AB_NAME"] = name
            subprocess.run([sys.executable, "-u", os.path.abspath(__file__)], env=env, check=True, timeout=300)
    sys.exit(0)
sys.path[:0] = [root]
import ray_tracing_amd as rt
for kind in os.environ.get("AB_KINDS", "field,soup").split(","):
    app = rt.App()
    if kind == "field": app.spawn_sphere_field(10000, 0x5EED0002, 5)
    else: app.spawn_icosphere_soup(1389, 0x5EED0003, 5)
    app.camera_frame_scene(1.0); app.camera_set_target(1920, 1080)
    sc, (m, p) = app.prepare_meshinfo(), app.camera_view()
    ctx = rt.Context(0); ctx.upload_scene(sc); ctx.set_view(m, p, 1920, 1080)
    ctx.render(flags=rt.FLAG_SHADOW | rt.FLAG_COUNTERS); c = ctx.counters()
    res = []
    for nf in (1, 4):
        ctx.set_frames_in_flight(nf)
        for _ in range(8): ctx.render(flags=rt.FLAG_SHADOW)
        ctx.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(60): ctx.render(flags=rt.FLAG_SHADOW)
            ctx.synchronize()
            best = min(best, (time.perf_counter() - t0) / 60 * 1e3)
        res.append(best)
    import zlib
    print("%-6s %-34s 1 in flight %.4f ms   4 in flight %.4f ms   nodes/ray %.2f tris/ray %.2f  prim crc %08x rgba crc %08x" % (
        kind, os.environ.get("AB_NAME", "default"), res[0], res[1], c.node_visits / (c.rays_primary + c.rays_shadow),
        c.tri_tests / (c.rays_primary + c.rays_shadow), zlib.crc32(ctx.read_prim_ids().tobytes()), zlib.crc32(ctx.read_radiance().tobytes())), flush=True)
    ctx.close()
