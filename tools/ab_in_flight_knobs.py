"""Experiment script (GPU): the k_stream knobs re-checked with 4 frames in flight (the loaded phase is what counts there, the
frame's tail does not): ms per frame on the bench scene and the 1 M-triangle soup.  Not a test."""
import os, sys, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import ray_tracing_amd as rt
SETTINGS = [{}, {"RT_TUNE_BLOCKS_PER_CU": "7"}, {"RT_TUNE_BLOCKS_PER_CU": "6"}, {"RT_TUNE_BLOCKS_PER_CU": "5"}, {"RT_TUNE_BLOCKS_PER_CU": "4"},
            {"RT_TUNE_REFILL_MIN": "48"}, {"RT_TUNE_REFILL_MIN": "32"}, {"RT_TUNE_FINISH_MIN": "48"}, {"RT_TUNE_FINISH_MIN": "32"},
            {"RT_TUNE_REFILL_MIN": "32", "RT_TUNE_FINISH_MIN": "32"}, {"RT_TUNE_WIDE_VOTE_NODE": "2"}, {"RT_TUNE_WIDE_VOTE_NODE": "4"},
            {"RT_TUNE_UNI_MIN": "65"}, {"RT_TUNE_UNI_MIN": "8"}]
KEYS = sorted({k for s in SETTINGS for k in s})
scenes = []
for kind in ("field", "soup"):
    app = rt.App()
    if kind == "field": app.spawn_sphere_field(10000, 0x5EED0002, 5)
    else: app.spawn_icosphere_soup(1389, 0x5EED0003, 5)
    app.camera_frame_scene(1.0); app.camera_set_target(1920, 1080)
    scenes.append((kind, app.prepare_meshinfo(), app.camera_view()))
for kind, sc, (m, p) in scenes:
    base = None
    for st in SETTINGS:
        for k in KEYS: os.environ.pop(k, None)
        os.environ.update(st)
        ctx = rt.Context(0); ctx.upload_scene(sc); ctx.set_view(m, p, 1920, 1080)
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
        prim = ctx.read_prim_ids()
        if base is None: base = prim
        print("%-6s %-58s 1 in flight %.4f ms   4 in flight %.4f ms   identical %s" % (kind, st or "defaults", res[0], res[1], bool((prim == base).all())), flush=True)
        ctx.close()
