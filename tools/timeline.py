"""Experiment script (GPU, needs a -DRT_DEBUG_TIMELINE build via RT_AMD_LIB): how many waves are running over the frame. Not a test."""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "oracle")]
import numpy as np, ray_tracing_amd as rt
for kind in ("field", "soup"):
    app = rt.App()
    if kind == "field": app.spawn_sphere_field(10000, 0x5EED0002, 5)
    else: app.spawn_icosphere_soup(1389, 0x5EED0003, 5)
    app.camera_frame_scene(1.0); app.camera_set_target(1920, 1080)
    sc = app.prepare_meshinfo(); m, p = app.camera_view()
    ctx = rt.Context(0); ctx.upload_scene(sc); ctx.set_view(m, p, 1920, 1080)
    for _ in range(3): ctx.render(flags=rt.FLAG_SHADOW)
    t = ctx.read_hit_t().view(np.uint32).reshape(-1)[:2 * 8192].reshape(-1, 2).astype(np.int64)
    t = t[t[:, 1] != 0]
    t0 = t[:, 0].min(); s = (t[:, 0] - t0) / 100.0; e = (t[:, 1] - t0) / 100.0     # microseconds
    total = e.max()
    print(kind, "waves", len(t), "frame %.0f us; wave start: median %.0f max %.0f; wave end: p10 %.0f median %.0f p90 %.0f max %.0f" % (
        total, np.median(s), s.max(), np.percentile(e, 10), np.median(e), np.percentile(e, 90), e.max()))
    grid = np.linspace(0, total, 21)
    running = [(int(((s <= g) & (e > g)).sum())) for g in grid]
    print("  waves running at 5 % steps of the frame:", running)
    busy = (e - s).sum() / (len(t) * total)
    print("  mean share of the frame a wave is running: %.3f" % busy, flush=True)
    ctx.close()
