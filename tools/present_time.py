"""Experiment script (GPU): what it costs to DELIVER every frame of the bench workload to the host as Bgra8UnormSrgb bytes
(src/pipeline.rs:56-60) -- rt_read_bgra8_srgb after every frame (stalls the pipeline) against rt_present_async /
rt_present_wait with 1, 2 and 4 frames in flight (encode + PCIe copy on the frame's own stream, collected n frames later),
against frames that stay in HBM.  Not a test."""
import os, sys, time
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT]
import numpy as np, torch, ray_tracing_amd as rt
W, H = 1920, 1080
app = rt.App(); app.spawn_sphere_field(10000, 0x5EED0002, 5); app.camera_frame_scene(1.0); app.camera_set_target(W, H)
sc = app.prepare_meshinfo(); m, p = app.camera_view()
ctx = rt.Context(0); ctx.upload_scene(sc); ctx.set_view(m, p, W, H)
ctx.render(flags=rt.FLAG_SHADOW | rt.FLAG_COUNTERS); cnt = ctx.counters(); rays = cnt.rays_primary + cnt.rays_shadow
want = ctx.read_bgra8_srgb()
N = 200
bufs = [torch.empty(H * W * 4, dtype=torch.uint8).pin_memory().numpy().reshape(H, W, 4) for _ in range(4)]


def line(what, dt):
    print("%-58s %.3f ms/frame = %5.0f Mrays/s = %4.0f frames/s" % (what, dt / N * 1e3, rays * N / dt / 1e6, N / dt), flush=True)


for n in (1, 2, 4):
    ctx.set_frames_in_flight(n)
    for _ in range(8): ctx.render(flags=rt.FLAG_SHADOW)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(N): ctx.render(flags=rt.FLAG_SHADOW)
    ctx.synchronize(); line("%d in flight, frames stay in HBM" % n, time.perf_counter() - t0)
    t0 = time.perf_counter()
    for _ in range(N):
        ctx.render(flags=rt.FLAG_SHADOW)
        ctx.read_bgra8_srgb()
    line("%d in flight, rt_read_bgra8_srgb after every frame" % n, time.perf_counter() - t0)
    tickets = []
    t0 = time.perf_counter()
    for k in range(N):
        ctx.render(flags=rt.FLAG_SHADOW)
        if len(tickets) == n:
            ctx.present_wait(tickets.pop(0))
        tickets.append(ctx.present_async(bufs[k % n]))
    for t in tickets: ctx.present_wait(t)
    line("%d in flight, rt_present_async, collected %d frame(s) later" % (n, n), time.perf_counter() - t0)
    assert all(np.array_equal(b, want) for b in bufs[:n])
ctx.close()
