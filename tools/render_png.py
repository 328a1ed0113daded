"""Demo (GPU): the reference's app flow through the host mirror -- App + RayTracingPlugin, the default scene of src/main.rs (or a
synthetic one), an optional fly-cam script, RayTracingPassNode::run every frame -- and the last frame written as the PNG the
Bgra8UnormSrgb target would show (rt_host.h: rth_write_png).

    python tools/render_png.py out.png [--scene default|field|soup] [--size 1280x720] [--spp 64] [--frames 30] [--script keys.txt]

--spp 0 (default): the live path (primary rays + the build-defined shadow ray); --spp N: N path-traced samples per pixel,
accumulated over the frames while the camera rests.  The cube of the default scene rotates between frames as in src/main.rs:122-126."""
import argparse, os, sys, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import ray_tracing_amd as rt

ap = argparse.ArgumentParser()
ap.add_argument("out")
ap.add_argument("--scene", default="default", choices=["default", "field", "soup"])
ap.add_argument("--size", default="1280x720")
ap.add_argument("--spp", type=int, default=0)
ap.add_argument("--frames", type=int, default=1)
ap.add_argument("--script", help="fly-cam script: one 'dt mouse_dx mouse_dy [keys]' line per step, run before the first frame")
a = ap.parse_args()
W, H = (int(v) for v in a.size.split("x"))
app = rt.App()
if a.scene == "default": app.spawn_default_scene()
elif a.scene == "field": app.spawn_sphere_field(10000, 0x5EED0002, 5); app.camera_frame_scene(1.0)
else: app.spawn_icosphere_soup(1389, 0x5EED0003, 5); app.camera_frame_scene(1.0)
app.camera_set_target(W, H)
if a.script:
    print("fly-cam steps:", app.fly_cam_run_script(open(a.script).read()))
app.finish(0)                                   # RayTracingPlugin::finish: pipeline (rt_create)
t0 = time.perf_counter()
for f in range(a.frames):
    if a.spp:
        per = max(1, a.spp // a.frames)
        app.update(flags=rt.FLAG_PATH_TRACE | (rt.FLAG_ACCUMULATE if f else 0), spp=per, sample_first=f * per, max_segments=4, seed=1)
    else:
        if a.scene == "default" and f: app.rotate_tagged(1.0 / 60.0)
        app.update(flags=rt.FLAG_SHADOW)
ctx = app.ctx()
bgra = ctx.read_bgra8_srgb()
print("%d frame(s) of %dx%d in %.1f ms (first frame includes the BVH build)" % (a.frames, W, H, (time.perf_counter() - t0) * 1e3))
rt.write_image(a.out, bgra)
print("wrote", a.out, os.path.getsize(a.out), "bytes; mean byte", float(bgra[..., :3].mean()))
