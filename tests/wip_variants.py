import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np, ray_tracing_amd as rt
ctx = rt.Context(0)
for kind,n,seed in (("field",10000,0x5EED0002),("soup",1389,0x5EED0003)):
    app = rt.App()
    (app.spawn_sphere_field if kind=="field" else app.spawn_icosphere_soup)(n, seed, 5)
    sc = app.prepare_meshinfo(); ctx.upload_scene(sc)
    app.camera_set_target(1920,1080)
    app.camera_frame_scene(1.0); m,p = app.camera_view(); ctx.set_view(m,p,1920,1080)
    for name, fl in (("static", 0), ("persist", rt.FLAG_PERSISTENT)):
        f = rt.FLAG_SHADOW | fl
        for _ in range(3): ctx.render(flags=f)
        ctx.counters()
        for _ in range(10): ctx.render(flags=f|rt.FLAG_TIME_KERNELS)
        k=ctx.counters(); ms=k.kernel_ms/k.kernel_launches
        rays=k.rays_primary+k.rays_shadow
        print(os.environ.get("RT_AMD_LIB","default")[-14:], kind, name, "ms %.3f Mrays/s %.0f"%(ms, rays/ms/1e3), flush=True)
