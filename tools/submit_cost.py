"""Experiment script (GPU): host time to SUBMIT one step (rt_render of a 1/8-frame shard + rt_pack_tiles [+ a one-rank RCCL gather]),
i.e. the loop's wall time before any synchronisation, against the step's time on the GPU.  Not a test."""
import os, sys, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import torch
import ray_tracing_amd as rt
GATHER = bool(os.environ.get("SHARD_GATHER"))
if GATHER:
    import torch.distributed as dist
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29546")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
app = rt.App(); app.spawn_sphere_field(10000, 0x5EED0002, 5); app.camera_frame_scene(1.0); app.camera_set_target(1920, 1080)
sc = app.prepare_meshinfo(); m, p = app.camera_view()
ctx = rt.Context(0); ctx.upload_scene(sc); ctx.set_view(m, p, 1920, 1080)
ctx.set_frames_in_flight(4)
n = 8
slot = ctx.tile_slot_bytes(n, 64)
bufs = [torch.zeros(slot // 4, device="cuda") for _ in range(4)]
recv = [[torch.zeros_like(bufs[0])] for _ in range(4)]
streams = {}
def step(k, what):
    ctx.render(flags=rt.FLAG_SHADOW, shard_rank=0, shard_count=n, tile_size=64)
    if what == "render": return
    if what == "pack":
        ctx.pack_tiles(0, n, 64, bufs[k].data_ptr(), slot); return
    ptr = ctx.get_stream()
    st = streams.get(ptr) or streams.setdefault(ptr, torch.cuda.ExternalStream(ptr))
    with torch.cuda.stream(st):
        ctx.pack_tiles(0, n, 64, bufs[k].data_ptr(), slot)
        dist.gather(bufs[k], recv[k], dst=0)
for what in ["render", "pack"] + (["gather"] if GATHER else []):
    for i in range(20): step(i % 4, what)
    ctx.synchronize(); torch.cuda.synchronize()
    for N in (20, 200):
        t0 = time.perf_counter()
        for i in range(N): step(i % 4, what)
        t1 = time.perf_counter()
        ctx.synchronize(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%-7s %4d steps: submitted in %.4f ms/step, done in %.4f ms/step" % (what, N, (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3), flush=True)
ctx.close()
