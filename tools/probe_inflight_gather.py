"""Experiment script (GPU): what keeps two frames in flight from overlapping once a pack kernel / a collective follows each render. Not a test."""
import os, sys, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import torch
import ray_tracing_amd as rt
torch.cuda.set_device(0)
NCCL = os.environ.get("PROBE_NCCL", "") in ("before", "after")
def init_nccl():
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
    print("RCCL initialised", os.environ["PROBE_NCCL"], "the context", flush=True)
if os.environ.get("PROBE_NCCL") == "before": init_nccl()
app = rt.App(); app.spawn_sphere_field(10000, 0x5EED0002, 5); app.camera_frame_scene(1.0); app.camera_set_target(1920, 1080)
sc = app.prepare_meshinfo(); m, p = app.camera_view()
ctx = rt.Context(0); ctx.set_frames_in_flight(2); ctx.upload_scene(sc); ctx.set_view(m, p, 1920, 1080)
if os.environ.get("PROBE_NCCL") == "after": init_nccl()
slot = ctx.tile_slot_bytes(1, 64)
bufs = [torch.zeros(slot // 4, device="cuda") for _ in range(2)]
torch.cuda.synchronize()
def run(name, fn, n=60):
    for _ in range(6): fn(0)
    ctx.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n): fn(i)
    ctx.synchronize(); torch.cuda.synchronize()
    print(name, "ms/step %.4f" % ((time.perf_counter() - t0) / n * 1e3), flush=True)
def render(i, **kw): ctx.render(flags=rt.FLAG_SHADOW, **kw)
run("render only (default tile)", lambda i: render(i))
run("render only, tile 64 shard 1/1", lambda i: render(i, shard_rank=0, shard_count=1, tile_size=64))
def rp(i):
    render(i); ctx.pack_tiles(0, 1, 64, bufs[i & 1].data_ptr(), slot)
run("render + pack", rp)
def rs(i):
    render(i)
    st = torch.cuda.ExternalStream(ctx.get_stream())
    with torch.cuda.stream(st): bufs[i & 1].add_(1.0)
run("render + torch add_ on the frame's stream", rs)
if NCCL:
    import torch.distributed as dist
    recv = [[torch.zeros_like(bufs[0])] for _ in range(2)]
    def rpg(i):
        render(i)
        st = torch.cuda.ExternalStream(ctx.get_stream())
        with torch.cuda.stream(st):
            ctx.pack_tiles(0, 1, 64, bufs[i & 1].data_ptr(), slot)
            dist.gather(bufs[i & 1], recv[i & 1], dst=0)
    run("render + pack + RCCL gather (one rank)", rpg)
    def rpg_t(i):
        ctx.render(flags=rt.FLAG_SHADOW | rt.FLAG_TIME_KERNELS, shard_rank=0, shard_count=1, tile_size=64)
        st = torch.cuda.ExternalStream(ctx.get_stream())
        with torch.cuda.stream(st):
            ctx.pack_tiles(0, 1, 64, bufs[i & 1].data_ptr(), slot)
            dist.gather(bufs[i & 1], recv[i & 1], dst=0)
    run("the same with RT_FLAG_TIME_KERNELS", rpg_t)
    ctx.counters()
    side = torch.cuda.Stream(device=0)
    torch.cuda.set_stream(side)
    run("the same, torch's current stream a side stream", rpg)
    from ray_tracing_amd import multi_gpu
    g = multi_gpu.TileGather(1920, 1080, 64, 1, 0, ctx=ctx, device="cuda:0", in_flight=2)
    def tg(i):
        ctx.render(flags=rt.FLAG_SHADOW, shard_rank=0, shard_count=1, tile_size=64)
        g.combine_in_flight()
    run("TileGather.combine_in_flight", tg)
    torch.cuda.set_stream(torch.cuda.default_stream())
ctx.set_frames_in_flight(1)
run("one frame in flight: render only", lambda i: render(i))
run("one frame in flight: render + pack", rp)
