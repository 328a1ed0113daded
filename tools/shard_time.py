"""Experiment script (GPU): what one rank of N does per frame in tiles mode -- render of its 64x64 tiles + pack -- on the
bench scene, N = 1, 2, 4, 8 (the strong-scaling floor a one-GPU box can measure).  Not a test."""
import os, sys, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import torch
import ray_tracing_amd as rt
GATHER = bool(os.environ.get("SHARD_GATHER"))          # add a one-rank RCCL gather of the packed tiles on the frame's stream
if GATHER:
    import torch.distributed as dist
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
app = rt.App(); app.spawn_sphere_field(10000, 0x5EED0002, 5); app.camera_frame_scene(1.0); app.camera_set_target(1920, 1080)
sc = app.prepare_meshinfo(); m, p = app.camera_view()
ctx = rt.Context(0); ctx.upload_scene(sc); ctx.set_view(m, p, 1920, 1080)
for inflight in (1, 2, 3, 4):
    ctx.set_frames_in_flight(inflight)
    for n in (1, 2, 4, 8):
        slot = ctx.tile_slot_bytes(n, 64)
        bufs = [torch.zeros(slot // 4, device="cuda") for _ in range(4)]
        recv = [[torch.zeros_like(bufs[0])] for _ in range(4)]
        turn = [0]
        for rank in sorted({0, n - 1}):
            def step(flags=rt.FLAG_SHADOW):
                ctx.render(flags=flags, shard_rank=rank, shard_count=n, tile_size=64)
                k = turn[0] = (turn[0] + 1) % inflight
                if GATHER:
                    with torch.cuda.stream(torch.cuda.ExternalStream(ctx.get_stream())):
                        ctx.pack_tiles(rank, n, 64, bufs[k].data_ptr(), slot)
                        dist.gather(bufs[k], recv[k], dst=0)
                else:
                    ctx.pack_tiles(rank, n, 64, bufs[k].data_ptr(), slot)
            for _ in range(10): step()
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(100): step()
            ctx.synchronize()
            dt = (time.perf_counter() - t0) / 100 * 1e3
            ctx.counters()
            for _ in range(20): step(rt.FLAG_SHADOW | rt.FLAG_TIME_KERNELS)
            ctx.synchronize()
            k = ctx.counters()
            print("in flight %d  N %d rank %d: render+pack %.4f ms/step, kernel %.4f ms, ideal %.4f" % (inflight, n, rank, dt, k.kernel_ms / 20, 0.774 / n), flush=True)
