#!/usr/bin/env python3
"""bench.py -- Mrays/s of the per-pixel hot path on MI355X (BASELINE.json metric).

Workload at N=1 (BASELINE.json configs[1]): synthetic 10k-sphere field,
reference-faithful tessellation (ico(5): 7.2 M triangles + ground plane), 1920x1080, one primary
ray per pixel + one shadow ray per hit.  A step = one frame through rt_render (C ABI).
N>1: scene replicated, one process per GPU (launched by torch.distributed.run, or -- when WORLD_SIZE is not set --
by this script itself, which starts N rank processes before anything touches a GPU).  --mode tiles (the default,
BASELINE.json north_star / SURVEY 8e): ONE frame sharded by 64x64 screen tiles round-robin over the ranks, every rank
packs its tiles with a HIP kernel and ONE gather over RCCL (one xGMI link per peer) brings them to rank 0, which
scatters them into its frame with one more kernel: strong scaling.  --mode samples (path-tracing workloads): every
rank its own sample slice of ONE frame, accumulation buffers combined by a direct reduce-scatter + gather (weak
scaling in samples, configs[4]).  --mode frames: every rank its own camera on an orbit, no data-path collective.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and
`cpu_baseline` objects.  Never reads /root/reference.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

CAMERA_RADII = 1.0
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    # name: (kind, count, seed, width, height[, spp, max_segments])   spp > 0 => path tracing (wgsl:180-202 shape)
    "sphere_field_10k_1080p_primary_shadow": ("field", 10000, 0x5EED0002, 1920, 1080),     # configs[1], the default
    "sphere_field_10k_analytic_1080p_primary_shadow": ("field_analytic", 10000, 0x5EED0002, 1920, 1080),   # configs[1] variant (ii): extension
    "soup_1m_1080p_primary_shadow": ("soup", 1389, 0x5EED0003, 1920, 1080),
    "soup_1m_1080p_4spp": ("soup", 1389, 0x5EED0003, 1920, 1080, 4, 1),                    # configs[2]: jittered primaries
    "soup_1m_4k_path_4spp_4seg": ("soup", 1389, 0x5EED0003, 3840, 2160, 4, 4),             # configs[3] per 4 of its 64 spp
    "soup_10m_1080p_path_1spp_8seg": ("soup", 13889, 0x5EED0005, 1920, 1080, 1, 8),        # configs[4] scene, one GPU
    "default_640x360": ("default", 0, 0, 640, 360),
    "sphere_field_200_small": ("field", 200, 0x5EED0002, 640, 360),
}


def build_app(rt, workload, rank=0, world=1):
    kind, count, seed, W, H = WORKLOADS[workload][:5]
    app = rt.App()
    # Camera: eye ON the scene's bounding sphere (1.0 radii) looking at the centroid.  SURVEY 8d
    # suggested 2.5 radii, but from there only 14 % of the primary rays hit anything (measured),
    # which flatters Mrays/s; at 1.0 radii 66 % (field) / 48 % (soup) of the pixels hit geometry.
    if kind == "field":
        app.spawn_sphere_field(count, seed, 5)
        app.camera_frame_scene(CAMERA_RADII)
    elif kind == "field_analytic":
        app.spawn_sphere_field(count, seed, 0)   # same entities (icosahedra stand in): only the camera framing is taken from it
        app.camera_frame_scene(CAMERA_RADII)
    elif kind == "soup":
        app.spawn_icosphere_soup(count, seed, 5)
        app.camera_frame_scene(CAMERA_RADII)
    else:
        app.spawn_default_scene()
    app.camera_set_target(W, H)
    return app, W, H


def orbit_camera(app, rank, world):
    """Rank r looks at the scene from its own azimuth (frames mode): same eye distance and height."""
    if world <= 1:
        return
    m, p = app.camera_view()
    # forward = -third rotation column = world_from_clip column 3 (xyz); keep the look-at target
    fwd = np.array(m[12:15], np.float64)
    # target = eye + fwd * distance, distance chosen as the eye's distance to the y=centre plane
    dist = float(np.linalg.norm(p)) if np.linalg.norm(p) > 0 else 1.0
    target = np.array(p, np.float64) + fwd * dist
    rel = np.array(p, np.float64) - target
    ang = 2.0 * math.pi * rank / world
    c, s = math.cos(ang), math.sin(ang)
    rel2 = np.array([c * rel[0] + s * rel[2], rel[1], -s * rel[0] + c * rel[2]])
    app.camera_look_at((target + rel2).astype(np.float32), target.astype(np.float32), (0, 1, 0))


def cpu_baseline(orc, scene, view, W, H, flags, target_seconds=15.0):
    """Time the oracle's reference-shaped brute-force loop on a bounded seeded pixel sample."""
    threads = max(1, min(16, os.cpu_count() or 1))
    osc = orc.Scene(scene.triangles, scene.meshes, scene.vertices, scene.materials,
                    getattr(scene, "spheres", None), getattr(scene, "sphere_materials", None))
    rng = np.random.default_rng(0x5EED)
    perm = rng.permutation(W * H).astype(np.uint32)
    oflags = orc.FLAG_SHADOW if flags & 1 else 0
    # calibrate on a small slice, then size the sample for ~target_seconds
    n0 = min(len(perm), 512 * threads)
    t0 = time.perf_counter()
    orc.render_primary(osc, view, oflags, threads=threads, pixel_list=perm[:n0])
    dt0 = max(time.perf_counter() - t0, 1e-6)
    n = int(max(n0, min(len(perm), n0 * target_seconds / dt0)))   # ~target_seconds of CPU work, at most the frame
    t0 = time.perf_counter()
    rgba, prim, t, rays = orc.render_primary(osc, view, oflags, threads=threads, pixel_list=perm[:n])
    dt = time.perf_counter() - t0
    return dict(value=rays / dt / 1e6, unit="Mrays/s", cores=threads, kind="port",
                sample=f"seeded random {n} of {W * H} pixels ({100.0 * n / (W * H):.2f} %), primary+shadow, "
                       f"reference-shaped brute-force loop (oracle/rt_oracle.c), {dt:.1f} s"), perm[:n], rgba, prim


def self_launch(n):
    """`bench.py --gpus N` without a launcher: start N rank processes (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment) from THIS process, which has not touched a GPU and never will; rank 0's stdout is
    the bench line."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for pr in procs:
        rc = max(rc, abs(pr.wait()))
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)      # 0.12 s of frames: the four-deep pipeline's fill and drain (inside the timed region) weigh 1 %, not 4
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="sphere_field_10k_1080p_primary_shadow", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="tiles", choices=["frames", "tiles", "samples"], help="N > 1 only; tiles = north_star's partitioning")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the extra two-frames-in-flight measurement (N=1, after the timed region)")
    ap.add_argument("--no-north-star", action="store_true", help="skip the extra 1M-triangle soup measurement (N=1, default workload, after the timed region)")
    ap.add_argument("--in-flight", type=int, default=4, choices=[1, 2, 3, 4],
                    help="primary/shadow workloads: frames in flight inside the context during the timed region (rt_set_frames_in_flight: "
                         "the render node submits frame after frame without waiting, as the reference's does; in tiles mode frame k's "
                         "pack -> gather -> unpack overlaps with the next frames' renders).  The same at every N.  1: one frame at a time "
                         "(always measured too, after the timed region: one_frame_at_a_time, and the roofline's launch duration)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--extra-flags", type=lambda v: int(v, 0), default=0,
                    help="experiments only: OR these RT_FLAG_* bits into every render (e.g. 0x100 = static grid, 0x200 = binary tree)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    # stdout carries ONE JSON line and nothing else: libraries that print banners to file descriptor 1 (RCCL prints its version
    # block there when a communicator is created) get stderr instead, the line goes to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: ray_tracing_amd has no CPU fallback")
    # rehearsal hooks (one-GPU box): BENCH_FORCE_DEVICE puts every rank on one card, BENCH_BACKEND=gloo
    # replaces RCCL, which refuses two ranks on one device.  Never set by the driver.
    if "BENCH_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["BENCH_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    # BENCH_FORCE_DIST=1 (rehearsal hook): a ONE-rank process group, so that the RCCL code path of the collective modes
    # (device tensors, pack -> gather -> unpack on one stream) can run on a one-GPU box.  Never set by the driver.
    force_dist = world == 1 and bool(os.environ.get("BENCH_FORCE_DIST"))
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import ray_tracing_amd as rt

    app, W, H = build_app(rt, args.workload, rank, world)
    distributed = world > 1 or force_dist
    tiles = args.mode == "tiles" and distributed
    samples = args.mode == "samples" and distributed
    if samples and len(WORKLOADS[args.workload]) <= 5:
        raise SystemExit("--mode samples needs a path-tracing workload")
    if world > 1 and not tiles and not samples:
        orbit_camera(app, rank, world)
    scene = app.prepare_meshinfo()            # P1: scene extraction (host)
    if WORKLOADS[args.workload][0] == "field_analytic":
        scene = rt.analytic_sphere_field(WORKLOADS[args.workload][1], WORKLOADS[args.workload][2])
    ctx = rt.Context(local_rank)
    # ONE stream for the library's kernels, torch's copies and the collectives' hand-over: a torch side stream (non-zero
    # handle; the default stream's handle is 0, which rt_set_stream reads as "the context's own stream"), made current
    # for the whole run so that render -> pack -> gather -> unpack are ordered without host synchronisation
    stream = torch.cuda.Stream(device=local_rank)
    torch.cuda.set_stream(stream)
    gloo_rehearsal = distributed and os.environ.get("BENCH_BACKEND", "nccl") != "nccl"

    def frames_in_flight():
        """The gloo rehearsal stages through the host, and the sample-slice reduction reads the accumulation buffer on torch's
        stream: one frame at a time there.  Path-traced frames overlap two at a time whatever n > 1 says (rt_render)."""
        if gloo_rehearsal or samples:
            return 1
        return args.in_flight
    n_in_flight = frames_in_flight()
    overlapped = n_in_flight > 1

    def configure(n):
        """n > 1: the context's own streams, one per frame set (consumers order against rt_get_stream).  1: torch's side stream."""
        if n > 1:
            ctx.set_stream(0)
            ctx.set_frames_in_flight(n)
        else:
            ctx.set_frames_in_flight(1)
            ctx.set_stream(stream.cuda_stream)
    configure(n_in_flight)
    ctx.upload_scene(scene)                   # upload + BVH build, outside the timed region
    m, p = app.camera_view()
    ctx.set_view(m, p, W, H)
    info = ctx.bvh_info()

    wl = WORKLOADS[args.workload]
    spp, segs = (wl[5], wl[6]) if len(wl) > 5 else (0, 0)
    flags = (rt.FLAG_PATH_TRACE if spp else rt.FLAG_SHADOW) | args.extra_flags
    shard = dict(shard_rank=rank, shard_count=world, tile_size=64) if tiles else {}
    if spp:   # one frame sharded by tiles: every rank the same RNG stream (the frame of one GPU); own cameras: own streams
        shard.update(spp=spp, max_segments=segs, seed=1, stream_id=0 if tiles else rank)
    if samples:   # one frame, rank r renders samples [r*spp, (r+1)*spp) of the same RNG stream
        shard.update(sample_first=rank * spp, stream_id=0)

    # instrumented launch (same view): counted node visits / triangle tests for the roofline
    ctx.render(flags=flags | rt.FLAG_COUNTERS, **shard)
    cnt = ctx.counters()
    rays_step = cnt.rays_primary + cnt.rays_shadow + cnt.rays_bounce
    hits = cnt.rays_shadow
    if spp:   # path tracing: 44 B of path state written + read per continued segment, 16 B accumulation RMW per path end
        alg_bytes = (cnt.node_visits * info.node_bytes + cnt.tri_tests * info.tri_bytes + cnt.mesh_aabb_tests * 32 +
                     cnt.rays_bounce * (88 + 80) + cnt.rays_primary * 32)
    else:
        alg_bytes = (cnt.node_visits * info.node_bytes + cnt.tri_tests * info.tri_bytes + cnt.mesh_aabb_tests * 32 +
                     cnt.rays_primary * 24 + hits * 80)

    gather = None
    if tiles:
        from ray_tracing_amd import multi_gpu
        gloo = dist.get_backend() == "gloo"
        # RCCL: pack / unpack are HIP kernels on the context's stream, the packed messages torch tensors on this device.
        # gloo (one-GPU rehearsal only): gloo moves host tensors, so the frame is read back and the host restatement of
        # the same packed layout is used.
        gather = multi_gpu.TileGather(W, H, 64, world, rank, ctx=None if gloo else ctx, device="cpu" if gloo else f"cuda:{local_rank}",
                                      in_flight=n_in_flight)

    acc_t = None
    if samples:
        d_rad_s, _, d_acc = ctx.device_buffers()

        class _PtrA:
            __cuda_array_interface__ = dict(shape=(H * W * 4,), typestr="<f4", data=(d_acc, False), version=2)

        class _PtrR:
            __cuda_array_interface__ = dict(shape=(H * W * 4,), typestr="<f4", data=(d_rad_s, False), version=2)
        acc_t = torch.as_tensor(_PtrA(), device=f"cuda:{local_rank}")
        rad_s = torch.as_tensor(_PtrR(), device=f"cuda:{local_rank}")
        from ray_tracing_amd import multi_gpu
        gloo_s = dist.get_backend() == "gloo"
        # direct reduce-scatter over the point-to-point links + gather of the resolved slices to rank 0
        reducer = multi_gpu.SampleReduce(H * W * 4, world, rank, "cpu" if gloo_s else f"cuda:{local_rank}", gather_to=0)

    def step(f):
        ctx.render(flags=f, **shard)
        if samples:
            # running sums of all ranks' sample slices -> radiance = sum / total samples in rank 0's radiance buffer
            res = reducer.combine(acc_t.cpu() if gloo_s else acc_t, spp * world)   # gloo (rehearsal only) moves host tensors
            if rank == 0:
                rad_s.copy_(res)
        if tiles:
            # every rank sends its own tiles, packed, straight to rank 0: one xGMI link per peer, 1/N of a frame each;
            # ordered on the stream of the frame just rendered, so the next frame's render overlaps with it
            if gloo:
                gather.combine(ctx.read_radiance())
            else:
                gather.combine_in_flight()

    def quiesce():
        ctx.synchronize()                          # every frame set's stream
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n_frames, steps, warmup):
        """steps frames with n_frames in flight, bracketed by barrier + synchronize; (seconds, max over ranks; kernel ms per launch)."""
        configure(n_frames)
        for _ in range(warmup):
            step(flags)
        quiesce()
        ctx.counters()                             # start the HIP-event sums afresh
        t0 = time.perf_counter()
        for _ in range(steps):
            step(flags | rt.FLAG_TIME_KERNELS)
        quiesce()
        dt_ = time.perf_counter() - t0
        kc = ctx.counters()                        # sums the per-launch HIP event pairs
        k_ms = kc.kernel_ms / (steps if spp else max(kc.kernel_launches, 1))   # per launch; per frame when path tracing
        if distributed:
            tt = torch.tensor([dt_], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_ = float(tt.item())
        return dt_, k_ms

    dt, kernel_ms = timed(n_in_flight, args.steps, args.warmup)
    # the same frames one at a time (every rank takes part: the collectives must match): what round 1's `value` was, and the
    # unoverlapped launch duration the counter-based figures refer to
    one_at_a_time = timed(1, args.steps, min(args.warmup, 3)) if overlapped else None
    if overlapped:
        configure(n_in_flight)

    total_rays = float(rays_step)
    if distributed:
        rr = torch.tensor([total_rays], dtype=torch.float64, device="cuda")
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
        total_rays = float(rr.item())

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = total_rays * args.steps / dt / 1e6
        # The roofline's launch duration is the kernel ALONE on the GPU (the one-frame-at-a-time pass, HIP events around
        # every launch): with frames in flight a launch's begin-to-end time includes the time it shares the chip with its
        # neighbours (n frames in flight: ~n x longer) and says nothing about the kernel.  Both are printed.
        kernel_ms_in_flight = kernel_ms
        if one_at_a_time:
            kernel_ms = one_at_a_time[1]
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        # which kernel the flags select (rtk_launch_primary)
        wide = bool(info.n_wide_nodes) and not (flags & rt.FLAG_BINARY_TREE)
        if spp:
            kernel = "k_path_start+k_path_bounce (whole frame)"
        elif flags & rt.FLAG_BRUTE_FORCE or flags & rt.FLAG_STATIC_GRID:
            kernel = "k_primary<false, false>"
        elif flags & rt.FLAG_WAVE_TRAVERSAL:
            kernel = "k_primary<false, true>"
        else:
            kernel = f"k_stream<false, {'true' if wide else 'false'}, {'true' if os.environ.get('RT_TUNE_POOL_DEP', '0') not in ('', '0') else 'false'}>"
        # Counters of the committed PMC passes (profiles/pmc_traffic.json; rocprofv3 --pmc cannot run inside this process):
        # quoted only for the workload AND the very kernel binary they were collected on -- the file records the kernel's
        # symbol, registers, spills and the hash of the device code object (tools/kernel_resources.py); a library built
        # from other kernel sources makes them stale, and stale counters are not printed.
        traffic, valu, hbm_frac, pmc = None, None, None, {"measured_in_this_run": False}
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import kernel_resources
            tj = json.load(open(tpath))
            now = kernel_resources.fingerprint(rt.LIB_PATH, tj.get("fingerprint_kernels", "k_stream"))
            pmc["source"] = tj.get("pmc_source")
            if tj.get("workload") != args.workload or tj.get("kernel") != kernel or world != 1:
                pmc["quoted"] = False
                pmc["why"] = "committed passes are for another workload / kernel / GPU count"
            elif tj.get("fingerprint") != now:
                pmc["quoted"] = False
                pmc["stale"] = True
                pmc["why"] = "librt_amd.so's device code differs from the one the passes were collected on: rerun profiles/pmc_passes.sh"
            else:
                pmc["quoted"] = True
                traffic = tj.get("hbm_bytes_per_launch")
                valu = {"busy": tj["valu_busy"], "lanes_active": tj["valu_lanes_active"],
                        "frac": round(tj["valu_busy"] * tj["valu_lanes_active"], 4),
                        "busy_is": "VALU wave-instructions per second and SIMD over the issue rate profiles/calib/valu_mask reaches (one per ~2.5 cycles)",
                        "wave_time": tj.get("wave_time"), "waves_per_simd_avg": tj.get("waves_per_simd_avg")}
                # counters are collected with launches serialised (rocprofv3 --pmc always does): HBM bytes per launch over
                # the UNOVERLAPPED launch duration
                if traffic and kernel_ms > 0:
                    hbm_frac = round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        except Exception as ex:
            pmc["quoted"] = False
            pmc["why"] = f"no usable profiles/pmc_traffic.json: {str(ex)[:80]}"
        # The bound: SURVEY 8d names HBM as the nominal roofline and `achieved`/`frac` keep that definition (ALGORITHMIC
        # bytes -- every lane's node / record fetch, most of them served by L1 / L2 / Infinity Cache -- over kernel time),
        # but the counters decide the label: a pipe that is >= 0.75 busy names the bound; when neither the VALU nor HBM is,
        # the launch is bound by the LATENCY of its dependent gathers (waves parked at s_waitcnt, wave_time.waiting_for_memory).
        if valu is not None and hbm_frac is not None:
            bound = "valu" if valu["busy"] >= 0.75 and valu["busy"] > hbm_frac else ("hbm" if hbm_frac >= 0.75 else "latency")
        else:
            bound = "hbm (nominal; no current counters)"
        out = {
            "metric": "Mrays/s", "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong" if tiles else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "resolution": f"{W}x{H}", "rays": (f"path tracing, {spp} spp, <= {segs} segments" if spp else "primary+shadow, 1 spp"),
                       "triangles": int(info.n_triangles), "bvh_nodes": int(info.n_nodes),
                       "rays_per_step_rank0": int(rays_step),
                       "primary_hit_fraction": round(hits / max(cnt.rays_primary, 1), 4), "camera_radii": CAMERA_RADII, "frames_in_flight": n_in_flight,
                       "backend": (dist.get_backend() if distributed else None), "rccl_ranks": (dist.get_world_size() if distributed and dist.get_backend() == "nccl" else None),
                       "parallelism": ("single-gpu" if not distributed else ("tiles64-roundrobin+host-pack+gather-to-rank0 (gloo rehearsal)" if gloo else "tiles64-roundrobin+hip-pack+one-gather-to-rank0+hip-unpack") if tiles else
                                       (f"sample-slices{world}+direct-reduce-scatter+gather-to-rank0" if samples else f"frames-dp{world}"))},
            "roofline": {"bound": bound, "bound_nominal": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "achieved_is": "ALGORITHMIC bytes per launch / kernel time (SURVEY 8d); caches serve most of them, so it can exceed what HBM moves",
                         "hbm_frac": hbm_frac, "valu": valu, "pmc": pmc,
                         "kernel": kernel, "kernel_ms": round(kernel_ms, 4),
                         **({"kernel_ms_is": "launch duration one frame at a time (HIP events around every launch of the pass after the timed region)",
                             "kernel_ms_in_flight": round(kernel_ms_in_flight, 4),
                             "kernel_ms_in_flight_is": f"the same events in the timed region, {n_in_flight} frames in flight: begin-to-end time of launches that share the GPU",
                             **({"kernel_in_flight": f"k_stream8<false, {'true' if wide else 'false'}>: the same kernel built for 8 waves per SIMD, launched with three or more frames in flight"}
                                if kernel.startswith("k_stream<") and kernel.endswith("false>") and n_in_flight >= 3 and not os.environ.get("RT_TUNE_STREAM_WAVES") else {})} if overlapped else {}),
                         "algorithmic_bytes_per_launch": int(alg_bytes),
                         "node_visits": int(cnt.node_visits), "tri_tests": int(cnt.tri_tests),
                         "bytes_per_ray": round(alg_bytes / max(rays_step, 1), 1)},
        }
        if one_at_a_time:
            dt1, k1 = one_at_a_time
            a1 = alg_bytes / (k1 * 1e-3) / 1e9 if k1 > 0 else 0.0
            out["one_frame_at_a_time"] = {"value": round(total_rays * args.steps / dt1 / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(dt1 / args.steps * 1e3, 4),
                                          "steps": args.steps, "kernel_ms": round(k1, 4), "achieved": round(a1, 2), "frac": round(a1 / HBM_PEAK_GBS, 5),
                                          "note": "the same frames with --in-flight 1 (round 1's definition of value), measured after the timed region"}
        if not args.no_pipelined and not spp and world == 1 and not overlapped:
            # Outside the contract's timed region and never `value`: the same frames with TWO in flight inside this one
            # context (rt_set_frames_in_flight: a second set of frame buffers and a second stream, taken in turn).  The waves
            # of one frame fill the chip while the other frame's last regions drain ("the end of the frame").
            try:
                ctx.set_stream(0)                      # the context's own streams: one per frame set
                ctx.set_frames_in_flight(2)
                for _ in range(4):
                    ctx.render(flags=flags)
                ctx.synchronize()
                n2 = max(2, args.steps - args.steps % 2)
                t0 = time.perf_counter()
                for _ in range(n2):
                    ctx.render(flags=flags)
                ctx.synchronize()
                dt2 = time.perf_counter() - t0
                prim_b = ctx.read_prim_ids()
                ctx.set_frames_in_flight(1)
                ctx.render(flags=flags)
                same = bool((ctx.read_prim_ids() == prim_b).all())
                ctx.set_stream(stream.cuda_stream)
                out["pipelined"] = {"frames_in_flight": 2, "value": round(rays_step * n2 / dt2 / 1e6, 2), "unit": "Mrays/s",
                                    "ms_per_step": round(dt2 / n2 * 1e3, 4), "steps": n2, "frames_identical": same,
                                    "note": "ONE context, rt_set_frames_in_flight(2): two sets of frame buffers and streams taken in turn by rt_render; not the contract's value"}
            except Exception as ex:                                              # never fail the bench line for it
                out["pipelined"] = {"error": str(ex)[:120]}
        if not args.no_north_star and not spp and world == 1 and args.workload == "sphere_field_10k_1080p_primary_shadow":
            # north_star's own target scene (1 M-triangle icosphere soup, 1080p, primary + shadow, ">= 1 Grays/s"): a
            # driver-timed figure for it, measured after the timed region; never `value`.  Same frames in flight as the
            # timed region, and one frame at a time beside it.
            try:
                app_s, Ws, Hs = build_app(rt, "soup_1m_1080p_primary_shadow")
                ctx_s = rt.Context(local_rank)
                ctx_s.upload_scene(app_s.prepare_meshinfo())
                ms_, ps_ = app_s.camera_view()
                ctx_s.set_view(ms_, ps_, Ws, Hs)
                ctx_s.render(flags=rt.FLAG_SHADOW | rt.FLAG_COUNTERS)
                cs = ctx_s.counters()
                rays_s = cs.rays_primary + cs.rays_shadow

                def soup_ms(n_frames):
                    ctx_s.set_frames_in_flight(n_frames)
                    for _ in range(5):
                        ctx_s.render(flags=rt.FLAG_SHADOW)
                    ctx_s.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(args.steps):
                        ctx_s.render(flags=rt.FLAG_SHADOW)
                    ctx_s.synchronize()
                    return (time.perf_counter() - t0) / args.steps * 1e3
                ms_n = soup_ms(n_in_flight)
                out["north_star_scene"] = {"workload": "soup_1m_1080p_primary_shadow", "value": round(rays_s / ms_n / 1e3, 2),
                                           "unit": "Mrays/s", "ms_per_step": round(ms_n, 4), "steps": args.steps, "frames_in_flight": n_in_flight,
                                           "rays_per_step": int(rays_s), "triangles": int(ctx_s.bvh_info().n_triangles),
                                           "target": ">= 1000 Mrays/s primary+shadow at 1080p on a 1M-triangle scene (BASELINE.json north_star)"}
                if overlapped:
                    ms_1 = soup_ms(1)
                    out["north_star_scene"]["one_frame_at_a_time"] = {"value": round(rays_s / ms_1 / 1e3, 2), "ms_per_step": round(ms_1, 4)}
                ctx_s.close(); app_s.close()
            except Exception as ex:                                              # never fail the bench line for it
                out["north_star_scene"] = {"error": str(ex)[:120]}
        if not args.no_cpu_baseline and not spp and world == 1:
            import oracle_py as orc
            view = orc.make_view(m, p, W, H)
            cb, pix, rgba_o, prim_o = cpu_baseline(orc, scene, view, W, H, flags, args.cpu_seconds)
            out["cpu_baseline"] = cb
            if not tiles:
                # parity gate on the same sample
                prim_all, rgba_all = ctx.read_prim_ids(), ctx.read_radiance()
                prim_g = prim_all.ravel()[pix]
                rgba_g = rgba_all.reshape(-1, 4)[pix]
                d = rgba_g.astype(np.float64) - rgba_o.astype(np.float64)
                out["parity"] = {"prim_id_agreement": float((prim_g == prim_o).mean()),
                                 "radiance_rmse": float(np.sqrt(np.mean(d * d))), "pixels": int(len(pix))}
                # "additionally time the CPU BVH traversal on full frames" -- the oracle's scalar walk of the
                # very tree the device holds (read back through the C ABI), whole frame: a second CPU figure, a
                # full-frame parity check, and the cross-check of the counted visits behind roofline.achieved
                try:
                    nodes_h, recs_h = ctx.read_bvh()
                    wide_h = ctx.read_bvh_wide() if info.n_wide_nodes and not (flags & rt.FLAG_BINARY_TREE) else None
                    tree = orc.Tree(nodes_h, recs_h, wide_h)
                    osc = orc.Scene(scene.triangles, scene.meshes, scene.vertices, scene.materials,
                                    getattr(scene, "spheres", None), getattr(scene, "sphere_materials", None))
                    t0 = time.perf_counter()
                    rgba_w, prim_w, _, rays_w, (nv, nt, nc) = orc.render_primary_tree(osc, tree, view, flags & 1, threads=cb["cores"])
                    dt_w = time.perf_counter() - t0
                    dw = rgba_all.astype(np.float64) - rgba_w.astype(np.float64)
                    cb["tree_walk"] = {"value": round(rays_w / dt_w / 1e6, 3), "unit": "Mrays/s", "cores": cb["cores"],
                                       "sample": f"full frame, {rays_w} rays, oracle walk (orc_hit_tree) of the device's own tree, {dt_w:.1f} s"}
                    out["parity"]["full_frame"] = {
                        "prim_id_agreement": float((prim_all == prim_w).mean()), "radiance_rmse": float(np.sqrt(np.mean(dw * dw))),
                        "pixels": int(prim_all.size),
                        "counters_equal_oracle_walk": bool((nv, nt, nc) == (cnt.node_visits, cnt.tri_tests, cnt.mesh_aabb_tests))}
                    del nodes_h, recs_h, wide_h, tree, rgba_w, prim_w, dw
                except Exception as ex:                                          # never fail the bench line for it
                    cb["tree_walk"] = {"error": str(ex)[:120]}
        if not args.no_cpu_baseline and spp and world == 1 and not args.extra_flags:
            # path-tracing workloads: the reference-shaped loop cannot reach these sizes, so the CPU figure is the oracle's
            # scalar walk of the device's own tree (oracle/rt_oracle.c orc_render_path_tree), the whole frame, which is
            # also a whole-frame parity check of the running sums
            try:
                import oracle_py as orc
                view = orc.make_view(m, p, W, H)
                threads = max(1, min(16, os.cpu_count() or 1))
                tree = orc.Tree(*ctx.read_bvh(), ctx.read_bvh_wide() if info.n_wide_nodes else None)
                osc = orc.Scene(scene.triangles, scene.meshes, scene.vertices, scene.materials)
                ctx.render(flags=flags, **shard)                      # one frame of exactly these samples, no accumulation carried over
                acc_g = ctx.read_accumulation()
                t0 = time.perf_counter()
                acc_w, segs_w, _ = orc.render_path_tree(osc, tree, view, spp=spp, max_segments=segs, seed=1, stream=rank, threads=threads)
                dt_w = time.perf_counter() - t0
                out["cpu_baseline"] = {"value": round(segs_w / dt_w / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
                                       "sample": f"full frame, {segs_w} ray segments, oracle walk of the device's own tree (orc_render_path_tree), {dt_w:.1f} s"}
                dd = acc_g.astype(np.float64) - acc_w.astype(np.float64)
                out["parity"] = {"accumulation_identical": bool((acc_g.view(np.uint32) == acc_w.view(np.uint32)).all()),
                                 "radiance_rmse": float(np.sqrt(np.mean(dd * dd))) / spp, "pixels": int(W * H),
                                 "segments_equal": bool(segs_w == rays_step)}
                del tree, acc_w, acc_g, dd
            except Exception as ex:                                              # never fail the bench line for it
                out["cpu_baseline"] = {"error": str(ex)[:120]}
        # second denominator: what a plain device-to-device copy reaches on THIS box, read + write bytes
        try:
            src = torch.empty(1 << 28, dtype=torch.float32, device="cuda")       # 1 GiB: beyond the 256 MiB Infinity Cache
            dst = torch.empty_like(src)
            dst.copy_(src)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                dst.copy_(src)
            e1.record()
            torch.cuda.synchronize()
            copy_gbs = 5 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            out["roofline"]["stream_copy_GBps"] = round(copy_gbs, 1)
            out["roofline"]["frac_of_stream_copy"] = round(achieved / copy_gbs, 5)
            del src, dst
        except Exception as ex:                                                  # never fail the bench line for it
            out["roofline"]["stream_copy_GBps"] = None
            out["roofline"]["stream_copy_error"] = str(ex)[:80]
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    ctx.close()
    if distributed:
        dist.barrier()                 # rank 0's CPU baseline is done before anyone tears the group down
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
