#!/usr/bin/env python3
"""bench.py -- Mrays/s of the per-pixel hot path on MI355X (BASELINE.json metric).

Workload at N=1 (BASELINE.json configs[1]): synthetic 10k-sphere field,
reference-faithful tessellation (ico(5): 7.2 M triangles + ground plane), 1920x1080, one primary
ray per pixel + one shadow ray per hit.  The camera MOVES: frame k renders view k % 8 of an orbit around the scene (45
degrees apart, same distance and height; view 0 is the camera of rounds 1 and 2), so the frames in flight are different
frames -- four copies of one frame would share their whole working set in the caches.  A STEP = one pass of the hot path over
that batch of input: the orbit's 8 views, i.e. 8 frames through rt_render (C ABI); `ms_per_frame` is printed beside
`ms_per_step`.  The same-view figure is printed too (`same_view`).
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
N_VIEWS = 8             # the orbit of the timed region
N_SIMDS = 1024          # 256 CUs x 4 SIMDs

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


def orbit_views(app, scene, n):
    """n views on a horizontal circle around the scene: view 0 is the app's camera as it stands (the camera of rounds 1 and 2),
    view k the same eye turned by k * 360 / n degrees about the vertical through the scene's centre (the centre of the bounding
    box of its vertices), looking at that centre.  Same distance, same height, different sides of the scene."""
    m0, p0 = app.camera_view()
    views = [(np.array(m0, np.float32).copy(), np.array(p0, np.float32).copy())]
    if n <= 1:
        return views
    pos = scene.vertices["pos"]
    if len(pos):
        centre = (pos.min(axis=0).astype(np.float64) + pos.max(axis=0).astype(np.float64)) * 0.5
    else:
        centre = np.zeros(3)
    rel = np.array(p0, np.float64) - centre
    for k in range(1, n):
        ang = 2.0 * math.pi * k / n
        c, s = math.cos(ang), math.sin(ang)
        rel2 = np.array([c * rel[0] + s * rel[2], rel[1], -s * rel[0] + c * rel[2]])
        app.camera_look_at((centre + rel2).astype(np.float32), centre.astype(np.float32), (0, 1, 0))
        m, p = app.camera_view()
        views.append((np.array(m, np.float32).copy(), np.array(p, np.float32).copy()))
    app.camera_look_at(np.array(p0, np.float32), centre.astype(np.float32), (0, 1, 0))    # leave the app near where it was
    return views


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


def cpu_baseline(orc, scene, view, W, H, flags, target_seconds=10.0):
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
    """`bench.py --gpus N` without a launcher: start N rank processes (one per GPU) from THIS process, which has not
    touched a GPU and never will; rank 0's stdout is the bench line.  The ranks are polled: the first one that dies takes the
    others down with it and its status is handed on at once (a rank that dies at init would otherwise leave its siblings in a
    collective until RCCL's timeout, and the run would be killed at its limit with nothing written: VERDICT r2)."""
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location("rt_multi_gpu_launch", os.path.join(ROOT, "ray-tracing_amd", "multi_gpu.py"))
    mod = importlib.util.module_from_spec(spec)      # the module alone: importing the package would load the HIP library here
    spec.loader.exec_module(mod)
    return mod.launch_ranks(n, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:])


def alg_bytes_of(cnt, info, spp):
    """SURVEY 8d's algorithmic bytes of one instrumented launch (node visits x node record + prim tests x leaf record + ...)."""
    hits = cnt.rays_shadow
    if spp:   # path tracing: 44 B of path state written + read per continued segment, 16 B accumulation RMW per path end
        return (cnt.node_visits * info.node_bytes + cnt.tri_tests * info.tri_bytes + cnt.mesh_aabb_tests * 32 +
                cnt.rays_bounce * (88 + 80) + cnt.rays_primary * 32)
    return (cnt.node_visits * info.node_bytes + cnt.tri_tests * info.tri_bytes + cnt.mesh_aabb_tests * 32 +
            cnt.rays_primary * 24 + hits * 80)


def lane_gathers_of(cnt):
    """64-B gather equivalents of one instrumented launch: a node visit is 4 x dwordx4 per lane (64 B), a record test 3 (48 B),
    a per-mesh cull 2 (32 B).  The unit rt_measure_gather_rates' ceilings are in."""
    return cnt.node_visits + 0.75 * cnt.tri_tests + 0.5 * cnt.mesh_aabb_tests


def gather_block(rates, lane_gathers, ms_timed, ms_alone=None, beyond_l2_bytes=None):
    """The roofline in the kernel's own currency: lane-level 64-B gathers per second against what this
    GPU's memory system delivers to exactly that access pattern, measured in this run (rt_measure_gather_rates)."""
    if not rates or ms_timed <= 0:
        return None
    ach = lane_gathers / (ms_timed * 1e-3) / 1e9
    top = rates["l2_2MiB"]["independent"]
    out = {"lane_gathers_per_frame": int(lane_gathers),
           "lane_gathers_are": "64-B gather equivalents: node visits (4 x dwordx4 per lane) + 3/4 x record tests + 1/2 x mesh culls, counted",
           "achieved_Ggather_s": round(ach, 1), "ceiling_Ggather_s": round(top, 1),
           "ceiling_is": "independent random 64-B gathers from a buffer that every XCD's L2 holds, 8 waves per SIMD, every lane of a wave a DIFFERENT line: what the CUs' vector-memory "
                         "front end passes to fully divergent waves when every gather hits a cache.  Lanes of a wave that fetch the same node cost it one line, so partly coherent waves "
                         "(primary rays near the top of the tree) can exceed it: frac is an upper bound on the front end's load, TA_BUSY (in_flight.utilisation.vmem_front_end) the hardware's own figure",
           "frac": round(ach / top, 4),
           "ceilings_by_level_Ggather_s": {k: round(v["independent"], 1) for k, v in rates.items()},
           "dependent_chain_Ggather_s": {k: round(v["dependent"], 1) for k, v in rates.items()},
           "measured_in_this_run": True}
    if ms_alone:
        out["achieved_one_frame_at_a_time_Ggather_s"] = round(lane_gathers / (ms_alone * 1e-3) / 1e9, 1)
    if beyond_l2_bytes:
        b = beyond_l2_bytes / 64.0 / (ms_timed * 1e-3) / 1e9
        out["beyond_l2"] = {"Ggather_s": round(b, 1), "ceiling_Ggather_s": round(rates["hbm_1GiB"]["independent"], 1),
                            "frac": round(b / rates["hbm_1GiB"]["independent"], 4),
                            "is": "HBM + Infinity Cache bytes per frame (FETCH_SIZE + WRITE_SIZE of the committed passes) / 64 B over the frame time, against random 64-B gathers from 1 GiB"}
    return out


def load_pmc(rt):
    """profiles/pmc_traffic.json, and whether the library that runs is the one its passes were collected on."""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import kernel_resources
        tj = json.load(open(tpath))
        now = kernel_resources.fingerprint(rt.LIB_PATH, tj.get("fingerprint_kernels", "k_stream"))
        fp_path = None
        if "fingerprint_path" in tj:
            fp_path = kernel_resources.fingerprint(rt.LIB_PATH, tj.get("fingerprint_path_kernels", "k_path"))
        return tj, tj.get("fingerprint") == now, (fp_path is not None and tj.get("fingerprint_path") == fp_path), None
    except Exception as ex:
        return None, False, False, f"no usable profiles/pmc_traffic.json: {str(ex)[:80]}"


def side_workload(rt, orc, name, local_rank, steps, in_flight, pmc_section=None, check=True, rates=None):
    """A driver-timed figure for another BASELINE config, measured after the timed region on a context of its own; never
    `value`.  Frames in flight as asked (path-traced frames overlap two at a time whatever n > 1 says) and one at a time; the
    one-at-a-time launch time gives SURVEY 8d's algorithmic-bytes fraction; path workloads are compared with the oracle's walk
    of the device's own tree over the whole frame (running sums identical, segment counts equal)."""
    wl = WORKLOADS[name]
    spp, segs = (wl[5], wl[6]) if len(wl) > 5 else (0, 0)
    app, W, H = build_app(rt, name)
    scene = app.prepare_meshinfo()
    ctx = rt.Context(local_rank)
    try:
        ctx.upload_scene(scene)
        m, p = app.camera_view()
        ctx.set_view(m, p, W, H)
        info = ctx.bvh_info()
        flags = rt.FLAG_PATH_TRACE if spp else rt.FLAG_SHADOW
        kw = dict(spp=spp, max_segments=segs, seed=1, stream_id=0) if spp else {}
        ctx.render(flags=flags | rt.FLAG_COUNTERS, **kw)
        cnt = ctx.counters()
        rays = cnt.rays_primary + cnt.rays_shadow + cnt.rays_bounce
        alg = alg_bytes_of(cnt, info, spp)

        def run(n):
            ctx.set_frames_in_flight(n)
            for _ in range(3):
                ctx.render(flags=flags, **kw)
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                ctx.render(flags=flags, **kw)
            ctx.synchronize()
            return (time.perf_counter() - t0) / steps * 1e3
        ms_n = run(in_flight)
        ms_1 = run(1)
        out = {"workload": name, "value": round(rays / ms_n / 1e3, 2), "unit": "Mrays/s", "ms_per_frame": round(ms_n, 4), "frames": steps,
               "frames_in_flight": (min(in_flight, 2) if spp else in_flight), "rays_per_step": int(rays), "triangles": int(info.n_triangles),
               "resolution": f"{W}x{H}", "rays": (f"path tracing, {spp} spp, <= {segs} segments" if spp else "primary+shadow, 1 spp"),
               "one_frame_at_a_time": {"value": round(rays / ms_1 / 1e3, 2), "ms_per_frame": round(ms_1, 4)},
               "bytes_per_ray": round(alg / max(rays, 1), 1),
               "frac": round(alg / (ms_1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               "frac_is": "SURVEY 8d's algorithmic bytes per frame / frame time one at a time / 8 TB/s (the caches serve most of those bytes)",
               "frac_at_timed_rate": round(alg / (ms_n * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        g = gather_block(rates, lane_gathers_of(cnt), ms_n, ms_1)
        if g:
            for k in ("ceilings_by_level_Ggather_s", "dependent_chain_Ggather_s", "lane_gathers_are", "ceiling_is"):
                g.pop(k, None)
            out["gather"] = g
        if pmc_section is not None:
            out["counters"] = pmc_section
            ks = pmc_section.get("k_path_start<false>") if isinstance(pmc_section, dict) else None
            kb = pmc_section.get("k_path_bounce<false>") if isinstance(pmc_section, dict) else None
            if ks and rates and spp:
                # the whole frame's traffic beyond the L2 -- one k_path_start per batch of samples, segs - 1 bounce launches (the
                # committed passes give bytes per launch, averaged over a frame's launches) -- over the frame time as timed here
                n_start = 1 if spp in (1, 4, 16) else spp
                fb = ks["hbm_bytes_per_launch"] * n_start + (kb["hbm_bytes_per_launch"] * (segs - 1) * n_start if kb else 0)
                r = fb / 64.0 / (ms_n * 1e-3) / 1e9
                out["frame_beyond_l2"] = {"bytes_per_frame": int(fb), "Ggather_s": round(r, 1), "TBps": round(r * 64 / 1e3, 2),
                                          "ceiling_Ggather_s": round(rates["hbm_1GiB"]["independent"], 1), "frac": round(r / rates["hbm_1GiB"]["independent"], 4),
                                          "is": "HBM + Infinity Cache bytes of a frame's launches (committed counter passes) / 64 B over the frame time measured in THIS run with frames in flight, "
                                                "against this GPU's rate of random 64-B gathers from 1 GiB (rt_measure_gather_rates)"}
            if kb and rates and kb.get("avg_duration_us_under_pmc"):
                r = kb["hbm_bytes_per_launch"] / 64.0 / (kb["avg_duration_us_under_pmc"] * 1e-6) / 1e9
                out["bounce_launch_beyond_l2"] = {"Ggather_s": round(r, 1), "TBps": round(r * 64 / 1e3, 2),
                                                  "ceiling_Ggather_s": round(rates["hbm_1GiB"]["independent"], 1),
                                                  "frac": round(r / rates["hbm_1GiB"]["independent"], 4),
                                                  "is": "k_path_bounce alone: HBM + Infinity Cache bytes per launch / 64 B / launch duration, against random 64-B gathers from 1 GiB (from 4 GiB: "
                                                        f"{round(rates['hbm_4GiB']['independent'], 1)})"}
        if check and spp:
            tree = orc.Tree(*ctx.read_bvh(), ctx.read_bvh_wide() if info.n_wide_nodes else None)
            osc = orc.Scene(scene.triangles, scene.meshes, scene.vertices, scene.materials)
            ctx.render(flags=flags, **kw)
            acc_g = ctx.read_accumulation()
            threads = max(1, min(16, os.cpu_count() or 1))
            t0 = time.perf_counter()
            acc_w, segs_w, _ = orc.render_path_tree(osc, tree, orc.make_view(m, p, W, H), spp=spp, max_segments=segs, seed=1, stream=0, threads=threads)
            dt_w = time.perf_counter() - t0
            out["parity"] = {"accumulation_identical": bool((acc_g.view(np.uint32) == acc_w.view(np.uint32)).all()),
                             "segments_equal": bool(segs_w == rays), "pixels": int(W * H)}
            out["cpu_tree_walk"] = {"value": round(segs_w / dt_w / 1e6, 3), "unit": "Mrays/s", "cores": threads}
            del tree, acc_w, acc_g
        return out
    finally:
        ctx.close()
        app.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=25)       # a step = one pass over the orbit's views (8 frames): 200 frames, 0.12 s -- the four-deep pipeline's fill and drain (inside the timed region) weigh 1 %
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="sphere_field_10k_1080p_primary_shadow", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="tiles", choices=["frames", "tiles", "samples"], help="N > 1 only; tiles = north_star's partitioning")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipelined", action="store_true", help="(kept for old command lines; the two-frames figure is gone: --in-flight measures it)")
    ap.add_argument("--no-north-star", action="store_true", help="skip the extra workloads measured after the timed region (N=1, default workload): "
                    "north_star's 1M-triangle soup, configs[2] (4 spp) and the 10M-triangle path-tracing scene")
    ap.add_argument("--no-same-view", action="store_true", help="skip the same-view pass (every frame in flight the same camera, rounds 1-2's bench)")
    ap.add_argument("--views", type=int, default=N_VIEWS, help="views on the orbit the steps cycle through (1: one camera)")
    ap.add_argument("--in-flight", type=int, default=4, choices=[1, 2, 3, 4],
                    help="primary/shadow workloads: frames in flight inside the context during the timed region (rt_set_frames_in_flight: "
                         "the render node submits frame after frame without waiting, as the reference's does; in tiles mode frame k's "
                         "pack -> gather -> unpack overlaps with the next frames' renders).  The same at every N.  1: one frame at a time "
                         "(always measured too, after the timed region: one_frame_at_a_time, and the roofline's launch duration)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
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
    frames_mode = world > 1 and not tiles and not samples
    if frames_mode:
        orbit_camera(app, rank, world)
    scene = app.prepare_meshinfo()            # P1: scene extraction (host)
    if WORKLOADS[args.workload][0] == "field_analytic":
        scene = rt.analytic_sphere_field(WORKLOADS[args.workload][1], WORKLOADS[args.workload][2])
    # the cameras of the timed region: an orbit every rank steps through in the same order (frames mode: every rank keeps
    # its own one camera)
    views = orbit_views(app, scene, 1 if frames_mode else max(1, args.views))
    n_views = len(views)
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
    ctx.set_view(views[0][0], views[0][1], W, H)
    info = ctx.bvh_info()

    wl = WORKLOADS[args.workload]
    spp, segs = (wl[5], wl[6]) if len(wl) > 5 else (0, 0)
    flags = (rt.FLAG_PATH_TRACE if spp else rt.FLAG_SHADOW) | args.extra_flags
    shard = dict(shard_rank=rank, shard_count=world, tile_size=64) if tiles else {}
    if spp:   # one frame sharded by tiles: every rank the same RNG stream (the frame of one GPU); own cameras: own streams
        shard.update(spp=spp, max_segments=segs, seed=1, stream_id=0 if tiles else rank)
    if samples:   # one frame, rank r renders samples [r*spp, (r+1)*spp) of the same RNG stream
        shard.update(sample_first=rank * spp, stream_id=0)

    # instrumented launches, one per view: counted rays, node visits and record tests (what `value` counts and what the
    # roofline's algorithmic bytes are made of)
    cnts = []
    for (vm, vp) in views:
        ctx.set_view(vm, vp, W, H)
        ctx.render(flags=flags | rt.FLAG_COUNTERS, **shard)
        cnts.append(ctx.counters())
    rays_v = [c.rays_primary + c.rays_shadow + c.rays_bounce for c in cnts]
    alg_v = [alg_bytes_of(c, info, spp) for c in cnts]

    gather = None
    gloo = False
    if tiles:
        from ray_tracing_amd import multi_gpu
        gloo = dist.get_backend() == "gloo"
        # RCCL: pack / unpack are HIP kernels on the context's stream, the packed messages torch tensors on this device.
        # gloo (one-GPU rehearsal only): gloo moves host tensors, so the frame is read back and the host restatement of
        # the same packed layout is used.
        gather = multi_gpu.TileGather(W, H, 64, world, rank, ctx=None if gloo else ctx, device="cpu" if gloo else f"cuda:{local_rank}",
                                      in_flight=n_in_flight)

    acc_t = None
    gloo_s = False
    if samples:
        from ray_tracing_amd import multi_gpu
        d_rad_s, _, d_acc = ctx.device_buffers()
        acc_t = multi_gpu.wrap_device_buffer(d_acc, H * W * 4, local_rank)
        rad_s = multi_gpu.wrap_device_buffer(d_rad_s, H * W * 4, local_rank)
        gloo_s = dist.get_backend() == "gloo"
        # direct reduce-scatter over the point-to-point links + gather of the resolved slices to rank 0
        reducer = multi_gpu.SampleReduce(H * W * 4, world, rank, "cpu" if gloo_s else f"cuda:{local_rank}", gather_to=0)

    fps = n_views          # frames per step: one step = one pass over the orbit (BASELINE's "one pass over one batch of synthetic input": the batch is the orbit's views)
    step_no = [0]          # frames since the start: frame k renders view k % n_views on every rank
    host_combined = [None]  # gloo rehearsal only: rank 0's combined frame of the last step (a host array)

    def step(f, fixed_view=None):
        k = (step_no[0] % n_views) if fixed_view is None else fixed_view
        step_no[0] += 1
        if n_views > 1 or fixed_view is not None:
            ctx.set_view(views[k][0], views[k][1], W, H)       # a host-side struct: handed to the kernel by value, nothing waits
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
                host_combined[0] = gather.combine(ctx.read_radiance())   # the host path completes rank 0's HOST copy of the frame
            else:
                gather.combine_in_flight()
        return k

    def quiesce():
        ctx.synchronize()                          # every frame set's stream
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n_frames, steps, warmup, fixed_view=None):
        """steps frames with n_frames in flight, bracketed by barrier + synchronize; returns (seconds, max over ranks; kernel ms
        per launch; rays traced by this rank; algorithmic bytes of this rank's launches; view of the last frame)."""
        configure(n_frames)
        for _ in range(warmup * fps):
            step(flags, fixed_view)
        quiesce()
        ctx.counters()                             # start the HIP-event sums afresh
        rays = alg = 0
        last = 0
        t0 = time.perf_counter()
        for _ in range(steps * fps):
            last = step(flags | rt.FLAG_TIME_KERNELS, fixed_view)
            rays += rays_v[last]
            alg += alg_v[last]
        quiesce()
        dt_ = time.perf_counter() - t0
        kc = ctx.counters()                        # sums the per-launch HIP event pairs
        k_ms = kc.kernel_ms / (steps * fps if spp else max(kc.kernel_launches, 1))   # per launch; per frame when path tracing
        if distributed:
            tt = torch.tensor([dt_], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_ = float(tt.item())
        return dt_, k_ms, rays, alg, last

    dt, kernel_ms, rays_timed, alg_timed, last_view = timed(n_in_flight, args.steps, args.warmup)
    # the LAST frame of the timed region, as it stands on the device: compared with the oracle below (rank 0; in tiles mode it
    # is the frame the ranks combined)
    timed_frame = None
    if rank == 0:
        timed_frame = (ctx.read_prim_ids(), ctx.read_radiance(), ctx.read_hit_t()) if not spp else (None, ctx.read_radiance(), None)
        if spp and world == 1:
            timed_frame = (None, timed_frame[1], ctx.read_accumulation())
    # the same frames one at a time (every rank takes part: the collectives must match): what round 1's `value` was, and the
    # unoverlapped launch duration the counter-based figures refer to
    one_at_a_time = timed(1, args.steps, min(args.warmup, 3)) if overlapped else None
    # every frame in flight the SAME camera (view 0): rounds 1 and 2's timed region, for continuity
    same_view = None
    if overlapped and n_views > 1 and not args.no_same_view:
        same_view = timed(n_in_flight, args.steps, min(args.warmup, 3), fixed_view=0)
    if overlapped:
        configure(n_in_flight)

    total_rays = float(rays_timed)
    if distributed:
        rr = torch.tensor([total_rays], dtype=torch.float64, device="cuda")
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
        total_rays = float(rr.item())

    # ---- N > 1: the combined frame against the same frame from ONE rank --------------------------------
    parity_multi = None
    if tiles and world > 1:
        from ray_tracing_amd import multi_gpu
        # one more sharded frame of the view the timed region ended on, combined, with nothing else in flight; rank 0 then
        # renders the whole frame alone.  The frame rank 0 held right after the timed region is compared too.
        configure(1)
        k0 = step(flags, last_view)
        quiesce()
        if rank == 0:
            combined = host_combined[0] if gloo else ctx.read_radiance()
            ctx.set_view(views[k0][0], views[k0][1], W, H)
            ctx.render(flags=flags, **{k: v for k, v in shard.items() if k not in ("shard_rank", "shard_count", "tile_size")})
            alone = ctx.read_radiance()
            parity_multi = {"combined_equals_unsharded": multi_gpu.frames_identical(combined, alone),
                            "what": f"view {k0}: {world} ranks' 64x64 tiles gathered into rank 0's frame against rank 0 rendering every tile itself; bitwise"}
            if timed_frame is not None and timed_frame[1] is not None and not gloo:
                parity_multi["last_timed_frame_equals_unsharded"] = multi_gpu.frames_identical(timed_frame[1], alone)
        quiesce()
    if samples and world > 1:
        from ray_tracing_amd import multi_gpu
        # at a quarter of the resolution (same aspect, same world_from_clip): every rank its sample slice, combined, against
        # rank 0 tracing all world * spp samples alone.  Summation order differs: RMSE gate 1e-3, not bits.
        Wr, Hr = max(16, W // 4), max(16, H // 4)
        configure(1)
        ctx.set_view(views[0][0], views[0][1], Wr, Hr)
        d_rad_r, _, d_acc_r = ctx.device_buffers()
        acc_r = multi_gpu.wrap_device_buffer(d_acc_r, Hr * Wr * 4, local_rank)
        red_r = multi_gpu.SampleReduce(Hr * Wr * 4, world, rank, "cpu" if gloo_s else f"cuda:{local_rank}", gather_to=0)
        ctx.render(flags=flags, **shard)
        ctx.synchronize()
        res_r = red_r.combine(acc_r.cpu() if gloo_s else acc_r, spp * world)
        quiesce()
        if rank == 0:
            comb = res_r.cpu().numpy().reshape(Hr, Wr, 4)
            alone = dict(shard, spp=spp * world, sample_first=0, stream_id=0)
            ctx.render(flags=flags, **alone)
            parity_multi = {"combined_vs_one_rank_rmse": multi_gpu.frames_rmse(comb, ctx.read_radiance()), "gate": 1e-3,
                            "what": f"{Wr}x{Hr}: {world} sample slices of {spp} reduced over the links against rank 0 tracing all {spp * world} samples"}
        quiesce()
        ctx.set_view(views[0][0], views[0][1], W, H)
        del acc_r, red_r

    rates = None
    if rank == 0 and world == 1:
        try:
            ctx.synchronize()
            rates = rt.measure_gather_rates(local_rank)      # ~1 s, 4 GiB of scratch; in this process (no child, no exec)
        except Exception:
            rates = None
    if rank == 0:
        import oracle_py as orc
        n_frames_timed = args.steps * fps
        ms_per_step = dt / args.steps * 1e3
        ms_per_frame = ms_per_step / fps
        value = total_rays / dt / 1e6
        rays_step = rays_timed / args.steps                 # per step = per pass over the orbit (this rank)
        rays_frame = rays_timed / n_frames_timed
        alg_bytes = alg_timed / n_frames_timed              # per frame (= per launch of the primary / shadow kernel)
        cnt = cnts[last_view]
        hits_all = sum(c.rays_shadow for c in cnts)
        prim_all_v = sum(c.rays_primary for c in cnts)
        # The roofline's launch duration is the kernel ALONE on the GPU (the one-frame-at-a-time pass, HIP events around
        # every launch): with frames in flight a launch's begin-to-end time includes the time it shares the chip with its
        # neighbours (n frames in flight: ~n x longer) and says nothing about the kernel.  Both are printed.
        kernel_ms_in_flight = kernel_ms
        if one_at_a_time:
            kernel_ms = one_at_a_time[1]
            alg_one = one_at_a_time[3] / n_frames_timed
        else:
            alg_one = alg_bytes
        achieved = alg_one / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        # which kernel the flags select (rtk_launch_primary)
        wide = bool(info.n_wide_nodes) and not (flags & rt.FLAG_BINARY_TREE)
        forced = os.environ.get("RT_TUNE_STREAM_WAVES", "")
        k7 = f"k_stream<false, {'true' if wide else 'false'}>"
        k8 = f"k_stream8<false, {'true' if wide else 'false'}>"
        if spp:
            kernel = kernel_timed = "k_path_start+k_path_bounce (whole frame)"
        elif flags & rt.FLAG_BRUTE_FORCE or flags & rt.FLAG_STATIC_GRID:
            kernel = kernel_timed = "k_primary<false, false>"
        elif flags & rt.FLAG_WAVE_TRAVERSAL:
            kernel = kernel_timed = "k_primary<false, true>"
        else:
            kernel = k8 if forced == "8" else k7                                   # one frame at a time
            kernel_timed = k8 if forced == "8" or (forced in ("", "0") and n_in_flight >= 3) else k7   # the timed region
        # Counters of the committed PMC passes (profiles/pmc_traffic.json; rocprofv3 --pmc cannot run inside this process):
        # quoted only for the workload AND the very kernel binary they were collected on -- the file records the kernel's
        # symbol, registers, spills and the hash of the device code object (tools/kernel_resources.py); a library built
        # from other kernel sources makes them stale, and stale counters are not printed.
        traffic, valu, hbm_frac, pmc = None, None, None, {"measured_in_this_run": False}
        tj, fresh, fresh_path, why = load_pmc(rt)
        in_flight_block = None
        if tj is None:
            pmc["quoted"] = False
            pmc["why"] = why
        else:
            pmc["source"] = tj.get("pmc_source")
            if tj.get("workload") != args.workload or tj.get("kernel") != kernel or world != 1 or tj.get("views", 1) != n_views:
                pmc["quoted"] = False
                pmc["why"] = "committed passes are for another workload / kernel / orbit / GPU count"
            elif not fresh:
                pmc["quoted"] = False
                pmc["stale"] = True
                pmc["why"] = "librt_amd.so's device code differs from the one the passes were collected on: rerun tools/profile_round3.sh"
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
                # ... and the regime `value` is measured in: the kernel of the timed region (k_stream8 with >= 3 frames in
                # flight), its instruction and byte counts per launch from passes of THAT kernel (forced with
                # RT_TUNE_STREAM_WAVES=8 --in-flight 1: a counter pass serialises launches), over the time a frame takes in
                # the timed region.  No launch is alone there, so time per frame -- ms_per_step / frames_per_step -- is the denominator.
                tf = tj.get("in_flight")
                if overlapped and tf and tf.get("kernel") == kernel_timed:
                    peak_rate = tj.get("valu_peak_wave_insts_per_s_per_simd", 0.909e9)
                    busy = tf["valu_insts_per_launch"] / N_SIMDS / (ms_per_frame * 1e-3) / peak_rate
                    hbm_if = tf["hbm_bytes_per_launch"] / (ms_per_frame * 1e-3) / 1e9 / HBM_PEAK_GBS
                    util = {"valu": round(busy, 4), "hbm": round(hbm_if, 4)}
                    lg_ = sum(lane_gathers_of(c_) for c_ in cnts) / n_views
                    # the vector-memory front end by the hardware's own counter: TA_BUSY of the serialised pass, carried over to
                    # the timed frame time (the same fetches in less time)
                    if tf.get("ta_busy") and tf.get("avg_duration_us_under_pmc"):
                        util["vmem_front_end"] = round(tf["ta_busy"] * tf["avg_duration_us_under_pmc"] * 1e-3 / ms_per_frame, 4)
                    if rates:
                        util["gathers_vs_divergent_ceiling"] = round(lg_ / (ms_per_frame * 1e-3) / 1e9 / rates["l2_2MiB"]["independent"], 4)
                        util["beyond_l2_gathers"] = round(tf["hbm_bytes_per_launch"] / 64.0 / (ms_per_frame * 1e-3) / 1e9 / rates["hbm_1GiB"]["independent"], 4)
                    top = max((k_ for k_ in util if k_ != "gathers_vs_divergent_ceiling"), key=util.get)
                    in_flight_block = {
                        "kernel": kernel_timed, "frames_in_flight": n_in_flight, "ms_per_frame": round(ms_per_frame, 4),
                        "valu": {"busy": round(busy, 4), "lanes_active": tf["valu_lanes_active"], "frac": round(busy * tf["valu_lanes_active"], 4),
                                 "insts_per_launch": tf["valu_insts_per_launch"],
                                 "busy_is": "VALU wave-instructions per launch / (1,024 SIMDs x calibrated issue rate x ms_per_frame)"},
                        "hbm_frac": round(hbm_if, 4), "traffic": tf["hbm_bytes_per_launch"], "l2_hit_rate": tf.get("l2_hit_rate"),
                        "ta_busy_alone": tf.get("ta_busy"),
                        "utilisation": util,
                        "utilisation_is": "valu: issue rate against the calibrated peak; vmem_front_end: TA_BUSY of the serialised counter pass x its launch duration / ms_per_frame; "
                                          "gathers_vs_divergent_ceiling: counted lane-level 64-B gathers per second against rt_measure_gather_rates' cache-resident ceiling, which "
                                          "is for waves whose 64 lanes all fetch different lines -- lanes that share a node cost the front end one line (profiles/r3/gather_front_end_variants.txt), "
                                          "so this OVERSTATES its load; beyond_l2_gathers: HBM + Infinity Cache bytes / 64 B against random gathers from 1 GiB; hbm: the same bytes against the 8 TB/s streaming peak",
                        "bound": (top if util[top] >= 0.65 else "latency"),
                        "bound_rule": "the highest of valu / vmem_front_end / beyond_l2_gathers / hbm names the bound if it is >= 0.65, else latency of dependent gathers",
                        "counters_from": tf.get("collected_with")}
        # The bound: SURVEY 8d names HBM as the nominal roofline and `achieved`/`frac` keep that definition (ALGORITHMIC
        # bytes -- every lane's node / record fetch, most of them served by L1 / L2 / Infinity Cache -- over kernel time),
        # but the counters decide the label, and they are read in the regime `value` is measured in (roofline.in_flight)
        # when its passes exist; one frame at a time: a pipe that is >= 0.75 busy names the bound, else the launch is bound by
        # the LATENCY of its dependent gathers (waves parked at s_waitcnt, wave_time.waiting_for_memory).
        if valu is not None and hbm_frac is not None:
            bound_alone = "valu" if valu["busy"] >= 0.75 and valu["busy"] > hbm_frac else ("hbm" if hbm_frac >= 0.75 else "latency")
        else:
            bound_alone = "hbm (nominal; no current counters)"
        bound = in_flight_block["bound"] if in_flight_block else bound_alone
        frac_timed = alg_bytes / (ms_per_frame * 1e-3) / 1e9 / HBM_PEAK_GBS
        g_main = None
        if not spp:
            tf_ = (tj or {}).get("in_flight") if (tj is not None and pmc.get("quoted")) else None
            g_main = gather_block(rates, sum(lane_gathers_of(c_) for c_ in cnts) / n_views, ms_per_frame, kernel_ms if overlapped else None,
                                  tf_["hbm_bytes_per_launch"] if (overlapped and tf_ and tf_.get("kernel") == kernel_timed) else traffic)
        out = {
            "metric": "Mrays/s", "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "ms_per_frame": round(ms_per_frame, 4),
            "higher_is_better": True, "scaling": "strong" if args.mode == "tiles" else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "resolution": f"{W}x{H}", "rays": (f"path tracing, {spp} spp, <= {segs} segments" if spp else "primary+shadow, 1 spp"),
                       "triangles": int(info.n_triangles), "bvh_nodes": int(info.n_nodes),
                       "frames_per_step": fps, "step_is": (f"one pass over the orbit's {fps} views = {fps} frames through rt_render; {args.steps} steps = {n_frames_timed} frames timed" if fps > 1 else "one frame through rt_render"),
                       "views": n_views, "views_are": (f"frame k renders view k % {n_views} of an orbit around the scene ({360 // n_views} degrees apart); view 0 = the camera of rounds 1-2" if n_views > 1 else "one camera"),
                       "rays_per_step_rank0": int(round(rays_step)), "rays_per_view_rank0": [int(r) for r in rays_v],
                       "primary_hit_fraction": round(hits_all / max(prim_all_v, 1), 4), "camera_radii": CAMERA_RADII, "frames_in_flight": n_in_flight,
                       "backend": (dist.get_backend() if distributed else None), "rccl_ranks": (dist.get_world_size() if distributed and dist.get_backend() == "nccl" else None),
                       "parallelism": ("single-gpu" if not distributed else ("tiles64-roundrobin+host-pack+gather-to-rank0 (gloo rehearsal)" if gloo else "tiles64-roundrobin+hip-pack+one-gather-to-rank0+hip-unpack") if tiles else
                                       (f"sample-slices{world}+direct-reduce-scatter+gather-to-rank0" if samples else f"frames-dp{world}"))},
            "roofline": {"bound": bound, "bound_nominal": "hbm", "bound_one_frame_at_a_time": bound_alone,
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "achieved_is": "ALGORITHMIC bytes per launch / kernel time one frame at a time (SURVEY 8d); the caches serve most of those bytes, so this is not a bound: "
                                        "see frac_at_timed_rate and in_flight",
                         "frac_at_timed_rate": round(frac_timed, 5), "frac_exceeds_1_at_timed_rate": bool(frac_timed > 1.0),
                         "frac_at_timed_rate_is": "the same algorithmic bytes over ms_per_frame (the regime `value` is measured in): above 1 proves that algorithmic bytes / HBM peak does not bound this kernel",
                         "hbm_frac": hbm_frac, "valu": valu, "pmc": pmc,
                         "kernel": kernel, "kernel_ms": round(kernel_ms, 4),
                         **({"in_flight": in_flight_block} if in_flight_block else {}),
                         **({"kernel_ms_is": "launch duration one frame at a time (HIP events around every launch of the pass after the timed region)",
                             "kernel_ms_in_flight": round(kernel_ms_in_flight, 4),
                             "kernel_ms_in_flight_is": f"the same events in the timed region, {n_in_flight} frames in flight: begin-to-end time of launches that share the GPU",
                             "kernel_in_flight": kernel_timed} if overlapped else {}),
                         **({"gather": g_main} if g_main else {}),
                         "algorithmic_bytes_per_launch": int(alg_one),
                         "node_visits": int(sum(c.node_visits for c in cnts) / n_views), "tri_tests": int(sum(c.tri_tests for c in cnts) / n_views),
                         "bytes_per_ray": round(alg_bytes / max(rays_frame, 1), 1)},
        }
        if one_at_a_time:
            dt1, k1, rays1, alg1, _ = one_at_a_time
            a1 = alg1 / n_frames_timed / (k1 * 1e-3) / 1e9 if k1 > 0 else 0.0
            out["one_frame_at_a_time"] = {"value": round(rays1 * (total_rays / max(rays_timed, 1)) / dt1 / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(dt1 / args.steps * 1e3, 4), "ms_per_frame": round(dt1 / n_frames_timed * 1e3, 4),
                                          "steps": args.steps, "kernel_ms": round(k1, 4), "achieved": round(a1, 2), "frac": round(a1 / HBM_PEAK_GBS, 5),
                                          "note": "the same frames (same orbit) with --in-flight 1 (round 1's definition of value), measured after the timed region"}
        if same_view:
            dts, ks, rays_s, alg_s, _ = same_view
            out["same_view"] = {"value": round(rays_s * (total_rays / max(rays_timed, 1)) / dts / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(dts / args.steps * 1e3, 4), "ms_per_frame": round(dts / n_frames_timed * 1e3, 4),
                                "steps": args.steps, "frames_in_flight": n_in_flight, "rays_per_frame_rank0": int(rays_v[0]),
                                "note": "every frame the SAME camera (view 0), as rounds 1 and 2 timed it: frames in flight then share their working set in the caches"}
        if parity_multi:
            out["parity"] = parity_multi
        if not args.no_cpu_baseline and not spp and world == 1:
            vm, vp = views[last_view]
            view = orc.make_view(vm, vp, W, H)
            cb, pix, rgba_o, prim_o = cpu_baseline(orc, scene, view, W, H, flags, args.cpu_seconds)
            out["cpu_baseline"] = cb
            # parity gate on the same sample -- on the LAST FRAME OF THE TIMED REGION (read back
            # right after it, before anything else was rendered)
            prim_all, rgba_all, t_all = timed_frame
            prim_g = prim_all.ravel()[pix]
            rgba_g = rgba_all.reshape(-1, 4)[pix]
            d = rgba_g.astype(np.float64) - rgba_o.astype(np.float64)
            out["parity"] = {"frame": f"the last frame of the timed region (step {args.steps - 1}, frame {n_frames_timed - 1}, view {last_view}, {n_in_flight} in flight, {kernel_timed})",
                             "prim_id_agreement": float((prim_g == prim_o).mean()),
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
                rgba_w, prim_w, t_w, rays_w, (nv, nt, nc) = orc.render_primary_tree(osc, tree, view, flags & 1, threads=cb["cores"])
                dt_w = time.perf_counter() - t0
                dw = rgba_all.astype(np.float64) - rgba_w.astype(np.float64)
                cb["tree_walk"] = {"value": round(rays_w / dt_w / 1e6, 3), "unit": "Mrays/s", "cores": cb["cores"],
                                   "sample": f"full frame, {rays_w} rays, oracle walk (orc_hit_tree) of the device's own tree, {dt_w:.1f} s"}
                out["parity"]["full_frame"] = {
                    "prim_id_agreement": float((prim_all == prim_w).mean()), "radiance_rmse": float(np.sqrt(np.mean(dw * dw))),
                    "identical": bool(prim_all.tobytes() == prim_w.tobytes() and rgba_all.tobytes() == rgba_w.tobytes() and t_all.tobytes() == t_w.tobytes()),
                    "pixels": int(prim_all.size),
                    "counters_equal_oracle_walk": bool((nv, nt, nc) == (cnt.node_visits, cnt.tri_tests, cnt.mesh_aabb_tests) and rays_w == rays_v[last_view]),
                    "counters_are": "of the instrumented launch of the same view (the timed launches do not count)"}
                del nodes_h, recs_h, wide_h, tree, rgba_w, prim_w, dw
            except Exception as ex:                                          # never fail the bench line for it
                cb["tree_walk"] = {"error": str(ex)[:120]}
        if not args.no_cpu_baseline and spp and world == 1 and not args.extra_flags:
            # path-tracing workloads: the reference-shaped loop cannot reach these sizes, so the CPU figure is the oracle's
            # scalar walk of the device's own tree (oracle/rt_oracle.c orc_render_path_tree), the whole frame, which is
            # also a whole-frame parity check of the running sums -- of the last frame of the timed region
            try:
                vm, vp = views[last_view]
                view = orc.make_view(vm, vp, W, H)
                threads = max(1, min(16, os.cpu_count() or 1))
                tree = orc.Tree(*ctx.read_bvh(), ctx.read_bvh_wide() if info.n_wide_nodes else None)
                osc = orc.Scene(scene.triangles, scene.meshes, scene.vertices, scene.materials)
                acc_g = timed_frame[2]
                t0 = time.perf_counter()
                acc_w, segs_w, _ = orc.render_path_tree(osc, tree, view, spp=spp, max_segments=segs, seed=1, stream=rank, threads=threads)
                dt_w = time.perf_counter() - t0
                out["cpu_baseline"] = {"value": round(segs_w / dt_w / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
                                       "sample": f"full frame, {segs_w} ray segments, oracle walk of the device's own tree (orc_render_path_tree), {dt_w:.1f} s"}
                dd = acc_g.astype(np.float64) - acc_w.astype(np.float64)
                out["parity"] = {"frame": f"the last frame of the timed region (view {last_view})",
                                 "accumulation_identical": bool((acc_g.view(np.uint32) == acc_w.view(np.uint32)).all()),
                                 "radiance_rmse": float(np.sqrt(np.mean(dd * dd))) / spp, "pixels": int(W * H),
                                 "segments_equal": bool(segs_w == rays_v[last_view])}
                del tree, acc_w, acc_g, dd
            except Exception as ex:                                              # never fail the bench line for it
                out["cpu_baseline"] = {"error": str(ex)[:120]}
        if not args.no_north_star and not spp and world == 1 and args.workload == "sphere_field_10k_1080p_primary_shadow":
            # Driver-timed figures for the other BASELINE configs, measured after the timed region; never `value`:
            # north_star's own target scene (1 M-triangle icosphere soup, 1080p, primary + shadow, ">= 1 Grays/s"), configs[2]
            # (the same scene, 4 jittered samples per pixel) and configs[4]'s scene (10 M triangles, incoherent bounces) -- the
            # kernels furthest from any ceiling, with the counters of their committed passes quoted under
            # the same fingerprint guard.
            steps_x = max(10, min(n_frames_timed, 50))
            try:
                ns = side_workload(rt, orc, "soup_1m_1080p_primary_shadow", local_rank, max(10, min(n_frames_timed, 200)), n_in_flight, check=False, rates=rates)
                ns["target"] = ">= 1000 Mrays/s primary+shadow at 1080p on a 1M-triangle scene (BASELINE.json north_star)"
                out["north_star_scene"] = ns
            except Exception as ex:                                              # never fail the bench line for it
                out["north_star_scene"] = {"error": str(ex)[:120]}
            extra = {}
            for key, name, sec in (("configs2", "soup_1m_1080p_4spp", "configs2"), ("path_10m", "soup_10m_1080p_path_1spp_8seg", "path_10m")):
                try:
                    quoted = None
                    if tj is not None and sec in tj:
                        quoted = dict(tj[sec], quoted=True) if fresh_path else {"quoted": False, "stale": True, "why": "the path kernels' device code differs from the one the passes were collected on"}
                    extra[key] = side_workload(rt, orc, name, local_rank, steps_x, n_in_flight, pmc_section=quoted, check=not args.no_cpu_baseline, rates=rates)
                except Exception as ex:
                    extra[key] = {"error": str(ex)[:120]}
            out["extra"] = extra
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
