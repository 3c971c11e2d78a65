"""Headline benchmark: Mrays/s (primary + secondary + shadow) and ms/frame on BASELINE config 4 --
a 708x708-cell height field (1,002,528 triangles) with one glass sphere, 8 point lights with shadow rays,
depth-4 reflection/refraction, 3840x2160 -- tile-sharded over N GPUs with a gather of the frame to rank 0.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one frame: every rank renders its interleaved 16x16 tiles (rt_render_shard), the compact tile buffers are
gathered to rank 0 (RCCL) and scattered into the row-major frame (rt_detile).  The scene is resident in HBM before the
timed region; the frame stays in HBM.  A ray is one Scene::cast_ray_once call of the reference.

The K timed frames are a sequence (the reference's --sequence workload, sdl/src/main.rs:70-95, whose frames move with
`t`): the camera orbits the point it looks at by 0.25 degrees and moves 0.03 % of its distance closer per frame, so every
frame of the run is a different picture (--still holds the scene's own view instead: the same frame K times).  The warm-up
renders the stretch of the path BEFORE the timed frames, as a running sequence would have: the level sizes the timed sets
are queued with were learned from earlier frames, never from themselves (`frames_rendered_twice` counts the sets that
outgrew them).  --frames-in-flight F sets of launches (default
2) are in flight per GPU, each frame complete and delivered in order; F = 1 renders them strictly one after the other.
A set of launches covers --batch B consecutive frames (rt_render_shard_batch): by default two 4K frames' worth of pixels
on one GPU (whole frames fill the chip, and longer sets only hold more memory); this rank's shard of half of the K frames
when the frame is sharded over N GPUs, at most ten whole frames' worth of rays, which keeps a rank's launches large.  The line also carries the one-frame-at-a-time figure (`ms_per_frame_alone`) and, on one GPU, the time per
frame through rt_render, host pixels included (`ms_per_frame_end_to_end`).

Roofline of the dominant kernel (the budget-free shadow packet kernel, k_shadow<false, false, false>): its launches are timed
alone (a context whose kernels run one after the other, HIP events around the kernel on its own stream); numerators
are work the kernel really does -- BVH bytes its walks fetch (instrumented variant), HBM-side bytes and VALU-busy
cycles from the rocprofv3 counter passes kept under profiles/ -- and `bound` is the roof it comes closest to.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a queue
# run one after the other.  Whole frames, and batches of N shards of 1/N frame, fill the chip with the default; the
# variable is left alone unless RT_HW_QUEUES asks for an experiment.
if os.environ.get("RT_HW_QUEUES"):
    os.environ["GPU_MAX_HW_QUEUES"] = os.environ["RT_HW_QUEUES"]

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E, spec
SIMDS = 256 * 4         # compute units x SIMDs
CLOCK_GHZ = 2.4         # peak engine clock, spec


def cpu_leg(scene, vw, vh, seconds, cores, frame_rgb):
    """The CPU oracle (a line-for-line f64 port of the reference path, one thread per core, dynamic chunks) on a
    uniformly random pixel sample of the SAME frame, sized for about `seconds` of wall time.  Returns the timing,
    the per-frame work counts scaled from the sample, and the GPU-vs-CPU parity on the sampled pixels."""
    import oracle
    from raytracer_amd.compare import image_metrics
    total = vw * vh
    rng = np.random.default_rng(12345)
    pilot = rng.choice(total, size=min(20000, total), replace=False).astype(np.uint32)
    t0 = time.perf_counter()
    oracle.render(scene, cores, pixels=pilot, ids=False)
    dt = max(time.perf_counter() - t0, 1e-4)
    n = int(min(total, max(len(pilot), len(pilot) * seconds / dt)))
    px = rng.choice(total, size=n, replace=False).astype(np.uint32)
    t0 = time.perf_counter()
    rgb, _, st, oc = oracle.render(scene, cores, pixels=px, ids=False)
    dt = time.perf_counter() - t0
    scale = total / float(n)
    parity = image_metrics(frame_rgb.reshape(-1, 3)[px], rgb) if frame_rgb is not None else None
    if parity and parity["psnr"] == float("inf"):
        parity["psnr"] = None   # identical: PSNR is unbounded (JSON has no infinity)
    return {
        "seconds": dt, "pixels": n, "rays": int(st.rays), "mrays_s": st.rays / dt / 1e6, "scale": scale,
        "bytes_closest": oc.algorithmic_bytes(0) * scale,
        "bytes_shadow": oc.algorithmic_bytes(1) * scale,
        "bytes_helper": oc.algorithmic_bytes(2) * scale,
        "bytes_texels": 3.0 * (oc.texel_image + oc.texel_cubemap) * scale,
        "counts": oc.as_dict(), "parity": parity, "would_panic": int(oc.would_panic),
    }


GOLDEN = os.path.join(ROOT, "tests", "golden")


def _atlas():
    # the reference's cubemap, assets/storforsen_4.jpg, is 8192 x 6144 (151 MB as RGB): a procedural atlas of that size,
    # so that the sky lookups gather from as large an image (skybox.rs:64-86)
    from raytracer_amd import scenes
    return scenes.procedural_cubemap(2048)


def _config1(vw, vh, args):
    from raytracer_amd import scenes
    return scenes.cornell_scene(vw, vh), "scenes/cornell.sdl (5 planes, an emissive panel, 2 spheres, one of glass; one rectangular area light of 64 sample points)"


def _config2(vw, vh, args):
    from raytracer_amd import scenes
    return scenes.refraction_scene(vw, vh, cubemap=_atlas()), "scenes/refraction.sdl (3 spheres, 1 plane, sun, 8192x6144 cubemap)"


def _config3(vw, vh, args):
    import raytracer_amd as ra
    s = ra.LoadedScene(os.path.join(GOLDEN, "fedora_192x108.rtscene")).set_resolution(vw, vh).set_cubemap(_atlas())
    return s, "scenes/fedora.sdl (1038-triangle textured mesh over a mirror plane, sun, 8192x6144 cubemap)"


def _config3b(vw, vh, args):
    import raytracer_amd as ra
    s = ra.LoadedScene(os.path.join(GOLDEN, "teapot_128x72.rtscene")).set_resolution(vw, vh)
    return s, "scenes/bunny.sdl with assets/teapot.obj (6320 triangles, normals recalculated) for the missing bunny"


def _config4(vw, vh, args):
    from raytracer_amd import scenes
    s = scenes.synthetic_terrain_scene(vw, vh, cells=args.cells, n_lights=args.lights)
    return s, (f"synthetic height field ({args.cells}x{args.cells} cells) + 1 glass sphere + {args.lights} point lights with shadows")


def _config5a(vw, vh, args):
    from raytracer_amd import scenes
    return scenes.huge_grid_scene(vw, vh, n=12, cubemap=scenes.procedural_cubemap(1024, seed=3)), "scenes/huge_grid.sdl (1728 jittered glass spheres, sun, 4096x3072 star atlas)"


def _config5b(vw, vh, args):
    from raytracer_amd import scenes
    return scenes.terrain_scene(vw, vh), "scenes/terrain.sdl (72x72 grid = 10,082 triangles, reflectiveness 0.8, sun)"


# name -> (builder, default resolution)
CONFIGS = {"1": (_config1, (1920, 1080)), "2": (_config2, (1920, 1080)), "3": (_config3, (3840, 2160)), "3b": (_config3b, (3840, 2160)),
           "4": (_config4, (3840, 2160)), "5a": (_config5a, (7680, 4320)), "5b": (_config5b, (7680, 4320))}


# The roofline legs (the dominant kernel timed alone, the instrumented frame) and the counter passes kept under profiles/
# (tools/prof_frame.py) all render THIS frame of the camera path, whatever --steps and --warmup are: per-launch counters of
# one view divided by the launch duration of another would be no fraction of anything.
ROOFLINE_FRAME = 20


def camera_path(base, f, still=False):
    """Frame f of the benchmark sequence: the base camera moved round the vertical axis through the point it looks at
    on the ground (or 5 units ahead when it does not look down) by 0.25 degrees per frame, and 0.03 % of its distance
    nearer per frame (config 4: 0.1 units).  The view direction of camera.rs:55-59 for (yaw, pitch) is
    (cos p sin y, sin p, -cos p cos y), so the camera keeps looking at the pivot with yaw = atan2(dx, -dz)."""
    import math
    from raytracer_amd.scene import Camera
    if still:
        return Camera(vw=base.vw, vh=base.vh, origin=tuple(base.origin), yaw=base.yaw, pitch=base.pitch, fov=base.fov)
    ox, oy, oz = (float(v) for v in base.origin)
    cp, sp = math.cos(base.pitch), math.sin(base.pitch)
    reach = min(oy / -sp * cp, 1e4) if sp < -1e-3 and oy > 0 else 5.0          # horizontal distance to the pivot
    px, pz = ox + reach * math.sin(base.yaw), oz - reach * math.cos(base.yaw)
    a = math.radians(0.25) * f
    r = reach * (1.0 - 0.0003) ** f
    # the camera's place on the circle round the pivot, seen from the pivot
    bx, bz = ox - px, oz - pz
    nx, nz = (bx * math.cos(a) - bz * math.sin(a)) * r / reach, (bx * math.sin(a) + bz * math.cos(a)) * r / reach
    yaw = math.atan2(-nx, nz)
    return Camera(vw=base.vw, vh=base.vh, origin=(px + nx, oy * r / reach, pz + nz), yaw=yaw if abs(yaw) > 1e-9 else 1e-4,
                  pitch=base.pitch, fov=base.fov)


def default_batch(steps, world, small, pixels=3840 * 2160):
    """Frames per set of launches when --batch is not given.  One GPU: as many frames as make up the pixels of two 4K
    frames (2 at 4K, 8 at 1920x1080, 1 beyond 4K) -- larger sets hold more memory and are no faster (4K, 40 frames:
    7.16 ms per frame and 7.3 GB of wavefront levels as sets of 2, 7.17 ms and 35.5 GB as sets of 10).  Sharded over N
    GPUs a rank's launches are N times smaller, and only long sets restore their size: half of the run's frames, at most
    ten whole frames' worth of rays (10 x world shard frames) and at most RT_BATCH_MAX = 64; frames beyond 4K at most
    `world` per set."""
    half = (steps + 1) // 2
    if world == 1:
        return max(1, min(half, 64, (2 * 3840 * 2160) // max(1, pixels)))
    if small:
        return max(1, min(half, 10 * world, 64))
    return max(1, min(half, world))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", default="4", choices=sorted(CONFIGS),
                    help="BASELINE configuration: 1 cornell.sdl 1080p (area light); 2 refraction.sdl 1080p; 3 fedora.sdl 4K; 3b the teapot standing in for bunny.sdl 4K; "
                         "4 (default, the headline) 1M-triangle terrain 4K; 5a huge_grid.sdl 8K; 5b terrain.sdl 8K")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--cells", type=int, default=708)
    ap.add_argument("--lights", type=int, default=8)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--frames-in-flight", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None, help="frames per set of launches (default: two 4K frames' worth of pixels on one GPU; half of the steps, at most ten whole frames' worth of rays, when sharded)")
    ap.add_argument("--still", action="store_true", help="every frame through the scene's own camera (the same picture K times) instead of the camera path")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import raytracer_amd as ra
    from raytracer_amd import dist as rd
    from raytracer_amd import scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    backend = os.environ.get("RT_DIST_BACKEND", "nccl")     # "gloo": rehearsal of the N > 1 path with ranks sharing a GPU
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    builder, (dw, dh) = CONFIGS[args.config]
    vw, vh = args.width or dw, args.height or dh
    t0 = time.perf_counter()
    scene, what = builder(vw, vh, args)
    t_build = time.perf_counter() - t0
    t_upload = [0.0]

    def make_context():
        c = ra.Context(local_rank)
        t0 = time.perf_counter()
        c.upload(scene)
        t_upload[0] = time.perf_counter() - t0
        return c

    # A 1/N shard's launches are N times smaller than a whole frame's and fill the chip badly (a 1/8 shard takes 1.5-1.7
    # ms per frame one frame per set, however many sets are in flight, against 0.91 for an eighth of a whole frame);
    # several consecutive frames per set of launches restore their size.
    small = vw * vh <= 3840 * 2160          # (an 8K frame's wavefront levels take tens of GB per frame in flight)
    if args.batch is None:
        # Sharded, the K frames go as sets of launches of half of them each, two sets side by side: large sets fill the
        # chip through their deep levels too, and every further set would add its own tail of small launches (a 1/8 shard
        # of 20 frames is 20 ms of work all told).  Whole frames on one GPU fill the chip by themselves: two per set.
        args.batch = default_batch(args.steps, world, small, vw * vh)
    if args.frames_in_flight is None:
        args.frames_in_flight = 2
    depth, batch = max(1, args.frames_in_flight), max(1, args.batch)
    pipe = rd.FramePipeline(make_context, rank, world, device, depth=depth, batch=batch)
    t_upload = t_upload[0]
    info = ra.desc_info(scene.flatten())
    base_cam = scene.camera                          # the scene's own view: frame 0 of the path
    # Where on the path the next run() starts: every run() continues where the last one ended, and the warm-up runs
    # below count up to 0, so that the timed frames are frames 0 .. K-1 of the path -- the sequence starts at the
    # scene's own view -- and what precedes them is the stretch of the path that leads there.
    cursor = [0]
    cam_last = [None]

    def camera_at(f):
        cam_last[0] = camera_path(base_cam, f, args.still)
        return cam_last[0]

    def sync():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    # one frame at a time first (latency figure; also warms every context up)
    last = {}

    def keep(f, st, frame):
        last["st"], last["frame"] = st, frame

    def run(n, cb=keep):
        first = cursor[0]
        cursor[0] += n
        return pipe.run(n, cb, camera_of=lambda f: camera_at(first + f))

    def twice(stats_list):
        # rt_stats.respeculated counts per context; set j of a run went to context j % depth
        lastv = {}
        for j, x in enumerate(stats_list):
            lastv[j % depth] = int(x.respeculated)
        return lastv

    # warm-up: every context renders, and every shape of launch set the timed run will use (a last, shorter batch
    # included) has been seen once, so that its level sizes are known (see rt_stats.speculative)
    alone = 5
    cursor[0] = -(max(args.warmup, depth * batch) + (args.steps if batch > 1 else 0) + 3 + alone + max(args.warmup, depth))
    run(max(args.warmup, depth * batch))
    if batch > 1:
        run(args.steps)
    for _ in range(3):      # (the first single frames after sets of many are slower than the ones that follow)
        run(1)
    sync()
    t0 = time.perf_counter()
    for _ in range(alone):
        run(1)
    sync()
    ms_alone = (time.perf_counter() - t0) * 1e3 / alone

    before = twice(run(max(args.warmup, depth)))      # (every context once more, so that each one's count is known)
    assert cursor[0] == 0
    sync()
    t0 = time.perf_counter()
    stats = run(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    st, frame = last["st"], last["frame"]
    last_frame_index = cursor[0] - 1
    last_cam = camera_path(base_cam, last_frame_index, args.still)
    ms_trace = sum(x.ms_trace for x in stats); ms_shadow = sum(x.ms_shadow for x in stats)
    ms_shade = sum(x.ms_shade for x in stats); ms_resolve = sum(x.ms_resolve for x in stats)
    ms_kernels = sum(x.ms_render for x in stats)
    after = twice(stats)
    n_respeculated = sum(after[k] - before.get(k, 0) for k in after)

    # rays of all K timed frames (each set of launches reports the rays of its frames; frames differ along the path)
    rays = torch.tensor([sum(int(getattr(x, k)) for x in stats) for k in ("rays_primary", "rays_reflect", "rays_transmit", "rays_shadow")],
                        dtype=torch.int64, device=device)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    rays_run = [int(x) for x in rays.cpu()]
    rays = [r / args.steps for r in rays_run]        # per frame, averaged over the run
    elapsed = float(tmax.cpu()[0])
    rays_frame = sum(rays)
    ms_per_step = elapsed * 1e3 / args.steps
    value = sum(rays_run) / elapsed / 1e6

    # the untimed legs below, and the CPU leg, look at the LAST timed frame: the scene takes that frame's camera
    last_cam.vw, last_cam.vh = vw, vh
    scene.set_camera(last_cam)

    # ---- untimed legs on rank 0 of a one-GPU run: end-to-end frames, exclusive kernel durations, instrumented counts
    visits = end_to_end = serial = None
    if rank == 0 and world == 1:
        # (1) Scene::render's own contract: pixels in host memory (scene.rs:220-246) -- rt_render, one frame at a time
        ctx0 = pipe.ctxs[0]
        ctx0.upload(scene, keep_meshes=True)          # the last timed frame's camera; the meshes stay where they are
        ctx0.render()
        ctx0.render()
        t0 = time.perf_counter()
        n_e2e = 5
        for _ in range(n_e2e):
            ctx0.render()
        end_to_end = (time.perf_counter() - t0) * 1e3 / n_e2e
        pipe.close()
        pipe = None
        # (2) the dominant kernel alone: a context whose kernels run one after the other on one stream, so that the
        #     HIP events around k_shadow<false, false, false> (on that stream) bracket nothing else.  Frame ROOFLINE_FRAME
        #     of the path: the view the counter passes under profiles/ were taken on.
        roof_cam = camera_path(base_cam, ROOFLINE_FRAME, args.still)
        roof_cam.vw, roof_cam.vh = vw, vh
        scene.set_camera(roof_cam)
        os.environ["RTGPU_SERIAL"] = "1"
        sctx = ra.Context(local_rank)
        sctx.upload(scene)
        sctx.render()
        sst = [sctx.render()[2] for _ in range(4)]
        sctx.close()
        del os.environ["RTGPU_SERIAL"]
        n_main = sum(x.launches_shadow_main for x in sst)
        serial = {"frames": len(sst), "ms_per_frame_kernels": sum(x.ms_render for x in sst) / len(sst),
                  "k_shadow_main_launches_per_frame": n_main / len(sst),
                  "k_shadow_main_avg_launch_ms": sum(x.ms_shadow_main for x in sst) / max(n_main, 1),
                  "ms_per_frame": {"trace": sum(x.ms_trace for x in sst) / len(sst), "shadow": sum(x.ms_shadow for x in sst) / len(sst),
                                   "spawn+light": sum(x.ms_shade for x in sst) / len(sst), "resolve": sum(x.ms_resolve for x in sst) / len(sst)}}
        # (3) what the traversal really touched: one frame through the instrumented kernel variants (all kernels), and
        #     one with only the shadow kernels instrumented (the bytes of the dominant kernel's launches)
        def instrumented(which):
            os.environ["RTGPU_COUNT_VISITS"] = which
            cctx = ra.Context(local_rank)
            cctx.upload(scene)
            _, _, cst = cctx.render(ids=False)
            cctx.close()
            del os.environ["RTGPU_COUNT_VISITS"]
            return cst
        cst, shst = instrumented("1"), instrumented("shadow")
        scene.set_camera(last_cam)                    # (the CPU leg below looks at the last timed frame again)
        visits = {"lane_node_visits": int(cst.nodes_visited), "lane_tri_tests": int(cst.tris_tested),
                  # wave_tri_steps: triangle records fetched per wave by the closest-hit packets, and rounds of 64
                  # (ray, leaf) pairs tested by the shadow packets (whose lanes fetch their own triangles)
                  "wave_node_fetches": int(cst.wave_nodes), "wave_tri_steps": int(cst.wave_tris),
                  # B_touched: node and triangle records the walks really fetched, at their real sizes (112 / 80
                  # bytes), a packet's shared fetch counted once
                  "bytes_touched": int(cst.bytes_fetched),
                  "bytes_touched_k_shadow_main": int(shst.bytes_fetched_main), "k_shadow_main_launches": int(shst.launches_shadow_main)}

    if rank == 0:
        frame_rgb = frame.cpu().numpy().reshape(vh, vw, 3)
        cores = int(os.environ.get("OMP_NUM_THREADS") or min(os.cpu_count() or 1, 16))
        # N == 1: the timed CPU baseline.  N > 1: only a short pass for the parity figure.
        cpu = cpu_leg(scene, vw, vh, args.cpu_seconds if world == 1 else 3.0, cores, frame_rgb)

        # B_alg, the survey's per-frame figure: bytes the REFERENCE's exhaustive walk would read (32 B per BVH node
        # test, 36 per triangle, 16/28/24 per sphere/plane/box, 3 per texel), counted by the CPU port on the sampled
        # pixels and scaled to the frame.  A side figure: the device walk prunes by distance and ends a shadow cast at
        # the first blocker, so it does not do that work, and B_alg over time is not a fraction of any roof.
        b_alg = {"closest_casts": cpu["bytes_closest"], "shadow_casts": cpu["bytes_shadow"],
                 "helper_intersects_and_texels": cpu["bytes_helper"] + cpu["bytes_texels"]}
        b_alg["total"] = sum(b_alg.values())
        b_alg["GB/s_over_frame_time"] = b_alg["total"] / (ms_per_step * 1e-3) / 1e9

        roofline = None
        roofline_unavailable = None
        if serial and serial["k_shadow_main_avg_launch_ms"] > 0 and args.config == "4":
            dur = serial["k_shadow_main_avg_launch_ms"] * 1e-3          # exclusive, measured in this run
            n_l = serial["k_shadow_main_launches_per_frame"]
            fr = {}
            touched = visits["bytes_touched_k_shadow_main"] / max(visits["k_shadow_main_launches"], 1)
            fr["bvh_bytes_touched"] = {"achieved": touched / dur / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "per_launch": touched,
                                       "note": "node and triangle records the launch's walks fetch, a packet's shared fetch once and "
                                               "a queued pair's fetch per lane; mostly served by the caches, so a side figure against "
                                               "the HBM peak and never the bound"}
            fr["bvh_bytes_touched"]["frac"] = fr["bvh_bytes_touched"]["achieved"] / HBM_PEAK_GBS
            traffic = None
            stale = None
            ipath = os.path.join(ROOT, "profiles", "r3_roofline_inputs.json")
            inp = None
            if not os.path.exists(ipath):
                stale = "profiles/r3_roofline_inputs.json is missing (tools/round_profiles.sh writes it)"
            else:
                try:
                    from raytracer_amd._native import source_hash
                    inp = json.load(open(ipath))
                    if inp.get("code_sha256") != source_hash():
                        stale = ("the counter passes in profiles/r3_roofline_inputs.json were taken on another build of the kernels "
                                 f"(sources then {str(inp.get('code_sha256'))[:12]}, now {source_hash()[:12]}): run tools/round_profiles.sh")
                    elif not (inp.get("workload_cells") == args.cells and inp.get("resolution") == [vw, vh] and inp.get("lights") == args.lights):
                        stale = "the counter passes in profiles/r3_roofline_inputs.json are of another workload"
                except Exception as e:   # a malformed file must not lose the run
                    stale = "profiles/r3_roofline_inputs.json could not be read: " + repr(e)
            if not stale:
                k = inp["k_shadow_main"]
                traffic = k["hbm_bytes_per_launch"]
                fr["hbm"] = {"achieved": traffic / dur / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "per_launch": traffic,
                             "note": "(2 x FETCH_SIZE + WRITE_SIZE) of the counter passes in profiles/, per launch; an upper bound: the counter tallies vector-memory fetches at half and fetches through the scalar cache exactly (profiles/r2_fetch_calibration.json)"}
                fr["hbm"]["frac"] = fr["hbm"]["achieved"] / HBM_PEAK_GBS
                if "fetch_size_raw_bytes_per_launch" in k:
                    low = k["fetch_size_raw_bytes_per_launch"] + k.get("write_bytes_per_launch", 0)
                    fr["hbm"]["per_launch_lower_bound"] = low               # FETCH_SIZE as it reads + WRITE_SIZE
                    fr["hbm"]["frac_lower_bound"] = low / dur / 1e9 / HBM_PEAK_GBS
                busy = k["valu_busy_cycles_per_launch"]         # SQ_ACTIVE_INST_VALU x 4, summed over the chip's SIMDs
                peak = SIMDS * CLOCK_GHZ
                fr["valu"] = {"achieved": busy / dur / 1e9, "peak": peak, "unit": "G SIMD-cycles/s", "per_launch": busy,
                              "valu_instructions_per_launch": k.get("valu_insts_per_launch"),
                              "busy_at_measured_clock": k.get("valu_busy_frac_measured_clock"),
                              "note": "cycles the vector ALUs are busy (SQ_ACTIVE_INST_VALU x 4) against SIMDs x the 2.4 GHz "
                                      "spec clock; the clock under this load is lower (busy_at_measured_clock uses GRBM_GUI_ACTIVE)"}
                fr["valu"]["frac"] = fr["valu"]["achieved"] / peak
                # how much of that is a ray's arithmetic: the share of the 64 lanes that is live while the vector ALUs
                # are busy, and the useful fraction that follows; the scalar unit beside it
                fr["valu"]["valu_lane_utilisation"] = k.get("valu_lane_utilisation")
                if k.get("valu_lane_utilisation"):
                    fr["valu"]["frac_live_lanes"] = fr["valu"]["frac"] * k["valu_lane_utilisation"]
                fr["salu_busy"] = k.get("salu_busy")          # wave-cycles on scalar instructions per SIMD-cycle
                fr["salu_insts_per_cu_cycle"] = k.get("salu_insts_per_cu_cycle")
                bound = max(("valu", "hbm"), key=lambda r: fr[r]["frac"])
                roofline = {"bound": bound, "kernel": "k_shadow<false, false, false>", "achieved": fr[bound]["achieved"], "peak": fr[bound]["peak"],
                            "unit": fr[bound]["unit"], "frac": fr[bound]["frac"], "traffic": traffic,
                            "avg_launch_ms": serial["k_shadow_main_avg_launch_ms"], "launches_per_frame": n_l,
                            "share_of_frame_kernel_time": serial["k_shadow_main_avg_launch_ms"] * n_l / serial["ms_per_frame_kernels"],
                            "valu_lane_utilisation": k.get("valu_lane_utilisation"), "salu_busy": k.get("salu_busy"),
                            "roofs": fr,
                            "frame_of_the_path": "the scene's own view" if args.still else ROOFLINE_FRAME,
                            "note": "launch duration: HIP events around the kernel on its own stream in a context whose kernels run one "
                                    "after the other (this run); bound = the hardware roof (valu, hbm) with the largest fraction; "
                                    "counters: profiles/r3_roofline_inputs.json, taken on this build of the kernels (source hash checked)"}
            else:
                # no roofline from counters of another build: the line says why, and keeps what this run measured itself
                roofline_unavailable = {"reason": stale, "kernel": "k_shadow<false, false, false>",
                                        "avg_launch_ms": serial["k_shadow_main_avg_launch_ms"], "bvh_bytes_touched": fr["bvh_bytes_touched"]}
        out = {
            "metric": "Mrays/s (primary+secondary)", "value": value, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"{what}, {info['n_tris']} triangles, {vw}x{vh}, max_ray_depth 4 (BASELINE config {args.config})",
                "resolution": [vw, vh], "triangles": info["n_tris"], "bvh_nodes": info["n_nodes"], "lights": info["n_lights"],
                "parallelism": f"64x64-pixel supertiles interleaved over {world} GPU(s), gather to rank 0",
                "sets_of_launches_in_flight": depth, "frames_per_set_of_launches": batch,
            },
            "rays_per_frame": {"primary": rays[0], "reflect": rays[1], "transmit": rays[2], "shadow": rays[3], "total": rays_frame},
            "ms_per_frame_alone": ms_alone,
            "ms_per_frame_end_to_end": end_to_end,
            "ms_per_frame_kernels_rank0": ms_kernels / args.steps,
            # sums of the launches' own durations; the phases (and frames) overlap, so they exceed the frame time
            "ms_phase_rank0": {"trace": ms_trace / args.steps, "shadow": ms_shadow / args.steps, "spawn+light": ms_shade / args.steps,
                               "resolve": ms_resolve / args.steps},
            "frames_rendered_twice": n_respeculated,
            "sequence": ("still: the scene's own view for every frame" if args.still else
                         f"camera path: 0.25 degrees round the point looked at and 0.03 % nearer per frame; timed frames {last_frame_index - args.steps + 1}..{last_frame_index} "
                         "of the path, level sizes learned from the frames before them; parity and the untimed legs on the last frame"),
            "roofline": roofline,
            "roofline_unavailable": roofline_unavailable,
            "kernels_one_after_the_other": serial,
            "reference_walk_bytes_per_frame": b_alg,
            "gpu_visits_per_frame": visits,
            "parity_vs_cpu_sample": cpu["parity"],
            "setup_s": {"scene_build": t_build, "upload": t_upload}, "device_bytes": int(st.device_bytes),
        }
        if world == 1:
            out["cpu_baseline"] = {
                "value": cpu["mrays_s"], "unit": "Mrays/s", "cores": cores, "kind": "port",
                "sample": f"{cpu['pixels']} uniformly random pixels of the same frame ({cpu['rays']} rays) in {cpu['seconds']:.1f} s; "
                          f"exhaustive reference traversal (no closest-hit pruning), {cores} threads",
            }
        print(json.dumps(out), flush=True)
    if pipe is not None:
        pipe.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
