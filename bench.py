"""Headline benchmark: Mrays/s (primary + secondary + shadow) and ms/frame on BASELINE config 4 --
a 708x708-cell height field (1,002,528 triangles) with one glass sphere, 8 point lights with shadow rays,
depth-4 reflection/refraction, 3840x2160 -- tile-sharded over N GPUs with a gather of the frame to rank 0.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one frame: every rank renders its interleaved 16x16 tiles (rt_render_shard), the compact tile buffers are
gathered to rank 0 (RCCL) and scattered into the row-major frame (rt_detile).  The scene is resident in HBM before the
timed region; the frame stays in HBM.  A ray is one Scene::cast_ray_once call of the reference.

The K timed frames are a sequence (the reference's --sequence workload): --frames-in-flight F (default 4; 8 when the frame is sharded) of them are
in flight per GPU, each complete and delivered in order; F = 1 renders them strictly one after the other.  The line
also carries the one-frame-at-a-time figure (`ms_per_frame_alone`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# One HIP stream per concurrent chain of launches: a frame uses five (trace/spawn + four shadow/light chains) and
# several frames are in flight.  HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (its
# default, 4, is also what the environment of a GPU box usually holds); streams that share a queue run one after the
# other.  A whole frame's launches fill the chip on their own and 4 queues serve it as well as any number (1 % better
# than 24, in fact).  The launches of a 1/N shard are N times shorter and only fill the chip when the chains of the
# frames in flight really overlap: a 1/8 shard takes 2.2 ms with 4 queues and 1.6 ms with 24 (four frames in flight
# are 20 streams, plus torch's and RCCL's own).  So a multi-GPU run asks for 24; RT_HW_QUEUES overrides either way.
if os.environ.get("RT_HW_QUEUES"):
    os.environ["GPU_MAX_HW_QUEUES"] = os.environ["RT_HW_QUEUES"]
elif int(os.environ.get("WORLD_SIZE", "1")) > 1:
    os.environ["GPU_MAX_HW_QUEUES"] = "24"

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E, spec


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--cells", type=int, default=708)
    ap.add_argument("--lights", type=int, default=8)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--frames-in-flight", type=int, default=None)
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

    vw, vh = args.width, args.height
    t0 = time.perf_counter()
    scene = scenes.synthetic_terrain_scene(vw, vh, cells=args.cells, n_lights=args.lights)
    t_build = time.perf_counter() - t0
    t_upload = [0.0]

    def make_context():
        c = ra.Context(local_rank)
        t0 = time.perf_counter()
        c.upload(scene)
        t_upload[0] = time.perf_counter() - t0
        return c

    if args.frames_in_flight is None:
        # One GPU, whole frame: 4 frames in flight, each with four shadow/light chains of its own.  A shard's launches are
        # short, and what fills the chip then is more frames rather than more chains per frame: 8 frames with one
        # shadow/light chain each (16 streams) render a 1/8 shard 2 % faster, halves and quarters equally fast.
        args.frames_in_flight = 4 if world == 1 else 8
        if world > 1:
            os.environ.setdefault("RTGPU_SIDE_STREAMS", "1")
    depth = max(1, args.frames_in_flight)
    pipe = rd.FramePipeline(make_context, rank, world, device, depth=depth)
    t_upload = t_upload[0]
    info = ra.desc_info(scene.flatten())

    def sync():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    # one frame at a time first (latency figure; also warms every context up)
    last = {}

    def keep(f, st, frame):
        last["st"], last["frame"] = st, frame

    pipe.run(max(args.warmup, depth), keep)
    sync()
    t0 = time.perf_counter()
    alone = 3
    for _ in range(alone):
        pipe.run(1, keep)
    sync()
    ms_alone = (time.perf_counter() - t0) * 1e3 / alone

    pipe.run(args.warmup, keep)
    sync()
    t0 = time.perf_counter()
    stats = pipe.run(args.steps, keep)
    sync()
    elapsed = time.perf_counter() - t0
    st, frame = last["st"], last["frame"]
    ms_trace = sum(x.ms_trace for x in stats); ms_shadow = sum(x.ms_shadow for x in stats)
    ms_shade = sum(x.ms_shade for x in stats); ms_resolve = sum(x.ms_resolve for x in stats)
    ms_kernels = sum(x.ms_render for x in stats)
    launches = {"k_trace": sum(x.launches_trace for x in stats), "k_shadow": sum(x.launches_shadow for x in stats),
                "k_spawn": sum(x.launches_shade for x in stats)}

    rays = torch.tensor([st.rays_primary, st.rays_reflect, st.rays_transmit, st.rays_shadow], dtype=torch.int64, device=device)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    rays = [int(x) for x in rays.cpu()]
    elapsed = float(tmax.cpu()[0])
    rays_frame = sum(rays)
    ms_per_step = elapsed * 1e3 / args.steps
    value = rays_frame * args.steps / elapsed / 1e6

    visits = None
    if rank == 0 and world == 1:
        # what the GPU traversal really touched: one untimed frame through the instrumented kernel variants
        os.environ["RTGPU_COUNT_VISITS"] = "1"
        pipe.close()
        pipe = None
        cctx = ra.Context(local_rank)
        cctx.upload(scene)
        _, _, cst = cctx.render(ids=False)
        cctx.close()
        del os.environ["RTGPU_COUNT_VISITS"]
        visits = {"lane_node_visits": int(cst.nodes_visited), "lane_tri_tests": int(cst.tris_tested),
                  # wave_tri_steps: triangle records fetched per wave by the closest-hit packets, and rounds of 64
                  # (ray, leaf) pairs tested by the shadow packets (whose lanes fetch their own triangles)
                  "wave_node_fetches": int(cst.wave_nodes), "wave_tri_steps": int(cst.wave_tris),
                  # B_touched: node and triangle records the walks really fetched, at their real sizes (112 / 80
                  # bytes), a packet's shared fetch counted once
                  "bytes_touched": int(cst.bytes_fetched)}

    if rank == 0:
        frame_rgb = frame.cpu().numpy().reshape(vh, vw, 3)
        cores = int(os.environ.get("OMP_NUM_THREADS") or min(os.cpu_count() or 1, 16))
        # N == 1: the timed CPU baseline.  N > 1: only a short counting pass for the roofline numerator.
        cpu = cpu_leg(scene, vw, vh, args.cpu_seconds if world == 1 else 3.0, cores, frame_rgb)

        # Roofline of the dominant kernel.  Algorithmic bytes are counted under the reference's exhaustive semantics
        # (32 B per BVH node test, 36 per triangle, 16/28/24 per sphere/plane/box, 3 per texel) on the sampled pixels
        # and scaled to the frame: k_trace owns the closest-hit casts, k_shadow the shadow casts, k_shade the
        # refraction helper intersects and the texel fetches.  With N ranks each launch handles 1/N of the frame;
        # a frame launches each kernel once per trace_ray depth that has rays.
        per_kernel = {"k_trace": (cpu["bytes_closest"], ms_trace), "k_shadow": (cpu["bytes_shadow"], ms_shadow),
                      "k_spawn": (cpu["bytes_helper"] + cpu["bytes_texels"], ms_shade)}
        kernels = {}
        for name, (bytes_frame, ms) in per_kernel.items():
            n_l = max(launches[name], 1)
            avg_ms = ms / n_l
            bytes_launch = bytes_frame / world / (n_l / max(args.steps, 1))
            gbs = bytes_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            kernels[name] = {"avg_launch_ms": avg_ms, "launches_per_frame": n_l / max(args.steps, 1), "bytes_per_launch": bytes_launch,
                             "GB/s": gbs, "ms_per_frame": ms / max(args.steps, 1)}
        # The dominant kernel: the one that owns most of the frame's algorithmic bytes (the shadow casts: 8 per lit hit).
        # Launch durations cannot decide it: the launches of different kernels overlap (side streams, frames in flight),
        # each stretches by what it waits for the others, and the order of two sums flips from run to run.  With the
        # kernels of a frame run one after the other (RTGPU_SERIAL=1, tools/prof_frame.py) it is the same kernel: 60 %
        # of the frame's kernel time.
        dom = max(per_kernel, key=lambda k: per_kernel[k][0])
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("kernel") == dom and tj.get("n_gpus") == world and tj.get("workload_cells") == args.cells:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/s (primary+secondary)", "value": value, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"synthetic {info['n_tris']}-triangle height field ({args.cells}x{args.cells} cells) + 1 glass sphere + "
                            f"{args.lights} point lights with shadows, {vw}x{vh}, max_ray_depth 4 (BASELINE config 4)",
                "resolution": [vw, vh], "triangles": info["n_tris"], "bvh_nodes": info["n_nodes"], "lights": args.lights,
                "parallelism": f"64x64-pixel supertiles interleaved over {world} GPU(s), gather to rank 0",
                "frames_in_flight": depth,
            },
            "rays_per_frame": {"primary": rays[0], "reflect": rays[1], "transmit": rays[2], "shadow": rays[3], "total": rays_frame},
            "ms_per_frame_alone": ms_alone,
            "ms_per_frame_kernels_rank0": ms_kernels / args.steps,
            # sums of the launches' own durations; the phases (and frames) overlap, so they exceed the frame time
            "ms_phase_rank0": {"trace": ms_trace / args.steps, "shadow": ms_shadow / args.steps, "spawn+light": ms_shade / args.steps,
                               "resolve": ms_resolve / args.steps},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["GB/s"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": kernels[dom]["GB/s"] / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_ms": kernels[dom]["avg_launch_ms"], "algorithmic_bytes_per_launch": kernels[dom]["bytes_per_launch"],
                         "kernels": kernels,
                         "note": "achieved = bytes the reference's exhaustive walk would read for this launch's casts / launch time; "
                                 "the GPU walk prunes by distance and stops shadow casts at the first blocker, so it reads far less "
                                 "(gpu_visits) and the figure can exceed the HBM peak; the kernel is bound by instruction issue (profiles/)"},
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
