"""The collective step of a sharded render through RCCL itself (torch.distributed's "nccl" backend), on the one GPU a
test box has: a process group of one rank.  The CPU tests run the same step with two and three gloo ranks; what only
a GPU can show is that the calls the N > 1 path makes -- a process group bound to a device, barrier, all_reduce, and
gather into per-rank views of one device buffer -- are ones this RCCL build takes, and that the gathered bytes
de-tile to the frame one context renders alone."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent(
    """
    import sys
    import numpy as np
    import torch
    import torch.distributed as dist
    import raytracer_amd as ra
    from raytracer_amd import dist as rd, scenes

    port = int(sys.argv[1])
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=device)
    try:
        assert dist.get_backend() == "nccl"
        dist.barrier()
        t = torch.tensor([3.5], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        n = torch.tensor([1, 2, 3, 4], dtype=torch.int64, device=device)
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
        assert float(t.cpu()[0]) == 3.5 and n.cpu().tolist() == [1, 2, 3, 4]

        vw, vh = 150, 85
        s = scenes.small_mixed_scene(vw, vh, cells=8)
        ctx = ra.Context(0)
        ctx.upload(s)
        whole, _, st0 = ctx.render()
        r = rd.ShardedRenderer(ctx, 0, 1, device, batch=3)
        # three frames in one set of launches, gathered and put together at once; then a shorter set of one frame
        # through the same buffers (what the last set of a run looks like)
        for frames in (3, 1, 2):
            st = r.render_local(frames)
            assert st.rays == frames * st0.rays
            part = r.local[: frames * r.slots * 3]
            got = rd.gather_shards(part, 0, 1, lambda g: r.detile(g, frames))
            got = got.cpu().numpy()[: frames * vh * vw * 3].reshape(frames, vh, vw, 3)
            for f in range(frames):
                assert np.array_equal(got[f], whole), (frames, f)
        ctx.close()
        print("rccl gather ok")
    finally:
        dist.destroy_process_group()
    """
)


@pytest.mark.gpu
def test_gather_through_rccl_with_a_group_of_one_rank():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-u", "-c", CHILD, str(port)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=110)
    assert p.returncode == 0 and "rccl gather ok" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


TWO_RANKS = textwrap.dedent(
    """
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    import raytracer_amd as ra
    from raytracer_amd import dist as rd, scenes

    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    device = torch.device("cuda", 0)                  # every rank on the one GPU of the box
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        vw, vh, n_frames = 203, 117, 7
        s = scenes.small_mixed_scene(vw, vh, cells=12)

        def make():
            c = ra.Context(0)
            c.upload(s)
            return c

        def cam(f):
            return ra.Camera(origin=(0.3 + 0.15 * f, 2.5 + 0.05 * f, 9. - 0.1 * f), yaw=0.05 + 0.02 * f, pitch=-0.2 - 0.01 * f, fov=60.)

        # sets of at most three frames (7 = 3 + 2 + 2), two sets in flight: rt_render_shard_batch, the gather (staged
        # through the host under gloo), rt_detile_frames, delivery in frame order
        pipe = rd.FramePipeline(make, rank, world, device, depth=2, batch=3)
        frames, rays = [], []

        def on_frame(f, st, frame):
            if rank == 0:
                frames.append(frame.cpu().numpy().reshape(vh, vw, 3).copy())

        stats = pipe.run(n_frames, on_frame, camera_of=cam)
        rays = torch.tensor([sum(int(x.rays) for x in stats)], dtype=torch.int64)
        dist.all_reduce(rays)
        # single frames through the scene's own camera as well (rt_render_shard + rt_detile)
        single = rd.FramePipeline(make, rank, world, device, depth=1, batch=1)
        still = []
        single.run(2, lambda f, st, frame: still.append(None if frame is None else frame.cpu().numpy().reshape(vh, vw, 3).copy()))
        single.close()
        pipe.close()
        if rank == 0:
            assert len(frames) == n_frames and still[0] is not None
            np.savez(out, frames=np.stack(frames), still=np.stack(still), rays=rays.numpy())
        print(f"rank {rank} ok")
    finally:
        dist.destroy_process_group()
    """
)


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_deliver_the_single_process_frames(tmp_path, gpu_ctx):
    """The N > 1 path of the one-process-per-GPU runs -- shard batches, gather to rank 0, de-tile of a set of frames,
    frame order -- with real kernels: two fresh processes share the box's one GPU and exchange their shards over gloo
    (RCCL cannot put two ranks on one device).  Rank 0's frames must be the frames one context renders alone."""
    import numpy as np
    import raytracer_amd as ra
    from raytracer_amd import scenes
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "frames.npz")
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-u", "-c", TWO_RANKS, str(r), "2", str(port), out], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    logs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=150)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(so[-1500:] + se[-3000:])
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(logs)
    got = np.load(out)
    vw, vh = 203, 117
    s = scenes.small_mixed_scene(vw, vh, cells=12)
    gpu_ctx.upload(s)
    want_still, _, _ = gpu_ctx.render()
    assert all(np.array_equal(f, want_still) for f in got["still"])
    total = 0
    for f in range(7):
        cam = ra.Camera(vw=vw, vh=vh, origin=(0.3 + 0.15 * f, 2.5 + 0.05 * f, 9. - 0.1 * f), yaw=0.05 + 0.02 * f, pitch=-0.2 - 0.01 * f, fov=60.)
        s.set_camera(cam)
        gpu_ctx.upload(s)
        want, _, st = gpu_ctx.render()
        assert np.array_equal(got["frames"][f], want), f
        total += st.rays
    assert int(got["rays"][0]) == total
