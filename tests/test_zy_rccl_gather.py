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
