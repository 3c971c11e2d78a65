"""The N>1 path on CPU: world_size-2 (and 3) gloo process groups exercise the shard map, the equal-count gather and
the host statement of the de-tile step.  Each rank renders ITS shard with the CPU oracle (tests may use it), so the
assembled frame must equal the single-process frame bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, vw, vh, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from raytracer_amd import dist as rd
        from raytracer_amd import scenes
        s = scenes.small_mixed_scene(vw, vh, cells=6)
        slots = rd.max_shard_slots(vw, vh, world)
        idx = rd.shard_pixel_indices(vw, vh, rank, world)
        ok = idx >= 0
        local = np.zeros((slots, 3), dtype=np.uint8)
        rgb, _, st, _ = oracle.render(s, 1, pixels=idx[ok].astype(np.uint32), ids=False)
        local[: len(idx)][ok] = rgb
        t = torch.from_numpy(local.reshape(-1))

        def detile(g):
            return rd.detile_host(g.numpy(), vw, vh, world, slots, 3)

        frame = rd.gather_frame(t, vw, vh, rank, world, detile)
        # rays add up across ranks
        rays = torch.tensor([st.rays], dtype=torch.int64)
        dist.all_reduce(rays)
        if rank == 0:
            np.save(out_path, frame)
            np.save(out_path + ".rays.npy", rays.numpy())
        else:
            assert frame is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,vw,vh", [(2, 50, 37), (3, 33, 18)])
def test_gloo_sharded_frame_equals_single_process(tmp_path, world, vw, vh):
    sys.path.insert(0, ROOT)
    import oracle
    from raytracer_amd import scenes
    port = 29500 + (os.getpid() % 2000) + world
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, port, vw, vh, out), nprocs=world, join=True)
    frame = np.load(out)
    full, _, st, _ = oracle.render(scenes.small_mixed_scene(vw, vh, cells=6), 1)
    assert np.array_equal(frame, full)
    assert int(np.load(out + ".rays.npy")[0]) == st.rays


def test_shard_map_partitions_the_image():
    from raytracer_amd import dist as rd
    for vw, vh in [(1, 1), (16, 16), (17, 5), (50, 37), (256, 144)]:
        for n in (1, 2, 3, 8, 11):
            seen = np.zeros(vw * vh, dtype=np.int32)
            for s in range(n):
                idx = rd.shard_pixel_indices(vw, vh, s, n)
                assert len(idx) == rd.shard_slots(vw, vh, s, n) <= rd.max_shard_slots(vw, vh, n)
                np.add.at(seen, idx[idx >= 0], 1)
            assert (seen == 1).all()                        # every pixel in exactly one shard
