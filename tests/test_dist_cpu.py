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


def test_shard_rows_cycle_through_every_column_phase():
    """the numbering rotates each row of supertiles so that the owner of column x in row y is (x + g*y) mod n with
    g coprime to n: in n consecutive rows a shard owns every column phase once (4K frame, 8 shards: a shard no
    longer sits on the same columns in every other row)"""
    import math
    from raytracer_amd import dist as rd
    side = rd.TILE * rd.supertile()
    for vw, vh, n in [(3840, 2160, 8), (3840, 2160, 4), (1920, 1080, 2), (2560, 1440, 3), (7680, 4320, 8), (1237, 911, 5)]:
        supers_x, supers_y = -(-vw // side), -(-vh // side)
        owner = np.full((supers_y, supers_x), -1)
        for s in range(n):
            idx = rd.shard_pixel_indices(vw, vh, s, n)
            idx = idx[idx >= 0]
            owner[(idx // vw) // side, (idx % vw) // side] = s
        assert (owner >= 0).all()
        if supers_x <= (n - 1) * (n - 1) + 1 or supers_y < n:
            continue
        # right of the wrap-around seam of the rotated rows (it moves right by fewer than n columns a row), the owner
        # moves by one per column and by g per row
        x0 = supers_x - 1
        step = (owner[1, x0] - owner[0, x0]) % n
        assert math.gcd(int(step), n) == 1
        assert (owner[0, x0] - owner[0, x0 - 1]) % n == 1
        phases = {int(owner[y, x0]) for y in range(n)}
        assert phases == set(range(n))


class _OracleCtx:
    """Stands in for a GPU context on the CPU: the shard comes from the oracle, the de-tile from its host statement.
    Frame f is rendered with the sphere row shifted by f, so frames differ and their order can be checked."""

    def __init__(self, vw, vh):
        self.vw, self.vh = vw, vh
        self.frame_no = 0

    def render_shard(self, rank, world, ptr):
        import ctypes
        import oracle
        from raytracer_amd import dist as rd
        from raytracer_amd import scenes
        f = self.frame_no
        s = scenes.small_mixed_scene(self.vw, self.vh, cells=4 + f)
        slots = rd.max_shard_slots(self.vw, self.vh, world)
        idx = rd.shard_pixel_indices(self.vw, self.vh, rank, world)
        ok = idx >= 0
        local = np.zeros((slots, 3), dtype=np.uint8)
        rgb, _, st, _ = oracle.render(s, 1, pixels=idx[ok].astype(np.uint32), ids=False)
        local[: len(idx)][ok] = rgb
        ctypes.memmove(ptr, local.ctypes.data, local.nbytes)
        return st

    def detile(self, gptr, world, slots, elem, fptr):
        import ctypes
        from raytracer_amd import dist as rd
        g = np.ctypeslib.as_array((ctypes.c_uint8 * (world * slots * elem)).from_address(gptr))
        out = rd.detile_host(g, self.vw, self.vh, world, slots, elem)
        ctypes.memmove(fptr, out.ctypes.data, out.nbytes)

    # -- batches of frames that differ in their camera (rt_render_shard_batch / rt_detile_frames)
    def camera_array(self, cameras):
        return list(cameras)

    def render_shard_batch(self, rank, world, cams, ptr):
        import ctypes
        import oracle
        from raytracer_amd import dist as rd
        from raytracer_amd import scenes
        slots = rd.max_shard_slots(self.vw, self.vh, world)
        idx = rd.shard_pixel_indices(self.vw, self.vh, rank, world)
        ok = idx >= 0
        local = np.zeros((len(cams), slots, 3), dtype=np.uint8)
        st = None
        for f, cam in enumerate(cams):
            s = scenes.small_mixed_scene(self.vw, self.vh, cells=5)
            cam.vw, cam.vh = self.vw, self.vh
            s.set_camera(cam)
            rgb, _, st, _ = oracle.render(s, 1, pixels=idx[ok].astype(np.uint32), ids=False)
            local[f][: len(idx)][ok] = rgb
        ctypes.memmove(ptr, local.ctypes.data, local.nbytes)
        return st

    def detile_frames(self, gptr, world, n_frames, slots, elem, fptr):
        import ctypes
        from raytracer_amd import dist as rd
        g = np.ctypeslib.as_array((ctypes.c_uint8 * (world * n_frames * slots * elem)).from_address(gptr)).reshape(world, n_frames, slots, elem)
        out = np.stack([rd.detile_host(np.ascontiguousarray(g[:, f]), self.vw, self.vh, world, slots, elem) for f in range(n_frames)])
        ctypes.memmove(fptr, out.ctypes.data, out.nbytes)

    def close(self):
        pass


def _path_camera(f):
    from raytracer_amd.scene import Camera
    return Camera(origin=(0.2 * f, 0.1 * f, 0.), yaw=0.04 * f, pitch=-0.02 * f)


def _batch_worker(rank, world, port, vw, vh, n_frames, batch, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracer_amd import dist as rd
        pipe = rd.FramePipeline(lambda: _OracleCtx(vw, vh), rank, world, torch.device("cpu"), depth=2, batch=batch)
        frames, covered = [], []

        def on_frame(f, st, frame):
            covered.append(st.frames)
            if rank == 0:
                frames.append(frame.numpy().reshape(vh, vw, 3).copy())

        stats = pipe.run(n_frames, on_frame, camera_of=_path_camera)
        pipe.close()
        assert len(stats) == -(-n_frames // batch) and len(covered) == n_frames
        # sets of as equal a size as possible: 7 frames in sets of at most 3 are 3 + 2 + 2
        n_sets = -(-n_frames // batch)
        sizes = [n_frames // n_sets + (1 if j < n_frames % n_sets else 0) for j in range(n_sets)]
        assert covered == [b for b in sizes for _ in range(b)] and max(sizes) <= batch
        if rank == 0:
            np.save(out_path, np.stack(frames))
    finally:
        dist.destroy_process_group()


def test_gloo_batched_pipeline_delivers_every_frame_of_a_camera_path(tmp_path):
    """at most 3 frames per set of launches (7 frames: 3 + 2 + 2), two sets in flight, world size 2"""
    sys.path.insert(0, ROOT)
    import oracle
    from raytracer_amd import scenes
    vw, vh, n_frames = 36, 20, 7
    port = 33500 + (os.getpid() % 2000)
    out = str(tmp_path / "frames.npy")
    mp.spawn(_batch_worker, args=(2, port, vw, vh, n_frames, 3, out), nprocs=2, join=True)
    frames = np.load(out)
    assert frames.shape[0] == n_frames
    for f in range(n_frames):
        s = scenes.small_mixed_scene(vw, vh, cells=5)
        cam = _path_camera(f)
        cam.vw, cam.vh = vw, vh
        s.set_camera(cam)
        full, _, _, _ = oracle.render(s, 1)
        assert np.array_equal(frames[f], full), f


def _pipeline_worker(rank, world, port, vw, vh, n_frames, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracer_amd import dist as rd
        ctxs = []

        def make():
            ctxs.append(_OracleCtx(vw, vh))
            return ctxs[-1]

        pipe = rd.FramePipeline(make, rank, world, torch.device("cpu"), depth=2)
        frames = []

        def on_frame(f, st, frame):
            if rank == 0:
                frames.append(frame.numpy().reshape(vh, vw, 3).copy())

        # the fake contexts learn their frame number from the slot: slot k renders frames k, k+2, ...
        orig = [c.render_shard for c in ctxs]
        for k, c in enumerate(ctxs):
            def rs(rank_, world_, ptr, c=c, k=k, o=orig[k]):
                st = o(rank_, world_, ptr)
                c.frame_no += 2
                return st
            c.frame_no = k
            c.render_shard = rs
        stats = pipe.run(n_frames, on_frame)
        pipe.close()
        assert len(stats) == n_frames
        if rank == 0:
            np.save(out_path, np.stack(frames))
    finally:
        dist.destroy_process_group()


def test_gloo_frame_pipeline_keeps_frame_order(tmp_path):
    """two frames in flight per rank, world size 2: every frame equals its single-process render, in order"""
    sys.path.insert(0, ROOT)
    import oracle
    from raytracer_amd import scenes
    vw, vh, n_frames = 40, 24, 5
    port = 31500 + (os.getpid() % 2000)
    out = str(tmp_path / "frames.npy")
    mp.spawn(_pipeline_worker, args=(2, port, vw, vh, n_frames, out), nprocs=2, join=True)
    frames = np.load(out)
    assert frames.shape[0] == n_frames
    for f in range(n_frames):
        full, _, _, _ = oracle.render(scenes.small_mixed_scene(vw, vh, cells=4 + f), 1)
        assert np.array_equal(frames[f], full), f


def test_bench_default_sets_of_launches():
    """bench.py's choice of frames per set of launches: on one GPU two 4K frames' worth of pixels; sharded, half of the
    run, within ten whole frames' worth of rays and the library's limit"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.default_batch(20, 1, True) == 2 and bench.default_batch(40, 1, True, 1920 * 1080) == 8 and bench.default_batch(1, 1, True) == 1
    assert bench.default_batch(20, 8, True) == 10 and bench.default_batch(400, 8, True) == 64
    assert bench.default_batch(5, 2, True) == 3 and bench.default_batch(1, 4, True) == 1
    assert bench.default_batch(20, 1, False, 7680 * 4320) == 1 and bench.default_batch(20, 8, False) == 8
