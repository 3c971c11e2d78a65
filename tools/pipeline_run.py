"""One GPU plays rank 0 of N through the real FramePipeline (the gather left out): K frames of shard 0, B frames per set of
launches, D slots.  What bench.py --gpus N asks of one rank, without the other ranks.
usage: pipeline_run.py N K B D"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import raytracer_amd as ra  # noqa: E402
from raytracer_amd import dist as rd, scenes  # noqa: E402

n, k, b, d = (int(x) for x in sys.argv[1:5])
s = scenes.synthetic_terrain_scene(3840, 2160)
rd.gather_frame = lambda local, vw, vh, rank, world, detile: None      # (no other ranks here)
dev = torch.device("cuda", 0)


def make_context():
    c = ra.Context(0)
    c.upload(s)
    return c


pipe = rd.FramePipeline(make_context, 0, n, dev, depth=d, batch=b)
pipe.run(max(5, d * b))
pipe.run(k)
best = None
for _ in range(3):
    pipe.run(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.run(k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3 / k
    best = dt if best is None else min(best, dt)
print(f"shard 0 of {n}: {k} frames, sets of {b}, {d} slots: {best:.3f} ms per frame (best of 3)", flush=True)
pipe.close()
