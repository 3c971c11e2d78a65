"""Throughput of one GPU rendering shard 0 of N with 1..3 frames in flight (threads, one context each): how much of
the per-frame tail the next frame's work fills in at small shard sizes."""
import os
import sys
import threading
import time

os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("RT_HW_QUEUES", "24")   # as bench.py: every stream of every context on a queue of its own
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import raytracer_amd as ra  # noqa: E402
from raytracer_amd import dist as rd, scenes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
shard = int(os.environ.get("SHARD", "0"))     # which of the n shards
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 12
batch = int(os.environ.get("BATCH", "1"))      # frames per set of launches (rt_render_shard_batch)
s = scenes.synthetic_terrain_scene(3840, 2160)
for depth in [int(x) for x in os.environ.get("DEPTHS", "1,2,3,4").split(",")]:
    ctxs = [ra.Context(0) for _ in range(depth)]
    for c in ctxs:
        c.upload(s)
    bufs = [torch.zeros(batch * rd.max_shard_slots(3840, 2160, n) * 3, dtype=torch.uint8, device="cuda:0") for _ in range(depth)]
    cams = ctxs[0].camera_array([s.camera] * batch)

    last = [None]

    def work(k, count):
        for _ in range(count):
            if batch > 1:
                last[0] = ctxs[k].render_shard_batch(shard, n, cams, bufs[k].data_ptr())
            else:
                last[0] = ctxs[k].render_shard(shard, n, bufs[k].data_ptr())

    for k in range(depth):
        work(k, 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ts = [threading.Thread(target=work, args=(k, frames)) for k in range(depth)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3 / (frames * depth * batch)
    print(f"shard {shard} of {n}, {depth} x {batch} in flight: {dt:.3f} ms per frame, {last[0].rays // batch} rays per frame", flush=True)
    for c in ctxs:
        c.close()
