"""Per-ray cost of a 1/N shard against the whole frame at several max_ray_depth values, 3 frames in flight:
separates the cost of the deep, small levels from everything else."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import raytracer_amd as ra  # noqa: E402
from raytracer_amd import dist as rd, scenes  # noqa: E402

frames, depth_in_flight = 10, 3
for mrd in (0, 1, 2, 4):
    s = scenes.synthetic_terrain_scene(3840, 2160)
    s.set_options(max_ray_depth=mrd)
    res = {}
    for n in (1, 8):
        ctxs = [ra.Context(0) for _ in range(depth_in_flight)]
        for c in ctxs:
            c.upload(s)
        bufs = [torch.zeros(rd.max_shard_slots(3840, 2160, n) * 3, dtype=torch.uint8, device="cuda:0") for _ in ctxs]
        rays = [0]

        def work(k, count):
            for _ in range(count):
                st = ctxs[k].render_shard(0, n, bufs[k].data_ptr())
            rays[0] = st.rays

        for k in range(len(ctxs)):
            work(k, 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ts = [threading.Thread(target=work, args=(k, frames)) for k in range(len(ctxs))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / (frames * len(ctxs))
        res[n] = (ms, rays[0])
        for c in ctxs:
            c.close()
    (m1, r1), (m8, r8) = res[1], res[8]
    print(f"max_ray_depth {mrd}: whole {m1:.3f} ms ({r1 / m1 / 1e3:.0f} Mrays/s)  1/8 {m8:.3f} ms ({r8 / m8 / 1e3:.0f} Mrays/s)  ratio {m1 / m8:.2f}", flush=True)
