"""One GPU renders shard 0 of N (N = 1, 2, 4, 8) of the benchmark frame: kernel time against host wall time per
frame, i.e. the per-frame overhead (launches, per-level count read-backs, de-tile) that limits strong scaling."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import raytracer_amd as ra  # noqa: E402
from raytracer_amd import dist as rd, scenes  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 10
s = scenes.synthetic_terrain_scene(3840, 2160)
ctx = ra.Context(0)
ctx.upload(s)
dev = torch.device("cuda", 0)
for n in (1, 2, 4, 8):
    r = rd.ShardedRenderer(ctx, 0, 1, dev)
    slots = rd.max_shard_slots(3840, 2160, n)
    local = torch.zeros(slots * 3, dtype=torch.uint8, device=dev)
    for _ in range(3):
        ctx.render_shard(0, n, local.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ms_k = 0.0
    for _ in range(frames):
        st = ctx.render_shard(0, n, local.data_ptr())
        ms_k += st.ms_render
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3 / frames
    print(f"shard 0 of {n}: kernels {ms_k / frames:.3f} ms  wall {wall:.3f} ms  overhead {wall - ms_k / frames:.3f} ms  rays {st.rays}", flush=True)
ctx.close()
