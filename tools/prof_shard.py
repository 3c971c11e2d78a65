"""Renders shard 0 of N of the benchmark frame a few times: the program to put under rocprofv3 --kernel-trace when
looking at per-launch durations of a small shard (the strong-scaling regime)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import raytracer_amd as ra  # noqa: E402
from raytracer_amd import dist as rd, scenes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
s = scenes.synthetic_terrain_scene(3840, 2160)
ctx = ra.Context(0)
ctx.upload(s)
local = torch.zeros(rd.max_shard_slots(3840, 2160, n) * 3, dtype=torch.uint8, device="cuda:0")
for _ in range(frames):
    st = ctx.render_shard(0, n, local.data_ptr())
print(f"shard 0 of {n}: ms {st.ms_render:.3f} trace {st.ms_trace:.3f} shadow {st.ms_shadow:.3f} shade {st.ms_shade:.3f} per level {list(st.rays_per_level)} long {st.long_trace} {st.long_shadow} max steps lane {st.max_lane_steps} wave {st.max_wave_steps}")
ctx.close()
