"""Renders a few frames of one of the parity / BASELINE scenes: the program to put under rocprofv3.
usage: prof_scene.py <huge_grid|terrain5|refraction|mixed> [w h] [frames]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra  # noqa: E402
from raytracer_amd import scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "huge_grid"
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 3
s = {"huge_grid": lambda: scenes.huge_grid_scene(w, h, 12), "terrain5": lambda: scenes.terrain_scene(w, h),
     "refraction": lambda: scenes.refraction_scene(w, h), "mixed": lambda: scenes.small_mixed_scene(w, h)}[name]()
with ra.Context(0) as ctx:
    ctx.upload(s)
    for _ in range(frames):
        _, _, st = ctx.render()
    print(f"{name} {w}x{h}: {st.ms_render:.2f} ms (trace {st.ms_trace:.2f} shadow {st.ms_shadow:.2f} shade {st.ms_shade:.2f}) rays {st.rays} "
          f"levels {list(st.rays_per_level)[:6]} shadow rays {st.rays_shadow}")
