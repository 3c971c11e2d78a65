"""Scene upload time of the benchmark scene: everything, and with the meshes kept (a frame of a sequence)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra  # noqa: E402
from raytracer_amd import scenes  # noqa: E402

s = scenes.synthetic_terrain_scene(3840, 2160)
with ra.Context(0) as ctx:
    for keep in (False, False, True, True):
        ctx.upload(s, keep_meshes=keep)
        _, _, st = ctx.render()
        print(f"keep_meshes={keep}: upload {st.ms_upload:.2f} ms, frame {st.ms_render:.2f} ms, rays {st.rays}", flush=True)
