"""Config 4 (1,002,528 triangles): what it costs to get the mesh's tree onto the device, and what the frame costs
with each tree.  Host: Mesh::generate_sbvh (binned SAH, mesh.rs:257-272) + upload (four-wide re-emission on the host);
device: rt_mesh::nodes == NULL (linear BVH built by the library at upload); refit: rt_mesh::unchanged == 2."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import raytracer_amd as ra  # noqa: E402
from raytracer_amd import scenes  # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 708


def timed(f, n=1):
    t0 = time.perf_counter()
    for _ in range(n):
        r = f()
    return (time.perf_counter() - t0) * 1e3 / n, r


for device_bvh in (False, True):
    t_scene, s = timed(lambda: scenes.synthetic_terrain_scene(3840, 2160, cells=cells, device_bvh=device_bvh))
    t_tree = None
    if not device_bvh:
        t_tree, _ = timed(lambda: s.mesh.generate_sbvh())          # the host build alone, once more
    with ra.Context(0) as ctx:
        ctx.upload(s)
        t_up, _ = timed(lambda: ctx.upload(s), 3)
        ctx.render()
        t_frame, (rgb, _, st) = timed(lambda: ctx.render(), 5)
        v = s.mesh.verts
        v[:, 1] += 0.5 * np.sin(v[:, 0] * 0.05)
        s.mesh.set_verts(v)
        ctx.upload(s, keep_meshes="refit")
        t_refit, _ = timed(lambda: ctx.upload(s, keep_meshes="refit"), 3)
        t_frame2, (_, _, st2) = timed(lambda: ctx.render(), 3)
    print(f"{'device' if device_bvh else 'host  '}-built tree: scene construction {t_scene:.1f} ms"
          + (f" (of which generate_sbvh {t_tree:.1f})" if t_tree else "")
          + f"; rt_scene_upload {t_up:.1f} ms; frame {t_frame:.2f} ms ({st.rays} rays, kernels {st.ms_render:.2f} ms); "
          f"refit upload {t_refit:.1f} ms; frame after refit {t_frame2:.2f} ms", flush=True)
