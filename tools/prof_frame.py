"""Renders a few frames of the benchmark scene with nothing else around them: the program to put under rocprofv3.
The view is frame bench.ROOFLINE_FRAME of bench.py's camera path -- the frame whose launches bench.py times for its
roofline -- unless STILL=1 asks for the scene's own camera.
SHARDS=n in the environment renders shard 0 of n instead of the whole frame."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra  # noqa: E402
from raytracer_amd import scenes  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cells = int(sys.argv[2]) if len(sys.argv) > 2 else 708
w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (3840, 2160)
s = scenes.synthetic_terrain_scene(w, h, cells=cells)
if not os.environ.get("STILL"):
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cam = bench.camera_path(s.camera, bench.ROOFLINE_FRAME)
    cam.vw, cam.vh = w, h
    s.set_camera(cam)
ctx = ra.Context(0)
ctx.upload(s)
shards = int(os.environ.get("SHARDS", "1"))
if shards > 1:
    import torch
    from raytracer_amd import dist as rd
    buf = torch.zeros(rd.max_shard_slots(w, h, shards) * 3, dtype=torch.uint8, device="cuda:0")
for _ in range(frames):
    if shards > 1:
        st = ctx.render_shard(0, shards, buf.data_ptr())
    else:
        rgb, _, st = ctx.render()
print(f"frames {frames} ms {st.ms_render:.3f} trace {st.ms_trace:.3f} shadow {st.ms_shadow:.3f} shade {st.ms_shade:.3f} rays {st.rays} nodes {st.nodes_visited} tris {st.tris_tested} wnodes {st.wave_nodes} wtris {st.wave_tris} long {st.long_trace} {st.long_shadow} max steps lane {st.max_lane_steps} wave {st.max_wave_steps}")
ctx.close()
