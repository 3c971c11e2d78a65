"""One process, N GPUs, through the C ABI alone (rt_ctx_create_multi + rt_render_frames): the benchmark sequence
(BASELINE config 4) rendered in sets of frames that every device shards, gathered to the first device inside the
library (RCCL) and returned as whole frames in host memory.

    python tools/multi_bench.py N [frames_per_set] [sets]

For a node with several GPUs; on a one-GPU box N = 1 runs the same calls without a gather.  Prints one JSON line.
The last frame is compared with the one-device render of the same camera."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as ra  # noqa: E402
from raytracer_amd import scenes  # noqa: E402
import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
per_set = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sets = int(sys.argv[3]) if len(sys.argv) > 3 else 4
vw, vh = int(os.environ.get("VW", 3840)), int(os.environ.get("VH", 2160))
devices = [int(d) for d in os.environ["DEVICES"].split(",")] if os.environ.get("DEVICES") else list(range(n))
s = scenes.synthetic_terrain_scene(vw, vh, cells=int(os.environ.get("CELLS", 708)))
base = s.camera
ctx = ra.Context(devices=devices)
t0 = time.perf_counter()
ctx.upload(s)
t_upload = time.perf_counter() - t0
cams = lambda first: ctx.camera_array([bench.camera_path(base, first + f) for f in range(per_set)])
ctx.render_frames(cams(0))                      # measured set: level sizes learned
ctx.render_frames(cams(per_set))                # first speculative set, buffers in place
rays = 0
ms_kernels = 0.0
t0 = time.perf_counter()
for k in range(sets):
    frames, _, st = ctx.render_frames(cams((k + 2) * per_set))
    rays += st.rays
    ms_kernels += st.ms_render
dt = time.perf_counter() - t0
twice = int(st.respeculated)
last = bench.camera_path(base, (sets + 2) * per_set - 1)
last.vw, last.vh = vw, vh
s.set_camera(last)
with ra.Context(devices[0]) as one:
    one.upload(s)
    want = one.render()[0]
same = bool(np.array_equal(frames[-1], want))
ctx.close()
print(json.dumps({"metric": "Mrays/s (primary+secondary), pixels in host memory", "value": rays / dt / 1e6, "n_gpus": len(devices), "devices": devices,
                  "ms_per_frame": dt * 1e3 / (sets * per_set),
                  "ms_per_frame_kernels_slowest_device": ms_kernels / (sets * per_set), "frames_per_set": per_set, "sets": sets, "resolution": [vw, vh],
                  "upload_s": t_upload, "frames_rendered_twice": twice, "last_frame_equals_one_device_render": same}))
