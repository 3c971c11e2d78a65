"""How long rt_render's copy of the finished frame into the caller's memory takes, by thread count
(RTGPU_COPY_THREADS; 0 = one plain hipMemcpy), with the system's HIP runtime and with the one torch brings along.

    python tools/copy_sweep.py [width height]        (on the GPU box)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = """
import sys
sys.path.insert(0, %r)
if %d:
    import torch
import raytracer_amd as ra
from raytracer_amd import scenes
s = scenes.synthetic_terrain_scene(%d, %d, cells=64)
c = ra.Context(0); c.upload(s)
import time
for _ in range(8):
    t0 = time.perf_counter()
    _, _, st = c.render()
    if %d == 0:
        print("total %%.3f ms" %% ((time.perf_counter() - t0) * 1e3 - st.ms_render))
c.close()
"""
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
for with_torch in (0, 1):
    for chunk in (0,):
        for threads in (0, 1, 2, 4, 8):
            env = dict(os.environ, RTGPU_DEBUG_COPY="1", RTGPU_COPY_THREADS=str(threads))
            p = subprocess.run([sys.executable, "-c", CHILD % (ROOT, with_torch, w, h, threads)], env=env, capture_output=True, text=True, timeout=200)
            ms = [float(m) for m in re.findall(r"copy \d+ MB: ([0-9.]+) ms", p.stderr)] or [float(m) for m in re.findall(r"total ([0-9.]+) ms", p.stdout)]
            if not ms:
                print("no output", p.stderr[-300:]); continue
            ms = ms[2:]
            print(f"torch {with_torch} threads {threads}: min {min(ms):.2f} median {sorted(ms)[len(ms)//2]:.2f} max {max(ms):.2f} ms", flush=True)
