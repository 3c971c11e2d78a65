import sys, os
sys.path.insert(0, os.getcwd())
import importlib.util
spec = importlib.util.spec_from_file_location("bench_module", "bench.py"); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import raytracer_amd as ra
from raytracer_amd import scenes
cfg = sys.argv[1]
class A: cells=708; lights=8
builder,(w,h) = bench.CONFIGS[cfg]
s,_ = builder(w,h,A)
base = s.camera
ctx = ra.Context(0); ctx.upload(s)
for k in range(-3,5):
    cams = ctx.camera_array([bench.camera_path(base, k*10+f) for f in range(10)])
    fr,_,st = ctx.render_frames(cams)
    print(k, list(st.rays_per_level)[:6], "twice", st.respeculated, "spec", st.speculative, flush=True)
