import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import raytracer_amd as ra
from raytracer_amd import scenes
s = scenes.small_mixed_scene(96, 72)
ctx = ra.Context(0); ctx.upload(s)
for _ in range(3): rgb, _, st = ctx.render()
print(st.ms_render, st.rays)
ctx.close()
