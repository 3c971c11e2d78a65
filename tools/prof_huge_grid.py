import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import raytracer_amd as ra
from raytracer_amd import scenes
for (w,h) in [(480,270),(1920,1080)]:
    s = scenes.huge_grid_scene(w, h, 12)
    ctx = ra.Context(0); ctx.upload(s)
    rgb,_,st = ctx.render()
    rgb,_,st = ctx.render()
    print(w,h,"objects", ra.desc_info(s.flatten())["n_objects"], "ms", round(st.ms_render,2), "trace", round(st.ms_trace,2), "shadow", round(st.ms_shadow,2), "shade", round(st.ms_shade,2), "levels", list(st.rays_per_level), "rays", st.rays, "Mrays/s", round(st.rays/st.ms_render/1e3,1), flush=True)
    ctx.close()
