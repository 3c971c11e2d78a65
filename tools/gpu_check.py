"""Ad-hoc GPU-vs-oracle comparison over the parity scenes (development aid)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as ra
from raytracer_amd import scenes
from raytracer_amd.compare import image_metrics, id_match, ray_counts
import oracle

def check(name, s, sample=None):
    ctx = ra.Context(0)
    t = time.time(); ctx.upload(s); tu = time.time() - t
    rgb, ids, st = ctx.render(ids=True)
    rgb, ids, st = ctx.render(ids=True)
    t = time.time()
    if sample is None:
        orgb, oids, ost, oc = oracle.render(s, 0)
        g_rgb, g_ids = rgb, ids
    else:
        n = ctx.vw * ctx.vh
        px = np.random.default_rng(0).choice(n, size=min(sample, n), replace=False).astype(np.uint32)
        orgb, oids, ost, oc = oracle.render(s, 0, pixels=px)
        g_rgb, g_ids = rgb.reshape(-1, 3)[px], ids.reshape(-1, 2)[px]
    to = time.time() - t
    m = image_metrics(g_rgb, orgb)
    print(f"{name}: {ctx.vw}x{ctx.vh} upload {tu*1e3:.0f} ms gpu {st.ms_render:.2f} ms (trace {st.ms_trace:.2f} shadow {st.ms_shadow:.2f} shade {st.ms_shade:.2f} resolve {st.ms_resolve:.2f}; visits nodes {st.nodes_visited} tris {st.tris_tested} wave-nodes {st.wave_nodes} wave-tris {st.wave_tris}) "
          f"rays {st.rays} -> {st.rays / st.ms_render / 1e3:.1f} Mrays/s | oracle {to:.1f}s | exact {m['exact']:.6f} within1 {m['within1']:.6f} max {m['max_abs']} psnr {m['psnr']:.1f} ids {id_match(g_ids, oids):.6f}"
          + ("" if sample else f" | rays gpu {ray_counts(st)} cpu {ray_counts(ost)}"), flush=True)
    ctx.close()

which = sys.argv[1:] or ["loop", "refraction", "mixed", "terrain_small"]
if "loop" in which: check("loop", scenes.loop_scene())
if "refraction" in which: check("refraction", scenes.refraction_scene(480, 270))
if "mixed" in which: check("mixed", scenes.small_mixed_scene(320, 240))
if "terrain_small" in which: check("terrain_small", scenes.synthetic_terrain_scene(320, 180, cells=64))
if "terrain_med" in which: check("terrain_med", scenes.synthetic_terrain_scene(1280, 720, cells=256), sample=20000)
if "terrain_full" in which: check("terrain_full", scenes.synthetic_terrain_scene(3840, 2160, cells=708), sample=2000)
if "huge_grid" in which: check("huge_grid", scenes.huge_grid_scene(480, 270, n=6))
# the BASELINE configurations at their full sizes (oracle on a pixel sample)
GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
if "config2" in which: check("config2 refraction 1920x1080", scenes.refraction_scene(1920, 1080), sample=20000)
if "config3" in which: check("config3 fedora 3840x2160", ra.LoadedScene(os.path.join(GOLDEN, "fedora_192x108.rtscene")).set_resolution(3840, 2160), sample=20000)
if "config5a" in which: check("config5 huge_grid 7680x4320", scenes.huge_grid_scene(7680, 4320, n=12), sample=3000)
if "config5b" in which: check("config5 terrain 7680x4320", scenes.terrain_scene(7680, 4320), sample=20000)
