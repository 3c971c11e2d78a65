"""Renders a scene-description-language file on the GPU: the counterpart of `sdl SOURCE -o OUT [--sequence N]`
(sdl/src/main.rs).  With --sequence the frames are written to OUT/frame_<i>.png and the script sees `t` = i."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("source")
ap.add_argument("-o", "--output", default="render.png")
ap.add_argument("--sequence", type=int, default=0)
ap.add_argument("--base-dir", default=None)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()

ctx = ra.Context(0)
frames = [None] if args.sequence <= 0 else list(range(args.sequence))
if args.sequence > 0:
    os.makedirs(args.output, exist_ok=True)
for i in frames:
    t0 = time.perf_counter()
    scene = ra.SdlScene(args.source, path=True, base_dir=args.base_dir, t=float(i or 0), seed=args.seed)
    if scene.printed:
        print(scene.printed, end="")
    t1 = time.perf_counter()
    ctx.upload(scene)
    rgb, _, st = ctx.render()
    out = args.output if i is None else os.path.join(args.output, f"frame_{i}.png")
    ra.write_png(out, rgb)
    print(f"{out}: scene constructed in {t1 - t0:.3f}s, {st.rays} rays in {st.ms_render:.2f} ms")
ctx.close()
