"""Stitches six cube faces into the 4 x 3 atlas a Cubemap skybox reads: the counterpart of the reference's
`stitcher XPOS XNEG YPOS YNEG ZPOS ZNEG -o cubemap.png` (stitcher/src/main.rs).  Faces: PNG or JPEG."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra  # noqa: E402
from raytracer_amd import scenes  # noqa: E402

ap = argparse.ArgumentParser(description="Stitches cubemap images together into a cubemap atlas")
for name in ("XPOS", "XNEG", "YPOS", "YNEG", "ZPOS", "ZNEG"):
    ap.add_argument(name, help=f"The {name[0].lower()}-{'positive' if name.endswith('POS') else 'negative'} image file")
ap.add_argument("-o", "--output", default="cubemap.png")
a = ap.parse_args()
faces = [ra.read_image(getattr(a, n)) for n in ("XPOS", "XNEG", "YPOS", "YNEG", "ZPOS", "ZNEG")]
ra.write_png(a.output, scenes.stitch_cubemap(*faces))
