"""Writes the committed golden fixtures under tests/golden/.

Run in the development container (it reads the reference's asset files -- data, not source -- from
/root/reference/assets).  Each fixture is a flat scene (.rtscene) plus the oracle's output for it (.npz: rgb, ids,
ray counts), so the GPU box can check the HIP path against fixed expectations without the oracle or the assets.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
import raytracer_amd as ra  # noqa: E402
from raytracer_amd import scenes  # noqa: E402
from raytracer_amd.compare import ray_counts  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference/assets"


def real_cubemap(cell):
    """assets/storforsen_4.jpg (8192x6144, cell 2048) box-downsampled to 4*cell x 3*cell"""
    from PIL import Image
    Image.MAX_IMAGE_PIXELS = None
    im = Image.open(os.path.join(REF, "storforsen_4.jpg")).convert("RGB")
    return np.array(im.resize((4 * cell, 3 * cell), Image.BOX))


def emit(name, scene):
    path = os.path.join(OUT, name + ".rtscene")
    scene.save(path)
    loaded = ra.LoadedScene(path)
    rgb, ids, st, oc = oracle.render(loaded, 0)
    assert oc.would_panic == 0
    np.savez_compressed(os.path.join(OUT, name + ".npz"), rgb=rgb, ids=ids, rays=np.array(ray_counts(st), dtype=np.uint64),
                        per_level=np.array(list(st.rays_per_level), dtype=np.uint64))
    print(name, rgb.shape, ray_counts(st), os.path.getsize(path))


if __name__ == "__main__":
    cm = real_cubemap(64)
    emit("mixed_96x72", scenes.small_mixed_scene(96, 72, cells=10))
    emit("refraction_160x90", scenes.refraction_scene(160, 90, cubemap=cm))
    tex = ra.read_png(os.path.join(REF, "fedora.png"))
    emit("fedora_192x108", scenes.fedora_scene(os.path.join(REF, "fedora.obj"), tex, 192, 108, cubemap=cm))
    emit("teapot_128x72", scenes.teapot_scene(os.path.join(REF, "teapot.obj"), 128, 72, seed=1))
    emit("loop_200x150", scenes.loop_scene(200, 150))
