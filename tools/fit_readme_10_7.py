"""Fits the one free parameter of the reference's README scene of 2021-10-07 (README.md:172-247 = scenes/uwl_cube.sdl):
the cube's position, which the listing draws from random().  Coarse grid, then coordinate descent with halving steps
on the fraction of pixels that differ from images/readme/10_7_2021.png, first on a pixel sample, then on the whole
frame.  Also estimates that build's point-light falloff constant from the floor pixels.  Needs the reference checkout
(development container only); prints the position that raytracer_amd/scenes.py holds as README_10_7_BOX.

    python tools/fit_readme_10_7.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import oracle  # noqa: E402
from raytracer_amd import scenes  # noqa: E402
from raytracer_amd.scene import read_image  # noqa: E402

REF = "/root/reference"
ref = read_image(os.path.join(REF, "images/readme/10_7_2021.png"))
tex = read_image(os.path.join(REF, "assets/uwl.png"))
INTENSITY = 2. * (1.5 / 2.) ** 2     # see estimate_meter()


def estimate_meter(box):
    """ratio of the point light's contribution in the image to v2's, over floor pixels, by distance to the light"""
    import ctypes as C
    from raytracer_amd import _native as N
    c0, ids, _, _ = oracle.render(scenes.readme_10_7_scene(box, tex, point_intensity=0.), 0)
    c2 = oracle.render(scenes.readme_10_7_scene(box, tex, point_intensity=2.), 0, ids=False)[0]
    lib = oracle.lib()
    cam = N.RtCamera()
    cam.vw, cam.vh, cam.origin, cam.yaw, cam.pitch, cam.fov, cam.chf = 1920, 1080, (C.c_double * 3)(-3, 3, 0), 0.4, -0.5, 60., lib.orc_camera_chf(60.)
    ys, xs = np.where(ids[..., 0] == 4)
    rows = []
    for k in np.random.default_rng(0).choice(len(ys), 6000, replace=False):
        y, x = ys[k], xs[k]
        a, b, r = float(c0[y, x, 0]), float(c2[y, x, 0]), float(ref[y, x, 0])
        if b >= 254 or r >= 254 or b - a < 20:
            continue
        out = (C.c_double * 3)()
        lib.orc_camera_direction_at(C.byref(cam), float(x), float(y), out)
        d, o = np.array(out[:]), np.array([-3., 3., 0.])
        p = o + d * ((-1 - o[1]) / d[1])
        rows.append((np.linalg.norm(p - np.array([0, 0, -3.5])), (r - a) / (b - a)))
    rows = np.array(rows)
    for lo in range(1, 6):
        m = (rows[:, 0] >= lo) & (rows[:, 0] < lo + 1)
        if m.sum() > 5:
            print(f"distance {lo}-{lo + 1}: image / v2 = {np.median(rows[m, 1]):.4f}  (n = {m.sum()})")
    print("(1.5 / 2)^2 = 0.5625")


def fit():
    px = np.random.default_rng(1).choice(1920 * 1080, 150000, replace=False).astype(np.uint32)
    refs = ref.reshape(-1, 3)[px].astype(int)

    def cost_sample(box):
        rgb = oracle.render(scenes.readme_10_7_scene(box, tex, point_intensity=INTENSITY), 0, pixels=px, ids=False)[0]
        return (np.abs(rgb.astype(int) - refs).max(axis=1) > 1).mean()

    def cost_full(box):
        rgb = oracle.render(scenes.readme_10_7_scene(box, tex, point_intensity=INTENSITY), 0, ids=False)[0]
        return (np.abs(rgb.astype(int) - ref.astype(int)).max(axis=2) > 0).mean()

    best = min(((cost_sample((x, y, z)), (x, y, z)) for x in np.linspace(-1, 1, 9) for y in np.linspace(0, 1, 5) for z in np.linspace(-9, -7, 9)))
    for cost, step, rounds in ((cost_sample, 0.125, 14), (cost_full, 1 / 1024, 16)):
        b = np.array(best[1])
        best = (cost(tuple(b)), tuple(b))
        for _ in range(rounds):
            improved = False
            for k in range(3):
                for sgn in (-1, 1):
                    cand = b.copy()
                    cand[k] += sgn * step
                    c = cost(tuple(cand))
                    if c < best[0]:
                        best, b, improved = (c, tuple(cand)), cand, True
            if not improved:
                step /= 2
            print(best, step, flush=True)
    return best


if __name__ == "__main__":
    estimate_meter(scenes.README_10_7_BOX)
    print("fitted:", fit())
