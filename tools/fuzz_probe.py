"""Random probe rays into random scenes: rt_cast_rays against the CPU checker's cast_ray_once, field by field
(object, triangle, near, far, normal, both points bit for bit; texture coordinates to 2e-7, since a sphere's go through
atan2/asin).  Directions are unit, non-unit (as rays reflected off interpolated mesh normals are), axis-aligned and
with zero components; origins lie inside and outside the objects.

usage: fuzz_probe.py [n_scenes] [first_seed] [rays_per_scene]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
import raytracer_amd as ra  # noqa: E402
from raytracer_amd import _native as N  # noqa: E402
import fuzz_parity as fz  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    m = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
    rich = bool(int(os.environ.get("FUZZ_RICH", "0")))
    bad = 0
    with ra.Context(0) as ctx:
        for seed in range(first, first + n):
            s, _ = fz.random_scene(seed, rich)
            ctx.upload(s)
            rng = np.random.default_rng(seed + 7)
            o = rng.uniform(-7, 7, (m, 3)) + np.array([0., 1., -1.])
            d = rng.normal(size=(m, 3))
            unit = rng.random(m) < 0.6
            d[unit] /= np.linalg.norm(d[unit], axis=1, keepdims=True)
            d[~unit] *= rng.uniform(0.2, 3.0, (int((~unit).sum()), 1))
            d[:6] = [[0, 0, -1], [0, -1, 0], [1, 0, 0], [0, 0, 1], [0, 1, 0], [-1, 0, 0]]
            d[6:12, 1] = 0.
            g = ctx.cast_rays(o, d)
            c = oracle.cast_rays(s, o, d)
            for k, (a, b) in enumerate(zip(g, c)):
                same = a.object == b.object and a.prim == b.prim
                if same and a.object != N.RT_ID_NONE:
                    same = (a.near_t == b.near_t and a.far_t == b.far_t and list(a.normal) == list(b.normal)
                            and list(a.vnear) == list(b.vnear) and list(a.vfar) == list(b.vfar)
                            and abs(a.uv[0] - b.uv[0]) <= 2e-7 and abs(a.uv[1] - b.uv[1]) <= 2e-7)
                if not same:
                    bad += 1
                    if bad <= 10:
                        print(f"seed {seed} ray {k}: gpu ({a.object}, {a.prim}, {a.near_t!r}, {a.far_t!r}) cpu ({b.object}, {b.prim}, {b.near_t!r}, {b.far_t!r}) "
                              f"o {o[k].tolist()} d {d[k].tolist()}", flush=True)
            if (seed - first) % 20 == 0:
                print(f"seed {seed}: {m} rays, {bad} mismatches so far", flush=True)
    print(f"{n} scenes x {m} rays, {bad} mismatches", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
