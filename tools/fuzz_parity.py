"""Random scenes, GPU frame against the CPU checker: colours, primary ids and ray counts must be identical.

usage: fuzz_parity.py [n_scenes] [first_seed]      (FUZZ_RICH=1: image textures, larger meshes, more objects;
                                                    FUZZ_SHARDS=1: every frame also rendered as 2-9 shards and reassembled;
                                                    FUZZ_MULTI=1: every scene also through a context over 2-5 "devices" (all of them
                                                    GPU 0: rt_ctx_create_multi, rt_render and a 3-frame rt_render_frames);
                                                    FUZZ_AREA=1: 1-2 area lights per scene, spheres and rectangles of 0-9 sample
                                                    points with short and long reach, and a seed of its own per scene)
Every scene draws its own mix of objects (0-40 spheres and boxes, so both the linear object list and the object tree
occur; 0-2 planes; 0-2 height-field meshes, some see-through with ior 1, some refracting), materials, lights (suns
with and without shadows, point lights with short and long reach), skybox kind, ray depth, camera and frame size
(ragged sizes included).  A mismatch prints the seed and leaves the process with status 1.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
import raytracer_amd as ra  # noqa: E402
from raytracer_amd import scenes  # noqa: E402


def random_material(rng, allow_far=True, rich=False):
    kind = rng.integers(0, 6)
    tex = ra.Texture.solid(tuple(int(x) for x in rng.integers(0, 256, 3)))
    if kind == 0:
        tex = ra.Texture.checkerboard(tuple(int(x) for x in rng.integers(0, 256, 3)), tuple(int(x) for x in rng.integers(0, 256, 3)))
    elif kind == 1 and rich:
        tex = ra.Texture.image(rng.integers(0, 256, (int(rng.integers(1, 40)), int(rng.integers(1, 40)), 3), dtype=np.uint8))
    refl = float(rng.choice([0., 0., 0.3, 0.8, 1.0]))
    transp = float(rng.choice([0., 0., 0., 0.5, 0.9]))
    ior = float(rng.choice([1.5, 1.5, 1.33, 1.0 if allow_far else 1.5, 2.4]))
    emis = float(rng.choice([0., 0., 0., 0.5, 1.0]))
    return scenes.sdl_material(tex, reflectiveness=refl, transparency=transp, ior=ior, emissivity=emis)


def random_area_light(rng):
    common = dict(iterations=int(rng.choice([0, 1, 2, 3, 4, 4, 5, 9])), intensity=float(rng.uniform(1, 12)),
                  max_distance=float(rng.choice([3., 6., 10., 50.])), color=tuple(int(x) for x in rng.integers(100, 256, 3)))
    centre = np.array([rng.uniform(-5, 5), rng.uniform(1, 7), rng.uniform(-5, 5)])
    if rng.integers(0, 2):
        return ra.Area(surface="sphere", position=tuple(float(x) for x in centre), radius=float(rng.uniform(0., 1.5)), **common)
    a, b = rng.uniform(-1.5, 1.5, 3), rng.uniform(-1.5, 1.5, 3)
    corners = [centre, centre + a, centre + b, centre + a + b + rng.uniform(-0.2, 0.2, 3)]     # (not quite a parallelogram)
    return ra.Area(surface="rectangle", corners=[tuple(float(x) for x in c) for c in corners], **common)


def random_scene(seed, rich=False, area=False):
    """rich: image textures on planes, boxes and meshes (not on spheres, whose texture coordinates go through
    atan2/asin and may differ in the last bit between the two libms), larger meshes, more objects"""
    rng = np.random.default_rng(seed)
    vw, vh = int(rng.integers(17, 200)), int(rng.integers(9, 130))
    s = ra.Scene()
    s.set_camera(ra.Camera(vw=vw, vh=vh, origin=(float(rng.uniform(-2, 2)), float(rng.uniform(0.5, 4)), float(rng.uniform(6, 11))),
                           yaw=float(rng.uniform(-0.3, 0.3)), pitch=float(rng.uniform(-0.5, 0.1)), fov=float(rng.uniform(40, 90))))
    s.set_options(max_ray_depth=int(rng.integers(0, 6)), ambient=tuple(int(x) for x in rng.integers(0, 80, 3)))
    sky = rng.integers(0, 3)
    if sky == 0:
        s.skybox_solid(tuple(int(x) for x in rng.integers(0, 256, 3)))
    elif sky == 1:
        s.skybox_normal()
    else:
        s.skybox_cubemap(scenes.procedural_cubemap(int(rng.integers(4, 24)), seed=seed))
    for _ in range(int(rng.integers(0, 3))):
        s.add_light(scenes.sdl_sun(tuple(float(x) for x in (rng.uniform(-1, 1), rng.uniform(-1, -0.2), rng.uniform(-1, 1))),
                                   shadows=bool(rng.integers(0, 2)), intensity=float(rng.uniform(0.3, 1.2))))
    for _ in range(int(rng.integers(0, 4))):
        s.add_light(ra.Point(position=tuple(float(x) for x in (rng.uniform(-5, 5), rng.uniform(1, 7), rng.uniform(-5, 5))),
                             intensity=float(rng.uniform(2, 20)), max_distance=float(rng.choice([3., 10., 50.])),
                             color=tuple(int(x) for x in rng.integers(100, 256, 3))))
    if area:
        arng = np.random.default_rng([seed, 0xA5EA])     # (a stream of its own: the other draws stay what they are without area lights)
        for _ in range(int(arng.integers(1, 3))):
            s.add_light(random_area_light(arng))
    for _ in range(int(rng.integers(0, 3))):
        cells = int(rng.integers(1, 60 if rich else 20))
        mesh = ra.Mesh.heightfield(cells, cells, cell=float(rng.uniform(0.3, 1.0)), amplitude=float(rng.uniform(0., 1.5)),
                                   base_freq=float(rng.uniform(0.1, 0.5)), octaves=int(rng.integers(1, 4)), seed=int(rng.integers(0, 1000)),
                                   material=random_material(rng, rich=rich))
        scenes.sdl_mesh(mesh, position=(float(rng.uniform(-2, 2)), float(rng.uniform(-1.5, 1.0)), float(rng.uniform(-3, 2))),
                        rotate_xyz=tuple(float(x) for x in rng.uniform(-0.4, 0.4, 3)) if rng.integers(0, 2) else None)
        s.add(mesh)
    for _ in range(int(rng.integers(0, 3))):
        s.add(scenes.sdl_plane((0., float(rng.uniform(-3, -0.5)), 0.), normal=(float(rng.uniform(-0.2, 0.2)), 1., float(rng.uniform(-0.2, 0.2))),
                               material=random_material(rng, rich=rich), uv_wrap=float(rng.choice([1., 2., 0.5]))))
    n_small = int(rng.choice([0, 1, 3, 7, 8, 9, 20, 40, 150] if rich else [0, 1, 3, 7, 8, 9, 20, 40]))
    for _ in range(n_small):
        pos = tuple(float(x) for x in (rng.uniform(-5, 5), rng.uniform(-1, 4), rng.uniform(-6, 3)))
        if rng.integers(0, 3) == 0:
            s.add(ra.Aabb(pos, tuple(float(x) for x in rng.uniform(0.1, 0.9, 3)), random_material(rng, rich=rich)))
        else:
            s.add(ra.Sphere(pos, float(rng.uniform(0.1, 1.2)), random_material(rng)))
    return s, (vw, vh)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    rich = bool(int(os.environ.get("FUZZ_RICH", "0")))
    shards = bool(int(os.environ.get("FUZZ_SHARDS", "0")))
    area = bool(int(os.environ.get("FUZZ_AREA", "0")))
    multi = bool(int(os.environ.get("FUZZ_MULTI", "0")))
    bad = 0
    mctx = {k: ra.Context(devices=[0] * k) for k in range(2, 6)} if multi else {}
    with ra.Context(0) as ctx:
        for seed in range(first, first + n):
            s, (vw, vh) = random_scene(seed, rich, area)
            if area:
                area_seed = (seed * 0x9E3779B97F4A7C15 + 12345) & 0xFFFFFFFFFFFFFFFF
                oracle.set_area_seed(area_seed); ctx.set_area_seed(area_seed)
            ctx.upload(s)
            rgb, ids, st = ctx.render(ids=True)
            if shards:
                # the same frame again as 2-9 shards, gathered and de-tiled: must be the frame just rendered
                import torch
                from raytracer_amd import dist as rd
                n_sh = 2 + seed % 8
                slots = rd.max_shard_slots(vw, vh, n_sh)
                gathered = torch.zeros(n_sh * slots * 3, dtype=torch.uint8, device="cuda")
                frame = torch.zeros(vh * vw * 3, dtype=torch.uint8, device="cuda")
                sh_rays = sum(ctx.render_shard(k, n_sh, gathered.data_ptr() + k * slots * 3).rays for k in range(n_sh))
                torch.cuda.synchronize()
                ctx.detile(gathered.data_ptr(), n_sh, slots, 3, frame.data_ptr())
                if not (np.array_equal(frame.cpu().numpy().reshape(vh, vw, 3), rgb) and sh_rays == st.rays):
                    bad += 1
                    print(f"seed {seed}: {vw}x{vh} SHARD MISMATCH with {n_sh} shards", flush=True)
            if multi:
                # the same frame through one context over several shards-on-one-GPU; then three frames of a camera
                # path as one set of launches against the same frames rendered one by one
                m = mctx[2 + seed % 4]
                if area:
                    m.set_area_seed(area_seed)
                m.upload(s)
                m_rgb, m_ids, m_st = m.render(ids=True)
                if not (np.array_equal(m_rgb, rgb) and np.array_equal(m_ids, ids) and m_st.rays == st.rays):
                    bad += 1
                    print(f"seed {seed}: {vw}x{vh} MULTI MISMATCH over {m.n_devices} devices", flush=True)
                cam = s.camera
                cams = [ra.Camera(vw=vw, vh=vh, origin=tuple(float(x) + 0.2 * f for x in cam.origin), yaw=cam.yaw + 0.03 * f, pitch=cam.pitch - 0.02 * f, fov=cam.fov)
                        for f in range(3)]
                fr, fi, fst = m.render_frames(cams, ids=True)
                tot = 0
                for f, c in enumerate(cams):
                    s.set_camera(c)
                    ctx.upload(s)
                    one, one_ids, one_st = ctx.render(ids=True)
                    tot += one_st.rays
                    if not (np.array_equal(fr[f], one) and np.array_equal(fi[f], one_ids)):
                        bad += 1
                        print(f"seed {seed}: {vw}x{vh} MULTI FRAMES MISMATCH frame {f} over {m.n_devices} devices", flush=True)
                if tot != fst.rays:
                    bad += 1
                    print(f"seed {seed}: MULTI FRAMES ray count {fst.rays} != {tot}", flush=True)
                s.set_camera(cam)
            o_rgb, o_ids, o_st, oc = oracle.render(s, 0)
            same_rgb = np.array_equal(rgb, o_rgb)
            same_ids = np.array_equal(ids, o_ids)
            same_rays = (st.rays_primary, st.rays_reflect, st.rays_transmit, st.rays_shadow) == \
                        (o_st.rays_primary, o_st.rays_reflect, o_st.rays_transmit, o_st.rays_shadow)
            ok = same_rgb and same_ids and same_rays and oc.would_panic == 0
            if not ok:
                bad += 1
                diff = int((rgb != o_rgb).any(axis=2).sum())
                print(f"seed {seed}: {vw}x{vh} MISMATCH rgb {same_rgb} ({diff} pixels) ids {same_ids} rays {same_rays} panic {oc.would_panic}", flush=True)
            elif (seed - first) % 10 == 0:
                print(f"seed {seed}: {vw}x{vh} ok, {st.rays} rays", flush=True)
    for m in mctx.values():
        m.close()
    print(f"{n} scenes, {bad} mismatches", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
