"""The benchmark / parity scenes, built through the Scene API with the values
the reference's SDL interpreter would hand over (defaults and normalisations
of sdl/src/interpret.rs:604-1078).  `random()` calls are replaced by seeded
draws and missing assets by procedural stand-ins; each builder says where.
"""
import math
import os

import numpy as np

from .scene import (Aabb, Area, Camera, Material, Mesh, Plane, Point, Scene, Sphere, Sun, Texture)


def normalize(v):
    """Vector3::normalize (vector.rs:74-81) with the reference's operation order"""
    x, y, z = (float(c) for c in v)
    m = math.sqrt(x * x + y * y + z * z)
    return (x / m, y / m, z / m)


def as_u8(x):
    """Rust `f64 as u8` (interpret.rs:1249): truncate, saturate, NaN -> 0"""
    if not (x > 0.0):
        return 0
    return 255 if x >= 255.0 else int(x)


def sdl_material(texture=None, reflectiveness=0., transparency=0., ior=1.5, emissivity=0.):
    """a `material: {..}` dictionary (interpret.rs:1052-1073): ior defaults to 1.5 there"""
    return Material(texture or Texture.solid((255, 255, 255)), reflectiveness, transparency, ior, emissivity)


def sdl_sun(vector, **kw):
    return Sun(vector=normalize(vector), **kw)        # interpret.rs:949-951


def sdl_plane(origin, normal=(0., 1., 0.), material=None, uv_wrap=1.):
    return Plane(origin, normalize(normal), material or Material(), uv_wrap)   # interpret.rs:860-867


def procedural_cubemap(cell, seed=1):
    """Stand-in atlas (4*cell x 3*cell) for cubemap images that cannot travel or are missing."""
    rng = np.random.default_rng(seed)
    h, w = 3 * cell, 4 * cell
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.empty((h, w, 3), dtype=np.uint8)
    img[..., 0] = (40 + 60 * (xx // cell) + (yy % cell) * 50 // cell) % 256
    img[..., 1] = (30 + 70 * (yy // cell) + (xx % cell) * 60 // cell) % 256
    img[..., 2] = 90 + rng.integers(0, 80, size=(h, w))
    stars = rng.random((h, w)) < 0.02
    img[stars] = (255, 240, 220)
    return img


# where the reference's stitcher puts the faces in the 4 x 3 atlas, in cells: x+, x-, y+, y-, z+, z- (stitcher/src/main.rs:4)
CUBEMAP_TILE_OFFSETS = ((2, 1), (0, 1), (1, 0), (1, 2), (1, 1), (3, 1))


def stitch_cubemap(xpos, xneg, ypos, yneg, zpos, zneg):
    """Six square faces (h, w, 3 u8 each, all the same size) -> the 4w x 3h atlas Cubemap skyboxes read
    (stitcher/src/main.rs:56-98).  The cells no face covers stay black (the reference leaves them transparent black;
    the skybox never reads them)."""
    faces = [np.asarray(f) for f in (xpos, xneg, ypos, yneg, zpos, zneg)]
    h, w = faces[0].shape[:2]
    if h != w:
        raise ValueError("The width and height must be the same for each tile")                    # main.rs:69-72
    for f in faces:
        if f.shape[:2] != (h, w):
            raise ValueError("Not all images have the same width/height")                         # main.rs:73-82
    atlas = np.zeros((3 * h, 4 * w, 3), dtype=np.uint8)
    for f, (ox, oy) in zip(faces, CUBEMAP_TILE_OFFSETS):
        atlas[oy * h:(oy + 1) * h, ox * w:(ox + 1) * w] = f[..., :3]
    return atlas


def loop_scene(vw=800, vh=600):
    """README.md:253-274 == scenes/loop.sdl at the default 800x600 camera: 16 grey
    spheres on a ring, one default sun, default Normal skybox."""
    s = Scene()
    s.set_camera(Camera(vw=vw, vh=vh))
    s.add_light(sdl_sun((-0.8, -1., -0.2)))
    segments = 16
    for i in range(segments):
        frac = i / segments
        inner = frac * math.tau
        cc = as_u8(frac * 255)
        s.add(Sphere((math.cos(inner), math.sin(inner), -4.), 0.3, sdl_material(Texture.solid((cc, cc, cc)))))
    return s


def refraction_scene(vw=1920, vh=1080, cubemap=None):
    """scenes/refraction.sdl (config 2).  `plane{vector:..}` is ignored by the
    interpreter (the property is called `normal`), so the plane keeps (0,1,0)."""
    s = Scene()
    s.add_light(sdl_sun((-0.8, -1., -0.2)))
    s.add(sdl_plane((0., -1., 0.)))
    s.set_camera(Camera(vw=vw, vh=vh, yaw=0.0001, pitch=0.0002))
    s.skybox_cubemap(cubemap if cubemap is not None else procedural_cubemap(256))
    s.add(Sphere((0., 0., -5.), 1., sdl_material(transparency=0.6, reflectiveness=0.5)))
    s.add(Sphere((-1., 0., -9.), 1., sdl_material(Texture.solid((255, 0, 0)), reflectiveness=0.2)))
    s.add(Sphere((1.0, 0., -9.), 1., sdl_material(Texture.solid((0, 0, 255)), reflectiveness=0.2)))
    return s


def cornell_scene(vw=1920, vh=1080, glass_ior=1.5, icosphere_order=None):
    """scenes/cornell.sdl (config 1); glass_ior: the first sphere's index of
    refraction, the language's default unless a test wants another picture.  icosphere_order: the box's contents are
    those of scenes/cornell_icosahedron.sdl instead -- the icosphere of that many subdivisions (the file says 3), scale
    0.5 at <0.3, -0.5, -0.5>"""
    s = Scene()
    s.set_camera(Camera(vw=vw, vh=vh, origin=(0., 0., 3.), fov=45.))
    s.add(Aabb((0., 1., 0.), (0.2, 0.0001, 0.2), sdl_material(emissivity=1.)))
    s.add_light(Area(surface="rectangle", corners=[(-0.25, 0.95, -0.25), (-0.25, 0.95, 0.25), (0.25, 0.95, -0.25), (0.25, 0.95, 0.25)],
                     iterations=64, intensity=0.12))
    s.add(sdl_plane((0., -1., 0.), (0., 1., 0.)))
    s.add(sdl_plane((0., 1., 0.), (0., -1., 0.)))
    s.add(sdl_plane((0., 0., -1.), (0., 0., 1.)))
    s.add(sdl_plane((-1., 0., 0.), (1., 0., 0.), sdl_material(Texture.solid((255, 0, 0)))))
    s.add(sdl_plane((1., 0., 0.), (-1., 0., 0.), sdl_material(Texture.solid((0, 255, 0)))))
    if icosphere_order is not None:
        soup = icosphere_soup(icosphere_order)
        m = Mesh(sdl_material())
        m.set_geometry(soup, np.arange(len(soup), dtype=np.uint32).reshape(-1, 3))
        sdl_mesh(m, position=(0.3, -0.5, -0.5), scale=0.5)
        s.add(m)
        return s
    s.add(Sphere((-0.1, -0.6, 0.4), 0.4, sdl_material(transparency=0.6, reflectiveness=0.5, ior=glass_ior)))
    s.add(Sphere((0.3, -0.5, -0.4), 0.5, sdl_material(Texture.solid((0, 0, 255)))))
    return s


def sdl_mesh(mesh, position=(0., 0., 0.), scale=1., rotate_xyz=None, rotate_zyx=None, recenter=True, device_bvh=False):
    """the mesh pipeline of interpret.rs:822-855 (device_bvh: the tree is left to the render library, which builds it on
    the GPU at upload, instead of Mesh::generate_sbvh on the host)"""
    if scale != 1.:
        mesh.scale(scale)
    if recenter:
        mesh.center()
    if rotate_xyz is not None:
        mesh.rotate_xyz(rotate_xyz)
    if rotate_zyx is not None:
        mesh.rotate_zyx(rotate_zyx)
    if tuple(position) != (0., 0., 0.):
        mesh.shift(position)
    if mesh.n_normals == 0:
        mesh.recalculate_normals()
    mesh.generate_sbvh(device=device_bvh)
    return mesh


def fedora_scene(obj_path, texture_pixels, vw=3840, vh=2160, cubemap=None):
    """scenes/fedora.sdl (config 3): textured 1038-triangle mesh over a reflective plane"""
    s = Scene()
    s.set_camera(Camera(vw=vw, vh=vh, origin=(1.5, 0.3, 3.), yaw=-0.5, pitch=-0.3))
    s.skybox_cubemap(cubemap if cubemap is not None else procedural_cubemap(256))
    s.add_light(sdl_sun((-0.8, -1., -0.3), intensity=0.8, specular_power=64))
    mesh = Mesh.from_obj(obj_path, sdl_material(Texture.image(texture_pixels)))
    sdl_mesh(mesh, position=(0., -0.5, 0.))
    s.add(mesh)
    s.add(sdl_plane((0., -1., 0.), material=sdl_material(reflectiveness=0.6)))
    return s


def teapot_scene(obj_path, vw=3840, vh=2160, seed=1):
    """scenes/bunny.sdl with assets/teapot.obj standing in for the missing
    bunny.obj; its three random() angles (bunny.sdl:15-19) come from `seed`."""
    rng = np.random.default_rng(seed)
    rot = tuple(float(a) for a in rng.uniform(0., 2. * math.pi, 3))
    s = Scene()
    s.set_camera(Camera(vw=vw, vh=vh))
    s.add_light(sdl_sun((-0.4, -1., -0.2), intensity=0.6))
    mesh = Mesh.from_obj(obj_path, sdl_material(Texture.solid((0, 180, 255))))
    sdl_mesh(mesh, position=(0., 0., -10.), scale=2., rotate_xyz=rot)
    s.add(mesh)
    return s


def synthetic_terrain_scene(vw=3840, vh=2160, cells=708, n_lights=8, seed=42, device_bvh=False):
    """Config 4: a cells x cells fBm height field (2*cells^2 triangles; 708 ->
    1,002,528), reflective 0.3, one glass sphere above it, `n_lights` point
    lights on a ring, solid skybox, camera looking down ~30 degrees."""
    s = Scene()
    extent = float(cells)
    mesh = Mesh.heightfield(cells, cells, cell=1.0, amplitude=extent * 0.02, base_freq=8.0 / extent, octaves=3, seed=seed,
                            material=sdl_material(Texture.solid((200, 190, 170)), reflectiveness=0.3))
    sdl_mesh(mesh, position=(0., 0., 0.), device_bvh=device_bvh)           # recentred on the origin
    s.add(mesh)
    s.mesh = s.mesh_at(0)
    s.add(Sphere((0., extent * 0.06, 0.), extent * 0.04, sdl_material(transparency=0.6, reflectiveness=0.5)))
    s.skybox_solid((30, 40, 70))
    ring_r, ring_h = extent * 0.28, extent * 0.085
    for k in range(n_lights):
        a = 2. * math.pi * k / max(n_lights, 1)
        pos = (ring_r * math.cos(a), ring_h, ring_r * math.sin(a))
        d_centre = math.sqrt(pos[0] ** 2 + pos[1] ** 2 + pos[2] ** 2)
        # I / (d/2)^2 ~ 0.5 at the terrain centre (point.rs:102)
        s.add_light(Point(position=pos, intensity=0.5 * (d_centre / 2.) ** 2 / max(n_lights / 4., 1.), max_distance=1e9))
    cam_h = extent * 0.25
    s.set_camera(Camera(vw=vw, vh=vh, origin=(0.0001, cam_h, extent * 0.55), yaw=0.0001, pitch=-0.5236))
    return s


def huge_grid_scene(vw=7680, vh=4320, n=12, seed=7, cubemap=None):
    """scenes/huge_grid.sdl (config 5): n^3 jittered glass spheres; the jitter's
    random() (huge_grid.sdl:31) comes from `seed`; assets/space.png is missing,
    so a procedural atlas stands in."""
    rng = np.random.default_rng(seed)
    spacing, radius, jitter = 2., 0.5, 0.45
    s = Scene()
    s.add_light(sdl_sun((-0.8, -1., -0.2), shadows=True))
    s.skybox_cubemap(cubemap if cubemap is not None else procedural_cubemap(256, seed=3))
    s.set_camera(Camera(vw=vw, vh=vh, origin=(4.6, 0.8, 9.), yaw=-0.4, pitch=-0.2))
    hn = n * 0.5
    for z in range(n):
        for y in range(n):
            for x in range(n):
                j = rng.uniform(-jitter, jitter, 3)
                pos = ((x - hn) * spacing + float(j[0]), (y - hn) * spacing + float(j[1]), (z - hn) * spacing + float(j[2]))
                col = (as_u8(x / n * 255), as_u8(y / n * 255), as_u8(z / n * 255))
                s.add(Sphere(pos, radius, sdl_material(Texture.solid(col), transparency=0.4, reflectiveness=0.5)))
    return s


def terrain_scene(vw=7680, vh=4320, w=72, h=72, seed=11, cubemap=None):
    """scenes/terrain.sdl (config 5): (w-1)*(h-1)*2 triangles, reflectiveness 0.8.
    Heights come from this module's seeded value noise (the reference's `perlin`
    is an un-vendored dependency, so the exact heights are not reproducible)."""
    s = Scene()
    s.skybox_cubemap(cubemap if cubemap is not None else procedural_cubemap(256, seed=3))
    s.add_light(sdl_sun((-0.8, -1., -0.2)))
    s.set_camera(Camera(vw=vw, vh=vh, origin=(-4., 2., 6.), yaw=0.3, pitch=-0.3))
    mesh = Mesh.heightfield(w - 1, h - 1, cell=1.0, amplitude=2.0, base_freq=0.08, octaves=3, seed=seed,
                            material=sdl_material(reflectiveness=0.8))
    sdl_mesh(mesh, position=(0., -3., -5.))
    s.add(mesh)
    return s


def small_mixed_scene(vw=160, vh=120, cells=24, seed=5, device_bvh=False, area_lights=False):
    """A small scene touching every device code path: mesh (reflective), glass
    sphere (refraction + fresnel mix), see-through box (ior 1), checker plane,
    emissive box, sun + two point lights, cubemap skybox.  area_lights: also a spherical area light whose reach ends
    inside the scene (some hits see some of its sample points beyond max_distance, area.rs:126-128) and a rectangular
    one."""
    s = Scene()
    s.set_camera(Camera(vw=vw, vh=vh, origin=(0.3, 2.5, 9.), yaw=0.05, pitch=-0.2, fov=60.))
    s.skybox_cubemap(procedural_cubemap(32, seed=seed))
    s.add_light(sdl_sun((-0.6, -1., -0.3)))
    s.add_light(Point(position=(3., 4., 2.), intensity=8.))
    s.add_light(Point(position=(-4., 3., 1.), intensity=5., color=(255, 200, 160)))
    if area_lights:
        s.add_light(Area(surface="sphere", position=(1., 4., 3.), radius=0.9, iterations=5, intensity=5., max_distance=6.5, color=(200, 220, 255)))
        s.add_light(Area(surface="rectangle", corners=[(-3., 5., -1.), (-3., 5., 1.), (-1., 5., -1.), (-1., 5.5, 1.)], iterations=3, intensity=3.))
    mesh = Mesh.heightfield(cells, cells, cell=0.5, amplitude=0.6, base_freq=0.3, octaves=3, seed=seed,
                            material=sdl_material(Texture.solid((180, 200, 160)), reflectiveness=0.3))
    sdl_mesh(mesh, position=(0., -0.5, 0.), device_bvh=device_bvh)
    s.add(mesh)
    s.mesh = s.mesh_at(0)
    s.add(sdl_plane((0., -2., 0.), material=sdl_material(Texture.checkerboard((255, 255, 255), (60, 60, 60)), reflectiveness=0.2), uv_wrap=2.))
    s.add(Sphere((0., 1.5, 1.), 1., sdl_material(transparency=0.6, reflectiveness=0.5)))
    s.add(Sphere((-2.5, 1.0, 0.), 0.8, sdl_material(Texture.solid((255, 40, 40)), reflectiveness=0.4)))
    s.add(Aabb((2.5, 0.8, 0.5), (0.7, 0.7, 0.7), sdl_material(Texture.solid((90, 120, 255)), transparency=0.5, ior=1.)))
    s.add(Aabb((-1., 3.5, -2.), (0.4, 0.1, 0.4), sdl_material(Texture.solid((255, 255, 200)), emissivity=1.)))
    s.add(Aabb((1.2, 0.2, 3.), (0.3, 0.3, 0.3), sdl_material(Texture.solid((255, 160, 0)), emissivity=0.5)))
    return s


def reference_asset(name):
    """Path of a file under the reference checkout, or None where it is absent (the GPU box)."""
    p = os.path.join("/root/reference", name)
    return p if os.path.exists(p) else None


def spheres_over_terrain_scene(vw=192, vh=108, n=4, cells=24, seed=3):
    """n^3 spheres and a few boxes above a mirror-like height field: rays reflected off the mesh's interpolated,
    unnormalised normals (mesh.rs:79-89) are not unit vectors, which is the case the reference's sphere test
    (sphere.rs:33-48) treats differently from geometry -- and the device's object tree must not cull on."""
    rng = np.random.default_rng(seed)
    s = Scene()
    s.set_camera(Camera(vw=vw, vh=vh, origin=(cells * 0.5, 7.0, cells + 6.0), yaw=0.02, pitch=-0.45, fov=60.))
    s.set_options(max_ray_depth=3)
    s.skybox_normal()
    s.add_light(sdl_sun((-0.3, -1., -0.4)))
    s.add_light(Point(position=(cells * 0.5, 9., cells * 0.5), intensity=40.))
    mesh = Mesh.heightfield(cells, cells, cell=1.0, amplitude=1.5, base_freq=0.15, octaves=3, seed=seed,
                            material=sdl_material(reflectiveness=0.7))
    sdl_mesh(mesh, recenter=False)
    s.add(mesh)
    for i in range(n):
        for j in range(n):
            for k in range(n):
                pos = (cells * (0.2 + 0.6 * i / max(n - 1, 1)) + float(rng.uniform(-.3, .3)), 3.0 + 1.4 * j,
                       cells * (0.2 + 0.6 * k / max(n - 1, 1)) + float(rng.uniform(-.3, .3)))
                col = (as_u8(40 + 200 * i / n), as_u8(40 + 200 * j / n), as_u8(40 + 200 * k / n))
                if (i + j + k) % 5 == 0:
                    s.add(Aabb(pos, (0.35, 0.3, 0.4), sdl_material(Texture.solid(col), reflectiveness=0.3)))
                else:
                    s.add(Sphere(pos, 0.45, sdl_material(Texture.solid(col), reflectiveness=0.4,
                                                         transparency=0.5 if (i + k) % 3 == 0 else 0.)))
    return s


def overlapping_objects_scene(vw=160, vh=90, seed=5):
    """Spheres and boxes that overlap each other, the camera and the point light: rays start inside objects, which
    the reference's sphere test answers with a negative distance (sphere.rs:33-62) and its closest-hit sort then
    prefers (scene.rs:57-82) -- the case an object tree's walk has to order correctly."""
    rng = np.random.default_rng(seed)
    s = Scene()
    s.set_camera(Camera(vw=vw, vh=vh, origin=(0.2, 0.1, 0.3), yaw=0.3, pitch=-0.1, fov=70.))
    s.set_options(max_ray_depth=3)
    s.skybox_normal()
    s.add_light(sdl_sun((-0.4, -1., -0.3), shadows=True))
    s.add_light(Point(position=(1.7, 0.4, -1.5), intensity=6.))
    for i in range(3):
        for j in range(2):
            for k in range(3):
                pos = (1.6 * (i - 1) + float(rng.uniform(-.2, .2)), 1.4 * j - 0.5 + float(rng.uniform(-.2, .2)),
                       -1.6 * k + float(rng.uniform(-.2, .2)))
                col = (as_u8(60 + 90 * i), as_u8(80 + 120 * j), as_u8(60 + 90 * k))
                if (i + j + k) % 4 == 1:
                    s.add(Aabb(pos, (0.9, 0.8, 1.0), sdl_material(Texture.solid(col), reflectiveness=0.3, transparency=0.5 if k == 1 else 0., ior=1.)))
                else:
                    s.add(Sphere(pos, 1.1, sdl_material(Texture.solid(col), reflectiveness=0.4 if i != 1 else 0.,
                                                        transparency=0.6 if (i + k) % 2 == 0 else 0., ior=1. if j == 0 else 1.3)))
    return s


# The cube of images/readme/10_7_2021.png, fitted (tools/fit_readme_10_7.py): the listing draws it from random()
README_10_7_BOX = (0.7830276489257812, 0.9763870239257812, -7.986015319824219)


def readme_10_7_scene(box_position=README_10_7_BOX, texture_pixels=None, vw=1920, vh=1080, point_intensity=2.):
    """The test scene of the reference's progress log, README.md:172-247 = scenes/uwl_cube.sdl (rendered there as
    images/readme/10_7_2021.png): a textured cube, a mirror sphere, a plain sphere, a wall, a checkered ground plate, one
    sun and one reddish point light, default (Normal) skybox.  The cube's position is `random(-1, 1), random(0, 1),
    random(-9, -7)` in the listing, so it is a parameter here.  texture_pixels: assets/uwl.png as RGB."""
    s = Scene()
    s.set_camera(Camera(vw=vw, vh=vh, fov=60., origin=(-3., 3., 0.), pitch=-0.5, yaw=0.4))
    s.add_light(sdl_sun((-0.8, -1., -0.3), intensity=0.8, specular_power=64))
    s.add(Aabb(tuple(box_position), (1., 1., 1.), sdl_material(Texture.image(texture_pixels))))
    grey = Texture.solid((200, 200, 200))
    s.add(Sphere((2.5, 0., -5.), 1., sdl_material(grey, reflectiveness=0.7)))
    s.add(Sphere((-2.5, 0., -5.), 1., sdl_material(grey)))
    s.add(Aabb((0., 2., -12.), (10., 3., 1.), sdl_material()))
    s.add(sdl_plane((0., -1., 0.), material=sdl_material(Texture.checkerboard((128, 128, 128), (255, 255, 255)))))
    s.add_light(Point(position=(0., 0., -3.5), color=(255, 100, 100), intensity=point_intensity))
    return s


def icosphere_soup(order):
    """The generator of scenes/icosahedron.sdl: an icosahedron whose triangles are split in four `order` times
    (midpoints are never shared between neighbours), every vertex pushed onto the unit sphere at the end, emitted as a
    triangle soup in the script's order -- (3 * n_triangles, 3) positions."""
    f = (1 + 5 ** 0.5) / 2
    vertices = [float(x) for x in (-1, f, 0, 1, f, 0, -1, 0 - f, 0, 1, 0 - f, 0, 0, -1, f, 0, 1, f, 0, -1, 0 - f, 0, 1, 0 - f,
                                   f, 0, -1, f, 0, 1, 0 - f, 0, -1, 0 - f, 0, 1)]
    triangles = [0, 11, 5, 0, 5, 1, 0, 1, 7, 0, 7, 10, 0, 10, 11, 11, 10, 2, 5, 11, 4, 1, 5, 9, 7, 1, 8, 10, 7, 6,
                 3, 9, 4, 3, 4, 2, 3, 2, 6, 3, 6, 8, 3, 8, 9, 9, 8, 1, 4, 9, 5, 2, 4, 11, 6, 2, 10, 8, 6, 7]

    def add_midpoint(a, b):
        for k in range(3):
            vertices.append((vertices[3 * a + k] + vertices[3 * b + k]) / 2)
        return len(vertices) // 3 - 1

    prev = triangles
    for _ in range(order):
        triangles = []
        for k in range(0, len(prev), 3):
            v1, v2, v3 = prev[k], prev[k + 1], prev[k + 2]
            a, b, c = add_midpoint(v1, v2), add_midpoint(v2, v3), add_midpoint(v3, v1)
            triangles += [v1, a, c, v2, b, a, v3, c, b, a, b, c]
        prev = triangles
    v = np.array(vertices).reshape(-1, 3)
    for i in range(len(v)):
        m = 1 / math.sqrt(v[i, 0] ** 2 + v[i, 1] ** 2 + v[i, 2] ** 2)
        v[i] = v[i] * m
    return v[np.array(triangles)]


def icosphere_scene(order, vw=800, vh=600):
    """scenes/icosahedron.sdl with `order` subdivisions (the reference's README shows orders 0, 2 and 5 as
    images/readme/11_22_2021_{1,2,3}.png): default camera, one default sun along <-0.8, -1, -0.4>, default (Normal)
    skybox, the icosphere as a mesh of unshared triangles (flat normals from recalculate_normals, mesh.rs:208-254),
    scale 2 at <0, 0, -6>."""
    s = Scene()
    s.set_camera(Camera(vw=vw, vh=vh))
    s.add_light(sdl_sun((-0.8, -1., -0.4)))
    soup = icosphere_soup(order)
    m = Mesh(sdl_material())
    m.set_geometry(soup, np.arange(len(soup), dtype=np.uint32).reshape(-1, 3))
    sdl_mesh(m, position=(0., 0., -6.), scale=2.)
    s.add(m)
    return s
