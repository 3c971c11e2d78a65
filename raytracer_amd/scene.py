"""Python face of the reference's Scene API for the render path.

Class and field names follow raytracer/src (scene.rs:35-41, camera.rs:5-27,
material.rs:147-164, object/*.rs, lighting/*.rs, skybox.rs); the objects are
thin handles on the C++ mirror in librthost.so.  `Scene.render()` is the
reference's `Scene::render` (scene.rs:220-246) and runs on the MI355X.
"""
import ctypes as C

import numpy as np

from . import _native as N


def _d3(v):
    a = (C.c_double * 3)(*[float(x) for x in v])
    return a


def _u8x3(c):
    return (C.c_uint8 * 3)(*[int(x) & 255 for x in c])


class Texture:
    """material.rs:108-117"""
    SOLID, IMAGE, CHECKERBOARD = 0, 1, 2

    def __init__(self, kind, a=(255, 255, 255), b=(0, 0, 0), image=None):
        self.kind, self.a, self.b = kind, tuple(a), tuple(b)
        self.pixels = None
        if image is not None:
            img = np.ascontiguousarray(image, dtype=np.uint8)
            if img.ndim != 3 or img.shape[2] != 3:
                raise ValueError("image texture must be H x W x 3 uint8")
            self.pixels = img

    @staticmethod
    def solid(color):
        return Texture(Texture.SOLID, a=color)

    @staticmethod
    def checkerboard(a, b):
        return Texture(Texture.CHECKERBOARD, a=a, b=b)

    @staticmethod
    def image(pixels):
        return Texture(Texture.IMAGE, image=pixels)


class Material:
    """material.rs:147-176 (defaults of Material::default; the SDL's own default ior is 1.5,
    sdl/src/interpret.rs:1058)"""

    def __init__(self, texture=None, reflectiveness=0., transparency=0., ior=1.3, emissivity=0.):
        self.texture = texture or Texture.solid((255, 255, 255))
        self.reflectiveness, self.transparency, self.ior, self.emissivity = reflectiveness, transparency, ior, emissivity

    def _c(self):
        m = N.RthMaterial()
        t = self.texture
        m.tex_kind = t.kind
        m.a = _u8x3(t.a)
        m.b = _u8x3(t.b)
        if t.kind == Texture.IMAGE:
            m.h, m.w = t.pixels.shape[0], t.pixels.shape[1]
            m.rgb = t.pixels.ctypes.data
        m.reflectiveness, m.transparency, m.ior, m.emissivity = self.reflectiveness, self.transparency, self.ior, self.emissivity
        return m


class Sphere:
    """object/sphere.rs:12-21"""

    def __init__(self, origin, radius, material=None):
        self.origin, self.radius, self.material = origin, radius, material or Material()


class Plane:
    """object/plane.rs:11-47"""

    def __init__(self, origin, normal=(0., 1., 0.), material=None, uv_wrap=1.):
        self.origin, self.normal, self.material, self.uv_wrap = origin, normal, material or Material(), uv_wrap


class Aabb:
    """object/aabb.rs:86-99 -- `size` is the half extent"""

    def __init__(self, pos, size, material=None):
        self.pos, self.size, self.material = pos, size, material or Material()


class Mesh:
    """object/mesh.rs:91-119 and its construction steps (:135-324)"""

    def __init__(self, material=None, _handle=None):
        self.material = material or Material()
        self._h = _handle if _handle is not None else N.host().rth_mesh_new()

    def __del__(self):
        try:
            if self._h:
                N.host().rth_mesh_free(self._h)
                self._h = None
        except Exception:
            pass

    @staticmethod
    def from_obj(file_name, material=None):
        h = N.host().rth_mesh_from_obj(str(file_name).encode())
        if not h:
            raise RuntimeError(N.host_error())
        return Mesh(material, _handle=h)

    @staticmethod
    def heightfield(nx, nz, cell=1.0, amplitude=1.0, base_freq=0.08, octaves=3, seed=42, material=None):
        h = N.host().rth_mesh_heightfield(nx, nz, cell, amplitude, base_freq, octaves, seed)
        return Mesh(material, _handle=h)

    def set_geometry(self, verts, tris):
        v = np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 3)
        t = np.ascontiguousarray(tris, dtype=np.uint32).reshape(-1, 3)
        if N.host().rth_mesh_set_geometry(self._h, len(v), v.ctypes.data, len(t), t.ctypes.data):
            raise RuntimeError(N.host_error())
        return self

    def set_normals(self, normals, tri_normals):
        v = np.ascontiguousarray(normals, dtype=np.float64).reshape(-1, 3)
        t = np.ascontiguousarray(tri_normals, dtype=np.uint32).reshape(-1, 3)
        if N.host().rth_mesh_set_normals(self._h, len(v), v.ctypes.data, len(t), t.ctypes.data):
            raise RuntimeError(N.host_error())
        return self

    def set_texcoords(self, texcoords, tri_texcoords):
        v = np.ascontiguousarray(texcoords, dtype=np.float32).reshape(-1, 2)
        t = np.ascontiguousarray(tri_texcoords, dtype=np.uint32).reshape(-1, 3)
        if N.host().rth_mesh_set_texcoords(self._h, len(v), v.ctypes.data, len(t), t.ctypes.data):
            raise RuntimeError(N.host_error())
        return self

    def scale(self, delta):
        N.host().rth_mesh_scale(self._h, float(delta)); return self

    def center(self):
        N.host().rth_mesh_center(self._h); return self

    def shift(self, delta):
        N.host().rth_mesh_shift(self._h, _d3(delta)); return self

    def rotate_xyz(self, rot):
        N.host().rth_mesh_rotate_xyz(self._h, _d3(rot)); return self

    def rotate_zyx(self, rot):
        N.host().rth_mesh_rotate_zyx(self._h, _d3(rot)); return self

    def recalculate_normals(self):
        N.host().rth_mesh_recalculate_normals(self._h); return self

    def generate_sbvh(self, device=False):
        """Mesh::generate_sbvh (mesh.rs:257-272) on the host; device=True leaves the tree to the render library,
        which builds it on the GPU at upload (no host tree: such a mesh cannot be rendered by a CPU checker)."""
        if device:
            N.host().rth_mesh_use_device_bvh(self._h, 1)
            return self
        N.host().rth_mesh_use_device_bvh(self._h, 0)
        if N.host().rth_mesh_generate_sbvh(self._h):
            raise RuntimeError(N.host_error())
        return self

    def set_verts(self, verts):
        """new positions for the same connectivity (an animated mesh): count must match"""
        v = np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 3)
        if N.host().rth_mesh_set_verts(self._h, len(v), v.ctypes.data):
            raise RuntimeError(N.host_error())
        return self

    BVH_NODE = np.dtype([("min", "<f8", 3), ("max", "<f8", 3), ("right_or_first", "<u4"), ("count", "<u4")])

    def set_bvh(self, nodes, prim_indices, depth=0):
        """Installs a caller-built tree (the reference's `sbvh` field is public, mesh.rs:118).  `nodes`: array of
        BVH_NODE records in depth-first order (a node's first child is the next record, `right_or_first` its second;
        leaves have count > 0 and `right_or_first` = their first entry of `prim_indices`)."""
        nodes = np.ascontiguousarray(nodes, dtype=Mesh.BVH_NODE)
        prim = np.ascontiguousarray(prim_indices, dtype=np.uint32)
        if N.host().rth_mesh_set_bvh(self._h, nodes.ctypes.data, len(nodes), prim.ctypes.data, len(prim), int(depth)):
            raise RuntimeError(N.host_error())
        return self

    n_tris = property(lambda self: N.host().rth_mesh_n_tris(self._h))
    n_verts = property(lambda self: N.host().rth_mesh_n_verts(self._h))
    n_normals = property(lambda self: N.host().rth_mesh_n_normals(self._h))
    n_nodes = property(lambda self: N.host().rth_mesh_n_nodes(self._h))
    bvh_depth = property(lambda self: N.host().rth_mesh_bvh_depth(self._h))

    def _array(self, fn, n, dtype, cols):
        if n == 0:
            return np.zeros((0, cols), dtype=dtype)
        ptr = fn(self._h)
        ctype = C.c_double if dtype == np.float64 else C.c_uint32
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n, cols)).copy()

    verts = property(lambda self: self._array(N.host().rth_mesh_verts, self.n_verts, np.float64, 3))
    normals = property(lambda self: self._array(N.host().rth_mesh_normals, self.n_normals, np.float64, 3))
    tris = property(lambda self: self._array(N.host().rth_mesh_tris, self.n_tris, np.uint32, 3))
    tri_normals = property(lambda self: self._array(N.host().rth_mesh_tri_normals, self.n_tris if self.n_normals else 0, np.uint32, 3))


class Sun:
    """lighting/sun.rs:13-53"""

    def __init__(self, vector=(0., -1., 0.), color=(255, 255, 255), intensity=1., specular_power=32,
                 specular_strength=0.5, shadows=True, shadow_coefficient=0.5):
        self.vector, self.color, self.intensity = vector, color, intensity
        self.specular_power, self.specular_strength = specular_power, specular_strength
        self.shadows, self.shadow_coefficient = shadows, shadow_coefficient


class Point:
    """lighting/point.rs:12-47"""

    def __init__(self, position=(0., 0., 0.), color=(255, 255, 255), intensity=6., specular_power=32,
                 specular_strength=0.7, max_distance=50.):
        self.position, self.color, self.intensity = position, color, intensity
        self.specular_power, self.specular_strength, self.max_distance = specular_power, specular_strength, max_distance


class Area:
    """lighting/area.rs:55-95 (its sample points come from a seeded counter-based generator here, from an unseeded one there)"""

    def __init__(self, surface="sphere", position=(0., 0., 0.), radius=0., corners=None, color=(255, 255, 255),
                 intensity=6., specular_power=32, specular_strength=0.7, iterations=4, max_distance=50.):
        self.surface, self.position, self.radius, self.corners = surface, position, radius, corners
        self.color, self.intensity, self.specular_power = color, intensity, specular_power
        self.specular_strength, self.iterations, self.max_distance = specular_strength, iterations, max_distance


class Camera:
    """camera.rs:5-52"""

    def __init__(self, vw=800, vh=600, origin=(0., 0., 0.), yaw=0., pitch=0., fov=60.):
        self.vw, self.vh, self.origin, self.yaw, self.pitch, self.fov = vw, vh, origin, yaw, pitch, fov


class Scene:
    """scene.rs:35-53"""

    def __init__(self):
        self._h = N.host().rth_scene_new()
        self.camera = Camera()
        self._keep = []
        self.n_objects = 0
        self.n_lights = 0

    def __del__(self):
        try:
            if self._h:
                N.host().rth_scene_free(self._h)
                self._h = None
        except Exception:
            pass

    def set_camera(self, camera):
        self.camera = camera
        return self

    def set_options(self, max_ray_depth=4, ambient=(40, 40, 40)):
        N.host().rth_set_options(self._h, int(max_ray_depth), C.cast(_u8x3(ambient), C.c_void_p))
        return self

    def add(self, obj):
        H = N.host()
        m = obj.material._c()
        self._keep.append(obj.material)
        if isinstance(obj, Sphere):
            H.rth_add_sphere(self._h, _d3(obj.origin), float(obj.radius), C.byref(m))
        elif isinstance(obj, Plane):
            H.rth_add_plane(self._h, _d3(obj.origin), _d3(obj.normal), float(obj.uv_wrap), C.byref(m))
        elif isinstance(obj, Aabb):
            H.rth_add_aabb(self._h, _d3(obj.pos), _d3(obj.size), C.byref(m))
        elif isinstance(obj, Mesh):
            H.rth_add_mesh(self._h, obj._h, C.byref(m))
        else:
            raise TypeError(f"not a scene object: {obj!r}")
        self.n_objects += 1
        return self

    def mesh_at(self, index):
        """the scene's own copy of the mesh added as object `index` (add() copies): edits through it -- set_verts,
        recalculate_normals, generate_sbvh -- show in the next flatten()"""
        h = N.host().rth_scene_object_mesh(self._h, int(index))
        if not h:
            raise RuntimeError(N.host_error())
        return Mesh(_handle=h)

    def add_light(self, l):
        H = N.host()
        col = C.cast(_u8x3(l.color), C.c_void_p)
        if isinstance(l, Sun):
            H.rth_add_sun(self._h, col, l.intensity, l.specular_power, l.specular_strength, _d3(l.vector), int(bool(l.shadows)), l.shadow_coefficient)
        elif isinstance(l, Point):
            H.rth_add_point(self._h, col, l.intensity, l.specular_power, l.specular_strength, _d3(l.position), l.max_distance)
        elif isinstance(l, Area):
            corners = None
            if l.corners is not None:
                corners = (C.c_double * 12)(*[float(x) for c in l.corners for x in c])
            H.rth_add_area(self._h, col, l.intensity, l.specular_power, l.specular_strength, 0 if l.surface == "sphere" else 1,
                           _d3(l.position), float(l.radius), corners, int(l.iterations), l.max_distance)
        else:
            raise TypeError(f"not a light: {l!r}")
        self.n_lights += 1
        return self

    def skybox_solid(self, color):
        N.host().rth_set_skybox_solid(self._h, C.cast(_u8x3(color), C.c_void_p)); return self

    def skybox_normal(self):
        N.host().rth_set_skybox_normal(self._h); return self

    def skybox_cubemap(self, pixels):
        img = np.ascontiguousarray(pixels, dtype=np.uint8)
        if N.host().rth_set_skybox_cubemap(self._h, img.shape[1], img.shape[0], img.ctypes.data):
            raise RuntimeError(N.host_error())
        return self

    def skybox_star_atlas(self, cell, seed=1):
        N.host().rth_set_skybox_star_atlas(self._h, int(cell), int(seed)); return self

    def _sync_camera(self):
        c = self.camera
        N.host().rth_set_camera(self._h, int(c.vw), int(c.vh), _d3(c.origin), float(c.yaw), float(c.pitch), float(c.fov))

    def flatten(self):
        """-> address of the rt_scene_desc (include/rtgpu.h); valid until the next flatten()"""
        self._sync_camera()
        p = N.host().rth_scene_flatten(self._h)
        if not p:
            raise RuntimeError(N.host_error())
        return p

    def save(self, path):
        self._sync_camera()
        if N.host().rth_scene_save(self._h, str(path).encode()):
            raise RuntimeError(N.host_error())

    def render(self, device_id=0):
        """Scene::render (scene.rs:220-246) -> (vh, vw, 3) uint8"""
        self._sync_camera()
        out = np.empty((self.camera.vh, self.camera.vw, 3), dtype=np.uint8)
        if N.host().rth_scene_render(self._h, out.ctypes.data, device_id):
            raise RuntimeError(N.host_error())
        return out

    def render_to(self, path, device_id=0):
        """Scene::render_to (scene.rs:249-265), PNG"""
        self._sync_camera()
        if N.host().rth_scene_render_to(self._h, str(path).encode(), device_id):
            raise RuntimeError(N.host_error())


class LoadedScene:
    """A flat scene read back from a .rtscene file (written by Scene.save)."""

    def __init__(self, path):
        self._h = N.host().rth_desc_load(str(path).encode())
        if not self._h:
            raise RuntimeError(N.host_error())

    def __del__(self):
        try:
            if self._h:
                N.host().rth_desc_free(self._h)
                self._h = None
        except Exception:
            pass

    def set_resolution(self, vw, vh):
        N.host().rth_desc_set_resolution(self._h, int(vw), int(vh)); return self

    def set_cubemap(self, pixels):
        """replaces the skybox by a 4x3 cubemap atlas ((3*cell, 4*cell, 3) uint8)"""
        self._sky = np.ascontiguousarray(pixels, dtype=np.uint8)
        if N.host().rth_desc_set_cubemap(self._h, self._sky.shape[1], self._sky.shape[0], self._sky.ctypes.data):
            raise RuntimeError(N.host_error())
        return self

    def flatten(self):
        return N.host().rth_desc_get(self._h)

    @property
    def camera(self):
        """the stored description's camera (rt_camera is the first member of rt_scene_desc)"""
        c = N.RtCamera.from_address(self.flatten())
        return Camera(vw=c.vw, vh=c.vh, origin=tuple(c.origin), yaw=c.yaw, pitch=c.pitch, fov=c.fov)

    def set_camera(self, camera):
        """moves the camera of the stored description; the frame size stays the description's"""
        c = N.RtCamera.from_address(self.flatten())
        c.origin = (C.c_double * 3)(*[float(x) for x in camera.origin])
        c.yaw, c.pitch, c.fov = float(camera.yaw), float(camera.pitch), float(camera.fov)
        c.chf = N.host().rth_camera_chf(float(camera.fov))
        return self


class SdlScene:
    """A scene built by the scene description language of the reference (sdl/src/main.rs:49-53).

    source: the script text, or with path=True a file name.  t is the global `t` (the frame number of --sequence),
    seed feeds random() (unseeded in the reference), base_dir resolves relative asset paths.
    .printed holds what print() wrote."""

    def __init__(self, source, path=False, base_dir=None, t=0.0, seed=1):
        H = N.host()
        self._h = H.rth_sdl_load(str(source).encode(), int(bool(path)), None if base_dir is None else str(base_dir).encode(),
                                 float(t), int(seed))
        if not self._h:
            raise RuntimeError(N.host_error())
        self.printed = H.rth_sdl_printed().decode(errors="replace")
        out = (C.c_uint32 * 4)()
        H.rth_scene_counts(self._h, out)
        self.n_objects, self.n_lights, self.vw, self.vh = [int(x) for x in out]

    def __del__(self):
        try:
            if self._h:
                N.host().rth_scene_free(self._h)
                self._h = None
        except Exception:
            pass

    def set_resolution(self, vw, vh):
        N.host().rth_set_resolution(self._h, int(vw), int(vh))
        self.vw, self.vh = int(vw), int(vh)
        return self

    def flatten(self):
        p = N.host().rth_scene_flatten(self._h)
        if not p:
            raise RuntimeError(N.host_error())
        return p

    def save(self, path):
        if N.host().rth_scene_save(self._h, str(path).encode()):
            raise RuntimeError(N.host_error())


def desc_info(desc_ptr):
    out = (C.c_uint32 * 8)()
    N.host().rth_desc_info(desc_ptr, out)
    keys = ["vw", "vh", "n_objects", "n_lights", "n_meshes", "n_tris", "n_nodes", "skybox_kind"]
    return dict(zip(keys, list(out)))


def write_png(path, rgb):
    img = np.ascontiguousarray(rgb, dtype=np.uint8)
    if N.host().rth_write_png(str(path).encode(), img.shape[1], img.shape[0], img.ctypes.data):
        raise RuntimeError(N.host_error())


def read_image(path):
    """PNG or JPEG (told apart by content) -> (h, w, 3) uint8"""
    w, h = C.c_uint32(), C.c_uint32()
    if N.host().rth_read_image(str(path).encode(), C.byref(w), C.byref(h), None):
        raise RuntimeError(N.host_error())
    out = np.empty((h.value, w.value, 3), dtype=np.uint8)
    if N.host().rth_read_image(str(path).encode(), C.byref(w), C.byref(h), out.ctypes.data):
        raise RuntimeError(N.host_error())
    return out


def read_png(path):
    w, h = C.c_uint32(), C.c_uint32()
    if N.host().rth_read_png(str(path).encode(), C.byref(w), C.byref(h), None):
        raise RuntimeError(N.host_error())
    out = np.empty((h.value, w.value, 3), dtype=np.uint8)
    if N.host().rth_read_png(str(path).encode(), C.byref(w), C.byref(h), out.ctypes.data):
        raise RuntimeError(N.host_error())
    return out
