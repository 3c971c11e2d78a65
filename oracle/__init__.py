"""ctypes bindings of liboracle.so, the CPU checker.

TEST INFRASTRUCTURE ONLY (see oracle.cpp): imported by tests/, by
__graft_entry__.smoke() and by the cpu_baseline leg of bench.py -- never by
the raytracer_amd package.

Pinned by five renders the reference itself holds (its author's README images,
reproduced byte for byte or all but 14 pixels: tests/test_oracle_kat.py) and by
analytic known-answer tests.  PARITY UNPINNED for refraction, cubemap sampling,
texture coordinates on meshes, shared-vertex normals and emissive materials,
which none of those images shows (see the header of oracle.cpp).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class OrcCounts(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("node_tests", "tri_tests", "sphere_tests", "plane_tests", "box_tests",
                                           "texel_image", "texel_cubemap", "would_panic")] + [("by_ctx", (C.c_uint64 * 5) * 3)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "by_ctx"}
        for ci, cname in enumerate(("closest", "shadow", "helper")):
            d[cname] = dict(zip(("node", "tri", "sphere", "plane", "box"), [int(x) for x in self.by_ctx[ci]]))
        return d

    def algorithmic_bytes(self, ctx=None):
        """32 B per BVH node test, 36 per triangle, 16 per sphere, 28 per plane, 24 per box (fp32 payload sizes),
        for one cast context (0 closest, 1 shadow, 2 helper) or all of them"""
        w = (32, 36, 16, 28, 24)
        rows = range(3) if ctx is None else (ctx,)
        return sum(int(self.by_ctx[c][k]) * w[k] for c in rows for k in range(5))


def _stats_type():
    from raytracer_amd._native import RtStats
    return RtStats


def _hit_type():
    from raytracer_amd._native import RtRayHit
    return RtRayHit


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise OSError(f"{path} is missing: build it with `make oracle/liboracle.so`")
        L = C.CDLL(path)
        RtStats = _stats_type()
        vp = C.c_void_p
        L.rt_cpu_render.argtypes = [vp, C.c_int, vp, vp, C.POINTER(RtStats), C.POINTER(OrcCounts)]
        L.rt_cpu_render.restype = C.c_int
        L.rt_cpu_render_pixels.argtypes = [vp, C.c_int, C.c_uint64, vp, vp, vp, C.POINTER(RtStats), C.POINTER(OrcCounts)]
        L.rt_cpu_render_pixels.restype = C.c_int
        L.rt_cpu_cast_rays.argtypes = [vp, C.c_uint32, vp, vp, vp]
        L.rt_cpu_cast_rays.restype = C.c_int
        for name in ("orc_sphere_intersect", "orc_plane_intersect", "orc_box_intersect"):
            getattr(L, name).argtypes = [vp, vp, vp, vp]
            getattr(L, name).restype = C.c_int
        L.orc_aabb_bool.argtypes = [vp, vp, vp, vp]
        L.orc_aabb_bool.restype = C.c_int
        L.orc_triangle_intersect.argtypes = [vp, vp, vp, vp, vp, vp]
        L.orc_triangle_intersect.restype = C.c_int
        L.orc_refraction_vec.argtypes = [vp, vp, C.c_double, C.c_double, vp]
        L.orc_refraction_vec.restype = C.c_int
        L.orc_color_newf.argtypes = [C.c_double, C.c_double, C.c_double, vp]
        L.orc_color_newf.restype = None
        L.orc_camera_direction_at.argtypes = [vp, C.c_double, C.c_double, vp]
        L.orc_camera_direction_at.restype = None
        L.orc_camera_chf.argtypes = [C.c_double]
        L.orc_camera_chf.restype = C.c_double
        L.orc_texture_at.argtypes = [vp, C.c_float, C.c_float, vp]
        L.orc_texture_at.restype = None
        L.orc_skybox_ray_color.argtypes = [vp, vp, vp]
        L.orc_skybox_ray_color.restype = None
        L.orc_powi.argtypes = [C.c_double, C.c_int]
        L.orc_powi.restype = C.c_double
        L.orc_set_area_seed.argtypes = [C.c_uint64]
        L.orc_set_area_seed.restype = None
        L.orc_set_mistake.argtypes = [C.c_uint32]
        L.orc_set_mistake.restype = None
        _lib = L
    return _lib


DEFAULT_AREA_SEED = 0x9E3779B97F4A7C15


def set_area_seed(seed=DEFAULT_AREA_SEED):
    """Seed of the stand-in for area.rs:121's unseeded generator, for the renders that follow."""
    lib().orc_set_area_seed(seed)


def set_mistake(flags=0):
    """Negative controls: 1 = cubemap texels from the nearest texel instead of the four-tap blend, 2 = two entries of
    the cubemap's face table exchanged; 0 = the reference's behaviour.  A test against a reference-held render must
    fail with either switched on."""
    lib().orc_set_mistake(flags)


def render(scene, n_threads=0, pixels=None, ids=True):
    """Reference-semantics CPU render of a flat scene.

    scene: object with .flatten() -> rt_scene_desc address.
    pixels: optional uint32 array of linear pixel indices (y*vw + x); the result is then (n, 3).
    -> rgb, ids, RtStats, OrcCounts
    """
    from raytracer_amd.scene import desc_info
    p = scene.flatten()
    info = desc_info(p)
    vw, vh = info["vw"], info["vh"]
    st, oc = _stats_type()(), OrcCounts()
    if pixels is None:
        rgb = np.empty((vh, vw, 3), dtype=np.uint8)
        idb = np.empty((vh, vw, 2), dtype=np.uint32) if ids else None
        rc = lib().rt_cpu_render(p, n_threads, rgb.ctypes.data, idb.ctypes.data if ids else None, C.byref(st), C.byref(oc))
    else:
        px = np.ascontiguousarray(pixels, dtype=np.uint32)
        rgb = np.empty((len(px), 3), dtype=np.uint8)
        idb = np.empty((len(px), 2), dtype=np.uint32) if ids else None
        rc = lib().rt_cpu_render_pixels(p, n_threads, len(px), px.ctypes.data, rgb.ctypes.data,
                                        idb.ctypes.data if ids else None, C.byref(st), C.byref(oc))
    if rc != 0:
        raise RuntimeError(f"oracle render failed: {rc}")
    return rgb, idb, st, oc


def cast_rays(scene, origins, dirs):
    o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
    d = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
    hits = (_hit_type() * len(o))()
    rc = lib().rt_cpu_cast_rays(scene.flatten(), len(o), o.ctypes.data, d.ctypes.data, hits)
    if rc != 0:
        raise RuntimeError(f"oracle cast failed: {rc}")
    return hits
