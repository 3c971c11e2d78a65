"""ctypes bindings of the two product libraries.

librtgpu.so  -- the C ABI of include/rtgpu.h (HIP render path, gfx950)
librthost.so -- flat C API over the C++ mirror of the reference's Scene API

Nothing here falls back to a CPU renderer: a missing library raises, and
rendering without a GPU fails with RT_ERR_DEVICE from rt_ctx_create.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))

RT_OK = 0
RT_ERR_INVALID = -1
RT_ERR_UNSUPPORTED = -2
RT_ERR_DEVICE = -3
RT_ERR_NOMEM = -4
RT_ID_NONE = 0xFFFFFFFF
RT_TILE = 16


class RtStats(C.Structure):
    _fields_ = [
        ("rays_primary", C.c_uint64), ("rays_reflect", C.c_uint64),
        ("rays_transmit", C.c_uint64), ("rays_shadow", C.c_uint64),
        ("rays_per_level", C.c_uint64 * 8),
        ("ms_upload", C.c_double), ("ms_render", C.c_double), ("ms_total", C.c_double),
        ("ms_trace", C.c_double), ("ms_shadow", C.c_double), ("ms_shade", C.c_double), ("ms_resolve", C.c_double),
        ("device_bytes", C.c_uint64),
        ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64), ("wave_nodes", C.c_uint64), ("wave_tris", C.c_uint64),
        ("launches_trace", C.c_uint32), ("launches_shadow", C.c_uint32), ("launches_shade", C.c_uint32), ("_pad", C.c_uint32),
        ("long_trace", C.c_uint64), ("long_shadow", C.c_uint64),
        ("max_lane_steps", C.c_uint64), ("max_wave_steps", C.c_uint64),
        ("bytes_fetched", C.c_uint64),
        ("ms_shadow_main", C.c_double), ("bytes_fetched_main", C.c_uint64),
        ("launches_shadow_main", C.c_uint32), ("speculative", C.c_uint32), ("respeculated", C.c_uint64),
    ]

    @property
    def rays(self):
        return self.rays_primary + self.rays_reflect + self.rays_transmit + self.rays_shadow

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "rays_per_level"}
        d["rays_per_level"] = list(self.rays_per_level)
        d["rays"] = self.rays
        return d


class RtCamera(C.Structure):
    """rt_camera of include/rtgpu.h (camera.rs:5-27)"""
    _fields_ = [("vw", C.c_int32), ("vh", C.c_int32), ("origin", C.c_double * 3),
                ("yaw", C.c_double), ("pitch", C.c_double), ("fov", C.c_double), ("chf", C.c_double)]


class RtMultiPlan(C.Structure):
    """rt_multi_plan of include/rtgpu.h: how a context over several GPUs lays out a set of frames"""
    _fields_ = [("slots_per_shard", C.c_int64), ("part_bytes_rgb", C.c_int64), ("part_bytes_ids", C.c_int64),
                ("gathered_bytes_rgb", C.c_int64), ("gathered_bytes_ids", C.c_int64), ("frame_bytes_rgb", C.c_int64)]


class RtRayHit(C.Structure):
    _fields_ = [
        ("object", C.c_uint32), ("prim", C.c_uint32),
        ("near_t", C.c_double), ("far_t", C.c_double),
        ("normal", C.c_double * 3), ("vnear", C.c_double * 3), ("vfar", C.c_double * 3),
        ("uv", C.c_float * 2),
    ]


class RthMaterial(C.Structure):
    _fields_ = [
        ("tex_kind", C.c_uint32), ("a", C.c_uint8 * 3), ("b", C.c_uint8 * 3), ("_pad", C.c_uint8 * 2),
        ("w", C.c_uint32), ("h", C.c_uint32), ("rgb", C.c_void_p),
        ("reflectiveness", C.c_double), ("transparency", C.c_double), ("ior", C.c_double), ("emissivity", C.c_double),
    ]


def _load(name):
    path = os.path.join(_HERE, name)
    if name == "librtgpu.so" and os.environ.get("RTGPU_LIB"):
        path = os.environ["RTGPU_LIB"]          # development: an experimental build of the same ABI
    if not os.path.exists(path):
        raise OSError(f"{path} is missing: build it with `make` (or __graft_entry__.build()) before use")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


_gpu = None
_host = None

GPU_SYMBOLS = [
    "rt_ctx_create", "rt_ctx_destroy", "rt_last_error", "rt_scene_upload", "rt_render",
    "rt_shard_pixels", "rt_supertile", "rt_render_shard", "rt_detile", "rt_cast_rays", "rt_cast_rays_noalloc",
    "rt_render_shard_batch", "rt_detile_frames", "rt_set_area_seed",
    "rt_ctx_create_multi", "rt_ctx_device_count", "rt_render_frames", "rt_multi_layout",
]


def gpu():
    global _gpu
    if _gpu is None:
        L = _load("librtgpu.so")
        L.rt_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.rt_ctx_create.restype = C.c_int
        L.rt_ctx_destroy.argtypes = [C.c_void_p]
        L.rt_ctx_destroy.restype = None
        L.rt_last_error.argtypes = [C.c_void_p]
        L.rt_last_error.restype = C.c_char_p
        L.rt_scene_upload.argtypes = [C.c_void_p, C.c_void_p]
        L.rt_scene_upload.restype = C.c_int
        L.rt_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(RtStats)]
        L.rt_render.restype = C.c_int
        L.rt_shard_pixels.argtypes = [C.c_int32, C.c_int32, C.c_uint32, C.c_uint32]
        L.rt_shard_pixels.restype = C.c_int64
        L.rt_supertile.argtypes = []
        L.rt_supertile.restype = C.c_uint32
        L.rt_render_shard.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(RtStats)]
        L.rt_render_shard.restype = C.c_int
        L.rt_detile.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int64, C.c_uint32, C.c_void_p]
        L.rt_detile.restype = C.c_int
        L.rt_render_shard_batch.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(RtStats)]
        L.rt_render_shard_batch.restype = C.c_int
        L.rt_detile_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int64, C.c_uint32, C.c_void_p]
        L.rt_detile_frames.restype = C.c_int
        L.rt_cast_rays.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_cast_rays.restype = C.c_int
        L.rt_cast_rays_noalloc.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_cast_rays_noalloc.restype = C.c_int
        L.rt_set_area_seed.argtypes = [C.c_void_p, C.c_uint64]
        L.rt_set_area_seed.restype = C.c_int
        L.rt_ctx_create_multi.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
        L.rt_ctx_create_multi.restype = C.c_int
        L.rt_ctx_device_count.argtypes = [C.c_void_p]
        L.rt_ctx_device_count.restype = C.c_int
        L.rt_render_frames.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(RtStats)]
        L.rt_render_frames.restype = C.c_int
        L.rt_multi_layout.argtypes = [C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, C.POINTER(RtMultiPlan)]
        L.rt_multi_layout.restype = C.c_int
        _gpu = L
    return _gpu


def host():
    global _host
    if _host is None:
        gpu()  # librthost.so links librtgpu.so
        L = _load("librthost.so")
        vp, u32, i32, dbl, u8p = C.c_void_p, C.c_uint32, C.c_int32, C.c_double, C.c_void_p
        dp = C.POINTER(C.c_double)
        sig = {
            "rth_last_error": ([], C.c_char_p),
            "rth_scene_new": ([], vp), "rth_scene_free": ([vp], None),
            "rth_set_camera": ([vp, i32, i32, dp, dbl, dbl, dbl], None),
            "rth_set_resolution": ([vp, i32, i32], None),
            "rth_set_options": ([vp, u32, u8p], None),
            "rth_add_sphere": ([vp, dp, dbl, C.POINTER(RthMaterial)], C.c_int),
            "rth_add_plane": ([vp, dp, dp, C.c_float, C.POINTER(RthMaterial)], C.c_int),
            "rth_add_aabb": ([vp, dp, dp, C.POINTER(RthMaterial)], C.c_int),
            "rth_mesh_new": ([], vp), "rth_mesh_free": ([vp], None),
            "rth_mesh_from_obj": ([C.c_char_p], vp),
            "rth_mesh_heightfield": ([u32, u32, dbl, dbl, dbl, u32, C.c_uint64], vp),
            "rth_mesh_set_geometry": ([vp, u32, vp, u32, vp], C.c_int),
            "rth_mesh_set_normals": ([vp, u32, vp, u32, vp], C.c_int),
            "rth_mesh_set_texcoords": ([vp, u32, vp, u32, vp], C.c_int),
            "rth_mesh_scale": ([vp, dbl], None), "rth_mesh_center": ([vp], None),
            "rth_mesh_shift": ([vp, dp], None), "rth_mesh_rotate_xyz": ([vp, dp], None), "rth_mesh_rotate_zyx": ([vp, dp], None),
            "rth_mesh_recalculate_normals": ([vp], None), "rth_mesh_generate_sbvh": ([vp], C.c_int),
            "rth_mesh_set_bvh": ([vp, vp, u32, vp, u32, u32], C.c_int),
            "rth_mesh_use_device_bvh": ([vp, C.c_int], None),
            "rth_mesh_set_verts": ([vp, u32, vp], C.c_int),
            "rth_desc_keep_meshes": ([vp, C.c_int], None),
            "rth_mesh_n_tris": ([vp], u32), "rth_mesh_n_verts": ([vp], u32), "rth_mesh_n_normals": ([vp], u32),
            "rth_mesh_n_nodes": ([vp], u32), "rth_mesh_bvh_depth": ([vp], u32),
            "rth_mesh_verts": ([vp], vp), "rth_mesh_normals": ([vp], vp), "rth_mesh_tri_normals": ([vp], vp), "rth_mesh_tris": ([vp], vp),
            "rth_add_mesh": ([vp, vp, C.POINTER(RthMaterial)], C.c_int),
            "rth_scene_object_mesh": ([vp, u32], vp),
            "rth_add_sun": ([vp, u8p, dbl, i32, dbl, dp, C.c_int, dbl], C.c_int),
            "rth_add_point": ([vp, u8p, dbl, i32, dbl, dp, dbl], C.c_int),
            "rth_add_area": ([vp, u8p, dbl, i32, dbl, u32, dp, dbl, dp, u32, dbl], C.c_int),
            "rth_set_skybox_solid": ([vp, u8p], None), "rth_set_skybox_normal": ([vp], None),
            "rth_set_skybox_cubemap": ([vp, u32, u32, vp], C.c_int),
            "rth_set_skybox_star_atlas": ([vp, u32, C.c_uint64], C.c_int),
            "rth_scene_flatten": ([vp], vp), "rth_scene_save": ([vp, C.c_char_p], C.c_int),
            "rth_desc_load": ([C.c_char_p], vp), "rth_desc_get": ([vp], vp), "rth_desc_free": ([vp], None),
            "rth_desc_set_resolution": ([vp, i32, i32], None),
            "rth_desc_set_cubemap": ([vp, u32, u32, vp], C.c_int),
            "rth_desc_info": ([vp, C.POINTER(C.c_uint32)], None),
            "rth_scene_render": ([vp, vp, C.c_int], C.c_int),
            "rth_scene_render_to": ([vp, C.c_char_p, C.c_int], C.c_int),
            "rth_sdl_load": ([C.c_char_p, C.c_int, C.c_char_p, dbl, C.c_uint64], vp),
            "rth_sdl_printed": ([], C.c_char_p),
            "rth_scene_counts": ([vp, C.POINTER(u32)], None),
            "rth_camera_chf": ([dbl], dbl),
            "rth_write_png": ([C.c_char_p, u32, u32, vp], C.c_int),
            "rth_read_png": ([C.c_char_p, C.POINTER(u32), C.POINTER(u32), vp], C.c_int),
            "rth_read_image": ([C.c_char_p, C.POINTER(u32), C.POINTER(u32), vp], C.c_int),
        }
        for name, (args, res) in sig.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = res
        _host = L
    return _host


HOST_SYMBOLS = [
    "rth_last_error", "rth_scene_new", "rth_scene_free", "rth_set_camera", "rth_set_options", "rth_add_sphere",
    "rth_add_plane", "rth_add_aabb", "rth_mesh_new", "rth_mesh_from_obj", "rth_mesh_generate_sbvh", "rth_add_mesh",
    "rth_add_sun", "rth_add_point", "rth_scene_flatten", "rth_scene_save", "rth_desc_load", "rth_scene_render",
    "rth_scene_render_to", "rth_write_png", "rth_read_png", "rth_sdl_load", "rth_sdl_printed",
]


def source_hash():
    """sha256 over the device code's sources (csrc/*.h, csrc/*.hip and include/rtgpu.h: names and bytes, in name order).
    Profiles record it, so that counters measured on one build of the kernels are not read as another build's."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(_HERE, "csrc")
    files = sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith((".h", ".hip")))
    files.append(os.path.join(os.path.dirname(_HERE), "include", "rtgpu.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rtgpu error {code}: {msg}")
        self.code = code


def host_error():
    return host().rth_last_error().decode(errors="replace")
