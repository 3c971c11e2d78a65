"""Handle on one rt_ctx of librtgpu.so (include/rtgpu.h): upload a flat scene,
render frames, render image-space shards into device memory, probe rays."""
import ctypes as C

import numpy as np

from . import _native as N


class Context:
    def __init__(self, device_id=0, devices=None):
        """device_id: one GPU (rt_ctx_create).  devices=[ids]: one context over several GPUs of this node
        (rt_ctx_create_multi): upload() reaches every device, render() and render_frames() shard the frame over them,
        gather the shards to the first device inside the library (RCCL) and return whole frames."""
        self._c = C.c_void_p()
        if devices is None:
            rc = N.gpu().rt_ctx_create(int(device_id), C.byref(self._c))
        else:
            ids = (C.c_int * len(devices))(*[int(d) for d in devices])
            rc = N.gpu().rt_ctx_create_multi(ids, len(devices), C.byref(self._c))
            device_id = int(devices[0]) if len(devices) else 0
        if rc != N.RT_OK:
            raise N.RtError(rc, N.gpu().rt_last_error(None).decode(errors="replace"))
        self.device_id = device_id
        self.vw = self.vh = 0

    @property
    def n_devices(self):
        return N.gpu().rt_ctx_device_count(self._c)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if self._c:
            N.gpu().rt_ctx_destroy(self._c)
            self._c = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != N.RT_OK:
            raise N.RtError(rc, N.gpu().rt_last_error(self._c).decode(errors="replace"))

    def upload(self, scene, keep_meshes=False):
        """scene: anything with .flatten() -> rt_scene_desc address (Scene, LoadedScene).
        keep_meshes: the caller's promise that every mesh (geometry, normals, texture coordinates, tree) is the one
        of this context's previous upload, as in a sequence where only the camera, lights, materials or the other
        objects move: their device copies are kept (rt_mesh::unchanged).  keep_meshes="refit": connectivity and tree are
        the previous upload's, vertex positions and normals are new: boxes and triangle records are recomputed on the
        device (rt_mesh::unchanged = 2)."""
        from .scene import desc_info
        p = scene.flatten()
        N.host().rth_desc_keep_meshes(p, 2 if keep_meshes == "refit" else 1 if keep_meshes else 0)
        self._check(N.gpu().rt_scene_upload(self._c, p))
        info = desc_info(p)
        self.vw, self.vh = info["vw"], info["vh"]
        return self

    def set_area_seed(self, seed=0x9E3779B97F4A7C15):
        """rt_set_area_seed: the seed of the area lights' sample points (the default is the oracle's default too)"""
        self._check(N.gpu().rt_set_area_seed(self._c, C.c_uint64(seed)))

    def render(self, ids=False):
        """rt_render -> (rgb (vh, vw, 3) u8, ids (vh, vw, 2) u32 or None, RtStats)"""
        rgb = np.empty((self.vh, self.vw, 3), dtype=np.uint8)
        idb = np.empty((self.vh, self.vw, 2), dtype=np.uint32) if ids else None
        st = N.RtStats()
        self._check(N.gpu().rt_render(self._c, rgb.ctypes.data, idb.ctypes.data if ids else None, C.byref(st)))
        return rgb, idb, st

    def render_frames(self, cameras, ids=False):
        """rt_render_frames: the frames of a camera path through the uploaded scene as one set of launches (on every
        device of the context) -> (rgb (n, vh, vw, 3) u8, ids (n, vh, vw, 2) u32 or None, RtStats).  `cameras`: Camera
        objects, a ready rt_camera array (camera_array), or a frame count (the scene's own view that many times)."""
        if isinstance(cameras, int):
            n, arr = cameras, None
        else:
            arr = cameras if isinstance(cameras, C.Array) else self.camera_array(cameras)
            n = len(arr)
        rgb = np.empty((n, self.vh, self.vw, 3), dtype=np.uint8)
        idb = np.empty((n, self.vh, self.vw, 2), dtype=np.uint32) if ids else None
        st = N.RtStats()
        self._check(N.gpu().rt_render_frames(self._c, n, arr, rgb.ctypes.data, idb.ctypes.data if ids else None, C.byref(st)))
        return rgb, idb, st

    def shard_pixels(self, shard, n_shards):
        return N.gpu().rt_shard_pixels(self.vw, self.vh, shard, n_shards)

    def render_shard(self, shard, n_shards, d_rgb_ptr, d_ids_ptr=None):
        """rt_render_shard into device memory (raw device addresses, e.g. tensor.data_ptr())"""
        st = N.RtStats()
        self._check(N.gpu().rt_render_shard(self._c, shard, n_shards, C.c_void_p(d_rgb_ptr),
                                            C.c_void_p(d_ids_ptr) if d_ids_ptr else None, C.byref(st)))
        return st

    def camera_array(self, cameras):
        """rt_camera array from Camera objects (camera.rs:5-27); every frame has the context's frame size"""
        arr = (N.RtCamera * len(cameras))()
        for a, cam in zip(arr, cameras):
            a.vw, a.vh = self.vw, self.vh
            a.origin = (C.c_double * 3)(*[float(x) for x in cam.origin])
            a.yaw, a.pitch, a.fov = float(cam.yaw), float(cam.pitch), float(cam.fov)
            a.chf = N.host().rth_camera_chf(float(cam.fov))
        return arr

    def render_shard_batch(self, shard, n_shards, cameras, d_rgb_ptr, d_ids_ptr=None, frame_stride=None):
        """rt_render_shard_batch: this shard of len(cameras) frames that differ in their camera only, in one set of
        launches; frame f's compact shard starts at slot f * frame_stride (default: the largest shard's size, shard
        0's, which is what a gather across ranks needs).  `cameras`: Camera objects, a ready rt_camera array
        (camera_array), or a frame count (every frame through the scene's own camera)."""
        if isinstance(cameras, int):            # that many frames through the uploaded scene's own camera
            n, arr = cameras, None
        else:
            arr = cameras if isinstance(cameras, C.Array) else self.camera_array(cameras)
            n = len(arr)
        if frame_stride is None:
            frame_stride = self.shard_pixels(0, n_shards)
        st = N.RtStats()
        self._check(N.gpu().rt_render_shard_batch(self._c, shard, n_shards, n, arr, int(frame_stride), C.c_void_p(d_rgb_ptr),
                                                  C.c_void_p(d_ids_ptr) if d_ids_ptr else None, C.byref(st)))
        return st

    def detile_frames(self, d_gathered_ptr, n_shards, n_frames, slots_per_shard, elem_bytes, d_images_ptr):
        self._check(N.gpu().rt_detile_frames(self._c, C.c_void_p(d_gathered_ptr), n_shards, n_frames, slots_per_shard, elem_bytes,
                                             C.c_void_p(d_images_ptr)))

    def detile(self, d_gathered_ptr, n_shards, slots_per_shard, elem_bytes, d_image_ptr):
        self._check(N.gpu().rt_detile(self._c, C.c_void_p(d_gathered_ptr), n_shards, slots_per_shard, elem_bytes, C.c_void_p(d_image_ptr)))

    def cast_rays(self, origins, dirs):
        o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
        hits = (N.RtRayHit * len(o))()
        self._check(N.gpu().rt_cast_rays(self._c, len(o), o.ctypes.data, d.ctypes.data, hits))
        return hits
