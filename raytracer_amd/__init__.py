"""raytracer_amd -- the MI355X render path behind the reference's Scene::render.

csrc/  HIP kernels + the C ABI of include/rtgpu.h      -> librtgpu.so
host/  C++ mirror of the reference's Scene API          -> librthost.so
scene.py / render.py / scenes.py / dist.py              Python harness over both
"""
from .scene import (Aabb, Area, Camera, LoadedScene, SdlScene, Material, Mesh, Plane, Point, Scene, Sphere, Sun, Texture,
                    desc_info, read_image, read_png, write_png)
from .render import Context

__all__ = ["Aabb", "Area", "Camera", "Context", "LoadedScene", "SdlScene", "Material", "Mesh", "Plane", "Point", "Scene", "Sphere",
           "Sun", "Texture", "desc_info", "read_image", "read_png", "write_png"]
