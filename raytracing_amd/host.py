"""ctypes binding of raytracing_amd/librtamd_host.so (include/rtamd_host.h): the reference's scenes built by the C++
scene API, returned as flat scene descriptions.  Needs no GPU."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librtamd_host.so")
ASSET_DIR = os.path.join(_HERE, "assets")
EXPORTS = ["rtamd_host_build_scene", "rtamd_host_scene_desc", "rtamd_host_scene_camera", "rtamd_host_scene_save", "rtamd_host_scene_animation",
           "rtamd_host_load_image", "rtamd_host_scene_free"]
_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: build it with `make -C raytracing_amd/host`")
        L = C.CDLL(LIB_PATH)
        L.rtamd_host_build_scene.restype = C.c_void_p
        L.rtamd_host_build_scene.argtypes = [C.c_int] * 7 + [C.c_char_p]
        L.rtamd_host_scene_desc.restype = C.POINTER(abi.SceneDesc)
        L.rtamd_host_scene_desc.argtypes = [C.c_void_p]
        L.rtamd_host_scene_camera.restype = C.c_int
        L.rtamd_host_scene_camera.argtypes = [C.c_void_p, C.c_void_p]
        L.rtamd_host_scene_save.restype = C.c_int
        L.rtamd_host_scene_save.argtypes = [C.c_void_p, C.c_char_p]
        L.rtamd_host_scene_animation.restype = C.c_int
        L.rtamd_host_scene_animation.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.rtamd_host_load_image.restype = C.c_int
        L.rtamd_host_load_image.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_size_t]
        L.rtamd_host_scene_free.restype = None
        L.rtamd_host_scene_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _copy(ptr, n, dt):
    if not ptr or n == 0:
        return np.zeros(0, dtype=dt)
    buf = (C.c_uint8 * (n * dt.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dt, count=n).copy()


def build_scene(scene: int, image_width: int, image_height: int, samples_per_pixel: int, max_depth: int,
                use_bvh: bool = True, frame: int = 0, asset_dir: str | None = None,
                animation: int | None = None) -> abi.Scene:
    """Build reference scene `scene` (1..5) with the C++ scene API; the result carries the camera frame.  With
    animation = 0 (rotate + zoom) or 1 (translate) it also carries `.animation`: the camera frames of
    camera::render_animation (camera.h:46), one per frame."""
    L = load()
    h = L.rtamd_host_build_scene(scene, image_width, image_height, samples_per_pixel, max_depth, int(use_bvh), frame,
                                 (asset_dir or ASSET_DIR).encode())
    if not h:
        raise ValueError("rtamd_host_build_scene: bad arguments")
    try:
        d = L.rtamd_host_scene_desc(h).contents
        sc = abi.Scene(d.world_root, d.lights_root)
        for name, dt in abi._ARRAYS:
            sc.arrays[name] = _copy(getattr(d, name), getattr(d, "n_" + name), dt)
        cam = np.zeros(1, dtype=abi.camera_dtype)
        assert L.rtamd_host_scene_camera(h, cam.ctypes.data) == 0
        sc.camera = cam
        if animation is not None:
            n = L.rtamd_host_scene_animation(h, animation, None, 0)
            if n < 0:
                raise ValueError("rtamd_host_scene_animation: bad method")
            cams = np.zeros(n, dtype=abi.camera_dtype)
            assert L.rtamd_host_scene_animation(h, animation, cams.ctypes.data, n) == n
            sc.animation = cams
    finally:
        L.rtamd_host_scene_free(h)
    return sc


def load_image(path: str) -> np.ndarray:
    """rtw_image::load (stb_image_utils.h:46-55) for one file: (height, width, 3) linearised RGB8 texels.  JPEGs are
    decoded by include/rtamd/jpeg_baseline.h, PNGs by include/rtamd/png_decode.h; other formats are read from "<path>.rtwtex".  Raises RuntimeError for a
    file that is missing, unsupported (progressive JPEG ...) or corrupt."""
    L = load()
    w, h = C.c_int(0), C.c_int(0)
    if L.rtamd_host_load_image(path.encode(), C.byref(w), C.byref(h), None, 0) != 0:
        raise RuntimeError(f"cannot load image {path!r} (see stderr)")
    out = np.zeros((h.value, w.value, 3), dtype=np.uint8)
    if L.rtamd_host_load_image(path.encode(), C.byref(w), C.byref(h), out.ctypes.data, out.size) != 0:
        raise RuntimeError(f"cannot load image {path!r} (see stderr)")
    return out
