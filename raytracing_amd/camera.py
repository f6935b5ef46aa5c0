"""Host-side mirror of camera::initialize() (reference src/cpu/camera.h:205-247) producing the rt_camera frame.

The C++ host API (include/rtamd/scene_api.h) computes the same frame with the reference's own libm; this Python
version exists for scripts and tests and is checked against the reference-generated fixtures."""
from __future__ import annotations

import math

import numpy as np

from . import abi

PI = 3.1415926535897932385  # utils.h:20


def _unit(v):
    return v * (1.0 / math.sqrt(float(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])))


def _cross(u, v):
    return np.array([u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]])


def make_camera(image_width: int, image_height: int, samples_per_pixel: int, max_depth: int, vfov: float, lookfrom,
                lookat, vup=(0, 1, 0), defocus_angle: float = 0.0, focus_dist: float = 10.0,
                background=(0, 0, 0)) -> np.ndarray:
    """aspect_ratio = W/H as cpuRenderer.cpp:437 passes it; image_height = int(W/aspect) (camera.h:206)."""
    aspect_ratio = float(image_width) / float(image_height)
    H = int(image_width / aspect_ratio)
    H = 1 if H < 1 else H
    lookfrom = np.asarray(lookfrom, dtype=np.float64)
    lookat = np.asarray(lookat, dtype=np.float64)
    vup = np.asarray(vup, dtype=np.float64)
    sqrt_spp = int(math.sqrt(samples_per_pixel))
    theta = vfov * PI / 180.0
    h = math.tan(theta / 2)
    viewport_height = 2 * h * focus_dist
    viewport_width = viewport_height * (float(image_width) / H)
    w = _unit(lookfrom - lookat)
    u = _unit(_cross(vup, w))
    v = _cross(w, u)
    viewport_u = viewport_width * u
    viewport_v = viewport_height * -v
    du = (1 / image_width) * viewport_u
    dv = (1 / H) * viewport_v
    upper_left = lookfrom - (focus_dist * w) - (1 / 2) * viewport_u - (1 / 2) * viewport_v
    pixel00 = upper_left + 0.5 * (du + dv)
    defocus_radius = focus_dist * math.tan((defocus_angle / 2) * PI / 180.0)
    cam = np.zeros(1, dtype=abi.camera_dtype)
    cam["image_width"] = image_width
    cam["image_height"] = H
    cam["sqrt_spp"] = sqrt_spp
    cam["max_depth"] = max_depth
    cam["center"] = lookfrom
    cam["pixel00_loc"] = pixel00
    cam["pixel_delta_u"] = du
    cam["pixel_delta_v"] = dv
    cam["defocus_disk_u"] = u * defocus_radius
    cam["defocus_disk_v"] = v * defocus_radius
    cam["defocus_angle"] = defocus_angle
    cam["background"] = background
    return cam


# camera parameters of the reference scenes (cpuRenderer.cpp:137-155, 266-283, 376-393, 199-216, 63-81)
SCENE_CAMERAS = {
    1: dict(vfov=20, lookfrom=(13, 2, 3), lookat=(0, 0, 0), defocus_angle=0.6, focus_dist=10.0, background=(0.70, 0.80, 1.00)),
    2: dict(vfov=40, lookfrom=(278, 278, -800), lookat=(278, 278, 0), defocus_angle=0, background=(0, 0, 0)),
    3: dict(vfov=40, lookfrom=(478, 278, -600), lookat=(278, 278, 0), defocus_angle=0, background=(0, 0, 0)),
    4: dict(vfov=40, lookfrom=(278, 278, -800), lookat=(278, 278, 0), defocus_angle=0, background=(0, 0, 0)),
    5: dict(vfov=24, lookfrom=(17, 4, 5), lookat=(0, 0, 0), defocus_angle=0.6, focus_dist=15.0, background=(0.70, 0.80, 1.00)),
}


def scene_camera(scene: int, image_width: int, image_height: int, samples_per_pixel: int, max_depth: int) -> np.ndarray:
    return make_camera(image_width, image_height, samples_per_pixel, max_depth, **SCENE_CAMERAS[scene])
