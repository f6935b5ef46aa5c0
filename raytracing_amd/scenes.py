"""The reference's scenes as flat scene descriptions, built by the C++ host library (raytracing_amd/host)."""
from __future__ import annotations

from . import abi, host

SCENE_IDS = {"first_scene": 1, "cornell_box": 2, "final_scene": 3, "mesh_scene": 4, "first_scene_moving": 5,
             "cornell_smoke": 6}  # 6: not a reference scene function; a scene its API permits (general medium boundaries)


def build(name: str, image_width: int, image_height: int, samples_per_pixel: int, max_depth: int, use_bvh: bool = True,
          frame: int = 0, animation: int | None = None) -> abi.Scene:
    """Scene + camera frame (scene.camera) for one of the reference's scene functions (cpuRenderer.cpp:19-401).
    animation = 0 | 1 adds scene.animation, the camera frames of camera::render_animation (rotate + zoom | translate)."""
    return host.build_scene(SCENE_IDS[name], image_width, image_height, samples_per_pixel, max_depth, use_bvh, frame,
                            animation=animation)
