"""Animation sequences of the reference, sharded by frame over GPUs.

The reference has three sequences, all on first_scene's spheres (README "First Scene Animation"):
  method 0  rotate + zoom, 62 frames     camera::render_animation, camera.h:49-59, 93-96
  method 1  translate, 42 frames         camera::render_animation, camera.h:99-119
  method 2  bouncing spheres, 20 frames  the scene is rebuilt per frame, cpuRenderer.cpp:19-84, 447-451
Frames are independent, so frame f goes to rank f mod world and nothing is exchanged (distributed.frame_shards).
The camera paths come from the C++ scene API (include/rtamd/scene_api.h: camera::animation_step), which
tests/test_animation.py pins byte for byte against the frames of the reference's own render_animation() run."""
from __future__ import annotations

from . import abi, scenes
from .distributed import frame_shards

N_BOUNCE_FRAMES = 20  # cpuRenderer.cpp:448


def n_frames(method: int) -> int:
    if method not in (0, 1, 2):
        raise ValueError("animation method must be 0 (rotate), 1 (translate) or 2 (bounce)")
    return {0: 62, 1: 42, 2: N_BOUNCE_FRAMES}[method]


def frames(method: int, image_width: int, image_height: int, samples_per_pixel: int, max_depth: int,
           use_bvh: bool = True, rank: int = 0, world: int = 1, name: str = "first_scene"):
    """Yield (frame index, scene, camera) for this rank's frames of the sequence.  For methods 0 and 1 every frame
    shares ONE scene object (upload it once) -- `name` picks the scene function the camera moves through, as argv[1]
    does in the reference (cpuRenderer.cpp:453-458); method 2 builds a new scene per frame and ignores `name`."""
    mine = frame_shards(n_frames(method), world, rank)
    if method == 2:
        for f in mine:
            sc = scenes.build("first_scene_moving", image_width, image_height, samples_per_pixel, max_depth, use_bvh, frame=f)
            yield f, sc, sc.camera
        return
    sc = scenes.build(name, image_width, image_height, samples_per_pixel, max_depth, use_bvh, animation=method)
    for f in mine:
        yield f, sc, sc.animation[f:f + 1]


def render(ctx, method: int, image_width: int, image_height: int, samples_per_pixel: int, max_depth: int,
           use_bvh: bool = True, rank: int = 0, world: int = 1, seed: int = 1984, precision: int = abi.RT_PREC_F32,
           name: str = "first_scene"):
    """Render this rank's frames on `ctx` (raytracing_amd.lib.Context); yields (frame index, accum, rgb8, stats)."""
    ds, held = None, None
    try:
        for f, sc, cam in frames(method, image_width, image_height, samples_per_pixel, max_depth, use_bvh, rank, world,
                                 name):
            if sc is not held:
                if ds is not None:
                    ds.close()
                ds, held = ctx.upload(sc), sc
            accum, rgb, stats = ds.render(cam, seed=seed, precision=precision)
            yield f, accum, rgb, stats
    finally:
        if ds is not None:
            ds.close()
