"""Kernel time of the reference's benchmark configurations (BASELINE.json configs) on one GPU.
DEVICE_BUILD=1: the hierarchy built on the GPU (scenes it does not take keep the host build)."""
import os, sys
sys.path.insert(0, os.getcwd())
from raytracing_amd import abi, lib, scenes
ctx = lib.Context(0)
for name, W, H, spp in (("first_scene", 1920, 1080, 100), ("cornell_box", 600, 600, 1000), ("final_scene", 1920, 1080, 1000),
                        ("mesh_scene", 1920, 1080, 500), ("first_scene_moving", 1920, 1080, 100)):
    scene = scenes.build(name, W, H, spp, 50)
    try:
        ds = ctx.upload(scene, device_build=bool(os.environ.get("DEVICE_BUILD")))
    except lib.RtError:
        ds = ctx.upload(scene)
    best = 1e9
    for rep in range(2):
        _, _, st = ds.render(scene.camera, seed=1, precision=abi.RT_PREC_F32, want_rgb8=False)
        best = min(best, st["kernel_ms"])
    print(f"{name} {W}x{H} spp {spp}: {best:.1f} ms  {st['samples']/best/1e3:.1f} Msamples/s  rays/sample {st['rays']/st['samples']:.2f}  {ds.info()}", flush=True)
    ds.close()
