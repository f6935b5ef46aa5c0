"""Throughput of the render kernel vs sample count on final_scene 1080p (how much of the headline frame is start-up / tail)."""
import os, sys
sys.path.insert(0, os.getcwd())
from raytracing_amd import abi, lib, scenes
ctx = lib.Context(0)
for spp in (int(x) for x in (sys.argv[1:] or ["121", "961", "3844"])):
    scene = scenes.build("final_scene", 1920, 1080, spp, 50)
    ds = ctx.upload(scene)
    best = 1e9
    for rep in range(2):
        _, _, st = ds.render(scene.camera, seed=1, precision=abi.RT_PREC_F32, want_rgb8=False)
        best = min(best, st["kernel_ms"])
    print(f"spp {spp}: {best:.1f} ms  {st['samples']/best/1e3:.1f} Msamples/s", flush=True)
    ds.close()
