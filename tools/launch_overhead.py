"""Kernel time of the headline frame for small stratum ranges: the fixed cost of a launch (start-up, end-of-kernel tail)."""
import os, sys
sys.path.insert(0, os.getcwd())
from raytracing_amd import abi, lib, scenes
scene = scenes.build("final_scene", 1920, 1080, 1000, 50)
ctx = lib.Context(0); ds = ctx.upload(scene)
for s_end in (1, 2, 4, 8, 16, 32, 64, 120, 240):
    best = 1e9
    for rep in range(3):
        _, _, st = ds.render(scene.camera, seed=1, s_begin=0, s_end=s_end, precision=abi.RT_PREC_F32, want_rgb8=False)
        best = min(best, st["kernel_ms"])
    print(f"strata {s_end}: {best:.2f} ms ({best / s_end:.3f} ms per stratum)", flush=True)
