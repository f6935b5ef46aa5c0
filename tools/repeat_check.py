"""Do two identical launches trace the same paths?  (ray counts and images of repeated renders; run on the GPU box)"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from raytracing_amd import abi, lib, scenes
ctx = lib.Context(0)
for name, W, H, spp, depth, s_end in (("first_scene", 1920, 1080, 100, 50, 25), ("first_scene", 1920, 1080, 100, 8, 25), ("first_scene", 640, 360, 100, 50, 25),
                                      ("cornell_box", 600, 600, 1000, 50, 120), ("mesh_scene", 1920, 1080, 500, 50, 60)):
    sc = scenes.build(name, W, H, spp, depth)
    ds = ctx.upload(sc)
    out = []
    for rep in range(3):
        acc, _, st = ds.render(sc.camera, seed=7, s_begin=0, s_end=s_end, precision=abi.RT_PREC_F32, want_rgb8=False)
        out.append((acc, st["rays"]))
    d01 = int((out[0][0] != out[1][0]).sum() - (np.isnan(out[0][0]) & np.isnan(out[1][0])).sum())
    d02 = int((out[0][0] != out[2][0]).sum() - (np.isnan(out[0][0]) & np.isnan(out[2][0])).sum())
    print(name, W, H, "depth", depth, "rays", [o[1] for o in out], "values differing run0/run1:", d01, "run0/run2:", d02, flush=True)
    if d01:
        k = np.argwhere(out[0][0] != out[1][0])[:5]
        for j, i, c in k:
            print("   pixel", i, j, "channel", c, out[0][0][j, i, c], out[1][0][j, i, c])
    ds.close()
