import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from raytracing_amd import abi, lib, camera
from tests import fixtures as fx
ctx = lib.Context(0)
for n, (W, H, spp) in {3: (1920, 1080, 64), 2: (600, 600, 256), 4: (1920, 1080, 64), 1: (1920, 1080, 16)}.items():
    sc = fx.scene(n); ds = ctx.upload(sc)
    cam = camera.scene_camera(n, W, H, spp, 50)
    for prec, name in ((abi.RT_PREC_F32, "f32"), (abi.RT_PREC_F64, "f64")):
        for rep in range(2):
            acc, rgb, st = ds.render(cam, seed=1, precision=prec, want_rgb8=False)
        print(f"scene {n} {W}x{H} spp {spp} {name}: {st['kernel_ms']:.1f} ms  {st['samples']/st['kernel_ms']/1e3:.1f} Msamples/s  rays/sample {st['rays']/st['samples']:.2f}  Mrays/s {st['rays']/st['kernel_ms']/1e3:.1f}", flush=True)
    ds.close()
