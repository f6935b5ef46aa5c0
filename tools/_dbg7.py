import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from raytracing_amd import abi, lib
from tests import fixtures as fx
ctx = lib.Context(0)
sc = fx.scene(1); ds = ctx.upload(sc)
cam0, _, _, _ = fx.img(1)
cam = fx.with_size(cam0, sqrt_spp=1, max_depth=3)
acc, rgb, st = ds.render(cam, seed=3, precision=abi.RT_PREC_F64)
print(acc[2, 16], cam["image_width"])
