import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from raytracing_amd import abi, lib
from tests import fixtures as fx
from tests import oracle_lib as O
ctx = lib.Context(0)
for n in (1, 2, 3, 4):
    sc = fx.scene(n); ds = ctx.upload(sc)
    cam, mean, spp, runs = fx.conv(n)
    ref = O.write_color(mean, 1.0)
    k = 300
    c = fx.with_size(cam, sqrt_spp=k)
    acc, _, st = ds.render(c, seed=5, precision=abi.RT_PREC_F32)
    m = acc.astype(np.float64) / (k * k)
    img = O.write_color(m, 1.0)
    print(n, "f32 psnr", round(fx.psnr8(img, ref), 2), "mean ratio", np.nanmean(m) / np.nanmean(mean), "nan px", int(np.isnan(m).any(axis=2).sum()),
          "inf px", int(np.isinf(m).any(axis=2).sum()), "rays/sample", st["rays"] / st["samples"], "ms", st["kernel_ms"], flush=True)
    ds.close()
