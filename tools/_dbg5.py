import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from raytracing_amd import abi, lib
from tests import fixtures as fx
from tests import oracle_lib as O
ctx = lib.Context(0)
for n, d in ((1, 2), (2, 3)):
    sc = fx.scene(n); ds = ctx.upload(sc)
    cam0, _, _, _ = fx.img(n)
    for k in (1, 4):
        cam = fx.with_size(cam0, sqrt_spp=k, max_depth=d)
        want, cnt = O.render(sc, cam, seed=3, rng_mode=1, threads=8)
        acc, rgb, st = ds.render(cam, seed=3, precision=abi.RT_PREC_F64)
        acc32, _, _ = ds.render(cam, seed=3, precision=abi.RT_PREC_F32)
        rel = np.abs(acc.astype(np.float64) - want) / np.maximum(np.abs(want), 1e-3)
        bad = np.argwhere(rel.max(axis=2) > 1e-6)
        print(n, "depth", d, "sqrt_spp", k, "rays", st["rays"], int(cnt[4]), "bad px", len(bad), flush=True)
        for (j, i) in bad[:8]:
            print("   ", j, i, "f64", acc[j, i], "want", want[j, i], "f32", acc32[j, i])
    ds.close()
