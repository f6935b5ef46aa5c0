import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from raytracing_amd import abi, lib
from tests import fixtures as fx
from tests import oracle_lib as O
ctx = lib.Context(0)
for n in (1, 2, 3, 4):
    sc = fx.scene(n); ds = ctx.upload(sc)
    print(n, ds.info())
    cam, _, _, _ = fx.img(n)
    cam = fx.with_size(cam, sqrt_spp=4, max_depth=12)
    want, cnt = O.render(sc, cam, seed=3, rng_mode=1, threads=8)
    acc, rgb, st = ds.render(cam, seed=3, precision=abi.RT_PREC_F64)
    rel = np.abs(acc.astype(np.float64) - want) / np.maximum(np.abs(want), 1e-3)
    bad = np.argwhere(rel.max(axis=2) > 1e-6)
    print(n, "rays", st["rays"], int(cnt[4]), "bad px", len(bad), bad[:10].tolist(), flush=True)
    for (j, i) in bad[:3]:
        print("   ", j, i, acc[j, i], want[j, i])
    ds.close()
