import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from raytracing_amd import abi, lib
from tests import fixtures as fx
from tests import oracle_lib as O
ctx = lib.Context(0)
for n in (2, 1, 4, 3):
    sc = fx.scene(n); ds = ctx.upload(sc)
    cam, mean, spp, runs = fx.conv(n)
    ref = O.write_color(mean, 1.0)
    print(n, "ref nan px", int(np.isnan(mean).any(axis=2).sum()), "ref mean", np.nanmean(mean), "runs nan", [int(np.isnan(r).any(axis=2).sum()) for r in runs], flush=True)
    k = 130
    c = fx.with_size(cam, sqrt_spp=k)
    res = {}
    for prec, name in ((abi.RT_PREC_F32, "f32"), (abi.RT_PREC_F64, "f64")):
        acc, _, st = ds.render(c, seed=5, precision=prec)
        m = acc.astype(np.float64) / (k * k)
        res[name] = m
        img = O.write_color(m, 1.0)
        print(n, name, "psnr", round(fx.psnr8(img, ref), 2), "mean ratio", np.nanmean(m) / np.nanmean(mean), "nan px", int(np.isnan(m).any(axis=2).sum()),
          "rays/sample", st["rays"] / st["samples"], "ms", st["kernel_ms"], flush=True)
    d = res["f32"] - res["f64"]
    print(n, "f32-f64 psnr", round(fx.psnr8(O.write_color(res["f32"],1.0), O.write_color(res["f64"],1.0)),2))
    # per-run reference vs mean psnr
    for r in runs[:2]:
        print("   ref run vs mean psnr", round(fx.psnr8(O.write_color(r.astype(np.float64),1.0), ref),2))
    np.save(f"gpurun_out/dbg3_{n}_f32.npy", res["f32"]); np.save(f"gpurun_out/dbg3_{n}_f64.npy", res["f64"])
    ds.close()
