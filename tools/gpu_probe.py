#!/usr/bin/env python3
"""Exploratory GPU run: parity numbers of the HIP path against the oracle and the reference fixtures."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import abi, lib
from tests import fixtures as fx
from tests import oracle_lib as O

ctx = lib.Context(0)
for n in (1, 2, 3, 4):
    sc = fx.scene(n)
    ds = ctx.upload(sc)
    print(f"== scene {n} {fx.SCENE_NAMES[n]}: {ds.info()}", flush=True)
    rays, xi, hits = fx.trace(n)
    for prec, name in ((abi.RT_PREC_F64, "f64"), (abi.RT_PREC_F32, "f32")):
        got = ds.trace_batch(rays, xi, prec)
        same = got["prim"] == hits["prim"]
        hit = same & (hits["prim"] >= 0)
        rel_t = np.abs(got["t"][hit] - hits["t"][hit]) / np.abs(hits["t"][hit])
        dn = np.abs(got["normal"][hit] - hits["normal"][hit]).max(axis=1)
        dp = np.abs(got["p"][hit] - hits["p"][hit]).max(axis=1)
        duv = np.maximum(np.abs(got["u"][hit] - hits["u"][hit]), np.abs(got["v"][hit] - hits["v"][hit]))
        print(f"  trace {name}: id mismatches {int((~same).sum())}/{len(rays)}  max rel t {rel_t.max():.3e}  "
              f"p99.9 rel t {np.quantile(rel_t, 0.999):.3e} max|dn| {dn.max():.3e} max|dp| {dp.max():.3e} max|duv| {duv.max():.3e} "
              f"ff mism {int((got['front_face'][hit] != hits['front_face'][hit]).sum())}", flush=True)
    cam, seed, sums, by = fx.img(n)
    cam = fx.with_size(cam, sqrt_spp=4, max_depth=12)
    t0 = time.time(); oacc, cnt = O.render(sc, cam, seed=3, rng_mode=1, threads=8); t1 = time.time()
    for prec, name in ((abi.RT_PREC_F64, "f64"), (abi.RT_PREC_F32, "f32")):
        acc, rgb, st = ds.render(cam, seed=3, precision=prec)
        d = np.abs(acc.astype(np.float64) - oacc)
        rel = d / np.maximum(np.abs(oacc), 1e-3)
        nn = np.isnan(acc).sum(); on = np.isnan(oacc).sum()
        ok = np.isfinite(rel)
        print(f"  render {name}: rays gpu {st['rays']} oracle {int(cnt[4])}  kernel {st['kernel_ms']:.2f} ms  median rel {np.median(rel[ok]):.2e} "
              f"frac(rel<1e-4) {np.mean(rel[ok] < 1e-4):.4f} max rel {rel[ok].max():.2e} NaN gpu/oracle {nn}/{on}", flush=True)
        if rgb is not None:
            wc = O.write_color(acc.astype(np.float64), 1.0 / 16)
            print(f"    tonemap mismatches vs oracle write_color on same sums: {int((wc != rgb).sum())}")
    ds.close()
print("probe done")
