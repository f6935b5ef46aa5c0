import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from raytracing_amd import abi, lib
from tests import fixtures as fx
from tests import oracle_lib as O
ctx = lib.Context(0)
rng = np.random.default_rng(1)
for n in (1, 2):
    sc = fx.scene(n); ds = ctx.upload(sc)
    rays, xi, hits = fx.trace(n)
    got = ds.trace_batch(rays, xi, abi.RT_PREC_F64)
    hit = got["prim"] >= 0
    p = got["p"][hit]; nrm = got["normal"][hit]; d = rays["dir"][hit]
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    refl = d - 2 * (d * nrm).sum(1, keepdims=True) * nrm
    into = -nrm + 0.3 * rng.standard_normal(nrm.shape)
    graze = refl + 0.0
    new = np.zeros(3 * len(p), dtype=abi.ray_dtype)
    new["orig"] = np.concatenate([p, p, p]); new["dir"] = np.concatenate([refl, into, d]); new["time"] = 0.5
    xi2 = np.full(len(new), 0.5)
    g2 = ds.trace_batch(new, xi2, abi.RT_PREC_F64)
    w2, _ = O.trace_batch(sc, new, xi2)
    bad = np.nonzero(g2["prim"] != w2["prim"])[0]
    print(n, len(new), "mismatch", len(bad))
    for k in bad[:5]:
        print("  ", k, g2["prim"][k], w2["prim"][k], g2["t"][k], w2["t"][k], new["orig"][k], new["dir"][k])
    ok = (g2["prim"] == w2["prim"]) & (w2["prim"] >= 0)
    print("   max |dt|", np.abs(g2["t"][ok] - w2["t"][ok]).max(), "max |dn|", np.abs(g2["normal"][ok] - w2["normal"][ok]).max(), "ff mism", (g2["front_face"][ok] != w2["front_face"][ok]).sum())
    ds.close()
