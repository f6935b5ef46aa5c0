"""f32 rt_trace_batch against the reference fixtures: which rays pick another primitive, and how close the two hits are."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from raytracing_amd import abi, lib
from tests import fixtures as fx
ctx = lib.Context(0)
for n in (1, 2, 3, 4):
    sc = fx.scene(n); ds = ctx.upload(sc)
    rays, xi, hits = fx.trace(n)
    got = ds.trace_batch(rays, xi, abi.RT_PREC_F32)
    bad = np.nonzero(got["prim"] != hits["prim"])[0]
    print("scene", n, len(bad), "of", len(rays))
    for k in bad[:12]:
        print("  ray", k, "got", got["prim"][k], got["t"][k], "want", hits["prim"][k], hits["t"][k], "kinds", sc.nodes["kind"][got["prim"][k]] if got["prim"][k] >= 0 else -1, sc.nodes["kind"][hits["prim"][k]] if hits["prim"][k] >= 0 else -1)
    ds.close()
