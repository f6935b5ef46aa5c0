"""f32 rt_trace_batch against the reference fixtures, ray by ray (run on the GPU box; writes gpurun_out/f32_parity.json
and gpurun_out/f32_parity_table.json -- copy the latter to tests/golden/ after a change to the float kernels).

For every fixture ray: does the production (float) path pick the reference's primitive, and how far are t and the normal
from the reference's doubles?  Every ray beyond 1e-5 (north_star's bar) is listed with the analytic float bound that
explains it -- tests/f32_bounds.py -- and every id mismatch with the distance between the two candidates."""
import json
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402

from raytracing_amd import abi, lib  # noqa: E402
from tests import f32_bounds as fb  # noqa: E402
from tests import fixtures as fx  # noqa: E402

ctx = lib.Context(0)
os.makedirs("gpurun_out", exist_ok=True)
report = {}
for n in (1, 2, 3, 4, 6, "box3"):
    sc = fx.scene(3 if n == "box3" else n)
    ds = ctx.upload(sc)
    rays, xi, hits = fx.boxtrace(3) if n == "box3" else fx.trace(n)
    got = ds.trace_batch(rays, xi, abi.RT_PREC_F32)
    g64 = ds.trace_batch(rays, xi, abi.RT_PREC_F64)
    assert (g64["prim"] == hits["prim"]).all()
    np.save(f"gpurun_out/f32_got{n}.npy", got)
    rows = fb.classify(sc, rays, hits, got, xi)
    rep = fb.summarise(rows)
    report[str(n)] = rep
    print(f"scene {n}: {rep['n_rays']} rays, {rep['n_same']} same id, {rep['n_tie']} equal-t ties ({rep['n_same_surface']} of them coplanar faces of one material), {rep['n_grazing']} grazing, "
          f"{rep['n_unexplained_id']} unexplained ids; over 1e-5: t {rep['n_t_over']} (max {rep['max_rel_t']:.3g}, max/bound {rep['max_t_over_bound']:.3g}), "
          f"normal {rep['n_n_over']} (max {rep['max_dn']:.3g}, max/bound {rep['max_n_over_bound']:.3g})")
    print(f"    measured / bound (C = {fb.C}): t {rep['t_over_bound_quantiles']}  normal {rep['n_over_bound_quantiles']}  min cos {rep['min_cos']:.3g}  largest t-bound {rep['max_bound_t']:.3g}")
    for r in rep["exceptions"][:4]:
        print("   ", r)
    ds.close()
json.dump(report, open("gpurun_out/f32_parity.json", "w"), indent=1)
# the committed table: every ray beyond 1e-5 with its bound, every id mismatch with its explanation
table = {}
for name, rep in report.items():
    table[name] = {
        "C": fb.C, "C_N": fb.C_N, "t_over_bound": rep["t_over_bound_quantiles"], "n_over_bound": rep["n_over_bound_quantiles"], "min_cos": rep["min_cos"],
        "max_bound_t": rep["max_bound_t"],
        "n_rays": rep["n_rays"], "n_same_id": rep["n_same"], "n_tie_same_surface": rep["n_same_surface"], "max_rel_t": rep["max_rel_t"], "max_dn": rep["max_dn"],
        "beyond_1e-5": [{k: e[k] for k in ("index", "prim", "kind", "rel_t", "dn", "bound_t", "bound_n", "cos", "M", "dist")}
                        for e in rep["exceptions"]],
        "id_mismatches": [{k: m[k] for k in ("index", "prim", "got", "t_ref", "t_got", "status", "why")} for m in rep["mismatches"]],
    }
json.dump(table, open("gpurun_out/f32_parity_table.json", "w"), indent=0)
