"""Device-built stream against the host-built one: same record sequence? upload times, kernel time of a frame on each."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from raytracing_amd import abi, lib, scenes
from tests import stream_check
ctx = lib.Context(0)
cfgs = (("first_scene", 400, 225, 10), ("cornell_box", 600, 600, 1000), ("final_scene", 1920, 1080, 1000), ("mesh_scene", 1920, 1080, 500),
        ("cornell_smoke", 600, 600, 200), ("first_scene_moving", 1920, 1080, 100))
s_end = int(os.environ.get("S_END", "100"))
for name, W, H, spp in cfgs:
    scene = scenes.build(name, W, H, spp, 50)
    for rep in range(2):
        t0 = time.perf_counter(); ds = ctx.upload(scene, device_build=True); t_dev = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter(); host = ctx.upload(scene); t_host = (time.perf_counter() - t0) * 1e3
        if rep == 0:
            ds.close(); host.close()
    units, ids = ds.download_stream()
    h_units, h_ids = lib.flatten_stream(scene)
    rk = [k for _, k in stream_check.records(units)]
    hk = [k for _, k in stream_check.records(h_units)]
    same_len = len(units) == len(h_units)
    same_kinds = rk == hk
    same_ids = same_len and bool((np.asarray(ids) == np.asarray(h_ids)).all())
    n_diff = int((np.asarray(ids) != np.asarray(h_ids)).sum()) if same_len else -1
    se = min(s_end, int(spp ** 0.5) ** 2)
    def ms(d):
        return min(d.render(scene.camera, seed=1, s_begin=0, s_end=se, precision=abi.RT_PREC_F32, want_rgb8=False)[2]["kernel_ms"] for _ in range(2))
    print(f"{name}: units {len(units)} / host {len(h_units)}; kinds equal {same_kinds}; ids equal {same_ids} ({n_diff} differ); "
          f"upload device-build {t_dev:.2f} ms, host-build {t_host:.2f} ms; first {se} strata: device-built {ms(ds):.2f} ms, host-built {ms(host):.2f} ms", flush=True)
    ds.close(); host.close()
