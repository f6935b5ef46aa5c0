"""rt_scene_tune on the reference's configurations: probe times of the candidates, the winner, the full frame before / after."""
import os, sys, time
sys.path.insert(0, os.getcwd())
from raytracing_amd import abi, lib, scenes
ctx = lib.Context(0)
for name, W, H, spp in (("first_scene", 1920, 1080, 100), ("cornell_box", 600, 600, 1000), ("final_scene", 1920, 1080, 1000),
                        ("mesh_scene", 1920, 1080, 500), ("first_scene_moving", 1920, 1080, 100)):
    scene = scenes.build(name, W, H, spp, 50)
    ds = ctx.upload(scene)
    def full():
        return min(ds.render(scene.camera, seed=1, precision=abi.RT_PREC_F32, want_rgb8=False)[2]["kernel_ms"] for _ in range(2))
    before = full()
    t0 = time.perf_counter()
    rep = ds.tune(scene.camera)
    dt = (time.perf_counter() - t0) * 1e3
    after = full()
    print(f"{name}: full frame {before:.1f} -> {after:.1f} ms; chosen candidate {rep['chosen']}; probes {[round(x, 2) for x in rep['probe_ms']]}; tuning took {dt:.0f} ms", flush=True)
    ds.close()
