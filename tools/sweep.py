"""Development sweep of the kernel's scheduling knobs on the headline workload (kernel_ms from hip events)."""
import os, sys, itertools
import numpy as np
sys.path.insert(0, os.getcwd())
from raytracing_amd import abi, lib, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 100
scene = scenes.build("final_scene", 1920, 1080, spp, 50)
cam = scene.camera
def run(env):
    for k in ("RT_AMD_LDS_UNITS", "RT_AMD_LEAF_T", "RT_AMD_SHADE_T"):
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in env.items()})
    ctx = lib.Context(0); ds = ctx.upload(scene)
    best = 1e30
    for rep in range(2):
        _, _, st = ds.render(cam, seed=1, precision=abi.RT_PREC_F32, want_rgb8=False)
        best = min(best, st["kernel_ms"])
    ds.close(); ctx.close()
    print(env, f"{best:.1f} ms  {st['samples']/best/1e3:.1f} Msamples/s  rays/sample {st['rays']/st['samples']:.3f}", flush=True)
configs = [dict()]
for lt, stt in itertools.product((1, 8, 16, 24, 32, 48), (1, 8, 16, 24, 32, 48)):
    configs.append({"RT_AMD_LEAF_T": lt, "RT_AMD_SHADE_T": stt})
configs.append({"RT_AMD_LDS_UNITS": 0})
for c in configs:
    run(c)
