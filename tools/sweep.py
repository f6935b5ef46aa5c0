"""Development sweep of the kernel's scheduling knobs on the headline workload (kernel_ms from hip events)."""
import os, sys
sys.path.insert(0, os.getcwd())
from raytracing_amd import abi, lib, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
scene = scenes.build("final_scene", 1920, 1080, spp, 50)
cam = scene.camera
KEYS = ("RT_AMD_LDS_UNITS", "RT_AMD_WEIGHTS", "RT_AMD_SHADE_T", "RT_AMD_BOX_REPEAT")
def run(env, reps=1):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in env.items()})
    ctx = lib.Context(0); ds = ctx.upload(scene)
    best = 1e30
    for rep in range(reps):
        _, _, st = ds.render(cam, seed=1, precision=abi.RT_PREC_F32, want_rgb8=False)
        best = min(best, st["kernel_ms"])
    ds.close(); ctx.close()
    print(env, f"{best:.1f} ms  {st['samples']/best/1e3:.1f} Msamples/s", flush=True)
configs = [dict()]
for line in sys.argv[2:]:
    t, w, r = line.split(":")
    configs.append({"RT_AMD_SHADE_T": t, "RT_AMD_WEIGHTS": w, "RT_AMD_BOX_REPEAT": r})
for c in configs:
    run(c)
