"""Randomised hierarchy builds (RT_AMD_BVH_JITTER=seed, dev build): probe time at S_END strata for seeds 1..N, then the full
frame for the best few.  usage: python tools/jitter_sweep.py [scene] [n_seeds] [probe strata]"""
import os, sys
sys.path.insert(0, os.getcwd())
os.environ.setdefault("RT_AMD_LIB", os.path.join(os.getcwd(), "raytracing_amd", "librt_amd_dev.so"))
from raytracing_amd import abi, lib, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "final_scene"
n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 30
probe = int(sys.argv[3]) if len(sys.argv) > 3 else 12
W, H, spp = {"final_scene": (1920, 1080, 1000), "mesh_scene": (1920, 1080, 500), "cornell_box": (600, 600, 1000),
             "first_scene": (1920, 1080, 100), "first_scene_moving": (1920, 1080, 100)}[name]
scene = scenes.build(name, W, H, spp, 50)
ctx = lib.Context(0)
def timed(s_end, reps=2):
    ds = ctx.upload(scene)
    best = 1e9
    for _ in range(reps):
        _, _, st = ds.render(scene.camera, seed=1, s_begin=0, s_end=s_end, precision=abi.RT_PREC_F32, want_rgb8=False)
        best = min(best, st["kernel_ms"])
    ds.close()
    return best
res = []
for seed in range(0, n_seeds + 1):
    if seed: os.environ["RT_AMD_BVH_JITTER"] = str(seed)
    res.append((timed(probe), seed))
    print(f"seed {seed}: probe {res[-1][0]:.2f} ms", flush=True)
res.sort()
print("best probes:", res[:6])
for t, seed in res[:4] + [(0, 0)]:
    if seed: os.environ["RT_AMD_BVH_JITTER"] = str(seed)
    else: os.environ.pop("RT_AMD_BVH_JITTER", None)
    print(f"seed {seed}: probe {t:.2f} ms, full frame {timed(-1):.1f} ms", flush=True)
