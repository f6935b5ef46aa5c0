import os, sys
sys.path.insert(0, os.getcwd())
os.environ["RT_AMD_PROFILE_ACTIONS"] = "1"
os.environ["RT_AMD_PROFILE_SHADE"] = "1"
os.environ.setdefault("RT_AMD_LIB", os.path.join(os.getcwd(), "raytracing_amd", "librt_amd_prof.so"))
from raytracing_amd import abi, lib, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "final_scene"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 100
W, H = (1920, 1080) if name != "cornell_box" else (600, 600)
if len(sys.argv) > 4: W, H = int(sys.argv[3]), int(sys.argv[4])
scene = scenes.build(name, W, H, spp, 50)
ctx = lib.Context(0); ds = ctx.upload(scene)
s_end = int(os.environ.get('S_END', '-1'))
for rep in range(2):
    _, _, st = ds.render(scene.camera, seed=1, s_begin=0, s_end=s_end, precision=abi.RT_PREC_F32, want_rgb8=False)
print(name, f"{st['kernel_ms']:.1f} ms  {st['samples']/st['kernel_ms']/1e3:.1f} Msamples/s rays/sample {st['rays']/st['samples']:.3f}")
