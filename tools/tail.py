import os, sys
sys.path.insert(0, os.getcwd())
os.environ["RT_AMD_PROFILE_TAIL"] = "1"
from raytracing_amd import abi, lib, scenes
ctx = lib.Context(0)
for spp in (int(x) for x in sys.argv[1:]):
    scene = scenes.build("final_scene", 1920, 1080, spp, 50)
    ds = ctx.upload(scene)
    for rep in range(2):
        _, _, st = ds.render(scene.camera, seed=1, precision=abi.RT_PREC_F32, want_rgb8=False)
    print(f"spp {spp}: {st['kernel_ms']:.1f} ms", flush=True)
    ds.close()
