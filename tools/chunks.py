import os, sys
sys.path.insert(0, os.getcwd())
from raytracing_amd import abi, lib, scenes
scene = scenes.build("final_scene", 1920, 1080, 1000, 50)
for env in (None, "16,3", "16,4", "16,6", "16,3,160", "16,2,128", "16,2,64", "16,3,96", "16,4,200"):
    if env: os.environ["RT_AMD_CHUNKS"] = env
    ctx = lib.Context(0); ds = ctx.upload(scene)
    for s_end in (120, 961):
        best = 1e9
        for rep in range(2):
            _, _, st = ds.render(scene.camera, seed=1, s_begin=0, s_end=s_end, precision=abi.RT_PREC_F32, sample_chunks=(6 if s_end > 500 else 4) if env is None else 0, want_rgb8=False)
            best = min(best, st["kernel_ms"])
        print(f"sched {env} s_end {s_end}: {best:.1f} ms  {st['samples']/best/1e3:.1f} Msamples/s", flush=True)
    ds.close(); ctx.close()
