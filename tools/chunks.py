"""Chunk schedules (RT_AMD_CHUNKS=min,divisor,max,percent; dev build): kernel time of a 1/8 range and of the whole frame."""
import os, sys
sys.path.insert(0, os.getcwd())
os.environ.setdefault("RT_AMD_LIB", os.path.join(os.getcwd(), "raytracing_amd", "librt_amd_dev.so"))
from raytracing_amd import abi, lib, scenes
scene = scenes.build("final_scene", 1920, 1080, 1000, 50)
for env in (None, "16,3", "16,2", "16,0,0,67", "16,0,0,75", "16,0,0,85", "32,0,0,85", "16,0,0,90"):
    if env: os.environ["RT_AMD_CHUNKS"] = env
    ctx = lib.Context(0); ds = ctx.upload(scene)
    for s_end in (120, 961):
        best = 1e9
        for rep in range(2):
            _, _, st = ds.render(scene.camera, seed=1, s_begin=0, s_end=s_end, precision=abi.RT_PREC_F32, want_rgb8=False)
            best = min(best, st["kernel_ms"])
        print(f"sched {env} s_end {s_end}: {best:.1f} ms  {st['samples']/best/1e3:.1f} Msamples/s", flush=True)
    ds.close(); ctx.close()
