"""A/B timing of library builds / knob settings on the headline frame: args are LIB[:ENV=VAL[:ENV=VAL...]]"""
import os, sys, subprocess
code = r'''
import os, sys
sys.path.insert(0, os.getcwd())
from raytracing_amd import abi, lib, scenes
scene = scenes.build("final_scene", 1920, 1080, 1000, 50)
ctx = lib.Context(0); ds = ctx.upload(scene, device_build=bool(os.environ.get("DEVICE_BUILD")))  # DEVICE_BUILD=1: hierarchy built on the GPU
best = 1e9
for rep in range(2):
    _, _, st = ds.render(scene.camera, seed=1, s_begin=0, s_end=int(os.environ.get("S_END", "-1")), precision=abi.RT_PREC_F32, want_rgb8=False)
    best = min(best, st["kernel_ms"])
print(f"{best:.1f} ms  {st['samples']/best/1e3:.1f} Msamples/s  rays/sample {st['rays']/st['samples']:.4f}", flush=True)
'''
for spec in sys.argv[1:]:
    parts = spec.split(":")
    env = dict(os.environ)
    if parts[0] != "cur":
        env["RT_AMD_LIB"] = os.path.join(os.getcwd(), "raytracing_amd", f"librt_amd_{parts[0]}.so")
    for kv in parts[1:]:
        k, v = kv.split("=")
        env[k] = v
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=90)
    print(spec, "->", r.stdout.strip() or r.stderr.strip()[-300:], flush=True)
    if r.stdout.strip() and "PROFILE" in spec:  # diagnostic builds print their tables on stderr
        print(r.stderr.strip()[-6000:], flush=True)
