"""Frames of two library builds compared bit for bit (the path tracer's result does not depend on which lane traces what):
python tools/q_check.py LIB_A LIB_B [scene] [s_end]"""
import os, sys, subprocess
import numpy as np
code = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from raytracing_amd import abi, lib, scenes
name, s_end, outp = sys.argv[1], int(sys.argv[2]), sys.argv[3]
w, h = (1920, 1080) if name != "cornell_box" else (600, 600)
scene = scenes.build(name, w, h, 1000, 50)
ctx = lib.Context(0); ds = ctx.upload(scene)
best = 1e9
for rep in range(2):
    acc, _, st = ds.render(scene.camera, seed=1, s_begin=0, s_end=s_end, precision=abi.RT_PREC_F32, want_rgb8=False)
    best = min(best, st["kernel_ms"])
np.save(outp, np.asarray(acc))
print(f"{best:.1f} ms  rays/sample {st['rays']/st['samples']:.4f}", flush=True)
'''
a, b = sys.argv[1], sys.argv[2]
scene = sys.argv[3] if len(sys.argv) > 3 else "final_scene"
s_end = sys.argv[4] if len(sys.argv) > 4 else "64"
os.makedirs("gpurun_out", exist_ok=True)
outs = []
for name in (a, b):
    env = dict(os.environ)
    if name != "cur":
        env["RT_AMD_LIB"] = os.path.join(os.getcwd(), "raytracing_amd", f"librt_amd_{name}.so")
    outp = f"gpurun_out/qcheck_{name}.npy"
    r = subprocess.run([sys.executable, "-c", code, scene, s_end, outp], env=env, capture_output=True, text=True, timeout=120)
    print(name, "->", r.stdout.strip() or r.stderr.strip()[-1500:], flush=True)
    if r.returncode != 0:
        print(r.stderr.strip()[-1500:]); sys.exit(1)
    outs.append(np.load(outp))
    os.remove(outp)
x, y = outs
same = np.array_equal(x, y, equal_nan=True)
print("identical" if same else f"DIFFERENT: {np.sum(x != y)} of {x.size} cells, max abs diff {np.nanmax(np.abs(x - y))}")
sys.exit(0 if same else 2)
