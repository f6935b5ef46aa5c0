"""A/B of library builds / environment knobs over all reference configurations: args are LIB[:ENV=VAL...] as for tools/ab.py"""
import os, sys, subprocess
for spec in sys.argv[1:]:
    parts = spec.split(":")
    env = dict(os.environ)
    if parts[0] != "cur":
        env["RT_AMD_LIB"] = os.path.join(os.getcwd(), "raytracing_amd", f"librt_amd_{parts[0]}.so")
    for kv in parts[1:]:
        k, v = kv.split("=")
        env[k] = v
    r = subprocess.run([sys.executable, "tools/scenes_bench.py"], env=env, capture_output=True, text=True, timeout=120)
    print(spec, "->", " | ".join(l.split("rays/sample")[0].split(":")[0].split()[0] + " " + l.split(":")[1].split("ms")[0].strip() for l in r.stdout.strip().split("\n")) or r.stderr[-300:], flush=True)
