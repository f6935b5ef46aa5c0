"""The host-side scene API (include/rtamd/scene_api.h) and the scenes written against it (raytracing_amd/host).

CPU: scenes built through the API -- std::rand geometry, perlin tables, OBJ parse + normalisation, the host BVH build
(median split + std::sort) and the flattening -- are byte-identical to scenes serialised from the reference's own
object graph (fixtures from oracle/refgen.cpp).  Where /root/reference exists, the reference's own cpuRenderer.cpp is
compiled unchanged against the compatibility headers (drop-in check of the source-level boundary).
GPU: the reference-style command-line driver renders through camera::render() and matches the ctypes path."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from raytracing_amd import abi, scenes
from tests import fixtures as fx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CPP = "/root/reference/src/cpu/cpuRenderer.cpp"


@pytest.mark.parametrize("n,name", [(1, "first_scene"), (2, "cornell_box"), (3, "final_scene"), (4, "mesh_scene"), (6, "cornell_smoke")])
def test_scene_api_builds_reference_graph_byte_for_byte(n, name):
    built = scenes.build(name, 64, 36, 9, 10)
    ref = fx.scene(n)
    for key, _ in abi._ARRAYS:
        assert built.arrays[key].tobytes() == ref.arrays[key].tobytes(), key
    assert (built.world_root, built.lights_root) == (ref.world_root, ref.lights_root)
    assert built.camera.tobytes() == ref.camera.tobytes()


def test_bounce_frame_and_no_bvh_variants():
    ref = abi.Scene.load(os.path.join(fx.GOLD, "scene5_f7.npz"))
    built = scenes.build("first_scene_moving", 64, 36, 9, 10, frame=7)
    assert built.equal(ref) and built.camera.tobytes() == ref.camera.tobytes()
    ref = abi.Scene.load(os.path.join(fx.GOLD, "scene2_nobvh.npz"))
    built = scenes.build("cornell_box", 64, 36, 9, 10, use_bvh=False)
    assert built.equal(ref)
    assert (built.nodes["kind"] == abi.RT_NODE_BVH).sum() == 0


def test_build_is_repeatable_and_rejects_bad_arguments():
    a = scenes.build("final_scene", 32, 18, 4, 5)
    b = scenes.build("final_scene", 32, 18, 4, 5)
    assert a.equal(b)  # srand(1) per build: independent of what ran before
    from raytracing_amd import host
    with pytest.raises(ValueError):
        host.build_scene(9, 32, 18, 4, 5)
    with pytest.raises(ValueError):
        host.build_scene(1, 0, 18, 4, 5)


@pytest.mark.skipif(not os.path.exists(REF_CPP), reason="reference sources not present on this machine")
def test_reference_scene_code_compiles_unchanged_against_the_api(tmp_path):
    """Drop-in boundary: the reference's own main + scene functions, compiled against include/rtamd/compat and linked
    with the HIP library (the .cpp is copied to a scratch dir only so that its quoted includes resolve to our headers)."""
    src = tmp_path / "cpuRenderer.cpp"
    shutil.copy(REF_CPP, src)
    exe = tmp_path / "cpuRayTracer_amd"
    cmd = ["g++", "-std=gnu++17", "-O1", "-w", "-I", os.path.join(ROOT, "include", "rtamd", "compat"), str(src), "-o",
           str(exe), "-L", os.path.join(ROOT, "raytracing_amd"), "-lrt_amd",
           "-Wl,-rpath," + os.path.join(ROOT, "raytracing_amd")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert exe.exists()


def read_ppm(path):
    tok = open(path).read().split()
    assert tok[0] == "P3"
    w, h, mx = int(tok[1]), int(tok[2]), int(tok[3])
    return np.array(tok[4:], dtype=np.int32).reshape(h, w, 3)


@pytest.mark.gpu
def test_cli_driver_renders_like_the_ctypes_path(tmp_path, gpu_ctx):
    (tmp_path / "images").mkdir()
    exe = os.path.join(ROOT, "raytracing_amd", "rtamd_render")
    env = dict(os.environ, RTAMD_ASSETS=os.path.join(ROOT, "raytracing_amd", "assets"))
    args = "2 16 8 false 0 true true 8 64 48".split()  # reference argv contract (cpuRenderer.cpp:426-438)
    r = subprocess.run([exe, *args], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Total render time (ms):" in r.stderr
    img = read_ppm(tmp_path / "images" / "test.ppm")
    sc = scenes.build("cornell_box", 64, 48, 16, 8)
    ds = gpu_ctx.upload(sc)
    _, rgb, _ = ds.render(sc.camera, seed=1984, precision=abi.RT_PREC_F32)
    ds.close()
    assert img.shape == (48, 64, 3)
    assert (img == rgb.astype(np.int32)).all()
