"""Animation sequences: camera::render_animation's camera paths, the per-frame scenes of the
bounce sequence, frame sharding over ranks.

Fixtures (tests/golden/make_golden.py): anim_translate.npz is the reference's OWN first_scene(animation=true) run --
camera::render_animation() at 24x16, 4 spp, depth 6, single thread, default rand() seed -- every frame's camera and
write_color ints; anim_rotate.npz holds the camera frames of the rotate + zoom branch (dead code in the reference's CPU
build: camera.h:48 assigns, so refgen drives the reference's rotate()/zoom()/initialize() in that branch's order)."""
import os
import socket
import subprocess

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracing_amd import abi, animation, scenes
from raytracing_amd.distributed import frame_shards
from tests import fixtures as fx
from tests import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP, DEPTH = 24, 16, 4, 6


def _gold(name):
    g = np.load(os.path.join(fx.GOLD, name))
    return g["cameras"].view(abi.camera_dtype), (g["images"] if "images" in g.files else None)


def test_translate_sequence_reproduces_the_reference_run_bit_for_bit():
    """Scene build + 42 frames through the oracle with the glibc rand() stream running on from the scene construction,
    like the reference process: same cameras (incl. frame 0, where u = v = w = 0 and only focus_dist moves), same ints."""
    cams, imgs = _gold("anim_translate.npz")
    sc = scenes.build("first_scene", W, H, SPP, DEPTH, animation=1)  # srand(1) + the scene's draws
    assert sc.animation.tobytes() == cams.tobytes()
    assert sc.animation["center"][0].tolist() == [13.0, 2.0, 3.0]  # frame 0 did not move
    for f in range(len(cams)):
        acc, _ = O.render(sc, sc.animation[f:f + 1], rng_mode=0)
        got = O.write_color(acc, 1.0 / SPP).reshape(H, W, 3)
        assert (got == imgs[f]).all(), f"frame {f}"


def test_rotate_sequence_cameras_match_reference():
    cams, _ = _gold("anim_rotate.npz")
    sc = scenes.build("first_scene", W, H, SPP, DEPTH, animation=0)
    assert len(sc.animation) == 62 == animation.n_frames(0)
    assert sc.animation.tobytes() == cams.tobytes()
    # the sequence is replayed from the scene function's camera on every call, and leaves the still camera alone
    again = scenes.build("first_scene", W, H, SPP, DEPTH, animation=0)
    assert again.animation.tobytes() == cams.tobytes()
    assert again.camera.tobytes() == scenes.build("first_scene", W, H, SPP, DEPTH).camera.tobytes()


def test_bounce_sequence_scenes():
    """method 2: one scene per frame (cpuRenderer.cpp:19-84); frame 7 is pinned against the reference's graph in
    test_host_api.py, here: the heights move as the schedule says and the frame index reaches the scene."""
    ys = []
    for f, sc, cam in animation.frames(2, 32, 18, 4, 5):
        assert cam is sc.camera
        big = sc.spheres[np.isclose(sc.spheres["radius"], 1.0) & (np.abs(sc.spheres["center"][:, 1]) < 100)]
        ys.append(sorted(np.round(big["center"][:, 1], 6).tolist()))
    assert len(ys) == animation.N_BOUNCE_FRAMES == 20
    assert ys[0] == [1.0, 1.0, 1.0] and ys[4] == [1.0, 1.0, 1.8] and ys[19] == [1.0, 1.0, 1.0]
    assert max(max(y) for y in ys) == pytest.approx(1.8)


def test_frame_shards_partition_the_sequence():
    for n, world in ((42, 8), (62, 8), (20, 3), (5, 8), (0, 2)):
        got = [frame_shards(n, world, r) for r in range(world)]
        assert sorted(f for g in got for f in g) == list(range(n))
        assert max(map(len, got)) - min(map(len, got)) <= 1
    with pytest.raises(ValueError):
        animation.n_frames(3)
    with pytest.raises(ValueError):
        scenes.build("first_scene", W, H, SPP, DEPTH, animation=2)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    for f, sc, cam in animation.frames(1, W, H, SPP, DEPTH, rank=rank, world=world):
        acc, _ = O.render(sc, cam, seed=5, rng_mode=1)
        np.save(os.path.join(out_dir, f"frame{f}.npy"), acc)
    dist.barrier()  # the only synchronisation: "sequence complete"
    dist.destroy_process_group()


def test_two_rank_frame_sharding_matches_single(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    n = 0
    for f, sc, cam in animation.frames(1, W, H, SPP, DEPTH):
        want, _ = O.render(sc, cam, seed=5, rng_mode=1)
        got = np.load(tmp_path / f"frame{f}.npy")
        assert got.tobytes() == want.tobytes(), f"frame {f}"
        n += 1
    assert n == 42


# ---- GPU ------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("method", [0, 1, 2])
def test_gpu_animation_shard_f64_matches_oracle(gpu_ctx, method):
    """rank 1 of 8's frames of each sequence through the C ABI (f64 parity path) against the oracle, path for path."""
    want = {}
    for f, sc, cam in animation.frames(method, 48, 32, 9, 8, rank=1, world=8):
        want[f] = O.render(sc, cam, seed=21, rng_mode=1, threads=8)
    assert sorted(want) == frame_shards(animation.n_frames(method), 8, 1)
    n = 0
    for f, acc, rgb, st in animation.render(gpu_ctx, method, 48, 32, 9, 8, rank=1, world=8, seed=21,
                                            precision=abi.RT_PREC_F64):
        ref, cnt = want[f]
        assert st["rays"] == int(cnt[4]) and st["samples"] == int(cnt[5])
        rel = np.abs(acc.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-3)
        assert np.nanmax(rel) <= 1e-6, f"frame {f}"
        assert (O.write_color(acc.astype(np.float64), 1.0 / 9) == rgb).all()
        n += 1
    assert n == len(want)


def _read_ppm(path):
    tok = open(path).read().split()
    assert tok[0] == "P3"
    w, h = int(tok[1]), int(tok[2])
    return np.array(tok[4:], dtype=np.int32).reshape(h, w, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("method", [0, 1, 2])
def test_cli_animation_writes_the_frames_of_the_ctypes_path(tmp_path, gpu_ctx, method):
    """`rtamd_render 1 <spp> <depth> true <method> ...` (the reference's argv contract, cpuRenderer.cpp:426-451) writes
    ./images/animation/image<f>.ppm for every frame; each equals the frame rendered through raytracing_amd.animation."""
    (tmp_path / "images" / "animation").mkdir(parents=True)
    exe = os.path.join(ROOT, "raytracing_amd", "rtamd_render")
    env = dict(os.environ, RTAMD_ASSETS=os.path.join(ROOT, "raytracing_amd", "assets"))
    args = f"1 4 6 true {method} false true 1 {W} {H}".split()
    r = subprocess.run([exe, *args], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    n = 0
    for f, acc, rgb, st in animation.render(gpu_ctx, method, W, H, 4, 6, seed=1984, precision=abi.RT_PREC_F32):
        img = _read_ppm(tmp_path / "images" / "animation" / f"image{f}.ppm")
        assert (img == rgb.astype(np.int32)).all(), f"frame {f}"
        n += 1
    assert n == animation.n_frames(method)
    assert not (tmp_path / "images" / "animation" / f"image{n}.ppm").exists()
