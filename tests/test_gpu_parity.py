"""GPU (-m gpu): the HIP path, called through the C ABI, against the reference fixtures and the CPU oracle.

Bars (north_star): on the deterministic ray batch the f64 path must report the reference's primitive ids bit-exactly
and t / normals within 1e-5 relative (measured: identical to the last bit except the log() in constant_medium);
whole frames traced with the same counter-based RNG must match the oracle path for path."""
import numpy as np
import pytest

from raytracing_amd import abi
from tests import fixtures as fx
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scenes(gpu_ctx):
    out = {n: gpu_ctx.upload(fx.scene(n)) for n in (1, 2, 3, 4)}
    yield out
    for ds in out.values():
        ds.close()


@pytest.mark.parametrize("n", [1, 2, 3, 4])
def test_trace_batch_f64_matches_reference(scenes, n):
    rays, xi, hits = fx.trace(n)
    got = scenes[n].trace_batch(rays, xi, abi.RT_PREC_F64)
    assert (got["prim"] == hits["prim"]).all(), "primitive ids must be bit-exact"
    hit = hits["prim"] >= 0
    assert hit.sum() > 1000
    assert (got["front_face"][hit] == hits["front_face"][hit]).all()
    rel_t = np.abs(got["t"][hit] - hits["t"][hit]) / np.abs(hits["t"][hit])
    assert rel_t.max() <= 1e-12
    assert np.abs(got["normal"][hit] - hits["normal"][hit]).max() <= 1e-12
    scale = np.maximum(1.0, np.abs(hits["p"][hit]).max())
    assert np.abs(got["p"][hit] - hits["p"][hit]).max() <= 1e-11 * scale
    assert np.abs(got["u"][hit] - hits["u"][hit]).max() <= 1e-12
    assert np.abs(got["v"][hit] - hits["v"][hit]).max() <= 1e-12
    assert np.isinf(got["t"][~hit]).all()


@pytest.mark.parametrize("n", [1, 2, 3, 4])
def test_trace_batch_f32_close_to_reference(scenes, n):
    """Production arithmetic: ids may differ only on a handful of grazing rays; primary-ray t within 1e-5 relative."""
    rays, xi, hits = fx.trace(n)
    got = scenes[n].trace_batch(rays, xi, abi.RT_PREC_F32)
    same = got["prim"] == hits["prim"]
    # ids differ only (a) where two primitives are hit at the same distance -- the shared edge of two faces of a box: the
    # same t to float precision, float rounding decides the winner (tools/trace_mismatch.py lists them) -- and (b) on a
    # handful of rays that graze an edge or start on a surface
    both = ~same & (got["prim"] >= 0) & (hits["prim"] >= 0)
    with np.errstate(invalid="ignore"):  # inf - inf on rays that miss in both
        tie = both & (np.abs(got["t"] - hits["t"]) <= 1e-5 * np.abs(hits["t"]))
    assert (~same).sum() <= 0.005 * len(rays)
    assert (~same & ~tie).sum() <= 0.002 * len(rays)
    primary = np.arange(len(rays)) < 64 * 36
    hit = same & (hits["prim"] >= 0) & primary
    kinds = fx.scene(n).nodes["kind"][hits["prim"][hit]]
    solid = kinds != abi.RT_NODE_MEDIUM  # a medium's t carries log(float(xi)): 1e-4 relative
    rel_t = np.abs(got["t"][hit] - hits["t"][hit]) / np.abs(hits["t"][hit])
    assert np.quantile(rel_t[solid], 0.99) <= 1e-5
    assert rel_t[solid].max() <= 2e-4
    if (~solid).any():
        assert rel_t[~solid].max() <= 1e-3
    dn = np.abs(got["normal"][hit] - hits["normal"][hit]).max(axis=1)
    assert np.quantile(dn, 0.99) <= 1e-4


def test_trace_batch_keyed_medium_rng_matches_oracle(scenes):
    rays, _, _ = fx.trace(3)
    got = scenes[3].trace_batch(rays, None, abi.RT_PREC_F64)
    want, _ = O.trace_batch(fx.scene(3), rays, None)
    assert (got["prim"] == want["prim"]).all()
    hit = want["prim"] >= 0
    assert np.abs(got["t"][hit] - want["t"][hit]).max() <= 1e-9


def test_trace_batch_empty_and_ragged(scenes):
    rays, xi, hits = fx.trace(2)
    assert len(scenes[2].trace_batch(rays[:0], None)) == 0
    got = scenes[2].trace_batch(rays[:257], xi[:257])  # not a multiple of the block size
    assert (got["prim"] == hits["prim"][:257]).all()


@pytest.mark.parametrize("n", [1, 2, 3, 4])
def test_frame_f64_matches_oracle_path_for_path(scenes, n):
    sc = fx.scene(n)
    cam, _, _, _ = fx.img(n)
    cam = fx.with_size(cam, sqrt_spp=4, max_depth=12)
    want, cnt = O.render(sc, cam, seed=3, rng_mode=1, threads=8)
    acc, rgb, st = scenes[n].render(cam, seed=3, precision=abi.RT_PREC_F64)
    assert st["rays"] == int(cnt[4]), "same number of ray_color invocations as the oracle"
    assert st["samples"] == int(cnt[5])
    rel = np.abs(acc.astype(np.float64) - want) / np.maximum(np.abs(want), 1e-3)
    assert np.nanmax(rel) <= 1e-6  # float32 output rounding is 6e-8
    assert (np.isnan(acc) == np.isnan(want)).all()
    # tonemap == write_color on the same sums
    assert (O.write_color(acc.astype(np.float64), 1.0 / 16) == rgb).all()


@pytest.mark.parametrize("n", [1, 2, 3, 4])
def test_frame_f32_tracks_oracle(scenes, n):
    """float paths follow the double oracle's paths except where rounding flips a branch: most pixels agree to 1e-4."""
    sc = fx.scene(n)
    cam, _, _, _ = fx.img(n)
    cam = fx.with_size(cam, sqrt_spp=4, max_depth=12)
    want, cnt = O.render(sc, cam, seed=3, rng_mode=1, threads=8)
    acc, rgb, st = scenes[n].render(cam, seed=3, precision=abi.RT_PREC_F32)
    assert abs(st["rays"] - int(cnt[4])) <= 0.02 * int(cnt[4])
    rel = np.abs(acc.astype(np.float64) - want) / np.maximum(np.abs(want), 1e-3)
    ok = np.isfinite(rel)
    assert np.mean(rel[ok] < 1e-4) >= 0.90
    assert np.isnan(acc).sum() <= np.isnan(want).sum() + 3


def test_sample_range_split_sums_to_whole_and_is_chunk_invariant(scenes):
    """The multi-GPU spp split: disjoint stratum ranges sum to the full frame; chunking does not change results."""
    cam, _, _, _ = fx.img(2)
    cam = fx.with_size(cam, sqrt_spp=6, max_depth=8)
    whole, _, _ = scenes[2].render(cam, seed=9, precision=abi.RT_PREC_F32, sample_chunks=1)
    parts = [scenes[2].render(cam, seed=9, s_begin=a, s_end=b, precision=abi.RT_PREC_F32)[0]
             for a, b in ((0, 5), (5, 17), (17, 36))]
    np.testing.assert_allclose(sum(p.astype(np.float64) for p in parts), whole, rtol=2e-6, atol=1e-6)
    chunked, _, _ = scenes[2].render(cam, seed=9, precision=abi.RT_PREC_F32, sample_chunks=7)
    np.testing.assert_allclose(chunked, whole, rtol=2e-6, atol=1e-6)
    again, _, _ = scenes[2].render(cam, seed=9, precision=abi.RT_PREC_F32, sample_chunks=7)
    assert again.tobytes() == chunked.tobytes(), "repeats exactly while every wave holds one work item (small frames)"
    empty, _, st = scenes[2].render(cam, seed=9, s_begin=4, s_end=4)
    assert st["samples"] == 0 and not empty.any()


def test_ragged_image_sizes(scenes):
    """Widths/heights that are not multiples of the 8x8 tile, down to 1x1."""
    from raytracing_amd import camera
    for W, H in ((1, 1), (13, 7), (33, 17)):
        cam = camera.scene_camera(2, W, H, 4, 5)
        want, _ = O.render(fx.scene(2), cam, seed=1, rng_mode=1)
        acc, _, st = scenes[2].render(cam, seed=1, precision=abi.RT_PREC_F64)
        assert acc.shape == (H, W, 3) and st["samples"] == W * H * 4
        np.testing.assert_allclose(acc, want, rtol=1e-6, atol=1e-7)


def test_bad_arguments_return_status(gpu_ctx, scenes):
    from raytracing_amd import lib
    cam, _, _, _ = fx.img(2)
    with pytest.raises(lib.RtError):
        scenes[2].render(cam, s_begin=5, s_end=1000)
    bad = fx.scene(2)
    bad.world_root = 10 ** 6
    with pytest.raises(lib.RtError):
        gpu_ctx.upload(bad)


@pytest.mark.parametrize("n", [2, 3, 4, 1])
def test_converged_frame_psnr_vs_reference(scenes, n):
    """PSNR >= 40 dB (8-bit display space) of a converged f32 GPU frame against the converged frame of the compiled
    reference (mean of 5 independent runs = 84.5k spp; final_scene 25 runs = 422.5k spp).  The reference's own run-to-run noise floor at that sample count is
    measured from the fixture and printed next to it."""
    c = fx.conv(n)
    if c is None:
        pytest.skip("converged fixture not generated")
    cam, mean, spp, runs = c
    k = 1000 if n == 3 else 300  # final_scene is ~16 dB noisier than the others: 10^6 spp against the 422.5k-spp reference
    cam = fx.with_size(cam, sqrt_spp=k)
    acc, _, st = scenes[n].render(cam, seed=2024, precision=abi.RT_PREC_F32)
    img_gpu = O.write_color(acc.astype(np.float64), 1.0 / (k * k))
    img_ref = O.write_color(mean, 1.0)
    psnr = fx.psnr8(img_gpu, img_ref)
    half_a = O.write_color(runs[:2].mean(axis=0).astype(np.float64), 1.0)
    half_b = O.write_color(runs[2:4].mean(axis=0).astype(np.float64), 1.0)
    floor = fx.psnr8(half_a, half_b)
    print(f"scene {n}: PSNR gpu-vs-reference {psnr:.1f} dB; reference self-PSNR (2 x 33.8k spp halves) {floor:.1f} dB")
    assert psnr >= 40.0
