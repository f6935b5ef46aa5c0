"""The headline configuration at full size (BASELINE.json: final_scene 1920x1080 @ 1000 spp -> 961 strata, depth 50),
checked through properties that do not need a CPU run of 2 G samples:

* the multi-GPU decomposition: the 8 per-rank stratum ranges of `split_range(961, 8, r)` sum to the whole frame, with
  exactly the same number of rays (the RNG is keyed by (pixel, stratum), so every path is the same path);
* a rank's partial frame repeats run to run BIT FOR BIT (the float kernels accumulate in 64-bit fixed point: integer adds
  commute, so neither the atomic work-item cursor nor the order in which a wave's paths finish can show in a sum);
* agreement with the compiled reference: the 1080p frame box-averaged 40x40 -> 48x27 is the same integral as the
  reference's converged 48x27 frame of the same camera (tests/golden/conv3.npz, 422.5k spp): PSNR >= 40 dB in 8-bit
  display space.

The whole frame and the sum of its 8 per-rank frames are the same integer total; each is rounded to float once per
frame it passes through, so they agree to 8 float roundings of the partial sums: 1 ulp of the total (1.2e-7 relative)."""
import numpy as np
import pytest

from tests import fixtures as fx
from tests import oracle_lib as O
from raytracing_amd import abi, camera, scenes as scn
from raytracing_amd.distributed import split_range

pytestmark = pytest.mark.gpu

W, H, SPP, DEPTH = 1920, 1080, 1000, 50


@pytest.fixture(scope="module")
def headline(gpu_ctx):
    scene = scn.build("final_scene", W, H, SPP, DEPTH)
    return scene, gpu_ctx.upload(scene)


@pytest.fixture(scope="module")
def whole(headline):
    scene, ds = headline
    acc, _, st = ds.render(scene.camera, seed=1984, precision=abi.RT_PREC_F32, want_rgb8=False)
    return acc, st


def test_rank_ranges_sum_to_the_whole_frame(headline, whole):
    scene, ds = headline
    acc, st = whole
    n = int(scene.camera["sqrt_spp"][0])
    assert n * n == 961 and st["samples"] == W * H * 961
    total = np.zeros(acc.shape, dtype=np.float64)
    rays = 0
    covered = 0
    for r in range(8):
        s0, s1 = split_range(n * n, 8, r)
        part, _, pst = ds.render(scene.camera, seed=1984, s_begin=s0, s_end=s1, precision=abi.RT_PREC_F32, want_rgb8=False)
        assert pst["samples"] == W * H * (s1 - s0)
        total += part
        rays += pst["rays"]
        covered += s1 - s0
    assert covered == 961
    assert rays == st["rays"], "same paths whatever the decomposition"
    ok = np.isfinite(acc)
    assert (~ok).sum() <= 64 and np.array_equal(ok, np.isfinite(total))
    # every rank's frame is its exact fixed-point sum rounded to float (<= 2^-25 relative each): the double sum of the 8
    # is within 2^-25 of the exact total, whose own float rounding is another 2^-25
    np.testing.assert_allclose(total[ok], acc[ok].astype(np.float64), rtol=2.0 ** -23, atol=0)


def test_partial_frame_repeats_bit_for_bit(headline):
    """Two runs trace the same paths (same ray count); which wave picks up which work item is decided by an atomic cursor
    and the order of a wave's adds follows its schedule -- and none of that shows: the sums are integers until the last
    conversion.  The same holds across different cuts of the sample range into chunks."""
    scene, ds = headline
    s0, s1 = split_range(961, 8, 3)
    a, _, sa = ds.render(scene.camera, seed=7, s_begin=s0, s_end=s1, precision=abi.RT_PREC_F32, want_rgb8=False)
    b, _, sb = ds.render(scene.camera, seed=7, s_begin=s0, s_end=s1, precision=abi.RT_PREC_F32, want_rgb8=False)
    assert sa["rays"] == sb["rays"]
    assert a.tobytes() == b.tobytes(), f"{int((a != b).sum())} of {a.size} values differ between two runs"
    c, _, sc_ = ds.render(scene.camera, seed=7, s_begin=s0, s_end=s1, precision=abi.RT_PREC_F32, sample_chunks=5, want_rgb8=False)
    assert sc_["rays"] == sa["rays"] and c.tobytes() == a.tobytes(), "the cut into sample chunks must not show either"


def test_full_frame_downsampled_matches_converged_reference(headline, whole):
    c = fx.conv(3)
    if c is None:
        pytest.skip("converged fixture not generated")
    cam_small, mean, _, runs = c
    scene, _ = headline
    # same view: the fixture's camera is the 48x27 instance of the scene function's camera
    assert int(cam_small["image_width"][0]) * 40 == W and int(cam_small["image_height"][0]) * 40 == H
    big = scene.camera
    np.testing.assert_allclose(cam_small["center"][0], big["center"][0], rtol=0, atol=0)
    np.testing.assert_allclose(cam_small["pixel_delta_u"][0], 40 * big["pixel_delta_u"][0], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(cam_small["pixel_delta_v"][0], 40 * big["pixel_delta_v"][0], rtol=1e-12, atol=1e-15)
    acc, _ = whole
    lin = np.nan_to_num(acc.astype(np.float64), nan=0.0, posinf=0.0, neginf=0.0) / 961.0
    small = lin.reshape(27, 40, 48, 40, 3).mean(axis=(1, 3))
    img_gpu = O.write_color(small, 1.0)
    img_ref = O.write_color(mean, 1.0)
    psnr = fx.psnr8(img_gpu, img_ref)
    floor = fx.psnr8(O.write_color(runs[:2].mean(axis=0).astype(np.float64), 1.0), O.write_color(runs[2:4].mean(axis=0).astype(np.float64), 1.0))
    print(f"1080p frame averaged to 48x27 vs converged reference: PSNR {psnr:.1f} dB (reference self-PSNR {floor:.1f} dB)")
    assert psnr >= 40.0
