"""image_texture files are decoded by include/rtamd/jpeg_baseline.h instead of the vendored stb_image.

The expected texels (tests/golden/jpeg_expected.npz, raytracing_amd/assets/earthmap.jpg.rtwtex) were produced by the
compiled reference's own rtw_image::load (stb_image_utils.h:46-55) on the same files -- tests/golden/make_golden.py
--jpeg-only -- so equality here is equality with the reference's texture pipeline: Huffman decoding, the fixed-point
inverse DCT, progressive scans, chroma upsampling, YCbCr -> RGB, gamma-2.2 linearisation, float -> byte.  Bar: bit-exact.  CPU only."""
import ctypes
import os

import numpy as np
import pytest

from raytracing_amd import host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JPEG = os.path.join(ROOT, "tests", "golden", "jpeg")
EXPECTED = np.load(os.path.join(ROOT, "tests", "golden", "jpeg_expected.npz"))
CASES = sorted(EXPECTED.files)


@pytest.mark.parametrize("name", CASES)
def test_decode_matches_reference_texels(name):
    got = host.load_image(os.path.join(JPEG, name + ".jpg"))
    want = EXPECTED[name]
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"{int((got != want).sum())} of {want.size} bytes differ"


def test_fixture_set_covers_the_sampling_modes():
    names = set(EXPECTED.files)
    for need in ("c444_q90", "c422_q75", "c420_q75", "c411_q80", "grey_q80", "c420_rst", "c420_1x1", "c420_opt",
                 "c420_progressive", "c420_progressive_rst", "grey_progressive"):
        assert need in names


def test_earthmap_matches_reference_texels():
    """The reference's own texture (final_scene's globe, 1024x512 4:4:4 baseline)."""
    got = host.load_image(os.path.join(ROOT, "raytracing_amd", "assets", "earthmap.jpg"))
    raw = np.fromfile(os.path.join(ROOT, "raytracing_amd", "assets", "earthmap.jpg.rtwtex"), np.uint8)
    assert raw[:8].tobytes() == b"RTWTEX01"
    w, h = (int(v) for v in raw[8:16].view("<i4"))
    assert got.shape == (h, w, 3) == (512, 1024, 3)
    assert np.array_equal(got.reshape(-1), raw[16:])


def test_unsupported_and_broken_files_fail_loudly(tmp_path):
    data = bytearray(open(os.path.join(JPEG, "c420_q75.jpg"), "rb").read())
    # the same file declared arithmetic-coded (SOF9) / lossless (SOF3): neither decoder reads those
    sof = data.index(b"\xff\xc0")
    for marker in (0xC9, 0xC3):
        bad = tmp_path / f"sof_{marker:x}.jpg"
        data[sof + 1] = marker
        bad.write_bytes(bytes(data))
        with pytest.raises(RuntimeError):
            host.load_image(str(bad))
    data[sof + 1] = 0xC0
    data[sof + 4] = 12  # 12-bit samples
    bad = tmp_path / "twelve_bit.jpg"
    bad.write_bytes(bytes(data))
    with pytest.raises(RuntimeError):
        host.load_image(str(bad))
    data[sof + 4] = 8
    # non-integer chroma ratio: Y h=3 with Cb/Cr h=2 (the upsampler only knows whole ratios; ADVICE r1: was a heap over-read)
    ratio = bytearray(data)
    assert ratio[sof + 11] == 0x22 and ratio[sof + 14] == 0x11  # SOF: P, Y, X, Nf, then (id, hv, tq) per component
    ratio[sof + 11], ratio[sof + 14], ratio[sof + 17] = 0x32, 0x21, 0x21
    bad = tmp_path / "ratio_3_2.jpg"
    bad.write_bytes(bytes(ratio))
    with pytest.raises(RuntimeError):
        host.load_image(str(bad))
    bad = tmp_path / "truncated_header.jpg"
    bad.write_bytes(bytes(data[:40]))
    with pytest.raises(RuntimeError):
        host.load_image(str(bad))
    junk = tmp_path / "junk.jpg"
    junk.write_bytes(b"\xff\xd8\xff" + bytes(range(256)) * 4)
    with pytest.raises(RuntimeError):
        host.load_image(str(junk))
    with pytest.raises(RuntimeError):
        host.load_image(str(tmp_path / "missing.jpg"))


def test_truncated_scan_keeps_what_was_decoded(tmp_path):
    """Entropy-coded data that ends early reads as zero bits (as in the reference's decoder): same size, top rows intact."""
    data = open(os.path.join(JPEG, "c444_q90.jpg"), "rb").read()
    cut = tmp_path / "cut.jpg"
    cut.write_bytes(data[: len(data) * 2 // 3])
    got = host.load_image(str(cut))
    want = EXPECTED["c444_q90"]
    assert got.shape == want.shape
    assert np.array_equal(got[:8], want[:8])


def test_predecoded_fallback(tmp_path):
    """A file that is not a JPEG is taken from "<name>.rtwtex" (how PNG textures are supplied)."""
    tex = np.arange(2 * 3 * 3, dtype=np.uint8)
    with open(tmp_path / "t.png.rtwtex", "wb") as f:
        f.write(b"RTWTEX01" + np.array([3, 2], "<i4").tobytes() + tex.tobytes())
    got = host.load_image(str(tmp_path / "t.png"))
    assert got.shape == (2, 3, 3) and np.array_equal(got.reshape(-1), tex)


def test_final_scene_texture_comes_from_the_jpeg(tmp_path):
    """final_scene built from an asset directory that holds only the .jpg gives the same texels as the shipped fixture."""
    import shutil
    assets = os.path.join(ROOT, "raytracing_amd", "assets")
    shutil.copy(os.path.join(assets, "earthmap.jpg"), tmp_path / "earthmap.jpg")
    os.symlink(os.path.join(assets, "shapes"), tmp_path / "shapes")
    a = host.build_scene(3, 64, 36, 4, 5, asset_dir=str(tmp_path))
    b = host.build_scene(3, 64, 36, 4, 5)
    assert a.image_bytes.size == 1024 * 512 * 3
    assert np.array_equal(a.image_bytes, b.image_bytes)
