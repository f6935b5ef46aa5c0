"""PNG textures are decoded by include/rtamd/png_decode.h (inflate + unfilter + Adam7) instead of the vendored stb_image.

Expected texels (tests/golden/png_expected.npz, made by tests/golden/make_golden.py --png-only): for 8-bit files the
output of the compiled reference's rtw_image::load on the same file (which Pillow's decode confirms); for 1/2/4-bit files
("spec_" cases) Pillow's decode put through the reference's linearisation, because the reference's stb_image v2.06
unfilters such rows against unwritten memory.  Bar: bit-exact.  CPU only."""
import os

import numpy as np
import pytest

from raytracing_amd import host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PNG = os.path.join(ROOT, "tests", "golden", "png")
EXPECTED = np.load(os.path.join(ROOT, "tests", "golden", "png_expected.npz"))


@pytest.mark.parametrize("name", sorted(EXPECTED.files))
def test_decode_matches_expected_texels(name):
    got = host.load_image(os.path.join(PNG, name + ".png"))
    want = EXPECTED[name]
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"{int((got != want).sum())} of {want.size} bytes differ"


def test_fixture_set_covers_the_format():
    names = set(EXPECTED.files)
    for need in ("rgb8", "rgba8", "grey8", "greyalpha8", "pal8", "pal8_trns", "rgb8_filters", "rgb8_adam7", "rgb8_stored",
                 "grey8_fixed_huffman", "spec_grey1", "spec_grey2_adam7", "spec_pal4"):
        assert need in names


def test_unsupported_and_broken_files_fail_loudly(tmp_path):
    data = bytearray(open(os.path.join(PNG, "rgb8.png"), "rb").read())
    sixteen = bytearray(data)
    sixteen[24] = 16  # IHDR bit depth: the reference's decoder refuses 16-bit files too
    (tmp_path / "sixteen.png").write_bytes(bytes(sixteen))
    with pytest.raises(RuntimeError):
        host.load_image(str(tmp_path / "sixteen.png"))
    (tmp_path / "cut.png").write_bytes(bytes(data[: len(data) // 2]))
    with pytest.raises(RuntimeError):
        host.load_image(str(tmp_path / "cut.png"))
    broken = bytearray(data)
    idat = broken.index(b"IDAT")
    for k in range(idat + 10, idat + 40):
        broken[k] ^= 0x5A  # damage the compressed stream
    (tmp_path / "broken.png").write_bytes(bytes(broken))
    with pytest.raises(RuntimeError):
        host.load_image(str(tmp_path / "broken.png"))
