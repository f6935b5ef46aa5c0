#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz from the compiled reference (oracle/_ref/refgen).

Runs only where /root/reference exists (this container).  The fixtures are DATA produced by running the
reference's own code (scene graphs, hit records, rendered pixel sums); no reference source is stored.

  scene{N}.npz   reference scene N (1 first_scene, 2 cornell_box, 3 final_scene, 4 mesh_scene), use_bvh=true, flat
                 format of include/rt_amd.h, camera frame for 64x36 / spp 9 / depth 10
  trace{N}.npz   64x36 pixel-centre rays + 2000 secondary-like rays: rays, injected medium uniform, reference hit records
                 (prim = node id, t, p, normal, u, v, front_face), all from world.hit() of the reference
  img{N}.npz     32x18, spp 9, depth 12 frame rendered by the reference single-threaded from srand(7): per-pixel double
                 sums and write_color ints (cross-checked in refgen against camera::render()'s PPM), + camera frame
  conv{N}.npz    48x27, depth 50 converged frame: mean of 5 (final_scene: 25) independent reference runs x 16900 spp
                 (srand 11..15 / 11..35); `runs` keeps the first five for the noise-floor estimate
                 -- optional, produced with --converged (about 25 core-minutes)
  anim_translate.npz  what `cpuRayTracer 1 4 6 true <any> false true 1 24 16` produces: the reference's own
                 first_scene(animation=true) -> camera::render_animation() run, single-threaded from the process's
                 default rand() seed: camera frame and write_color ints of each of its 42 frames
  jpeg/*.jpg + jpeg_expected.npz  small JPEGs written with Pillow (4:4:4 / 4:2:2 / 4:2:0 / 4:1:1, grey, odd and
                 one-pixel sizes, restart intervals, optimised tables, progressive scans) and the linearised RGB8 texels the
                 reference's rtw_image::load gives for each (--jpeg-only)
  png/*.png + png_expected.npz  PNG files (Pillow-written and hand-assembled: every row filter, Adam7, 1/2/4/8-bit grey
                 and palette, alpha, stored / fixed / dynamic deflate blocks) and their linearised RGB8 texels (--png-only)
  win{N}.npz     a 64 x 64 PIXEL WINDOW of the 1920 x 1080 frame of scene N (3 final_scene, 4 mesh_scene) at full resolution:
                 8 independent reference runs x 50^2 = 2 500 spp (20 000 spp), depth 50, single-threaded each (REFGEN_WINDOW);
                 the full-size camera, the window origin, the mean and the eight runs (noise floor) -- --window (about 25 core-minutes)
  anim_rotate.npz     camera frames of the 62-frame rotate + zoom sequence (camera.h:49-59,93-96; the reference's
                 rotate()/zoom()/initialize() driven in that order by refgen), 24x16
"""
import argparse
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from raytracing_amd import abi  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
WORK = os.path.join(REF, "work")
OUT = os.path.join(REF, "out")
GOLD = os.path.dirname(os.path.abspath(__file__))
ENV = dict(os.environ, RTW_IMAGES="/root/reference/images/resource")


def refgen(*args):
    subprocess.run([os.path.join(REF, "refgen"), *map(str, args)], cwd=WORK, env=ENV, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def read_trace(path):
    b = open(path, "rb").read()
    assert b[:8] == b"RTTRACE1"
    n = int(np.frombuffer(b, "<i8", 1, 8)[0])
    off = 16
    rays = np.frombuffer(b, abi.ray_dtype, n, off).copy()
    off += n * abi.ray_dtype.itemsize
    xi = np.frombuffer(b, "<f8", n, off).copy()
    off += n * 8
    hits = np.frombuffer(b, abi.hit_dtype, n, off).copy()
    return rays, xi, hits


def read_img(path):
    b = open(path, "rb").read()
    assert b[:8] == b"RTIMG001"
    W, H, n, d = (int(x) for x in np.frombuffer(b, "<i4", 4, 8))
    seed = int(np.frombuffer(b, "<u4", 1, 24)[0])
    off = 28
    sums = np.frombuffer(b, "<f8", W * H * 3, off).reshape(H, W, 3).copy()
    off += W * H * 3 * 8
    by = np.frombuffer(b, "<i4", W * H * 3, off).reshape(H, W, 3).copy()
    return W, H, n, d, seed, sums, by


def read_anim(path):
    b = open(path, "rb").read()
    assert b[:8] == b"RTANIM01"
    n, W, H, with_images = (int(x) for x in np.frombuffer(b, "<i4", 4, 8))
    off = 24
    cams = np.frombuffer(b, abi.camera_dtype, n, off).copy()
    off += n * abi.camera_dtype.itemsize
    imgs = np.frombuffer(b, "<i4", n * W * H * 3, off).reshape(n, H, W, 3).copy() if with_images else None
    return cams, imgs


def make_anim():
    refgen("anim", 1, 24, 16, 4, 6, 1, f"{OUT}/anim_translate.bin")
    cams, imgs = read_anim(f"{OUT}/anim_translate.bin")
    assert len(cams) == 42 and imgs.max() < 256
    np.savez_compressed(f"{GOLD}/anim_translate.npz", cameras=cams.view("u1"), images=imgs.astype(np.uint8))
    refgen("anim_rotate", 1, 24, 16, 4, 6, 1, f"{OUT}/anim_rotate.bin")
    cams, _ = read_anim(f"{OUT}/anim_rotate.bin")
    assert len(cams) == 62
    np.savez_compressed(f"{GOLD}/anim_rotate.npz", cameras=cams.view("u1"))
    print("animation fixtures: 42 translate frames (cameras + images), 62 rotate frames (cameras)")


def test_picture(w, h, seed, grey=False):
    """Synthetic content with smooth gradients, sharp edges, saturated colours and noise (all four stress a decoder)."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([255 * x / max(w - 1, 1), 255 * y / max(h - 1, 1), 127 + 127 * np.sin(0.31 * x + 0.17 * y)], -1)
    img[(x // 5 + y // 3) % 4 == 0] = (255, 0, 0)
    img[(x // 7 + y // 4) % 5 == 1] = (0, 0, 255)
    img[(x + 2 * y) % 11 == 0] = (255, 255, 255)
    img += rng.normal(0, 12, img.shape)
    img = np.clip(img, 0, 255).astype(np.uint8)
    return img[..., 1] if grey else img


JPEG_CASES = [  # name, width, height, PIL save options  (mode L when "grey")
    ("c444_q90", 48, 32, dict(quality=90, subsampling=0)),
    ("c422_q75", 48, 32, dict(quality=75, subsampling=1)),
    ("c420_q75", 48, 32, dict(quality=75, subsampling=2)),
    ("c420_odd", 37, 21, dict(quality=80, subsampling=2)),
    ("c422_odd", 35, 9, dict(quality=80, subsampling=1)),
    ("c420_wide", 301, 40, dict(quality=85, subsampling=2)),
    ("c420_1x1", 1, 1, dict(quality=80, subsampling=2)),
    ("c420_2x3", 2, 3, dict(quality=80, subsampling=2)),
    ("c420_17x1", 17, 1, dict(quality=80, subsampling=2)),
    ("c422_1x17", 1, 17, dict(quality=80, subsampling=1)),
    ("c411_q80", 45, 19, dict(quality=80, subsampling="4:1:1")),
    ("c420_q5", 40, 40, dict(quality=5, subsampling=2)),
    ("c444_q100", 40, 24, dict(quality=100, subsampling=0)),
    ("c420_opt", 64, 48, dict(quality=70, subsampling=2, optimize=True)),
    ("c420_rst", 64, 48, dict(quality=70, subsampling=2, restart_marker_blocks=3)),
    ("c444_rst_rows", 50, 40, dict(quality=70, subsampling=0, restart_marker_rows=1)),
    ("grey_q80", 33, 19, dict(quality=80)),
    ("grey_rst", 40, 40, dict(quality=60, restart_marker_blocks=2)),
    ("c420_progressive", 48, 32, dict(quality=75, subsampling=2, progressive=True)),
    ("c444_progressive", 37, 29, dict(quality=90, subsampling=0, progressive=True)),
    ("c422_progressive_q20", 61, 18, dict(quality=20, subsampling=1, progressive=True)),
    ("c420_progressive_rst", 72, 40, dict(quality=80, subsampling=2, progressive=True, restart_marker_blocks=2)),
    ("c420_progressive_wide", 301, 24, dict(quality=95, subsampling=2, progressive=True, optimize=True)),
    ("grey_progressive", 33, 19, dict(quality=80, progressive=True)),
    ("c420_progressive_1x1", 1, 1, dict(quality=80, subsampling=2, progressive=True)),
]


def make_jpeg():
    """tests/golden/jpeg/<case>.jpg (written with Pillow) and jpeg_expected.npz: the texels the reference's
    rtw_image::load (vendored stb_image -> stbi_loadf -> bytes) gives for each of them."""
    from PIL import Image
    d = os.path.join(GOLD, "jpeg")
    os.makedirs(d, exist_ok=True)
    expected = {}
    for k, (name, w, h, opts) in enumerate(JPEG_CASES):
        grey = name.startswith("grey")
        path = os.path.join(d, name + ".jpg")
        try:
            Image.fromarray(test_picture(w, h, 100 + k, grey), "L" if grey else "RGB").save(path, "JPEG", **opts)
        except Exception as e:  # an option this Pillow does not know
            print("skipped", name, e)
            continue
        out = f"{OUT}/{name}.rtwtex"
        refgen("teximg", path, out)
        b = open(out, "rb").read()
        assert b[:8] == b"RTWTEX01"
        tw, th = (int(v) for v in np.frombuffer(b, "<i4", 2, 8))
        assert (tw, th) == (w, h)
        expected[name] = np.frombuffer(b, np.uint8, w * h * 3, 16).reshape(h, w, 3).copy()
    # the reference's own texture: its reference texels are the shipped raytracing_amd/assets/earthmap.jpg.rtwtex
    refgen("teximg", "/root/reference/images/resource/earthmap.jpg", f"{OUT}/earthmap.rtwtex")
    assert open(f"{OUT}/earthmap.rtwtex", "rb").read() == open(f"{ROOT}/raytracing_amd/assets/earthmap.jpg.rtwtex", "rb").read()
    np.savez_compressed(f"{GOLD}/jpeg_expected.npz", **expected)
    print("jpeg fixtures:", ", ".join(expected))


def write_png(path, arr, color, depth=8, interlace=False, filters=(0, 1, 2, 3, 4), palette=None, level=6):
    """Minimal PNG writer for the cases Pillow does not produce: chosen row filters, Adam7 interlace.  arr: (h, w) or
    (h, w, c) uint8 samples (for depth < 8: one sample per pixel, values < 2**depth)."""
    import struct
    import zlib
    arr = np.asarray(arr, np.uint8)
    h, w = arr.shape[:2]
    comps = 1 if arr.ndim == 2 else arr.shape[2]

    def pack_rows(a):
        rows = a.reshape(a.shape[0], -1)
        if depth == 8:
            return rows
        per = 8 // depth
        pad = (-rows.shape[1]) % per
        rows = np.pad(rows, ((0, 0), (0, pad)))
        out = np.zeros((rows.shape[0], rows.shape[1] // per), np.uint8)
        for k in range(per):
            out |= rows[:, k::per] << (8 - depth * (k + 1))
        return out

    def filt(rows):
        bpp = max(1, comps * depth // 8)
        out = bytearray()
        prev = np.zeros(rows.shape[1], np.int32)
        for y, r in enumerate(rows.astype(np.int32)):
            f = filters[y % len(filters)]
            a_ = np.concatenate([np.zeros(bpp, np.int32), r[:-bpp]]) if r.size > bpp else np.zeros_like(r)
            c_ = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]]) if r.size > bpp else np.zeros_like(r)
            if f == 0:
                e = r
            elif f == 1:
                e = r - a_
            elif f == 2:
                e = r - prev
            elif f == 3:
                e = r - ((a_ + prev) >> 1)
            else:
                pq = a_ + prev - c_
                pa, pb, pc = abs(pq - a_), abs(pq - prev), abs(pq - c_)
                e = r - np.where((pa <= pb) & (pa <= pc), a_, np.where(pb <= pc, prev, c_))
            out += bytes([f]) + (e & 255).astype(np.uint8).tobytes()
            prev = r
        return bytes(out)

    if interlace:
        xo, yo, xs, ys = (0, 4, 0, 2, 0, 1, 0), (0, 0, 4, 0, 2, 0, 1), (8, 8, 4, 4, 2, 2, 1), (8, 8, 8, 4, 4, 2, 2)
        data = b"".join(filt(pack_rows(arr[yo[k]::ys[k], xo[k]::xs[k]])) for k in range(7)
                        if arr[yo[k]::ys[k], xo[k]::xs[k]].size)
    else:
        data = filt(pack_rows(arr))

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body))
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color, 0, 0, int(interlace)))
    if palette is not None:
        out += chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    co = zlib.compressobj(level)
    z = co.compress(data) + co.flush()
    out += chunk(b"IDAT", z[: len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"tEXt", b"k\0v") + chunk(b"IEND", b"")
    open(path, "wb").write(out)


def make_png():
    """tests/golden/png/*.png and png_expected.npz.  8-bit files: expected texels from the reference's rtw_image::load
    (stb_image v2.06), cross-checked against Pillow's decode.  1/2/4-bit files: that stb version unfilters them against a
    wrong prior row (it reads unwritten memory), so the expected texels are Pillow's decode put through the reference's
    linearisation -- recorded as such in the archive ("spec_" prefix)."""
    from PIL import Image
    d = os.path.join(GOLD, "png")
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(77)
    lin = np.array([0 if v == 0 else min(255, int(256.0 * float(np.float32(np.float32(v) / np.float32(255.0)) ** np.float32(2.2)))) for v in range(256)], np.uint8)
    rgb = test_picture(37, 21, 5)
    big = test_picture(300, 64, 6)
    pal = rng.integers(0, 256, (16, 3), dtype=np.uint8)
    cases = {
        "rgb8": lambda p: Image.fromarray(rgb).save(p),
        "rgba8": lambda p: Image.fromarray(np.dstack([rgb, rng.integers(0, 256, rgb.shape[:2], dtype=np.uint8)]), "RGBA").save(p),
        "grey8": lambda p: Image.fromarray(rgb[..., 1]).save(p),
        "greyalpha8": lambda p: Image.fromarray(np.dstack([rgb[..., 1], rgb[..., 2]]), "LA").save(p),
        "pal8": lambda p: Image.fromarray(rgb).quantize(200).save(p),
        "pal8_trns": lambda p: Image.fromarray(rgb).quantize(40).save(p, transparency=3),
        "rgb8_big_l9": lambda p: Image.fromarray(big).save(p, compress_level=9),
        "rgb8_stored": lambda p: Image.fromarray(rgb).save(p, compress_level=0),
        "rgb8_filters": lambda p: write_png(p, big, 2),
        "rgb8_adam7": lambda p: write_png(p, rgb, 2, interlace=True),
        "rgba8_adam7_small": lambda p: write_png(p, np.dstack([rgb, rgb[..., 0]])[:5, :3], 6, interlace=True),
        "grey8_fixed_huffman": lambda p: write_png(p, rgb[:4, :9, 0], 0, level=1),
        "rgb8_1x1": lambda p: write_png(p, rgb[:1, :1], 2),
        "spec_grey1": lambda p: write_png(p, (rgb[..., 0] > 128).astype(np.uint8), 0, depth=1),
        "spec_grey2_adam7": lambda p: write_png(p, rgb[..., 0] >> 6, 0, depth=2, interlace=True),
        "spec_grey4": lambda p: write_png(p, rgb[..., 0] >> 4, 0, depth=4),
        "spec_pal4": lambda p: write_png(p, rgb[..., 1] >> 4, 3, depth=4, palette=pal),
        "spec_pal1": lambda p: write_png(p, rgb[..., 1] >> 7, 3, depth=1, palette=pal[:2], filters=(0,)),
    }
    expected = {}
    for name, make in cases.items():
        path = os.path.join(d, name + ".png")
        make(path)
        spec = lin[np.asarray(Image.open(path).convert("RGB"))]
        if name.startswith("spec_"):
            expected[name] = spec
            continue
        out = f"{OUT}/{name}.rtwtex"
        refgen("teximg", path, out)
        b = open(out, "rb").read()
        tw, th = (int(v) for v in np.frombuffer(b, "<i4", 2, 8))
        ref = np.frombuffer(b, np.uint8, tw * th * 3, 16).reshape(th, tw, 3).copy()
        assert np.array_equal(ref, spec), name  # the reference's decoder and Pillow agree on 8-bit files
        expected[name] = ref
    np.savez_compressed(f"{GOLD}/png_expected.npz", **expected)
    print("png fixtures:", ", ".join(expected))


def make_boxtrace():
    """final_scene's ground boxes (box() lists of six quads, quad.h:103) hit through their corners, edge midpoints, face
    centres and points just inside a face next to an edge, from the camera and from above: the reference's quad ids /
    normals / front_face for the rays the f32 stream answers from ONE fused slab test per box (refgen mode boxtrace)."""
    refgen("boxtrace", 3, 64, 36, 9, 10, 1, f"{OUT}/boxtrace3.bin", 0)
    rays, xi, hits = read_trace(f"{OUT}/boxtrace3.bin")
    np.savez_compressed(f"{GOLD}/boxtrace3.npz", rays=rays.view("u1"), xi=xi, hits=hits.view("u1"))
    print(f"boxtrace3: {len(rays)} rays, {(hits['prim'] >= 0).sum()} hits")


def make_converged_square(s=2, side=50, seeds=(11, 12, 13, 14, 15)):
    """cornell_box at 600x600 (BASELINE config C2) box-averaged 12x12 is the same integral as this 50x50 frame of the same
    camera: 5 independent reference runs of 130^2 = 16 900 spp each (84 500 spp), depth 50."""
    procs = [subprocess.Popen([os.path.join(REF, "refgen"), "render_nocheck", str(s), str(side), str(side), "16900", "50", "1",
                               f"{OUT}/convsq_{s}_{k}.bin", str(k)], cwd=WORK, env=ENV, stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL) for k in seeds]
    for p in procs:
        assert p.wait() == 0
    refgen("dump", s, side, side, 16900, 50, 1, f"{OUT}/scene{s}_sq.rtscene")
    cam = abi.Scene.load(f"{OUT}/scene{s}_sq.rtscene").camera
    tot, spp, runs = None, 0, []
    for k in seeds:
        W, H, n, d, seed, sums, by = read_img(f"{OUT}/convsq_{s}_{k}.bin")
        runs.append(sums / (n * n))
        tot = sums if tot is None else tot + sums
        spp += n * n
    np.savez_compressed(f"{GOLD}/conv{s}_sq.npz", camera=cam.view("u1"), mean=(tot / spp).astype(np.float64),
                        spp=np.array([spp]), runs=np.stack(runs).astype(np.float32))
    print(f"scene {s}: converged {side}x{side} fixture from {spp} spp")


def make_window(s, x0, y0, side=64, seeds=(21, 22, 23, 24, 25, 26, 27, 28), sqrt_spp=50):
    """Pixel-level fixture at the headline resolution: the reference's own get_ray / ray_color on pixels (x0 + i, y0 + j) of
    the 1920 x 1080 camera (the full-size checks saw only 40 x 40 box averages)."""
    env = dict(ENV, REFGEN_WINDOW=f"{x0},{y0},{side},{side}")
    procs = [subprocess.Popen([os.path.join(REF, "refgen"), "render_nocheck", str(s), "1920", "1080", str(sqrt_spp * sqrt_spp), "50", "1",
                               f"{OUT}/win_{s}_{k}.bin", str(k)], cwd=WORK, env=env, stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL) for k in seeds]
    for p in procs:
        assert p.wait() == 0
    refgen("dump", s, 1920, 1080, sqrt_spp * sqrt_spp, 50, 1, f"{OUT}/scene{s}_1080.rtscene")
    cam = abi.Scene.load(f"{OUT}/scene{s}_1080.rtscene").camera
    tot, spp, runs = None, 0, []
    for k in seeds:
        W, H, n, d, seed, sums, by = read_img(f"{OUT}/win_{s}_{k}.bin")
        assert (W, H) == (side, side)
        runs.append(sums / (n * n))
        tot = sums if tot is None else tot + sums
        spp += n * n
    np.savez_compressed(f"{GOLD}/win{s}.npz", camera=cam.view("u1"), origin=np.array([x0, y0]), mean=(tot / spp).astype(np.float64),
                        spp=np.array([spp]), runs=np.stack(runs).astype(np.float32))
    print(f"scene {s}: {side}x{side} window at ({x0}, {y0}) of the 1080p frame from {spp} spp")


def make_extras():
    refgen("dump", 5, 64, 36, 9, 10, 1, f"{OUT}/scene5_f7.rtscene", 0, 7)
    abi.Scene.load(f"{OUT}/scene5_f7.rtscene").save_npz(f"{GOLD}/scene5_f7.npz")
    refgen("dump", 2, 64, 36, 9, 10, 0, f"{OUT}/scene2_nobvh.rtscene")
    abi.Scene.load(f"{OUT}/scene2_nobvh.rtscene").save_npz(f"{GOLD}/scene2_nobvh.npz")
    refgen("trace", 2, 64, 36, 9, 10, 0, f"{OUT}/trace2_nobvh.bin", 500)
    rays, xi, hits = read_trace(f"{OUT}/trace2_nobvh.bin")
    np.savez_compressed(f"{GOLD}/trace2_nobvh.npz", rays=rays.view("u1"), xi=xi, hits=hits.view("u1"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--converged", action="store_true", help="also (re)run the long converged renders")
    ap.add_argument("--scenes", default="1,2,3,4,6")
    ap.add_argument("--anim-only", action="store_true", help="only (re)generate the animation fixtures")
    ap.add_argument("--jpeg-only", action="store_true", help="only (re)generate the JPEG decoding fixtures")
    ap.add_argument("--png-only", action="store_true", help="only (re)generate the PNG decoding fixtures")
    ap.add_argument("--boxtrace-only", action="store_true", help="only (re)generate the box corner / edge ray batch")
    ap.add_argument("--smoke-only", action="store_true", help="only (re)generate scene 6 (cornell smoke: general medium boundaries)")
    ap.add_argument("--window", action="store_true", help="only: the 64x64 full-resolution pixel windows of final_scene and mesh_scene")
    ap.add_argument("--converged-square", action="store_true",
                    help="only: the converged 50x50 cornell_box frame (the 1:1 view of BASELINE config C2, 600x600)")
    a = ap.parse_args()
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
    os.makedirs(OUT, exist_ok=True)
    if a.jpeg_only:
        make_jpeg()
        return
    if a.png_only:
        make_png()
        return
    if a.boxtrace_only:
        make_boxtrace()
        return
    if a.window:
        os.makedirs(OUT, exist_ok=True)
        make_window(3, 640, 640)   # final_scene: the globe (image texture) against the glass sphere with the blue medium, ground boxes below
        make_window(4, 900, 560)   # mesh_scene: the bunny's head against the back wall
        sys.exit(0)
    if a.converged_square:
        make_converged_square()
        return
    if a.smoke_only:
        a.scenes = "6"
    if not a.smoke_only:
        make_anim()
        if a.anim_only:
            return
        make_boxtrace()
    # extra graph shapes: one frame of the bounce animation (cpuRenderer.cpp:19) and a scene without BVH
    if not a.smoke_only:
        make_extras()
    for s in (int(x) for x in a.scenes.split(",")):
        refgen("dump", s, 64, 36, 9, 10, 1, f"{OUT}/scene{s}.rtscene")
        sc = abi.Scene.load(f"{OUT}/scene{s}.rtscene")
        sc.save_npz(f"{GOLD}/scene{s}.npz")
        refgen("trace", s, 64, 36, 9, 10, 1, f"{OUT}/trace{s}.bin", 2000)
        rays, xi, hits = read_trace(f"{OUT}/trace{s}.bin")
        np.savez_compressed(f"{GOLD}/trace{s}.npz", rays=rays.view("u1"), xi=xi, hits=hits.view("u1"))
        refgen("dump", s, 32, 18, 9, 12, 1, f"{OUT}/scene{s}_32.rtscene")
        cam = abi.Scene.load(f"{OUT}/scene{s}_32.rtscene").camera
        refgen("render", s, 32, 18, 9, 12, 1, f"{OUT}/img{s}.bin", 7)
        W, H, n, d, seed, sums, by = read_img(f"{OUT}/img{s}.bin")
        np.savez_compressed(f"{GOLD}/img{s}.npz", camera=cam.view("u1"), seed=np.array([seed]), sums=sums, bytes=by)
        print(f"scene {s}: {len(sc.nodes)} nodes, {len(rays)} traced rays, {W}x{H} frame")
        seeds = tuple(range(11, 36)) if s == 3 else (11, 12, 13, 14, 15)  # final_scene is ~16 dB noisier: 25 runs
        if a.converged:
            procs = [subprocess.Popen([os.path.join(REF, "refgen"), "render_nocheck", str(s), "48", "27", "16900", "50",
                                       "1", f"{OUT}/conv_{s}_{k}.bin", str(k)], cwd=WORK, env=ENV,
                                      stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for k in seeds]
            for p in procs:
                assert p.wait() == 0
        if all(os.path.exists(f"{OUT}/conv_{s}_{k}.bin") for k in seeds):
            refgen("dump", s, 48, 27, 16900, 50, 1, f"{OUT}/scene{s}_48.rtscene")
            cam = abi.Scene.load(f"{OUT}/scene{s}_48.rtscene").camera
            tot, spp = None, 0
            runs = []
            for k in seeds:
                W, H, n, d, seed, sums, by = read_img(f"{OUT}/conv_{s}_{k}.bin")
                runs.append(sums / (n * n))
                tot = sums if tot is None else tot + sums
                spp += n * n
            np.savez_compressed(f"{GOLD}/conv{s}.npz", camera=cam.view("u1"), mean=(tot / spp).astype(np.float64),
                                spp=np.array([spp]), runs=np.stack(runs[:5]).astype(np.float32))
            print(f"scene {s}: converged fixture from {spp} spp")


if __name__ == "__main__":
    main()
