"""GPU (-m gpu): rt_render_static, the entry point that allocates nothing (no counterpart in the reference; for callers
under HIP graph capture, serving loops, harnesses that forbid hipMalloc).  Called through the C ABI; its output must equal
rt_render_device's bit for bit (same kernels, same chunk schedule at these sizes) on two calls in a row, the second with a
smaller input -- the static scratch keeps the first call's contents --, in both precisions; a frame too large for two
partial frames in the static scratch renders as one sample chunk (no scratch but the counters) and still equals
rt_render_device; device memory in use does not change across the calls."""
import numpy as np
import pytest

from raytracing_amd import abi, lib
from tests import fixtures as fx

pytestmark = pytest.mark.gpu


def _free_bytes():
    import torch
    return torch.cuda.mem_get_info(0)[0]  # hipMemGetInfo through the runtime this process already uses


@pytest.mark.parametrize("n", [3, 2])
def test_static_entry_point_equals_render_device(gpu_ctx, n):
    import torch
    sc = fx.scene(n)
    ds = gpu_ctx.upload(sc)
    try:
        cam0, _, _, _ = fx.img(n)  # 32 x 18
        big = fx.with_size(sc.camera, sqrt_spp=5, max_depth=10)   # 64 x 36, 25 strata
        small = fx.with_size(cam0, sqrt_spp=3, max_depth=6)       # 32 x 18, 9 strata: the second, smaller call
        dev = torch.device("cuda", 0)
        for prec in (abi.RT_PREC_F32, abi.RT_PREC_F64):
            if prec == abi.RT_PREC_F64:
                scratch_probe = torch.empty(32 * 18 * 3, dtype=torch.float32, device=dev)
                with pytest.raises(lib.RtError, match="f64 stream"):  # nothing has built (allocated) the f64 stream yet
                    ds.render_static(small, scratch_probe.data_ptr(), seed=3, precision=prec)
            want = []
            for cam in (big, small):
                W, H = int(cam["image_width"][0]), int(cam["image_height"][0])
                out = torch.full((H * W * 3,), -7.0, dtype=torch.float32, device=dev)
                ds.render_device(cam, out.data_ptr(), seed=3, precision=prec)
                torch.cuda.synchronize()
                want.append(out.cpu().numpy())
            outs = [torch.full((w.size,), -9.0, dtype=torch.float32, device=dev) for w in want]
            warm = torch.empty(want[0].size, dtype=torch.float32, device=dev)
            ds.render_static(big, warm.data_ptr(), seed=3, precision=prec)  # first use: the runtime may load code lazily
            torch.cuda.synchronize()
            before = _free_bytes()
            ds.render_static(big, outs[0].data_ptr(), seed=3, precision=prec)
            ds.render_static(small, outs[1].data_ptr(), seed=3, precision=prec)  # back to back on the same stream
            torch.cuda.synchronize()
            assert _free_bytes() == before, "rt_render_static must not allocate device memory"
            for got, w in zip(outs, want):
                g = got.cpu().numpy()
                assert np.isfinite(w).any() and w.any()
                assert g.tobytes() == w.tobytes(), f"precision {prec}: {int((g != w).sum())} of {w.size} values differ"
            # a sub-range and forced chunking go the same way
            a = torch.empty(want[0].size, dtype=torch.float32, device=dev)
            b = torch.empty(want[0].size, dtype=torch.float32, device=dev)
            ds.render_device(big, a.data_ptr(), seed=3, s_begin=4, s_end=21, precision=prec, sample_chunks=3)
            ds.render_static(big, b.data_ptr(), seed=3, s_begin=4, s_end=21, precision=prec, sample_chunks=3)
            torch.cuda.synchronize()
            assert a.cpu().numpy().tobytes() == b.cpu().numpy().tobytes()
            ds.render_static(big, b.data_ptr(), seed=3, s_begin=6, s_end=6, precision=prec)  # empty range: zeros
            torch.cuda.synchronize()
            assert not b.cpu().numpy().any()
    finally:
        ds.close()


def test_static_entry_point_renders_oversize_frames_as_one_chunk(gpu_ctx):
    """A 4096 x 2048 frame is 201 MB of 64-bit cells per partial frame, more than the 64 MiB static scratch: the plan falls
    back to ONE sample chunk, written straight to d_accum (refuse only what cannot be rendered).  Same integer
    sums as rt_render_device's several chunks.  Null output pointers are still refused."""
    import torch
    ds = gpu_ctx.upload(fx.scene(2))
    try:
        cam = fx.with_size(fx.scene(2).camera, sqrt_spp=2, max_depth=4)
        W, H = 4096, 2048
        cam["image_width"], cam["image_height"] = W, H
        dev = torch.device("cuda", 0)
        a = torch.full((H * W * 3,), -3.0, dtype=torch.float32, device=dev)
        b = torch.full((H * W * 3,), -5.0, dtype=torch.float32, device=dev)
        ds.render_device(cam, a.data_ptr(), seed=1, precision=abi.RT_PREC_F32, sample_chunks=2)
        ds.render_static(cam, b.data_ptr(), seed=1, precision=abi.RT_PREC_F32, sample_chunks=2)
        torch.cuda.synchronize()
        x, y = a.cpu().numpy(), b.cpu().numpy()
        assert x.any() and y.tobytes() == x.tobytes()
        with pytest.raises(lib.RtError):
            ds.render_static(cam, 0, seed=1)
    finally:
        ds.close()


def test_static_entry_point_at_the_headline_size(gpu_ctx):
    """1920 x 1080 fits (one partial frame of 64-bit cells: the static variant renders it as a single sample chunk): one
    eighth of the headline frame's strata, against rt_render_device, which cuts them into several chunks."""
    import torch
    from raytracing_amd import scenes as scn
    scene = scn.build("final_scene", 1920, 1080, 1000, 50)
    ds = gpu_ctx.upload(scene)
    try:
        dev = torch.device("cuda", 0)
        a = torch.empty(1920 * 1080 * 3, dtype=torch.float32, device=dev)
        b = torch.empty(1920 * 1080 * 3, dtype=torch.float32, device=dev)
        st = ds.render_device(scene.camera, a.data_ptr(), seed=11, s_begin=0, s_end=40, precision=abi.RT_PREC_F32, want_stats=True)
        ds.render_static(scene.camera, b.data_ptr(), seed=11, s_begin=0, s_end=40, precision=abi.RT_PREC_F32)
        torch.cuda.synchronize()
        x, y = a.cpu().numpy(), b.cpu().numpy()
        ok = np.isfinite(x)
        assert st["samples"] == 1920 * 1080 * 40 and np.array_equal(ok, np.isfinite(y))
        # different chunk schedules, different wave schedules -- and the same integer sums
        assert y.tobytes() == x.tobytes()
    finally:
        ds.close()
