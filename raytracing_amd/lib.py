"""ctypes binding of raytracing_amd/librt_amd.so (the C ABI of include/rt_amd.h).

There is no CPU fallback: if the HIP library is missing or no GPU is present, loading / rt_init raises.
torch is used by callers only for device buffers, streams and torch.distributed; this module itself needs numpy only
(it imports torch, when installed, before loading the library so that the process holds one HIP runtime: see load()).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "librt_amd.so")  # RT_AMD_LIB: diagnostic builds

EXPORTS = ["rt_abi_sizeof", "rt_init", "rt_shutdown", "rt_last_error", "rt_scene_upload", "rt_scene_free",
           "rt_scene_upload_device_build", "rt_scene_tune", "rt_scene_download_stream", "rt_scene_info", "rt_scene_flatten_info", "rt_scene_flatten_stream", "rt_trace_batch", "rt_render_device", "rt_tonemap_device", "rt_render",
           "rt_render_multi", "rt_render_static"]

_lib = None


class RtError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the HIP library (raises if it has not been built: run `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtError(f"{LIB_PATH} not found: the HIP extension is not built (make -C raytracing_amd/csrc)")
    # One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so (soname libamdhip64.so.7, the one librt_amd.so
    # asks for); callers that hand this library torch device buffers need both to use the same copy.  The loader matches by
    # soname only against what is ALREADY loaded, so torch comes first: loaded after librt_amd.so it would bring a second
    # runtime into the process and find "No HIP GPUs".
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.rt_abi_sizeof.restype = C.c_int
    L.rt_abi_sizeof.argtypes = [C.c_char_p]
    L.rt_init.restype = C.c_int
    L.rt_init.argtypes = [C.c_int, C.POINTER(vp)]
    L.rt_shutdown.restype = None
    L.rt_shutdown.argtypes = [vp]
    L.rt_last_error.restype = C.c_char_p
    L.rt_last_error.argtypes = [vp]
    L.rt_scene_upload.restype = C.c_int
    L.rt_scene_upload.argtypes = [vp, C.POINTER(abi.SceneDesc), C.POINTER(vp)]
    L.rt_scene_upload_device_build.restype = C.c_int
    L.rt_scene_upload_device_build.argtypes = [vp, C.POINTER(abi.SceneDesc), C.POINTER(vp)]
    L.rt_scene_download_stream.restype = C.c_int
    L.rt_scene_download_stream.argtypes = [vp, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.rt_scene_free.restype = None
    L.rt_scene_free.argtypes = [vp]
    L.rt_scene_info.restype = C.c_int
    L.rt_scene_info.argtypes = [vp, vp]
    L.rt_scene_tune.restype = C.c_int
    L.rt_scene_tune.argtypes = [vp, vp, vp, C.c_int, vp, vp]
    L.rt_scene_flatten_info.restype = C.c_int
    L.rt_scene_flatten_info.argtypes = [C.POINTER(abi.SceneDesc), C.c_int, vp]
    L.rt_scene_flatten_stream.restype = C.c_int
    L.rt_scene_flatten_stream.argtypes = [C.POINTER(abi.SceneDesc), C.c_int, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.rt_trace_batch.restype = C.c_int
    L.rt_trace_batch.argtypes = [vp, vp, vp, vp, C.c_size_t, vp, C.c_int]
    L.rt_render_device.restype = C.c_int
    L.rt_render_device.argtypes = [vp, vp, vp, C.POINTER(abi.RenderParams), vp, vp, C.POINTER(abi.Stats)]
    L.rt_tonemap_device.restype = C.c_int
    L.rt_tonemap_device.argtypes = [vp, vp, C.c_double, C.c_size_t, vp, vp]
    L.rt_render.restype = C.c_int
    L.rt_render.argtypes = [vp, vp, vp, C.POINTER(abi.RenderParams), vp, vp, C.POINTER(abi.Stats)]
    L.rt_render_static.restype = C.c_int
    L.rt_render_static.argtypes = [vp, vp, vp, C.POINTER(abi.RenderParams), vp, vp]
    L.rt_render_multi.restype = C.c_int
    L.rt_render_multi.argtypes = [C.POINTER(vp), C.POINTER(vp), C.c_int, vp, C.POINTER(abi.RenderParams), vp, vp,
                                  C.POINTER(abi.Stats)]
    _lib = L
    return L


def flatten_info(scene: abi.Scene, reference_order: bool = False) -> dict:
    """Host-only: what rt_scene_upload would build for `scene` (raises RtError with the status it would return).
    reference_order: the f32 stream with the reference's own bvh_nodes and visiting order (structure tests)."""
    L = load()
    d = scene.desc()
    out = np.zeros(5, dtype=np.int32)
    rc = L.rt_scene_flatten_info(C.byref(d), int(reference_order), out.ctypes.data)
    if rc != abi.RT_OK:
        raise RtError(f"rt_scene_flatten_info failed with status {rc}")
    return {"f32_units": int(out[0]), "f64_units": int(out[1]), "fused_boxes": int(out[2]), "lights": int(out[3]),
            "baked_instances": int(out[4])}


def flatten_stream(scene: abi.Scene, reference_order: bool = False):
    """Host-only: the f32 record stream rt_scene_upload would build, as (units, node_ids): a (n, 4) uint32 array of
    16-byte units (layout: csrc/rt_scene.h) and the description node id of the record starting at each unit (-1: none)."""
    L = load()
    d = scene.desc()
    n = C.c_size_t(0)
    rc = L.rt_scene_flatten_stream(C.byref(d), int(reference_order), None, None, 0, C.byref(n))
    if rc != abi.RT_OK:
        raise RtError(f"rt_scene_flatten_stream failed with status {rc}")
    units = np.zeros((n.value, 4), dtype=np.uint32)
    ids = np.zeros(n.value, dtype=np.int32)
    rc = L.rt_scene_flatten_stream(C.byref(d), int(reference_order), units.ctypes.data, ids.ctypes.data, n.value, C.byref(n))
    if rc != abi.RT_OK:
        raise RtError(f"rt_scene_flatten_stream failed with status {rc}")
    return units, ids


class Context:
    """rt_ctx on one HIP device."""

    def __init__(self, device: int = 0):
        self.L = load()
        h = C.c_void_p()
        rc = self.L.rt_init(device, C.byref(h))
        if rc != abi.RT_OK:
            raise RtError(f"rt_init(device={device}) failed with status {rc} (is a HIP GPU visible?)")
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.L.rt_shutdown(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc: int, what: str):
        if rc != abi.RT_OK:
            msg = self.L.rt_last_error(self.h)
            raise RtError(f"{what} failed with status {rc}: {msg.decode() if msg else ''}")

    def upload(self, scene: abi.Scene, device_build: bool = False) -> "DeviceScene":
        """device_build: the float stream's hierarchy is built on the GPU (rt_scene_upload_device_build)."""
        return DeviceScene(self, scene, device_build)


class DeviceScene:
    def __init__(self, ctx: Context, scene: abi.Scene, device_build: bool = False):
        self.ctx = ctx
        self.host = scene
        d = scene.desc()
        h = C.c_void_p()
        if device_build:
            rc = ctx.L.rt_scene_upload_device_build(ctx.h, C.byref(d), C.byref(h))
            ctx.check(rc, "rt_scene_upload_device_build")
        else:
            rc = ctx.L.rt_scene_upload(ctx.h, C.byref(d), C.byref(h))
            ctx.check(rc, "rt_scene_upload")
        self.h = h

    def download_stream(self):
        """The f32 record stream as it stands on the device: (units (n, 4) uint32, node ids per unit)."""
        n = C.c_size_t(0)
        self.ctx.check(self.ctx.L.rt_scene_download_stream(self.h, None, None, 0, C.byref(n)), "rt_scene_download_stream")
        units = np.zeros((n.value, 4), dtype=np.uint32)
        ids = np.zeros(n.value, dtype=np.int32)
        self.ctx.check(self.ctx.L.rt_scene_download_stream(self.h, units.ctypes.data, ids.ctypes.data, n.value, C.byref(n)),
                       "rt_scene_download_stream")
        return units, ids

    def tune(self, camera: np.ndarray, probe_strata: int = 0) -> dict:
        """rt_scene_tune: time the hierarchy build's candidate settings on `camera` and keep the fastest stream."""
        cam = np.ascontiguousarray(camera, dtype=abi.camera_dtype).reshape(1)
        ms = np.zeros(abi.RT_TUNE_CANDIDATES, dtype=np.float32)
        chosen = np.zeros(1, dtype=np.int32)
        rc = self.ctx.L.rt_scene_tune(self.ctx.h, self.h, cam.ctypes.data, probe_strata, ms.ctypes.data, chosen.ctypes.data)
        self.ctx.check(rc, "rt_scene_tune")
        return {"probe_ms": [float(x) for x in ms], "chosen": int(chosen[0])}

    def info(self) -> dict:
        out = np.zeros(5, dtype=np.int32)
        self.ctx.check(self.ctx.L.rt_scene_info(self.h, out.ctypes.data), "rt_scene_info")
        return {"flat_nodes": int(out[0]), "lights": int(out[1]), "f32_bytes": int(out[2]), "f64_bytes": int(out[3])}

    def close(self):
        if getattr(self, "h", None):
            self.ctx.L.rt_scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def trace_batch(self, rays: np.ndarray, xi: np.ndarray | None, precision: int = abi.RT_PREC_F64) -> np.ndarray:
        rays = np.ascontiguousarray(rays, dtype=abi.ray_dtype)
        n = rays.shape[0]
        out = np.zeros(n, dtype=abi.hit_dtype)
        xi_p = None
        if xi is not None:
            xi = np.ascontiguousarray(xi, dtype=np.float64)
            assert xi.shape[0] == n
            xi_p = xi.ctypes.data
        rc = self.ctx.L.rt_trace_batch(self.ctx.h, self.h, rays.ctypes.data if n else None, xi_p, n,
                                       out.ctypes.data if n else None, precision)
        self.ctx.check(rc, "rt_trace_batch")
        return out

    def render(self, camera: np.ndarray, seed: int = 0, s_begin: int = 0, s_end: int = abi.RT_ALL_STRATA,
               precision: int = abi.RT_PREC_F32, sample_chunks: int = 0, want_rgb8: bool = True):
        """Host-buffer render through rt_render: returns (accum[H,W,3] float32 sums, rgb8[H,W,3] or None, stats)."""
        cam = np.ascontiguousarray(camera, dtype=abi.camera_dtype).reshape(1)
        W, H = int(cam["image_width"][0]), int(cam["image_height"][0])
        accum = np.zeros((H, W, 3), dtype=np.float32)
        rgb = np.zeros((H, W, 3), dtype=np.uint8) if want_rgb8 else None
        p = abi.RenderParams(seed, s_begin, s_end, precision, sample_chunks)
        st = abi.Stats()
        rc = self.ctx.L.rt_render(self.ctx.h, self.h, cam.ctypes.data, C.byref(p), accum.ctypes.data,
                                  rgb.ctypes.data if want_rgb8 else None, C.byref(st))
        self.ctx.check(rc, "rt_render")
        return accum, rgb, {"samples": st.samples, "rays": st.rays, "kernel_ms": st.kernel_ms, "sample_chunks": st.sample_chunks}

    def render_device(self, camera: np.ndarray, d_accum_ptr: int, seed: int = 0, s_begin: int = 0, s_end: int = abi.RT_ALL_STRATA,
                      precision: int = abi.RT_PREC_F32, sample_chunks: int = 0, stream: int = 0, want_stats: bool = False):
        """Device-buffer render through rt_render_device (d_accum_ptr: device address of H*W*3 floats)."""
        cam = np.ascontiguousarray(camera, dtype=abi.camera_dtype).reshape(1)
        p = abi.RenderParams(seed, s_begin, s_end, precision, sample_chunks)
        st = abi.Stats()
        rc = self.ctx.L.rt_render_device(self.ctx.h, self.h, cam.ctypes.data, C.byref(p), d_accum_ptr, stream or None,
                                         C.byref(st) if want_stats else None)
        self.ctx.check(rc, "rt_render_device")
        return {"samples": st.samples, "rays": st.rays, "kernel_ms": st.kernel_ms, "sample_chunks": st.sample_chunks} if want_stats else None

    def render_static(self, camera: np.ndarray, d_accum_ptr: int, seed: int = 0, s_begin: int = 0,
                      s_end: int = abi.RT_ALL_STRATA, precision: int = abi.RT_PREC_F32, sample_chunks: int = 0, stream: int = 0):
        """rt_render_static: rt_render_device without any allocation (scratch in a static device array; size-limited)."""
        cam = np.ascontiguousarray(camera, dtype=abi.camera_dtype).reshape(1)
        p = abi.RenderParams(seed, s_begin, s_end, precision, sample_chunks)
        rc = self.ctx.L.rt_render_static(self.ctx.h, self.h, cam.ctypes.data, C.byref(p), d_accum_ptr, stream or None)
        self.ctx.check(rc, "rt_render_static")

    def tonemap_device(self, d_accum_ptr: int, scale: float, n_pixels: int, d_rgb_ptr: int, stream: int = 0):
        rc = self.ctx.L.rt_tonemap_device(self.ctx.h, d_accum_ptr, scale, n_pixels, d_rgb_ptr, stream or None)
        self.ctx.check(rc, "rt_tonemap_device")


def render_multi(scenes: list, camera: np.ndarray, seed: int = 0, s_begin: int = 0, s_end: int = abi.RT_ALL_STRATA,
                 precision: int = abi.RT_PREC_F32, sample_chunks: int = 0, want_rgb8: bool = True):
    """One frame on several contexts from this process (rt_render_multi): `scenes` = the same scene uploaded through one
    Context per device (two contexts may share a device).  Returns what DeviceScene.render returns."""
    cam = np.ascontiguousarray(camera, dtype=abi.camera_dtype).reshape(1)
    W, H = int(cam["image_width"][0]), int(cam["image_height"][0])
    accum = np.zeros((H, W, 3), dtype=np.float32)
    rgb = np.zeros((H, W, 3), dtype=np.uint8) if want_rgb8 else None
    p = abi.RenderParams(seed, s_begin, s_end, precision, sample_chunks)
    st = abi.Stats()
    n = len(scenes)
    ctxs = (C.c_void_p * n)(*[ds.ctx.h for ds in scenes])
    hs = (C.c_void_p * n)(*[ds.h for ds in scenes])
    root = scenes[0].ctx
    rc = root.L.rt_render_multi(ctxs, hs, n, cam.ctypes.data, C.byref(p), accum.ctypes.data,
                                rgb.ctypes.data if want_rgb8 else None, C.byref(st))
    root.check(rc, "rt_render_multi")
    return accum, rgb, {"samples": st.samples, "rays": st.rays, "kernel_ms": st.kernel_ms, "sample_chunks": st.sample_chunks}
