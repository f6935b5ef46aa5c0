"""ctypes / numpy mirror of include/rt_amd.h (the C ABI of the HIP path tracer).

Every struct here has the exact layout of its C counterpart; `check_layout()` is verified against the
compiled library's `rt_abi_sizeof()` table by the CPU tests.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

RT_AMD_ABI_VERSION = 3
RT_TUNE_CANDIDATES = 6  # include/rt_amd.h
RT_ALL_STRATA = -1

RT_OK, RT_ERR_INVALID, RT_ERR_HIP, RT_ERR_UNSUPPORTED, RT_ERR_NOMEM = 0, -1, -2, -3, -4

(RT_NODE_BVH, RT_NODE_LIST, RT_NODE_TRANSLATE, RT_NODE_ROTATE_Y, RT_NODE_MEDIUM, RT_NODE_SPHERE, RT_NODE_QUAD,
 RT_NODE_TRIANGLE) = range(8)
RT_MAT_LAMBERTIAN, RT_MAT_METAL, RT_MAT_DIELECTRIC, RT_MAT_DIFFUSE_LIGHT, RT_MAT_ISOTROPIC = range(5)
RT_TEX_SOLID, RT_TEX_CHECKER, RT_TEX_IMAGE, RT_TEX_NOISE = range(4)
RT_PREC_F32, RT_PREC_F64 = 0, 1

node_dtype = np.dtype([("kind", "<i4"), ("a", "<i4"), ("b", "<i4"), ("reserved", "<i4"), ("bbox", "<f8", (6,)),
                       ("p", "<f8", (4,))], align=True)
sphere_dtype = np.dtype([("center", "<f8", (3,)), ("dcenter", "<f8", (3,)), ("radius", "<f8"), ("mat", "<i4"),
                         ("reserved", "<i4")], align=True)
quad_dtype = np.dtype([("Q", "<f8", (3,)), ("u", "<f8", (3,)), ("v", "<f8", (3,)), ("w", "<f8", (3,)),
                       ("normal", "<f8", (3,)), ("D", "<f8"), ("area", "<f8"), ("mat", "<i4"), ("reserved", "<i4")],
                      align=True)
triangle_dtype = np.dtype([("v0", "<f8", (3,)), ("v1", "<f8", (3,)), ("v2", "<f8", (3,)), ("normal", "<f8", (3,)),
                           ("mat", "<i4"), ("cull", "<i4")], align=True)
material_dtype = np.dtype([("kind", "<i4"), ("tex", "<i4"), ("albedo", "<f8", (3,)), ("fuzz", "<f8"), ("ior", "<f8")],
                          align=True)
texture_dtype = np.dtype([("kind", "<i4"), ("a", "<i4"), ("b", "<i4"), ("reserved", "<i4"), ("color", "<f8", (3,)),
                          ("scale", "<f8")], align=True)
image_dtype = np.dtype([("width", "<i4"), ("height", "<i4"), ("offset", "<i8")], align=True)
perlin_dtype = np.dtype([("randvec", "<f8", (256, 3)), ("perm_x", "<i4", (256,)), ("perm_y", "<i4", (256,)),
                         ("perm_z", "<i4", (256,))], align=True)
camera_dtype = np.dtype([("image_width", "<i4"), ("image_height", "<i4"), ("sqrt_spp", "<i4"), ("max_depth", "<i4"),
                         ("center", "<f8", (3,)), ("pixel00_loc", "<f8", (3,)), ("pixel_delta_u", "<f8", (3,)),
                         ("pixel_delta_v", "<f8", (3,)), ("defocus_disk_u", "<f8", (3,)),
                         ("defocus_disk_v", "<f8", (3,)), ("defocus_angle", "<f8"), ("background", "<f8", (3,))],
                        align=True)
ray_dtype = np.dtype([("orig", "<f8", (3,)), ("dir", "<f8", (3,)), ("time", "<f8")], align=True)
hit_dtype = np.dtype([("prim", "<i4"), ("front_face", "<i4"), ("t", "<f8"), ("p", "<f8", (3,)),
                      ("normal", "<f8", (3,)), ("u", "<f8"), ("v", "<f8")], align=True)

EXPECTED_SIZES = {
    "rt_node": (node_dtype, 96), "rt_sphere": (sphere_dtype, 64), "rt_quad": (quad_dtype, 144),
    "rt_triangle": (triangle_dtype, 104), "rt_material": (material_dtype, 48), "rt_texture": (texture_dtype, 48),
    "rt_image": (image_dtype, 16), "rt_perlin": (perlin_dtype, 9216), "rt_camera": (camera_dtype, 192),
    "rt_ray": (ray_dtype, 56), "rt_hit": (hit_dtype, 80),
}


def check_layout() -> None:
    for name, (dt, size) in EXPECTED_SIZES.items():
        if dt.itemsize != size:
            raise AssertionError(f"{name}: numpy itemsize {dt.itemsize} != {size}")


class SceneDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("world_root", C.c_int32), ("lights_root", C.c_int32), ("n_nodes", C.c_int32),
        ("nodes", C.c_void_p),
        ("n_child_index", C.c_int32), ("child_index", C.c_void_p),
        ("n_spheres", C.c_int32), ("spheres", C.c_void_p),
        ("n_quads", C.c_int32), ("quads", C.c_void_p),
        ("n_triangles", C.c_int32), ("triangles", C.c_void_p),
        ("n_materials", C.c_int32), ("materials", C.c_void_p),
        ("n_textures", C.c_int32), ("textures", C.c_void_p),
        ("n_images", C.c_int32), ("images", C.c_void_p),
        ("n_image_bytes", C.c_int64), ("image_bytes", C.c_void_p),
        ("n_perlins", C.c_int32), ("perlins", C.c_void_p),
    ]


class RenderParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("s_begin", C.c_int32), ("s_end", C.c_int32), ("precision", C.c_int32),
                ("sample_chunks", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays", C.c_uint64), ("kernel_ms", C.c_double), ("sample_chunks", C.c_uint32), ("reserved", C.c_uint32)]


_ARRAYS = [("nodes", node_dtype), ("child_index", np.dtype("<i4")), ("spheres", sphere_dtype), ("quads", quad_dtype),
           ("triangles", triangle_dtype), ("materials", material_dtype), ("textures", texture_dtype),
           ("images", image_dtype), ("image_bytes", np.dtype("u1")), ("perlins", perlin_dtype)]


@dataclass
class Scene:
    """A flat scene description held in numpy arrays (see rt_scene_desc in include/rt_amd.h)."""
    world_root: int = -1
    lights_root: int = -1
    arrays: dict = field(default_factory=dict)
    camera: np.ndarray | None = None  # shape (1,) of camera_dtype, optional

    def __post_init__(self):
        for name, dt in _ARRAYS:
            self.arrays.setdefault(name, np.zeros(0, dtype=dt))

    def __getattr__(self, name):
        arrays = self.__dict__.get("arrays", {})
        if name in arrays:
            return arrays[name]
        raise AttributeError(name)

    def desc(self) -> SceneDesc:
        """Build the ctypes rt_scene_desc; keeps the numpy arrays alive via self."""
        d = SceneDesc()
        d.abi_version = RT_AMD_ABI_VERSION
        d.world_root = self.world_root
        d.lights_root = self.lights_root
        for name, dt in _ARRAYS:
            arr = np.ascontiguousarray(self.arrays[name], dtype=dt)
            self.arrays[name] = arr
            setattr(d, "n_" + name, arr.shape[0])
            setattr(d, name, arr.ctypes.data if arr.size else None)
        return d

    # ---- "RTSCENE1" container (written by the host library and by oracle/refgen) ----
    @staticmethod
    def load(path: str) -> "Scene":
        if path.endswith(".npz"):
            z = np.load(path, allow_pickle=False)
            sc = Scene(int(z["roots"][0]), int(z["roots"][1]))
            for name, dt in _ARRAYS:
                sc.arrays[name] = np.ascontiguousarray(z[name].view(dt) if z[name].dtype != dt else z[name])
            if "camera" in z.files:
                sc.camera = z["camera"].view(camera_dtype)
            return sc
        with open(path, "rb") as f:
            buf = f.read()
        if buf[:8] != b"RTSCENE1":
            raise ValueError(f"{path}: not an RTSCENE1 file")
        off = 8
        roots = np.frombuffer(buf, "<i4", 2, off)
        off += 8
        sc = Scene(int(roots[0]), int(roots[1]))
        for name, dt in _ARRAYS:
            n = int(np.frombuffer(buf, "<i8", 1, off)[0])
            off += 8
            sc.arrays[name] = np.frombuffer(buf, dt, n, off).copy()
            off += n * dt.itemsize
        if len(buf) - off >= camera_dtype.itemsize:
            sc.camera = np.frombuffer(buf, camera_dtype, 1, off).copy()
        return sc

    def save_npz(self, path: str) -> None:
        out = {"roots": np.array([self.world_root, self.lights_root], dtype="<i4")}
        for name, dt in _ARRAYS:
            a = self.arrays[name]
            # structured dtypes are stored as raw bytes so np.load needs no pickle
            out[name] = a.view("u1") if a.dtype.fields else a
        if self.camera is not None:
            out["camera"] = self.camera.view("u1")
        np.savez_compressed(path, **out)

    def equal(self, other: "Scene") -> bool:
        if (self.world_root, self.lights_root) != (other.world_root, other.lights_root):
            return False
        return all(self.arrays[n].tobytes() == other.arrays[n].tobytes() for n, _ in _ARRAYS)


def _fix_npz_views(sc: Scene) -> None:
    for name, dt in _ARRAYS:
        a = sc.arrays[name]
        if a.dtype != dt:
            sc.arrays[name] = a.view(dt)
