"""Synthetic scene graphs for tests of the flattener's own hierarchy (test infrastructure): a flat hittable_list of random
spheres (clusters, exact duplicates, concentric shells, huge and tiny radii), quads, triangles, a sphere-bounded and a
quad-bounded constant_medium, a bakeable instance (translate(rotate_y(list of spheres))) and one that is not
(translate(rotate_y(list of quads))).  No bvh_node: the double (parity) stream then needs no bounding boxes from the
description, so the graph can be written down directly; the float stream builds its boxes from the geometry."""
from __future__ import annotations

import math

import numpy as np

from raytracing_amd import abi


class _Builder:
    def __init__(self):
        self.nodes, self.child_index, self.spheres, self.quads, self.triangles = [], [], [], [], []

    def node(self, kind, a=0, b=0, p=(0, 0, 0, 0), bbox=None):
        bb = bbox if bbox is not None else (0, 0, 0, 0, 0, 0)
        self.nodes.append((kind, a, b, 0, tuple(bb), tuple(p)))
        return len(self.nodes) - 1

    def sphere(self, c, r, mat=0, dc=(0, 0, 0)):
        self.spheres.append((tuple(c), tuple(dc), r, mat, 0))
        lo = np.minimum(c, np.add(c, dc)) - abs(r)
        hi = np.maximum(c, np.add(c, dc)) + abs(r)
        return self.node(abi.RT_NODE_SPHERE, len(self.spheres) - 1, bbox=(lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]))

    def quad(self, Q, u, v, mat=0):
        Q, u, v = (np.asarray(x, dtype=np.float64) for x in (Q, u, v))
        n = np.cross(u, v)
        normal = n / math.sqrt(n @ n)
        self.quads.append((tuple(Q), tuple(u), tuple(v), tuple(n / (n @ n)), tuple(normal), float(normal @ Q), math.sqrt(n @ n), mat, 0))
        pts = np.array([Q, Q + u, Q + v, Q + u + v])
        lo, hi = pts.min(0) - 5e-5, pts.max(0) + 5e-5
        return self.node(abi.RT_NODE_QUAD, len(self.quads) - 1, bbox=(lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]))

    def triangle(self, v0, v1, v2, mat=0):
        v0, v1, v2 = (np.asarray(x, dtype=np.float64) for x in (v0, v1, v2))
        n = np.cross(v1 - v0, v2 - v0)
        self.triangles.append((tuple(v0), tuple(v1), tuple(v2), tuple(n / math.sqrt(n @ n)), mat, 0))
        pts = np.array([v0, v1, v2])
        lo, hi = pts.min(0) - 5e-5, pts.max(0) + 5e-5
        return self.node(abi.RT_NODE_TRIANGLE, len(self.triangles) - 1, bbox=(lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]))

    def box(self, a, b, mat=0):
        """quad.h:103 box(): the six quads in the reference's order, as a list"""
        lo, hi = np.minimum(a, b).astype(np.float64), np.maximum(a, b).astype(np.float64)
        dx, dy, dz = np.array([hi[0] - lo[0], 0, 0]), np.array([0, hi[1] - lo[1], 0]), np.array([0, 0, hi[2] - lo[2]])
        return self.list([self.quad((lo[0], lo[1], hi[2]), dx, dy, mat), self.quad((hi[0], lo[1], hi[2]), -dz, dy, mat),
                          self.quad((hi[0], lo[1], lo[2]), -dx, dy, mat), self.quad((lo[0], lo[1], lo[2]), dz, dy, mat),
                          self.quad((lo[0], hi[1], hi[2]), dx, -dz, mat), self.quad((lo[0], lo[1], lo[2]), dx, dz, mat)])

    def list(self, kids):
        a = len(self.child_index)
        self.child_index += list(kids)
        return self.node(abi.RT_NODE_LIST, a, len(kids))


def build(seed: int, n_spheres: int = 240, single_records: bool = False) -> abi.Scene:
    """single_records: leave out what is more than one record in the float stream (the quad-bounded medium, the unbaked
    instance): what the device build accepts."""
    rng = np.random.default_rng(seed)
    b = _Builder()
    top = []
    centres = rng.uniform(-40, 40, size=(6, 3))
    for k in range(n_spheres):
        c = centres[rng.integers(0, 6)] + rng.normal(size=3) * rng.choice([0.5, 4.0, 15.0])
        top.append(b.sphere(c, float(rng.choice([0.05, 0.7, 2.5])) * (1 + rng.random())))
    dup = b.sphere((3.0, 2.0, 1.0), 1.25)
    top += [dup] * 9                                                   # one node listed ten times: equal centroids, a shared subtree
    top += [b.sphere((3.0, 2.0, 1.0), 1.25 + 0.5 * k) for k in range(6)]  # concentric shells
    top.append(b.sphere((0, -3000, 0), 2950.0))                        # a ground much larger than everything else
    top.append(b.sphere((10, 5, -10), 1.5, dc=(4.0, 0.5, -2.0)))       # moving
    for k in range(25):
        Q = rng.uniform(-50, 50, size=3)
        if k % 2 == 0:                                                 # axis-aligned (flat boxes), else tilted
            ax = rng.integers(0, 3)
            u, v = np.zeros(3), np.zeros(3)
            u[(ax + 1) % 3], v[(ax + 2) % 3] = rng.uniform(2, 30), rng.uniform(2, 30)
        else:
            u, v = rng.normal(size=3) * 10, rng.normal(size=3) * 10
        top.append(b.quad(Q, u, v))
    for k in range(30):
        p = rng.uniform(-45, 45, size=3)
        top.append(b.triangle(p, p + rng.normal(size=3) * 6, p + rng.normal(size=3) * 6))
    # media: material 1 (isotropic); a sphere boundary that encloses the scene's core, and a boundary made of two quads
    top.append(b.node(abi.RT_NODE_MEDIUM, b.sphere((0, 0, 0), 60.0, mat=-1), 1, p=(-1.0 / 0.004, 0, 0, 0)))
    slab = b.list([b.quad((-20, 8, -20), (40, 0, 0), (0, 0, 40)), b.quad((-20, 14, -20), (40, 0, 0), (0, 0, 40))])
    if not single_records:
        top.append(b.node(abi.RT_NODE_MEDIUM, slab, 1, p=(-1.0 / 0.05, 0, 0, 0)))
    # instances
    th = math.radians(25.0)
    cluster = b.list([b.sphere(rng.normal(size=3) * 3, 0.4 + rng.random()) for _ in range(20)])
    top.append(b.node(abi.RT_NODE_TRANSLATE, b.node(abi.RT_NODE_ROTATE_Y, cluster, p=(math.sin(th), math.cos(th), 0, 0)), p=(25, 10, 5, 0)))
    walls = b.list([b.quad(rng.uniform(-4, 4, size=3), rng.normal(size=3) * 3, rng.normal(size=3) * 3) for _ in range(6)])
    if not single_records:
        top.append(b.node(abi.RT_NODE_TRANSLATE, b.node(abi.RT_NODE_ROTATE_Y, walls, p=(math.sin(-th), math.cos(-th), 0, 0)), p=(-25, 6, 12, 0)))
    # boxes: a plain box() (one AABOX record), and the ways the reference places one (one OBOX record each)
    top.append(b.box((-38, -6, 20), (-30, 4, 31)))
    th2 = math.radians(-18.0)
    top.append(b.node(abi.RT_NODE_TRANSLATE, b.node(abi.RT_NODE_ROTATE_Y, b.box((0, 0, 0), (9, 17, 8)), p=(math.sin(th2), math.cos(th2), 0, 0)), p=(12, -4, -30, 0)))
    top.append(b.node(abi.RT_NODE_ROTATE_Y, b.box((20, 0, 20), (26, 5, 29)), p=(math.sin(th), math.cos(th), 0, 0)))
    top.append(b.node(abi.RT_NODE_TRANSLATE, b.box((0, 0, 0), (5, 5, 5)), p=(-12, 15, 3, 0)))
    world = b.list(top)
    light = b.quad((-5, 70, -5), (10, 0, 0), (0, 0, 10), mat=2)
    arrays = {
        "nodes": np.array(b.nodes, dtype=abi.node_dtype), "child_index": np.array(b.child_index, dtype="<i4"),
        "spheres": np.array(b.spheres, dtype=abi.sphere_dtype), "quads": np.array(b.quads, dtype=abi.quad_dtype),
        "triangles": np.array(b.triangles, dtype=abi.triangle_dtype),
        "materials": np.array([(abi.RT_MAT_LAMBERTIAN, 0, (0, 0, 0), 0, 0), (abi.RT_MAT_ISOTROPIC, 0, (0, 0, 0), 0, 0),
                               (abi.RT_MAT_DIFFUSE_LIGHT, 1, (0, 0, 0), 0, 0)], dtype=abi.material_dtype),
        "textures": np.array([(abi.RT_TEX_SOLID, 0, 0, 0, (0.7, 0.6, 0.5), 0), (abi.RT_TEX_SOLID, 0, 0, 0, (9, 9, 9), 0)], dtype=abi.texture_dtype),
    }
    return abi.Scene(world_root=world, lights_root=light, arrays=arrays)


def rays(seed: int, n: int = 3000) -> np.ndarray:
    rng = np.random.default_rng(seed + 1000)
    r = np.zeros(n, dtype=abi.ray_dtype)
    o = rng.uniform(-55, 55, size=(n, 3))
    o[: n // 3] = rng.normal(size=(n // 3, 3)) * 150          # from outside
    d = rng.normal(size=(n, 3))
    d[: n // 3] = -o[: n // 3] + rng.normal(size=(n // 3, 3)) * 30
    r["orig"], r["dir"], r["time"] = o, d, rng.random(n)
    return r
