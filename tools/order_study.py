"""Study (CPU, host-only): how many records does a ray visit under different CHILD ORDERS of the f32 stream's hierarchy?
Parses the stream into a tree and walks the fixture rays with (a) the stream's own order, (b) children ascending along
the axis that separates them, (c) per ray: ascending or descending as a whole, by the majority of the direction's signs,
(d) per node front to back along the separating axis (what a walk with a stack could do).
usage: python tools/order_study.py [scene] [max rays]"""
import math
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402

from raytracing_amd import lib  # noqa: E402
from tests import fixtures as fx  # noqa: E402
from tools.stream_stats import UNITS, K_BOX, K_READY, K_SPHERE, K_MSPHERE, K_AABOX, K_MEDIUM, K_QUAD, slab, sphere_roots  # noqa: E402


class Node:
    __slots__ = ("off", "kind", "lo", "hi", "kids", "axis", "key", "pax")


def leaf_bounds(uf, i, kind):
    u0, u1 = uf[i], uf[i + 1]
    if kind in (K_SPHERE, K_MEDIUM, K_MSPHERE):
        c, r = u0[:3], abs(u0[3])
        lo, hi = c - r, c + r
        if kind == K_MSPHERE:
            d = uf[i + 2][:3]
            lo, hi = np.minimum(lo, lo + d), np.maximum(hi, hi + d)
        return lo, hi
    if kind == K_AABOX:
        return np.array([u0[0], u0[2], u1[0]]), np.array([u0[1], u0[3], u1[1]])
    if kind == K_QUAD:
        Q, u, v = u1[:3], uf[i + 2][:3], uf[i + 3][:3]
        pts = np.array([Q, Q + u, Q + v, Q + u + v])
        return pts.min(0), pts.max(0)
    return None, None


def parse(uf, ui, i, end):
    out = []
    while i < end:
        kind = int(ui[i + 1, 3]) & 15
        if kind == K_READY:
            break
        n = Node()
        n.off, n.kind, n.kids, n.axis = i, kind, [], 0
        if kind == K_BOX:
            link = int(ui[i + 1, 2]) // 16
            n.lo = np.array([uf[i, 0], uf[i, 2], uf[i + 1, 0]])
            n.hi = np.array([uf[i, 1], uf[i, 3], uf[i + 1, 1]])
            n.kids = parse(uf, ui, i + 2, link)
            i = link
        else:
            n.lo, n.hi = leaf_bounds(uf, i, kind)
            i += UNITS[kind]
        out.append(n)
    return out


def hit_leaf(uf, n, o, d, tm, xi, best_t, tmin=0.001):
    i, kind = n.off, n.kind
    u0, u1 = uf[i], uf[i + 1]
    if kind in (K_SPHERE, K_MSPHERE):
        c = u0[:3] + (tm * uf[i + 2][:3] if kind == K_MSPHERE else 0)
        r = sphere_roots(c, float(u0[3]), o, d)
        if r:
            for cand in r:
                if tmin < cand < best_t:
                    return cand
    elif kind == K_AABOX:
        s = slab(n.lo, n.hi, o, d, -math.inf, math.inf)
        if s:
            for cand in s:
                if tmin < cand < best_t:
                    return cand
    elif kind == K_MEDIUM:
        r = sphere_roots(u0[:3], float(u0[3]), o, d)
        if r and r[0] + 0.0001 < r[1]:
            t1, t2 = max(r[0], tmin), min(r[1], best_t)
            if t1 < t2:
                t1 = max(t1, 0.0)
                ln = math.sqrt(d @ d)
                hd = float(u1[0]) * math.log(xi)
                if hd <= (t2 - t1) * ln:
                    return t1 + hd / ln
    elif kind == K_QUAD:
        nn, D, Q = u0[:3], float(u0[3]), u1[:3]
        u, v, w = uf[i + 2][:3], uf[i + 3][:3], uf[i + 4][:3]
        den = nn @ d
        if abs(den) >= 1e-8:
            tt = (D - nn @ o) / den
            if tmin < tt < best_t:
                ph = o + tt * d - Q
                al, be = w @ np.cross(ph, v), w @ np.cross(u, ph)
                if 0 <= al <= 1 and 0 <= be <= 1:
                    return tt
    return None


def walk(uf, nodes, o, d, tm, xi, mode, flip, st):
    """st = [best_t, box steps, leaf steps]"""
    seq = nodes
    if mode == "asc":
        seq = sorted(nodes, key=lambda n: n.key)
    elif mode == "two":
        seq = sorted(nodes, key=lambda n: n.key, reverse=flip)
    elif mode == "ideal":
        ax = nodes[0].pax
        seq = sorted(nodes, key=lambda n: n.key, reverse=d[ax] < 0)
    for n in seq:
        if n.kind == K_BOX:
            st[1] += 1
            if slab(n.lo, n.hi, o, d, 0.001, st[0]):
                walk(uf, n.kids, o, d, tm, xi, mode, flip, st)
        else:
            st[2] += 1
            t = hit_leaf(uf, n, o, d, tm, xi, st[0])
            if t is not None:
                st[0] = t


def annotate(nodes):
    """per sibling list: the axis that separates the siblings' centroids most, key = centroid along it"""
    cs = [(0.5 * (n.lo + n.hi)) if n.lo is not None else np.zeros(3) for n in nodes]
    ext = np.ptp(np.array(cs), axis=0) if len(cs) > 1 else np.zeros(3)
    ax = int(np.argmax(ext))
    for n, c in zip(nodes, cs):
        n.key, n.pax = float(c[ax]), ax
        if n.kids:
            annotate(n.kids)


def main():
    scn = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    cap = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
    sc = fx.scene(scn)
    units, ids = lib.flatten_stream(sc)
    uf = units.view(np.float32).astype(np.float64)
    ui = units.view(np.int32)
    rays, xi, hits = fx.trace(scn)
    top = parse(uf, ui, 0, len(units))
    annotate(top)
    m = min(cap, len(rays))
    for mode in ("stream", "asc", "two", "ideal"):
        box = leaf = 0
        for k in range(m):
            o, d = np.array(rays[k]["orig"], dtype=np.float64), np.array(rays[k]["dir"], dtype=np.float64)
            flip = (d < 0).sum() >= 2
            st = [math.inf, 0, 0]
            walk(uf, top, o, d, float(rays[k]["time"]), float(xi[k]), mode, flip, st)
            box += st[1]
            leaf += st[2]
        print(f"{mode:7s} box {box / m:6.2f} leaf {leaf / m:5.2f} total {(box + leaf) / m:6.2f}")


if __name__ == "__main__":
    main()
