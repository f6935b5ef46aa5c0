"""Walk the f32 record stream on the CPU for the fixture rays of a scene (test infrastructure, double arithmetic on the
float records): steps per ray by record kind, and the closest hit against the fixture's.  Host-only: no GPU.

usage: python tools/stream_stats.py [scene number, default 3] [max rays]        (REF_ORDER=1: the reference's order)
"""
import math
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402

from raytracing_amd import lib  # noqa: E402
from tests import fixtures as fx  # noqa: E402

K_BOX, K_SPHERE, K_AABOX, K_MEDIUM, K_QUAD, K_TRI, K_XF_ENTER, K_XF_EXIT, K_MSPHERE, K_MEDIUM_GEN, K_OBOX, K_READY = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 15
NAMES = {0: "box", 1: "sphere", 2: "aabox", 3: "medium", 4: "quad", 5: "tri", 6: "xf_enter", 7: "xf_exit", 8: "msphere", 9: "medium_gen", 10: "obox"}
UNITS = {K_SPHERE: 2, K_AABOX: 2, K_MEDIUM: 2, K_QUAD: 5, K_TRI: 3, K_XF_ENTER: 2, K_XF_EXIT: 2, K_MSPHERE: 3, K_MEDIUM_GEN: 2, K_OBOX: 4}


def slab(lo, hi, o, d, tmin, tmax):
    for a in range(3):
        inv = 1.0 / d[a] if d[a] != 0 else math.copysign(1e30, d[a])
        t0, t1 = (lo[a] - o[a]) * inv, (hi[a] - o[a]) * inv
        if t0 > t1:
            t0, t1 = t1, t0
        tmin, tmax = max(tmin, t0), min(tmax, t1)
        if tmax < tmin:
            return None
    return tmin, tmax


def sphere_roots(c, r, o, d):
    oc = c - o
    a = d @ d
    h = d @ oc
    disc = h * h - a * (oc @ oc - r * r)
    if disc < 0:
        return None
    s = math.sqrt(disc)
    return (h - s) / a, (h + s) / a


def walk(units_f, units_i, ids, ray, xi, counts, start=0, tmin=0.001):
    o, d, tm = np.array(ray["orig"], dtype=np.float64), np.array(ray["dir"], dtype=np.float64), float(ray["time"])
    best_t, best = math.inf, -1
    stack = []
    i = start
    while True:
        kind = int(units_i[i + 1, 3]) & 15
        if kind == K_READY:
            return best_t, best
        counts[kind] = counts.get(kind, 0) + 1
        u0, u1 = units_f[i], units_f[i + 1]
        if kind == K_BOX:
            hit = slab((u0[0], u0[2], u1[0]), (u0[1], u0[3], u1[1]), o, d, tmin, best_t)
            i = i + 2 if hit else int(units_i[i + 1, 2]) // 16
            continue
        t = None
        if kind in (K_SPHERE, K_MSPHERE):
            c = np.array(u0[:3], dtype=np.float64)
            if kind == K_MSPHERE:
                c = c + tm * np.array(units_f[i + 2][:3], dtype=np.float64)
            r = sphere_roots(c, float(u0[3]), o, d)
            if r:
                for cand in r:
                    if tmin < cand < best_t:
                        t = cand
                        break
        elif kind == K_AABOX:
            s = slab((u0[0], u0[2], u1[0]), (u0[1], u0[3], u1[1]), o, d, -math.inf, math.inf)
            if s:
                for cand in s:
                    if tmin < cand < best_t:
                        t = cand
                        break
        elif kind == K_OBOX:  # translate(rotate_y(box())): the ray moved by -offset and turned, then the slab test
            u2, u3 = units_f[i + 2], units_f[i + 3]
            sn, cs = float(u2[0]), float(u2[1])
            oo = o - np.array([u2[2], u2[3], u3[0]], dtype=np.float64)
            oo = np.array([cs * oo[0] - sn * oo[2], oo[1], sn * oo[0] + cs * oo[2]])
            dd = np.array([cs * d[0] - sn * d[2], d[1], sn * d[0] + cs * d[2]])
            s = slab((u0[0], u0[2], u1[0]), (u0[1], u0[3], u1[1]), oo, dd, -math.inf, math.inf)
            if s:
                for cand in s:
                    if tmin < cand < best_t:
                        t = cand
                        break
        elif kind == K_MEDIUM:
            r = sphere_roots(np.array(u0[:3], dtype=np.float64), float(u0[3]), o, d)
            if r and r[0] + 0.0001 < r[1]:
                t1, t2 = max(r[0], tmin), min(r[1], best_t)
                if t1 < t2:
                    t1 = max(t1, 0.0)
                    ln = math.sqrt(d @ d)
                    hd = float(u1[0]) * math.log(xi)
                    if hd <= (t2 - t1) * ln:
                        t = t1 + hd / ln
        elif kind == K_QUAD:
            n, D, Q = np.array(u0[:3], dtype=np.float64), float(u0[3]), np.array(u1[:3], dtype=np.float64)
            u, v, w = (np.array(units_f[i + k][:3], dtype=np.float64) for k in (2, 3, 4))
            den = n @ d
            if abs(den) >= 1e-8:
                tt = (D - n @ o) / den
                if tmin < tt < best_t:
                    ph = o + tt * d - Q
                    al, be = w @ np.cross(ph, v), w @ np.cross(u, ph)
                    if 0 <= al <= 1 and 0 <= be <= 1:
                        t = tt
        elif kind in (K_XF_ENTER, K_XF_EXIT):
            xk = int(units_i[i + 1, 0])
            if kind == K_XF_ENTER:
                stack.append((o.copy(), d.copy()))
                if xk == 0:  # XF_TRANSLATE
                    o = o - np.array(u0[:3], dtype=np.float64)
                else:
                    s, c = float(u0[0]), float(u0[1])
                    o = np.array([c * o[0] - s * o[2], o[1], s * o[0] + c * o[2]])
                    d = np.array([c * d[0] - s * d[2], d[1], s * d[0] + c * d[2]])
            else:
                o, d = stack.pop()
        elif kind == K_TRI:
            v0 = np.array(u0[:3], dtype=np.float64)
            e1 = np.array([u0[3], u1[0], u1[1]], dtype=np.float64)
            e2 = np.array([u1[2], units_f[i + 2][0], units_f[i + 2][1]], dtype=np.float64)
            h = np.cross(d, e2)
            a = e1 @ h
            if abs(a) >= 1e-6:
                f = 1.0 / a
                s = o - v0
                uu = f * (s @ h)
                q = np.cross(s, e1)
                vv = f * (d @ q)
                tt = f * (e2 @ q)
                if uu >= 0 and vv >= 0 and uu + vv <= 1 and tmin <= tt <= best_t:
                    t = tt
        if t is not None:
            best_t, best = t, i
        i += UNITS[kind]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    cap = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    sc = fx.scene(n)
    units, ids = lib.flatten_stream(sc, reference_order=bool(os.environ.get('REF_ORDER')))
    uf = units.view(np.float32).astype(np.float64)
    ui = units.view(np.int32)
    rays, xi, hits = fx.trace(n)
    k_all = {}
    n_box_records = int(((ui[1:, 3] & 15) == K_BOX).sum())
    print(f"scene {n}: {len(units)} units")
    counts = {}
    wrong = 0
    m = min(cap, len(rays))
    for k in range(m):
        t, off = walk(uf, ui, ids, rays[k], float(xi[k]), counts)
        prim = int(ids[off]) if off >= 0 else -1
        want = int(hits["prim"][k])
        if prim != want and (ui[off + 1, 3] & 15) not in (K_AABOX, K_OBOX):  # (an AABOX record stands for six quads)
            if not (want >= 0 and prim >= 0 and abs(t - float(hits["t"][k])) <= 1e-4 * abs(t)):
                wrong += 1
    print(f"{m} rays; steps per ray: " + ", ".join(f"{NAMES[k]} {v / m:.2f}" for k, v in sorted(counts.items())) +
          f"; total {sum(counts.values()) / m:.2f}; leaf {sum(v for k, v in counts.items() if k) / m:.2f}; id mismatches {wrong}")


if __name__ == "__main__":
    main()
