// rt_kernels.h -- HIP device code of the path tracer (gfx950), templated on the arithmetic type R.
//
//   R = float  : production arithmetic (rt_kernels_f32.hip)
//   R = double : reference arithmetic, compiled with -ffp-contract=off (rt_kernels_f64.hip); every expression keeps
//                the operation order of the reference C++ so that closest-hit results are bit-identical to it.
//
// Reference functions restated here (src/cpu/...):
//   aabb::hit aabb.h:36 | bvh_node::hit bvh.h:43 | hittable_list::hit hittable_list.h:24 | translate/rotate_y::hit
//   hittable.h:53,110 | constant_medium::hit constant_medium.h:21 | sphere::hit/pdf_value/random sphere.h:29-108 |
//   quad::hit/pdf_value/random quad.h:32-89 | triangle::hit triangle.h:23 | materials material.h:39-160 |
//   textures texture.h:16-92, perlin.h:16-92 | pdfs pdf.h:17-89, onb.h:7 | camera::get_ray/ray_color camera.h:249-326 |
//   write_color color.h:20.
#ifndef RT_KERNELS_H
#define RT_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_amd.h"
#include "rt_scene.h"

namespace rtamd {

#define RT_DEV __device__ __forceinline__

// ------------------------------------------------------------------------------------------------ scalar helpers
RT_DEV float m_sqrt(float x) { return sqrtf(x); }
RT_DEV double m_sqrt(double x) { return sqrt(x); }
RT_DEV float m_sin(float x) { return sinf(x); }
RT_DEV double m_sin(double x) { return sin(x); }
RT_DEV float m_cos(float x) { return cosf(x); }
RT_DEV double m_cos(double x) { return cos(x); }
RT_DEV float m_acos(float x) { return acosf(x); }
RT_DEV double m_acos(double x) { return acos(x); }
RT_DEV float m_atan2(float y, float x) { return atan2f(y, x); }
RT_DEV double m_atan2(double y, double x) { return atan2(y, x); }
RT_DEV float m_log(float x) { return logf(x); }
RT_DEV double m_log(double x) { return log(x); }
RT_DEV float m_floor(float x) { return floorf(x); }
RT_DEV double m_floor(double x) { return floor(x); }
RT_DEV float m_abs(float x) { return fabsf(x); }
RT_DEV double m_abs(double x) { return fabs(x); }
RT_DEV float m_min(float a, float b) { return fminf(a, b); }
RT_DEV double m_min(double a, double b) { return fmin(a, b); }
RT_DEV float m_max(float a, float b) { return fmaxf(a, b); }
RT_DEV double m_max(double a, double b) { return fmax(a, b); }
RT_DEV float m_pow5(float x) { float x2 = x * x; return x2 * x2 * x; }
RT_DEV double m_pow5(double x) { return pow(x, 5.0); } // material.h:119 std::pow((1 - cosine),5)

template <class R> struct Const;
template <> struct Const<float> {
    static constexpr float inf = __builtin_huge_valf();
    static constexpr float pi = 3.14159265358979323846f;
};
template <> struct Const<double> {
    static constexpr double inf = __builtin_huge_val();
    static constexpr double pi = 3.1415926535897932385; // utils.h:20
};

// ------------------------------------------------------------------------------------------------ vec3.h
template <class R> struct V3 {
    R x, y, z;
};
template <class R> RT_DEV V3<R> mk(R a, R b, R c) { return V3<R>{a, b, c}; }
template <class R> RT_DEV V3<R> ld3(const R *p) { return V3<R>{p[0], p[1], p[2]}; }
template <class R> RT_DEV V3<R> operator+(V3<R> a, V3<R> b) { return mk<R>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <class R> RT_DEV V3<R> operator-(V3<R> a, V3<R> b) { return mk<R>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <class R> RT_DEV V3<R> operator-(V3<R> a) { return mk<R>(-a.x, -a.y, -a.z); }
template <class R> RT_DEV V3<R> operator*(V3<R> a, V3<R> b) { return mk<R>(a.x * b.x, a.y * b.y, a.z * b.z); }
template <class R> RT_DEV V3<R> scale(R t, V3<R> v) { return mk<R>(t * v.x, t * v.y, t * v.z); }           // vec3.h:78
template <class R> RT_DEV V3<R> divide(V3<R> v, R t) { return scale<R>(R(1) / t, v); }                      // vec3.h:86
template <class R> RT_DEV R dot(V3<R> u, V3<R> v) { return u.x * v.x + u.y * v.y + u.z * v.z; }             // vec3.h:90
template <class R> RT_DEV V3<R> cross(V3<R> u, V3<R> v) {                                                   // vec3.h:96
    return mk<R>(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
template <class R> RT_DEV R len2(V3<R> v) { return v.x * v.x + v.y * v.y + v.z * v.z; }                     // vec3.h:41
template <class R> RT_DEV R len(V3<R> v) { return m_sqrt(len2(v)); }
template <class R> RT_DEV V3<R> unit(V3<R> v) { return divide<R>(v, len(v)); }                              // vec3.h:102
template <class R> RT_DEV V3<R> reflect(V3<R> v, V3<R> n) { return v - scale<R>(R(2) * dot(v, n), n); }     // vec3.h:131
template <class R> RT_DEV V3<R> refract(V3<R> uv, V3<R> n, R etai_over_etat) {                              // vec3.h:135
    R cos_theta = m_min(dot(-uv, n), R(1));
    V3<R> r_out_perp = scale<R>(etai_over_etat, uv + scale<R>(cos_theta, n));
    V3<R> r_out_parallel = scale<R>(-m_sqrt(m_abs(R(1) - len2(r_out_perp))), n);
    return r_out_perp + r_out_parallel;
}

template <class R> struct RayT { // ray.h:7 (direction is NOT normalised)
    V3<R> o, d;
    R tm;
};
template <class R> RT_DEV V3<R> ray_at(const RayT<R> &r, R t) { return r.o + scale<R>(t, r.d); } // ray.h:22

// ------------------------------------------------------------------------------------------------ RNG
// Philox4x32-10, keyed by the 64-bit seed; counter = (block, sample, pixel, stream).  Identical to
// oracle/rt_oracle.cpp's so that f64 renders can be compared path by path.
RT_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t &o0,
                          uint32_t &o1, uint32_t &o2, uint32_t &o3) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}

template <class R> RT_DEV R u01(uint32_t x);
template <> RT_DEV double u01<double>(uint32_t x) { return x * (1.0 / 4294967296.0); }
template <> RT_DEV float u01<float>(uint32_t x) { return (x >> 8) * (1.0f / 16777216.0f); }

template <class R> struct Rng {
    uint32_t k0, k1, pixel, sample, block;
    uint32_t b0, b1, b2, b3;
    int have;
    RT_DEV void begin(uint32_t px, uint32_t s) { pixel = px; sample = s; block = 0; have = 0; }
    RT_DEV R next() { // utils.h:33 random_double()
        if (have == 0) {
            philox4x32_10(block, sample, pixel, 0u, k0, k1, b0, b1, b2, b3);
            block++;
            have = 4;
        }
        uint32_t x = b0;
        b0 = b1; b1 = b2; b2 = b3;
        have--;
        return u01<R>(x);
    }
    RT_DEV R next(R mn, R mx) { return mn + (mx - mn) * next(); } // utils.h:38
    RT_DEV R medium(int node_id, int bounce) const {              // keyed draw for constant_medium.h:41
        uint32_t o0, o1, o2, o3;
        philox4x32_10((uint32_t)node_id, sample, pixel, 1u + (uint32_t)bounce, k0, k1, o0, o1, o2, o3);
        return u01<R>(o0);
    }
};

// ------------------------------------------------------------------------------------------------ intersection
template <class R> RT_DEV bool box_hit(const NodeT<R> &nd, const RayT<R> &r, V3<R> inv, R tmin, R tmax) { // aabb.h:36
    {
        R t0 = (nd.bmin[0] - r.o.x) * inv.x, t1 = (nd.bmax[0] - r.o.x) * inv.x;
        if (t0 < t1) { if (t0 > tmin) tmin = t0; if (t1 < tmax) tmax = t1; }
        else { if (t1 > tmin) tmin = t1; if (t0 < tmax) tmax = t0; }
        if (tmax <= tmin) return false;
    }
    {
        R t0 = (nd.bmin[1] - r.o.y) * inv.y, t1 = (nd.bmax[1] - r.o.y) * inv.y;
        if (t0 < t1) { if (t0 > tmin) tmin = t0; if (t1 < tmax) tmax = t1; }
        else { if (t1 > tmin) tmin = t1; if (t0 < tmax) tmax = t0; }
        if (tmax <= tmin) return false;
    }
    {
        R t0 = (nd.bmin[2] - r.o.z) * inv.z, t1 = (nd.bmax[2] - r.o.z) * inv.z;
        if (t0 < t1) { if (t0 > tmin) tmin = t0; if (t1 < tmax) tmax = t1; }
        else { if (t1 > tmin) tmin = t1; if (t0 < tmax) tmax = t0; }
        if (tmax <= tmin) return false;
    }
    return true;
}

template <class R> RT_DEV bool sphere_t(const SphereT<R> &s, const RayT<R> &r, R tmin, R tmax, R &t_out) { // sphere.h:29
    V3<R> cc = ld3(s.c) + scale<R>(r.tm, ld3(s.d));
    V3<R> oc = cc - r.o;
    R a = len2(r.d);
    R h = dot(r.d, oc);
    R c = len2(oc) - s.radius * s.radius;
    if constexpr (sizeof(R) == 4) {
        // Same quadratic, conditioned for float: h*h - a*c cancels catastrophically when |oc| >> radius (first_scene's
        // r=1 light 1000 units away, the r=1000 ground, final_scene's r=5000 fog).  disc = a*(r^2 - |oc - (h/a)d|^2) and
        // the roots c/q, q/a with q = h + sign(h)*sqrt(disc) are algebraically identical and lose no digits.
        R inva = R(1) / a;
        V3<R> perp = oc - scale<R>(h * inva, r.d);
        R disc = a * (s.radius * s.radius - len2(perp));
        if (disc < R(0)) return false;
        R sq = m_sqrt(disc);
        R q = h + (h < R(0) ? -sq : sq);
        R ta = c / q, tb = q * inva;
        R t0 = m_min(ta, tb), t1 = m_max(ta, tb);
        if (q == R(0)) { t0 = R(0); t1 = R(0); }
        R root = t0;
        if (!(tmin < root && root < tmax)) {
            root = t1;
            if (!(tmin < root && root < tmax)) return false;
        }
        t_out = root;
        return true;
    } else {
        R disc = h * h - a * c;
        if (disc < R(0)) return false;
        R sq = m_sqrt(disc);
        R root = (h - sq) / a;
        if (!(tmin < root && root < tmax)) { // interval::surrounds
            root = (h + sq) / a;
            if (!(tmin < root && root < tmax)) return false;
        }
        t_out = root;
        return true;
    }
}

template <class R>
RT_DEV bool quad_t(const QuadT<R> &q, const RayT<R> &r, R tmin, R tmax, R &t_out, R &alpha, R &beta) { // quad.h:32
    V3<R> n = ld3(q.n);
    R denom = dot(n, r.d);
    if (m_abs(denom) < R(1e-8)) return false;
    R t = (q.D - dot(n, r.o)) / denom;
    if (!(tmin <= t && t <= tmax)) return false; // interval::contains
    V3<R> ph = ray_at(r, t) - ld3(q.Q);
    V3<R> w = ld3(q.w);
    alpha = dot(w, cross(ph, ld3(q.v)));
    beta = dot(w, cross(ld3(q.u), ph));
    if (!(R(0) <= alpha && alpha <= R(1)) || !(R(0) <= beta && beta <= R(1))) return false; // quad.h:62
    t_out = t;
    return true;
}

template <class R>
RT_DEV bool tri_t(const TriT<R> &tr, const RayT<R> &r, R tmin, R tmax, R &t_out, float &u_out, float &v_out) { // triangle.h:23
    const float EPSILON = 0.000001; // triangle.h:16 (const float): a, f, u, v, t are float in the reference
    V3<R> v0 = ld3(tr.v0);
    V3<R> edge1 = ld3(tr.v1) - v0;
    V3<R> edge2 = ld3(tr.v2) - v0;
    V3<R> h = cross(r.d, edge2);
    float a = (float)dot(edge1, h);
    if (tr.cull && a < EPSILON) return false;
    if (fabsf(a) < EPSILON) return false;
    float f = (float)(R(1.0) / R(a));
    V3<R> s = r.o - v0;
    float u = (float)(R(f) * dot(s, h));
    if (u < 0.0f || u > 1.0f) return false;
    V3<R> q = cross(s, edge1);
    float v = (float)(R(f) * dot(r.d, q));
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = (float)(R(f) * dot(edge2, q));
    if (!(tmin <= R(t) && R(t) <= tmax)) return false;
    t_out = R(t);
    u_out = u;
    v_out = v;
    return true;
}

template <class R, class XiFn>
RT_DEV bool medium_t(const SceneT<R> &sc, const MediumT<R> &m, const RayT<R> &r, R tmin, R tmax, XiFn xi_fn, R &t_out) {
    // constant_medium.h:21 (boundary is a sphere); the uniform is drawn only where the reference draws it (line 41)
    const SphereT<R> s = sc.spheres[m.sphere];
    R t1, t2;
    if (!sphere_t(s, r, -Const<R>::inf, Const<R>::inf, t1)) return false;
    if (!sphere_t(s, r, t1 + R(0.0001), Const<R>::inf, t2)) return false;
    if (t1 < tmin) t1 = tmin;
    if (t2 > tmax) t2 = tmax;
    if (t1 >= t2) return false;
    if (t1 < R(0)) t1 = R(0);
    R ray_length = len(r.d);
    R distance_inside_boundary = (t2 - t1) * ray_length;
    R hit_distance = m.neg_inv_density * m_log(xi_fn(m.node_id));
    if (hit_distance > distance_inside_boundary) return false;
    t_out = t1 + hit_distance / ray_length;
    return true;
}

template <class R> RT_DEV RayT<R> xf_apply(const XformT<R> &x, const RayT<R> &r) {
    RayT<R> o;
    o.tm = r.tm;
    if (x.kind == XF_TRANSLATE) { // hittable.h:55
        o.o = r.o - mk<R>(x.p[0], x.p[1], x.p[2]);
        o.d = r.d;
    } else { // hittable.h:114-124
        R s = x.p[0], c = x.p[1];
        o.o = mk<R>((c * r.o.x) - (s * r.o.z), r.o.y, (s * r.o.x) + (c * r.o.z));
        o.d = mk<R>((c * r.d.x) - (s * r.d.z), r.d.y, (s * r.d.x) + (c * r.d.z));
    }
    return o;
}

template <class R> RT_DEV RayT<R> ray_in_ctx(const SceneT<R> &sc, const RayT<R> &world, int ctx) {
    RayT<R> r = world;
    if (ctx != 0) {
        const int depth = sc.xforms[ctx].depth;
        for (int k = 0; k < depth; k++) r = xf_apply(sc.xforms[sc.xforms[ctx].chain[k]], r);
    }
    return r;
}

template <class R> RT_DEV V3<R> inv_dir(V3<R> d) { return mk<R>(R(1.0) / d.x, R(1.0) / d.y, R(1.0) / d.z); } // aabb.h:42

template <class R> struct Best {
    R t;
    int meta; // -1 = miss
    int ctx;
};

// world.hit(r, interval(tmin, tmax), rec): threaded pre-order walk == the reference's recursion order.
template <class R, class MediumXi>
RT_DEV Best<R> traverse(const SceneT<R> &sc, const RayT<R> &world, R tmin, R tmax, MediumXi medium_xi) {
    Best<R> best;
    best.t = tmax;
    best.meta = -1;
    best.ctx = 0;
    RayT<R> ray = world;
    V3<R> inv = inv_dir(ray.d);
    int ctx = 0;
    int i = 0;
    const int n = sc.n_nodes;
    while (i < n) {
        const NodeT<R> nd = sc.nodes[i];
        const int kind = nd.meta & 15, idx = nd.meta >> 4;
        if (kind == K_BOX) {
            i = box_hit(nd, ray, inv, tmin, best.t) ? i + 1 : nd.skip;
            continue;
        }
        i++;
        if (kind == K_SPHERE) {
            R t;
            if (sphere_t(sc.spheres[idx], ray, tmin, best.t, t)) { best.t = t; best.meta = nd.meta; best.ctx = ctx; }
        } else if (kind == K_QUAD) {
            R t, al, be;
            if (quad_t(sc.quads[idx], ray, tmin, best.t, t, al, be)) { best.t = t; best.meta = nd.meta; best.ctx = ctx; }
        } else if (kind == K_TRI) {
            R t;
            float u, v;
            if (tri_t(sc.tris[idx], ray, tmin, best.t, t, u, v)) { best.t = t; best.meta = nd.meta; best.ctx = ctx; }
        } else if (kind == K_MEDIUM) {
            const MediumT<R> m = sc.media[idx];
            R t;
            if (medium_t(sc, m, ray, tmin, best.t, medium_xi, t)) { best.t = t; best.meta = nd.meta; best.ctx = ctx; }
        } else if (kind == K_XF_ENTER) {
            ray = xf_apply(sc.xforms[idx], ray);
            inv = inv_dir(ray.d);
            ctx = idx;
        } else { // K_XF_EXIT
            ctx = sc.xforms[idx].parent;
            ray = ray_in_ctx(sc, world, ctx);
            inv = inv_dir(ray.d);
        }
    }
    return best;
}

template <class R> struct Hit { // hittable.h:10 hit_record
    V3<R> p, n;
    R t, u, v;
    int mat;
    int kind, idx;
    bool front;
};

template <class R> RT_DEV void sphere_uv(V3<R> p, R &u, R &v) { // sphere.h:89
    R ny = -p.y;
    if constexpr (sizeof(R) == 4) ny = m_max(R(-1), m_min(R(1), ny)); // (p-c)/r can exceed 1 by an ulp in float
    R theta = m_acos(ny);
    R phi = m_atan2(-p.z, p.x) + Const<R>::pi;
    u = phi / (R(2) * Const<R>::pi);
    v = theta / Const<R>::pi;
}

// Re-derive the full hit record of the winning primitive (same arithmetic as the reference's success path).
template <class R> RT_DEV Hit<R> finalize(const SceneT<R> &sc, const RayT<R> &world, const Best<R> &b, bool want_uv) {
    Hit<R> h;
    h.t = b.t;
    h.kind = b.meta & 15;
    h.idx = b.meta >> 4;
    h.u = R(0);
    h.v = R(0);
    const RayT<R> r = ray_in_ctx(sc, world, b.ctx);
    h.p = ray_at(r, b.t);
    if (h.kind == K_SPHERE) {
        const SphereT<R> s = sc.spheres[h.idx];
        V3<R> cc = ld3(s.c) + scale<R>(r.tm, ld3(s.d));
        V3<R> outward = divide<R>(h.p - cc, s.radius);
        h.front = dot(r.d, outward) < R(0);
        h.n = h.front ? outward : -outward;
        h.mat = s.mat;
        if (want_uv || sc.mats[s.mat].needs_uv) sphere_uv(outward, h.u, h.v);
    } else if (h.kind == K_QUAD) {
        const QuadT<R> q = sc.quads[h.idx];
        V3<R> ph = h.p - ld3(q.Q);
        V3<R> w = ld3(q.w);
        h.u = dot(w, cross(ph, ld3(q.v)));
        h.v = dot(w, cross(ld3(q.u), ph));
        V3<R> n = ld3(q.n);
        h.front = dot(r.d, n) < R(0);
        h.n = h.front ? n : -n;
        h.mat = q.mat;
    } else if (h.kind == K_TRI) {
        const TriT<R> tr = sc.tris[h.idx];
        R t;
        float u = 0.f, v = 0.f;
        tri_t(tr, r, -Const<R>::inf, Const<R>::inf, t, u, v);
        h.u = R(u);
        h.v = R(v);
        V3<R> n = ld3(tr.n);
        h.front = dot(r.d, n) < R(0);
        h.n = h.front ? n : -n;
        h.mat = tr.mat;
    } else { // K_MEDIUM, constant_medium.h:46-51
        h.n = mk<R>(R(1), R(0), R(0));
        h.front = true;
        h.mat = sc.media[h.idx].mat;
    }
    if (b.ctx != 0) { // leave object space, innermost transform first (hittable.h:62, 135-145)
        const XformT<R> X = sc.xforms[b.ctx];
        for (int k = X.depth - 1; k >= 0; k--) {
            const XformT<R> x = sc.xforms[X.chain[k]];
            if (x.kind == XF_TRANSLATE) {
                h.p = h.p + mk<R>(x.p[0], x.p[1], x.p[2]);
            } else {
                R s = x.p[0], c = x.p[1];
                h.p = mk<R>((c * h.p.x) + (s * h.p.z), h.p.y, (-s * h.p.x) + (c * h.p.z));
                h.n = mk<R>((c * h.n.x) + (s * h.n.z), h.n.y, (-s * h.n.x) + (c * h.n.z));
            }
        }
    }
    return h;
}

// ------------------------------------------------------------------------------------------------ textures
template <class R> RT_DEV R perlin_noise(const PerlinT<R> &pn, V3<R> p) { // perlin.h:16, 75
    R fx = m_floor(p.x), fy = m_floor(p.y), fz = m_floor(p.z);
    R u = p.x - fx, v = p.y - fy, w = p.z - fz;
    int i = int(fx), j = int(fy), k = int(fz);
    R uu = u * u * (R(3) - R(2) * u);
    R vv = v * v * (R(3) - R(2) * v);
    R ww = w * w * (R(3) - R(2) * w);
    R accum = R(0);
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int d = 0; d < 2; d++) {
                int g = pn.perm_x[(i + a) & 255] ^ pn.perm_y[(j + b) & 255] ^ pn.perm_z[(k + d) & 255];
                V3<R> c = ld3(pn.randvec[g]);
                V3<R> wv = mk<R>(u - a, v - b, w - d);
                accum += (a * uu + (1 - a) * (R(1) - uu)) * (b * vv + (1 - b) * (R(1) - vv)) *
                         (d * ww + (1 - d) * (R(1) - ww)) * dot(c, wv);
            }
    return accum;
}

template <class R> RT_DEV R perlin_turb(const PerlinT<R> &pn, V3<R> p, int depth) { // perlin.h:38
    R accum = R(0);
    R weight = R(1);
    for (int i = 0; i < depth; i++) {
        accum += weight * perlin_noise(pn, p);
        weight *= R(0.5);
        p = mk<R>(p.x * R(2), p.y * R(2), p.z * R(2));
    }
    return m_abs(accum);
}

template <class R> RT_DEV V3<R> tex_value(const SceneT<R> &sc, int tex, R u, R v, V3<R> p) {
    TexT<R> t = sc.texs[tex];
    for (int guard = 0; guard < 16 && t.kind == RT_TEX_CHECKER; guard++) { // texture.h:39
        int xi = int(m_floor(t.scale * p.x));
        int yi = int(m_floor(t.scale * p.y));
        int zi = int(m_floor(t.scale * p.z));
        bool even = (xi + yi + zi) % 2 == 0;
        t = sc.texs[even ? t.a : t.b];
    }
    if (t.kind == RT_TEX_SOLID) return ld3(t.color); // texture.h:22
    if (t.kind == RT_TEX_IMAGE) {                    // texture.h:60
        if (t.a < 0) return mk<R>(R(0), R(1), R(1));
        const ImageT im = sc.images[t.a];
        if (im.height <= 0) return mk<R>(R(0), R(1), R(1));
        u = u < R(0) ? R(0) : (u > R(1) ? R(1) : u);
        v = R(1.0) - (v < R(0) ? R(0) : (v > R(1) ? R(1) : v));
        int i = int(u * im.width);
        int j = int(v * im.height);
        i = i < 0 ? 0 : (i < im.width ? i : im.width - 1);
        j = j < 0 ? 0 : (j < im.height ? j : im.height - 1);
        const uint8_t *px = sc.image_bytes + im.offset + (size_t)j * (size_t)(im.width * 3) + (size_t)i * 3;
        R cs = R(1.0) / R(255.0);
        return mk<R>(cs * px[0], cs * px[1], cs * px[2]);
    }
    if (t.kind == RT_TEX_NOISE) { // texture.h:85
        R s = R(1) + m_sin(t.scale * p.z + R(10) * perlin_turb(sc.perlins[t.a], p, 7));
        return mk<R>(R(.5) * s, R(.5) * s, R(.5) * s);
    }
    return mk<R>(R(0), R(0), R(0));
}

// ------------------------------------------------------------------------------------------------ sampling
template <class R> struct Onb { // onb.h:5
    V3<R> u, v, w;
    RT_DEV explicit Onb(V3<R> n) {
        w = unit(n);
        V3<R> a = (m_abs(w.x) > R(0.9)) ? mk<R>(R(0), R(1), R(0)) : mk<R>(R(1), R(0), R(0));
        v = unit(cross(w, a));
        u = cross(w, v);
    }
    RT_DEV V3<R> transform(V3<R> q) const { return scale<R>(q.x, u) + scale<R>(q.y, v) + scale<R>(q.z, w); }
};

// Draw orders below follow oracle/rt_oracle.cpp (which is pinned to the compiled reference): vec3::random(-1,1)
// consumes z, y, x; random_in_unit_disk y, x; quad::random v then u.
template <class R> RT_DEV V3<R> random_unit_vector(Rng<R> &g) { // vec3.h:114
    while (true) {
        R z = g.next(R(-1), R(1));
        R y = g.next(R(-1), R(1));
        R x = g.next(R(-1), R(1));
        V3<R> p = mk<R>(x, y, z);
        R lensq = len2(p);
        if (R(1e-160) < lensq && lensq <= R(1.0)) return divide<R>(p, m_sqrt(lensq));
    }
}
template <> RT_DEV V3<float> random_unit_vector<float>(Rng<float> &g) {
    while (true) {
        float z = g.next(-1.f, 1.f);
        float y = g.next(-1.f, 1.f);
        float x = g.next(-1.f, 1.f);
        V3<float> p = mk<float>(x, y, z);
        float lensq = len2(p);
        if (1e-30f < lensq && lensq <= 1.0f) return divide<float>(p, m_sqrt(lensq)); // 1e-160 underflows float
    }
}

template <class R> RT_DEV V3<R> random_cosine_direction(Rng<R> &g) { // vec3.h:142
    R r1 = g.next();
    R r2 = g.next();
    R phi = R(2) * Const<R>::pi * r1;
    R sr = m_sqrt(r2);
    return mk<R>(m_cos(phi) * sr, m_sin(phi) * sr, m_sqrt(R(1) - r2));
}

// `generated` = this light produced `dir` (light_random picked it), so in exact arithmetic the direction hits it.  In
// float the containment test of a direction drawn on the rim of the sphere's cone / the quad's edge can fail by
// rounding, which would make the mixture pdf 0 and the sample 0/0; the float path therefore skips the test for the
// generating light.  The double path always runs the reference's test.
template <class R> RT_DEV R light_pdf_one(const LightT<R> &L, V3<R> origin, V3<R> dir, bool generated) {
    RayT<R> r{origin, dir, R(0)}; // 2-argument ray ctor: time 0 (ray.h:14)
    const bool skip_test = (sizeof(R) == 4) && generated;
    if (L.kind == LIGHT_SPHERE) { // sphere.h:62
        R t;
        if (!skip_test && !sphere_t(L.s, r, R(0.001), Const<R>::inf, t)) return R(0);
        R dist_squared = len2(ld3(L.s.c) - origin);
        R x = L.s.radius * L.s.radius / dist_squared;
        if constexpr (sizeof(R) == 4) {
            // an origin ON the sphere (the floor point a sphere rests on) lands a few ulp inside it in float: x = 1 + eps
            // would give sqrt(<0).  Genuinely interior origins (x well above 1) keep the reference's NaN (sphere.h:70).
            if (x > R(1) && x < R(1.0001)) x = R(1);
        }
        R cos_theta_max = m_sqrt(R(1) - x);
        R one_minus_cos = R(1) - cos_theta_max;
        if constexpr (sizeof(R) == 4) one_minus_cos = x / (R(1) + cos_theta_max); // same value, no cancellation in float
        R solid_angle = R(2) * Const<R>::pi * one_minus_cos;
        return R(1) / solid_angle;
    }
    if (L.kind == LIGHT_QUAD) { // quad.h:75
        R t, al, be;
        if (skip_test) {
            V3<R> n = ld3(L.q.n);
            R denom = dot(n, dir);
            if (m_abs(denom) < R(1e-8)) return R(0);
            t = (L.q.D - dot(n, origin)) / denom;
            if (!(t >= R(0.001))) return R(0);
        } else {
            if (!quad_t(L.q, r, R(0.001), Const<R>::inf, t, al, be)) return R(0);
        }
        R distance_squared = t * t * len2(dir);
        R cosine = m_abs(dot(dir, ld3(L.q.n)) / len(dir)); // |.| makes the face flip of rec.normal irrelevant
        return distance_squared / (cosine * L.q.area);
    }
    return R(0); // hittable.h:35
}

// `picked` = index of the light that generated `dir`, or -1 when the direction came from the BRDF sampler.
template <class R> RT_DEV R light_pdf(const SceneT<R> &sc, V3<R> origin, V3<R> dir, int picked) {
    if (!sc.lights_is_list) return light_pdf_one(sc.lights[0], origin, dir, picked == 0);
    R weight = R(1.0) / R(sc.n_lights); // hittable_list.h:42
    R sum = R(0);
    for (int k = 0; k < sc.n_lights; k++) sum += weight * light_pdf_one(sc.lights[k], origin, dir, picked == k);
    return sum;
}

template <class R> RT_DEV V3<R> light_random(const SceneT<R> &sc, V3<R> origin, Rng<R> &g, int &pick) {
    pick = 0;
    if (sc.lights_is_list) { // hittable_list.h:52: random_int(0, n-1) = int(0 + n*xi)
        pick = int(R(0) + (R(sc.n_lights) - R(0)) * g.next());
        if (pick >= sc.n_lights) pick = sc.n_lights - 1; // f32 rounding guard (xi < 1 always in f64)
    }
    const LightT<R> &L = sc.lights[pick];
    if (L.kind == LIGHT_SPHERE) { // sphere.h:76, 98
        V3<R> direction = ld3(L.s.c) - origin;
        R distance_squared = len2(direction);
        Onb<R> uvw(direction);
        R r1 = g.next();
        R r2 = g.next();
        R x = L.s.radius * L.s.radius / distance_squared;
        if constexpr (sizeof(R) == 4) {
            if (x > R(1) && x < R(1.0001)) x = R(1); // see light_pdf_one
        }
        R cm = m_sqrt(R(1) - x);
        R z = R(1) + r2 * (cm - R(1));
        R phi = R(2) * Const<R>::pi * r1;
        R s = m_sqrt(R(1) - z * z);
        if constexpr (sizeof(R) == 4) { // 1-z = r2*x/(1+cm) exactly; sqrt(1-z^2) = sqrt((1-z)(1+z)) keeps the small cone angle
            R omz = r2 * x / (R(1) + cm);
            z = R(1) - omz;
            s = m_sqrt(omz * (R(1) + z));
        }
        return uvw.transform(mk<R>(m_cos(phi) * s, m_sin(phi) * s, z));
    }
    if (L.kind == LIGHT_QUAD) { // quad.h:86
        R rv = g.next();
        R ru = g.next();
        V3<R> p = ld3(L.q.Q) + scale<R>(ru, ld3(L.q.u)) + scale<R>(rv, ld3(L.q.v));
        return p - origin;
    }
    return mk<R>(R(1), R(0), R(0)); // hittable.h:39
}

// ------------------------------------------------------------------------------------------------ camera
template <class R> RT_DEV RayT<R> get_ray(const CameraT<R> &cam, int i, int j, int s_i, int s_j, Rng<R> &g) { // camera.h:249
    R px = ((s_i + g.next()) * cam.recip_sqrt_spp) - R(0.5);
    R py = ((s_j + g.next()) * cam.recip_sqrt_spp) - R(0.5);
    V3<R> pixel_sample = ld3(cam.pixel00) + scale<R>(R(i) + px, ld3(cam.du)) + scale<R>(R(j) + py, ld3(cam.dv));
    V3<R> origin = ld3(cam.center);
    if (cam.defocus_angle > R(0)) { // camera.h:285, vec3.h:106
        R dx, dy;
        while (true) {
            dy = g.next(R(-1), R(1));
            dx = g.next(R(-1), R(1));
            if (dx * dx + dy * dy + R(0) * R(0) < R(1)) break;
        }
        origin = ld3(cam.center) + scale<R>(dx, ld3(cam.ddu)) + scale<R>(dy, ld3(cam.ddv));
    }
    RayT<R> r;
    r.o = origin;
    r.d = pixel_sample - origin;
    r.tm = g.next();
    return r;
}

// One bounce of camera::ray_color (camera.h:291-326), written iteratively: radiance L += beta * (emitted | background),
// beta *= attenuation * scattering_pdf / pdf.  Returns false when the path ends.
template <class R>
RT_DEV bool bounce(const SceneT<R> &sc, const CameraT<R> &cam, RayT<R> &ray, V3<R> &beta, V3<R> &L, Rng<R> &g, int bounce_idx) {
    auto mxi = [&](int node_id) { return g.medium(node_id, bounce_idx); };
    Best<R> b = traverse(sc, ray, R(0.001), Const<R>::inf, mxi);
    if (b.meta < 0) { // camera.h:300
        L = L + beta * ld3(cam.background);
        return false;
    }
    // which material? (needed before finalize to know whether u,v matter)
    Hit<R> h = finalize(sc, ray, b, false);
    const MatT<R> m = sc.mats[h.mat];
    V3<R> emitted = mk<R>(R(0), R(0), R(0));
    if (m.kind == RT_MAT_DIFFUSE_LIGHT) { // material.h:129
        if (h.front) emitted = tex_value(sc, m.tex, h.u, h.v, h.p);
        L = L + beta * emitted;
        return false; // material.h:28: no scatter
    }
    L = L + beta * emitted; // keeps inf*0 -> NaN semantics of the recursive form
    if (m.kind == RT_MAT_METAL) { // material.h:66
        V3<R> reflected = reflect(ray.d, h.n);
        V3<R> ruv = random_unit_vector(g);
        reflected = unit(reflected) + scale<R>(m.fuzz, ruv);
        beta = beta * ld3(m.albedo);
        ray.o = h.p;
        ray.d = reflected;
        return true;
    }
    if (m.kind == RT_MAT_DIELECTRIC) { // material.h:88
        R ri = h.front ? (R(1.0) / m.ior) : m.ior;
        V3<R> ud = unit(ray.d);
        R cos_theta = m_min(dot(-ud, h.n), R(1.0));
        R sin_theta = m_sqrt(R(1.0) - cos_theta * cos_theta);
        bool cannot_refract = ri * sin_theta > R(1.0);
        bool do_reflect = cannot_refract;
        if (!do_reflect) {
            R r0 = (R(1) - ri) / (R(1) + ri);
            r0 = r0 * r0;
            R refl = r0 + (R(1) - r0) * m_pow5(R(1) - cos_theta);
            do_reflect = refl > g.next();
        }
        V3<R> dir = do_reflect ? reflect(ud, h.n) : refract(ud, h.n, ri);
        // attenuation (1,1,1): beta unchanged up to a multiplication by one
        ray.o = h.p;
        ray.d = dir;
        return true;
    }
    // lambertian / isotropic: mixture of light sampling and BRDF sampling (camera.h:313-325, pdf.h:69-85)
    const bool lambert = (m.kind == RT_MAT_LAMBERTIAN);
    V3<R> atten = tex_value(sc, m.tex, h.u, h.v, h.p);
    V3<R> dir;
    int picked = -1;
    if (g.next() < R(0.5)) dir = light_random(sc, h.p, g, picked);
    else if (lambert) dir = Onb<R>(h.n).transform(random_cosine_direction(g)); // pdf.h:40
    else dir = random_unit_vector(g);                                          // pdf.h:25
    R v0 = light_pdf(sc, h.p, dir, picked);
    R v1, spdf;
    if (lambert) {
        R cosine_theta = dot(unit(dir), unit(h.n)); // cosine_pdf::value pdf.h:35 (onb re-normalises w)
        v1 = m_max(R(0), cosine_theta / Const<R>::pi);
        R cos_theta = dot(h.n, unit(dir)); // lambertian::scattering_pdf material.h:51
        spdf = cos_theta < R(0) ? R(0) : cos_theta / Const<R>::pi;
    } else {
        v1 = R(1) / (R(4) * Const<R>::pi);   // pdf.h:21
        spdf = R(1) / (R(4) * Const<R>::pi); // material.h:153
    }
    R pdf_value = R(0.5) * v0 + R(0.5) * v1;
    R wgt = spdf / pdf_value;
#ifdef RT_DEBUG_NAN
    if (wgt != wgt || atten.x != atten.x)
        printf("NAN diffuse: bounce %d mat %d kind %d v0 %g v1 %g spdf %g dir %g %g %g n %g %g %g p %g %g %g atten %g t %g primkind %d\n", bounce_idx,
               h.mat, m.kind, (double)v0, (double)v1, (double)spdf, (double)dir.x, (double)dir.y, (double)dir.z, (double)h.n.x,
               (double)h.n.y, (double)h.n.z, (double)h.p.x, (double)h.p.y, (double)h.p.z, (double)atten.x, (double)h.t, h.kind);
#endif
    beta = beta * scale<R>(wgt, atten);
    ray.o = h.p;
    ray.d = dir;
    return true;
}

struct RenderArgs {
    uint32_t k0, k1;
    int32_t s_begin, s_end;
    int32_t chunk_size;
    int32_t tiles_x, tiles_y;
    int32_t n_chunks;
};

// One wave = one 8x8 pixel tile x one chunk of the stratified sample range.  Each lane owns a pixel and runs its
// samples back to back ("path regeneration"): the loop body is one bounce of whichever path the lane is on, so
// lanes whose path ended start their next sample instead of idling until the longest path of the wave finishes.
template <class R>
__global__ void __launch_bounds__(256) render_kernel(SceneT<R> sc, CameraT<R> cam, RenderArgs a, float *__restrict__ out,
                                                     unsigned long long *__restrict__ ray_counter) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= a.tiles_x * a.tiles_y) return;
    const int chunk = blockIdx.y;
    const int ti = tile % a.tiles_x, tj = tile / a.tiles_x;
    const int i = ti * 8 + (lane & 7), j = tj * 8 + (lane >> 3);
    const bool inside = (i < cam.W) && (j < cam.H);
    int s = a.s_begin + chunk * a.chunk_size;
    int s_end = s + a.chunk_size;
    if (s_end > a.s_end) s_end = a.s_end;
    if (!inside) s = s_end;

    Rng<R> g;
    g.k0 = a.k0;
    g.k1 = a.k1;
    g.have = 0;
    const uint32_t pixel = (uint32_t)(j * cam.W + i);
    double ax = 0, ay = 0, az = 0; // per-pixel sum over samples (the reference sums in double: camera.h:175-181)
    unsigned rays = 0;
    RayT<R> ray;
    V3<R> beta, L;
    int depth = 0; // remaining ray_color depth; 0 = need a new camera path
    bool active = false;
    while (true) {
        if (!active) {
            if (s >= s_end) break;
            g.begin(pixel, (uint32_t)s);
            ray = get_ray(cam, i, j, s % cam.sqrt_spp, s / cam.sqrt_spp, g);
            beta = mk<R>(R(1), R(1), R(1));
            L = mk<R>(R(0), R(0), R(0));
            depth = cam.max_depth;
            active = true;
        }
        bool alive = false;
        if (depth > 0) { // camera.h:294
            rays++;
    #ifdef RT_DEBUG_NAN
            bool was_ok = !(beta.x != beta.x) && !(ray.d.x != ray.d.x) && !(L.x != L.x);
#endif
            alive = bounce(sc, cam, ray, beta, L, g, cam.max_depth - depth);
#ifdef RT_DEBUG_NAN
            if (was_ok && ((beta.x != beta.x) || (ray.d.x != ray.d.x) || (ray.o.x != ray.o.x) || (L.x != L.x)))
                printf("NAN after bounce %d: beta %g dir %g %g %g o %g %g %g L %g alive %d\n", cam.max_depth - depth, (double)beta.x,
                       (double)ray.d.x, (double)ray.d.y, (double)ray.d.z, (double)ray.o.x, (double)ray.o.y, (double)ray.o.z, (double)L.x, (int)alive);
#endif
            depth--;
            if (alive && depth == 0) { // depth exhausted: the recursion returns 0 here (camera.h:294-295)
                L = L + beta * mk<R>(R(0), R(0), R(0));
                alive = false;
            }
        }
        if (!alive) {
            ax += (double)L.x; ay += (double)L.y; az += (double)L.z;
            s++;
            active = false;
        }
    }
    if (inside) {
        size_t o = ((size_t)chunk * cam.H * cam.W + (size_t)j * cam.W + i) * 3;
        out[o] = (float)ax; out[o + 1] = (float)ay; out[o + 2] = (float)az;
    }
    // wave-level ray count -> one atomic per wave
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_xor(rays, off);
    if (lane == 0 && ray_counter) atomicAdd(ray_counter, (unsigned long long)rays);
}

struct TraceArgs {
    uint32_t k0, k1;
    int has_xi;
};

// rt_trace_batch: closest hit of each ray over [0.001, inf) (camera.h:300).
template <class R>
__global__ void __launch_bounds__(256) trace_kernel(SceneT<R> sc, const rt_ray *__restrict__ rays, const double *__restrict__ xi, size_t n,
                                                    rt_hit *__restrict__ out, TraceArgs a) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    RayT<R> r;
    r.o = mk<R>(R(rays[k].orig[0]), R(rays[k].orig[1]), R(rays[k].orig[2]));
    r.d = mk<R>(R(rays[k].dir[0]), R(rays[k].dir[1]), R(rays[k].dir[2]));
    r.tm = R(rays[k].time);
    const R xik = a.has_xi ? R(xi[k]) : R(0);
    auto mxi = [&](int node_id) {
        if (a.has_xi) return xik;
        uint32_t o0, o1, o2, o3;
        philox4x32_10((uint32_t)node_id, (uint32_t)k, (uint32_t)((uint64_t)k >> 32), 0x7fffffffu, a.k0, a.k1, o0, o1, o2, o3);
        return u01<R>(o0);
    };
    Best<R> b = traverse(sc, r, R(0.001), Const<R>::inf, mxi);
    rt_hit o;
    o.prim = -1;
    o.front_face = 0;
    o.t = (double)Const<double>::inf;
    o.p[0] = o.p[1] = o.p[2] = 0;
    o.normal[0] = o.normal[1] = o.normal[2] = 0;
    o.u = o.v = 0;
    if (b.meta >= 0) {
        Hit<R> h = finalize(sc, r, b, true);
        int id;
        if (h.kind == K_SPHERE) id = sc.sphere_ids[h.idx];
        else if (h.kind == K_QUAD) id = sc.quad_ids[h.idx];
        else if (h.kind == K_TRI) id = sc.tri_ids[h.idx];
        else id = sc.media[h.idx].node_id;
        o.prim = id;
        o.front_face = h.front ? 1 : 0;
        o.t = (double)h.t;
        o.p[0] = h.p.x; o.p[1] = h.p.y; o.p[2] = h.p.z;
        o.normal[0] = h.n.x; o.normal[1] = h.n.y; o.normal[2] = h.n.z;
        o.u = (double)h.u;
        o.v = (double)h.v;
    }
    out[k] = o;
}

} // namespace rtamd
#endif
