// rt_oracle.cpp -- CPU restatement of the reference's per-sample ray loop.  TEST INFRASTRUCTURE ONLY.
//
// This file is the parity oracle: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load it.  The product (raytracing_amd/, include/) never includes, links or calls anything in oracle/.
//
// It restates, in IEEE double and in the reference's own evaluation order, the hot path of
// Jiaqi1song/RayTracing's CPU renderer (src/cpu/*): camera::render -> get_ray -> ray_color -> world.hit ->
// material::scatter.  It walks the flat scene description of include/rt_amd.h recursively, exactly as the
// reference walks its object graph, so that with RNG mode "glibc" (std::rand in the reference's draw order) it
// reproduces the compiled reference (oracle/_ref) bit for bit.  Each function cites the reference lines it follows.
//
// Pinned by tests/test_oracle_vs_reference.py against fixtures generated from the compiled reference
// (oracle/refgen.cpp): hit records of deterministic ray batches and whole rendered frames, all bit-exact.
//
// Build: g++ -O2 -ffp-contract=off -fopenmp -shared -fPIC (see oracle/Makefile).
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "../include/rt_amd.h"

namespace {

const double infinity = std::numeric_limits<double>::infinity(); // utils.h:19
const double pi = 3.1415926535897932385;                         // utils.h:20

// ---- vec3.h ------------------------------------------------------------------------------------
struct V {
    double x, y, z;
};
inline V mk(double a, double b, double c) { return V{a, b, c}; }
inline V mk(const double *p) { return V{p[0], p[1], p[2]}; }
inline V operator+(V a, V b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }          // vec3.h:66
inline V operator-(V a, V b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }          // vec3.h:70
inline V operator-(V a) { return mk(-a.x, -a.y, -a.z); }                              // vec3.h:15
inline V operator*(V a, V b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }          // vec3.h:74
inline V operator*(double t, V v) { return mk(t * v.x, t * v.y, t * v.z); }           // vec3.h:78
inline V operator*(V v, double t) { return t * v; }                                   // vec3.h:82
inline V operator/(V v, double t) { return (1 / t) * v; }                             // vec3.h:86
inline double dot(V u, V v) { return u.x * v.x + u.y * v.y + u.z * v.z; }             // vec3.h:90
inline V cross(V u, V v) {                                                            // vec3.h:96
    return mk(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
inline double length_squared(V v) { return v.x * v.x + v.y * v.y + v.z * v.z; }       // vec3.h:41
inline double length(V v) { return std::sqrt(length_squared(v)); }                    // vec3.h:37
inline V unit_vector(V v) { return v / length(v); }                                   // vec3.h:102
inline V reflect(V v, V n) { return v - 2 * dot(v, n) * n; }                          // vec3.h:131
inline V refract(V uv, V n, double etai_over_etat) {                                  // vec3.h:135
    double cos_theta = std::fmin(dot(-uv, n), 1.0);
    V r_out_perp = etai_over_etat * (uv + cos_theta * n);
    V r_out_parallel = -std::sqrt(std::fabs(1.0 - length_squared(r_out_perp))) * n;
    return r_out_perp + r_out_parallel;
}

struct Ray { // ray.h:7
    V o, d;
    double tm;
    V at(double t) const { return o + t * d; } // ray.h:22
};

struct Interval { // interval.h:4
    double min, max;
    bool contains(double x) const { return min <= x && x <= max; }  // interval.h:22
    bool surrounds(double x) const { return min < x && x < max; }   // interval.h:26
    double clamp(double x) const {                                  // interval.h:30
        if (x < min) return min;
        if (x > max) return max;
        return x;
    }
};

struct Rec { // hittable.h:10
    V p{0, 0, 0}, normal{0, 0, 0};
    int mat = -1;
    double t = 0, u = 0, v = 0;
    bool front_face = false;
    int prim = -1; // node id of the writer (not in the reference: recovered there via unique materials)
    void set_face_normal(const Ray &r, V outward) { // hittable.h:20
        front_face = dot(r.d, outward) < 0;
        normal = front_face ? outward : -outward;
    }
};

// ---- Philox4x32 (Salmon et al., SC'11) : the counter-based RNG shared with the HIP kernels ----
// The sample streams use SEVEN rounds (RT_STREAM_ROUNDS: the fewest the paper reports as Crush-resistant for 4x32; the
// generator is 9 % of the GPU frame at ten); the ten-round function is kept for the Random123 known-answer vectors.
enum { RT_STREAM_ROUNDS = 7 };
inline void philox4x32(int rounds, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                       uint32_t out[4]) {
    for (int r = 0; r < rounds; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    philox4x32(10, c0, c1, c2, c3, k0, k1, out);
}
inline void philox_stream(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    philox4x32(RT_STREAM_ROUNDS, c0, c1, c2, c3, k0, k1, out);
}

enum { RNG_GLIBC = 0, RNG_PHILOX = 1, RNG_INJECT = 2 };

struct Counters {
    uint64_t rays = 0, aabb = 0, sphere = 0, quad = 0, tri = 0, samples = 0;
};

struct Ctx {
    const rt_scene_desc *sc;
    int mode;
    // philox state
    // counter = (block, sample, pixel, stream): stream 0 = the camera ray's draws (get_ray), 0x80000000|b = the draws of
    // ray_color invocation b of the path (scatter + mixture pdf).  A block is four 32-bit words = four uniforms, plus a
    // FIFTH uniform of 24 bits made of the low bytes of words 0..2 (bits the float kernels' 24-bit uniforms never use):
    // the fifth uniform of the block that starts ray b's predecessor -- the camera block for ray 0, bounce b-1's block 0
    // otherwise -- serves the scene's PRIMARY constant_medium while ray b is traced (primary_medium() below: the medium
    // with the largest boundary box; in final_scene the fog, which draws on every ray), so that draw costs no generator
    // call.  Every other medium draws from stream 1+b keyed by its node id.  Which uniform a medium gets therefore does
    // not depend on the order in which a ray meets the scene's objects (the production path walks a bounding-volume
    // hierarchy of its own).  A span-1 bvh_node has left == right (bvh.h:30) and the reference hits that child twice, a
    // constant_medium below it drawing a fresh uniform each time: `dup` is the path of second visits above the current
    // node (2*dup for the first visit, 2*dup+1 for the second); only visit 0 of the primary medium takes the fifth
    // uniform, any other visit -- like any other medium -- takes a hash of the same 24 bits keyed by (node id, dup).
    uint32_t k0 = 0, k1 = 0, pixel = 0, sample = 0, block = 0, have = 0, stream = 0;
    uint32_t buf[4];
    int bounce = 0;
    double inject = 0; // RNG_INJECT: every draw returns this
    uint32_t ray_lo = 0, ray_hi = 0;
    bool trace_keyed = false; // rt_trace_batch with xi == NULL
    Counters cnt;

    int primary = -1;  // primary_medium(sc)
    uint32_t dup = 0;  // second-visit path of span-1 bvh_nodes above the node being hit
    void begin_sample(uint32_t px, uint32_t s) { pixel = px; sample = s; block = 0; have = 0; stream = 0; }
    void begin_bounce(int b) { block = 0; have = 0; stream = 0x80000000u | (uint32_t)b; }

    // vec3.h:106 random_in_unit_disk and vec3.h:114 random_unit_vector are rejection loops in the reference.  Under the
    // counter-based generator (protocol v4) both take ONE word of the stream, which seeds a xorshift32 sub-sequence of two
    // uniforms, and sample their distribution DIRECTLY -- the unit sphere by (z, phi), the unit disk by (sqrt(u), phi) --
    // the same distributions as the loops' (a point uniform in the ball, normalised, is uniform on the sphere), without a
    // loop that a wave of 64 lanes runs until its unluckiest lane accepts (protocol v3 ran the loops on the sub-sequence).
    // std::rand / injected draws keep the reference's loops and its plain sequence.
    uint32_t sub = 0;
    bool in_sub = false;
    void sub_begin() {
        if (mode != RNG_PHILOX) return; // std::rand / injected draws stay the reference's plain sequence
        sub = next_word();
        if (sub == 0) sub = 0x9e3779b9u;
        in_sub = true;
    }
    void sub_end() { in_sub = false; }
    uint32_t next_word() {
        if (have == 0) {
            philox_stream(block++, sample, pixel, stream, k0, k1, buf);
            have = 4;
        }
        uint32_t x = buf[4 - have];
        have--;
        return x;
    }
    double next() { // utils.h:33 random_double()
        if (mode == RNG_GLIBC) return std::rand() / (RAND_MAX + 1.0);
        if (mode == RNG_INJECT) return inject;
        if (in_sub) {
            sub ^= sub << 13;
            sub ^= sub >> 17;
            sub ^= sub << 5;
            return sub * (1.0 / 4294967296.0);
        }
        return next_word() * (1.0 / 4294967296.0);
    }
    double medium_draw(int node_id) { // constant_medium.h:41's random_double()
        if (mode == RNG_GLIBC) return std::rand() / (RAND_MAX + 1.0);
        if (mode == RNG_INJECT && !trace_keyed) return inject;
        uint32_t o[4];
        if (trace_keyed) {
            philox_stream((uint32_t)node_id, ray_lo, ray_hi, 0x7fffffffu - dup, k0, k1, o);
            return o[0] * (1.0 / 4294967296.0);
        }
        if (node_id == primary && dup == 0) { // the fifth uniform of the block that started this ray's predecessor
            philox_stream(0u, sample, pixel, bounce == 0 ? 0u : (0x80000000u | (uint32_t)(bounce - 1)), k0, k1, o);
            return (double)(((o[0] & 255u) << 16) | ((o[1] & 255u) << 8) | (o[2] & 255u)) * (1.0 / 16777216.0);
        }
        // any other medium (or visit): the same 24 bits, hashed with the medium's node id and visit path (protocol v4:
        // a full generator block per medium record was 3 % of final_scene's frame, for the blue sphere inside the glass one)
        philox_stream(0u, sample, pixel, bounce == 0 ? 0u : (0x80000000u | (uint32_t)(bounce - 1)), k0, k1, o);
        const uint32_t k24 = ((o[0] & 255u) << 16) | ((o[1] & 255u) << 8) | (o[2] & 255u);
        return medium_hash(k24, (uint32_t)node_id, dup) * (1.0 / 4294967296.0);
    }
    static uint32_t medium_hash(uint32_t k24, uint32_t node_id, uint32_t dup) { // lowbias32 (Wellons) of the keyed word
        uint32_t x = k24 ^ (node_id * 0x9E3779B9u) ^ (dup * 0x85EBCA6Bu);
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        return x;
    }
    double next(double mn, double mx) { return mn + (mx - mn) * next(); }          // utils.h:38
    int next_int(int mn, int mx) { return int(next(mn, mx + 1)); }                 // utils.h:43
};

// vec3.h:106.  Argument evaluation order of the reference's `vec3(random_double(-1,1), random_double(-1,1), 0)`
// is compiler-chosen; g++ 11 -O2 evaluates these constructor arguments right to left (pinned against
// oracle/_ref by the whole-frame fixtures: any other order desynchronises the std::rand stream).
#ifndef ORC_ARGS_RIGHT_TO_LEFT
#define ORC_ARGS_RIGHT_TO_LEFT 1
#endif

inline V random_in_unit_disk(Ctx &c) {
    struct Sub { Ctx &c; Sub(Ctx &c_) : c(c_) { c.sub_begin(); } ~Sub() { c.sub_end(); } } sub(c);
    if (c.mode == RNG_PHILOX) { // direct: radius sqrt(u1), angle 2 pi u2
        const double u1 = c.next(), u2 = c.next();
        const double r = std::sqrt(u1), phi = 2 * pi * u2;
        return mk(r * std::cos(phi), r * std::sin(phi), 0);
    }
    while (true) {
#if ORC_ARGS_RIGHT_TO_LEFT
        double y = c.next(-1, 1);
        double x = c.next(-1, 1);
#else
        double x = c.next(-1, 1);
        double y = c.next(-1, 1);
#endif
        V p = mk(x, y, 0);
        if (length_squared(p) < 1) return p;
    }
}
inline V random_vec(Ctx &c, double mn, double mx) { // vec3.h:55
#if ORC_ARGS_RIGHT_TO_LEFT
    double z = c.next(mn, mx);
    double y = c.next(mn, mx);
    double x = c.next(mn, mx);
#else
    double x = c.next(mn, mx);
    double y = c.next(mn, mx);
    double z = c.next(mn, mx);
#endif
    return mk(x, y, z);
}
inline V random_unit_vector(Ctx &c) { // vec3.h:114
    struct Sub { Ctx &c; Sub(Ctx &c_) : c(c_) { c.sub_begin(); } ~Sub() { c.sub_end(); } } sub(c);
    if (c.mode == RNG_PHILOX) { // direct: z uniform in [-1, 1], angle 2 pi u2
        const double u1 = c.next(), u2 = c.next();
        const double z = 1 - 2 * u1, r = std::sqrt(std::fmax(0.0, 1 - z * z)), phi = 2 * pi * u2;
        return mk(r * std::cos(phi), r * std::sin(phi), z);
    }
    while (true) {
        V p = random_vec(c, -1, 1);
        double lensq = length_squared(p);
        if (1e-160 < lensq && lensq <= 1.0) return p / std::sqrt(lensq);
    }
}
inline V random_cosine_direction(Ctx &c) { // vec3.h:142
    double r1 = c.next();
    double r2 = c.next();
    double phi = 2 * pi * r1;
    double x = std::cos(phi) * std::sqrt(r2);
    double y = std::sin(phi) * std::sqrt(r2);
    double z = std::sqrt(1 - r2);
    return mk(x, y, z);
}

struct Onb { // onb.h:5
    V axis[3];
    explicit Onb(V n) {
        axis[2] = unit_vector(n);
        V a = (std::fabs(axis[2].x) > 0.9) ? mk(0, 1, 0) : mk(1, 0, 0);
        axis[1] = unit_vector(cross(axis[2], a));
        axis[0] = cross(axis[2], axis[1]);
    }
    V transform(V v) const { return (v.x * axis[0]) + (v.y * axis[1]) + (v.z * axis[2]); } // onb.h:18
};

// ---- aabb.h:36 ------------------------------------------------------------------------------------
inline bool aabb_hit(const double *bb, const Ray &r, Interval ray_t, Ctx &c) {
    c.cnt.aabb++;
    const double orig[3] = {r.o.x, r.o.y, r.o.z};
    const double dir[3] = {r.d.x, r.d.y, r.d.z};
    for (int axis = 0; axis < 3; axis++) {
        const double amin = bb[2 * axis], amax = bb[2 * axis + 1];
        const double adinv = 1.0 / dir[axis];
        double t0 = (amin - orig[axis]) * adinv;
        double t1 = (amax - orig[axis]) * adinv;
        if (t0 < t1) {
            if (t0 > ray_t.min) ray_t.min = t0;
            if (t1 < ray_t.max) ray_t.max = t1;
        } else {
            if (t1 > ray_t.min) ray_t.min = t1;
            if (t0 < ray_t.max) ray_t.max = t0;
        }
        if (ray_t.max <= ray_t.min) return false;
    }
    return true;
}

// ---- primitives ------------------------------------------------------------------------------------
inline void get_sphere_uv(V p, double &u, double &v) { // sphere.h:89
    double theta = std::acos(-p.y);
    double phi = std::atan2(-p.z, p.x) + pi;
    u = phi / (2 * pi);
    v = theta / pi;
}

bool sphere_hit(const rt_sphere &s, int node, const Ray &r, Interval ray_t, Rec &rec, Ctx &c) { // sphere.h:29
    c.cnt.sphere++;
    V current_center = mk(s.center) + r.tm * mk(s.dcenter); // center.at(r.time())
    V oc = current_center - r.o;
    double a = length_squared(r.d);
    double h = dot(r.d, oc);
    double cc = length_squared(oc) - s.radius * s.radius;
    double discriminant = h * h - a * cc;
    if (discriminant < 0) return false;
    double sqrtd = std::sqrt(discriminant);
    double root = (h - sqrtd) / a;
    if (!ray_t.surrounds(root)) {
        root = (h + sqrtd) / a;
        if (!ray_t.surrounds(root)) return false;
    }
    rec.t = root;
    rec.p = r.at(rec.t);
    V outward_normal = (rec.p - current_center) / s.radius;
    rec.set_face_normal(r, outward_normal);
    get_sphere_uv(outward_normal, rec.u, rec.v);
    rec.mat = s.mat;
    rec.prim = node;
    return true;
}

bool quad_hit(const rt_quad &q, int node, const Ray &r, Interval ray_t, Rec &rec, Ctx &c) { // quad.h:32
    c.cnt.quad++;
    V normal = mk(q.normal);
    double denom = dot(normal, r.d);
    if (std::fabs(denom) < 1e-8) return false;
    double t = (q.D - dot(normal, r.o)) / denom;
    if (!ray_t.contains(t)) return false;
    V intersection = r.at(t);
    V planar_hitpt_vector = intersection - mk(q.Q);
    double alpha = dot(mk(q.w), cross(planar_hitpt_vector, mk(q.v)));
    double beta = dot(mk(q.w), cross(mk(q.u), planar_hitpt_vector));
    Interval unit_interval{0, 1}; // quad.h:62 is_interior
    if (!unit_interval.contains(alpha) || !unit_interval.contains(beta)) return false;
    rec.u = alpha;
    rec.v = beta;
    rec.t = t;
    rec.p = intersection;
    rec.mat = q.mat;
    rec.set_face_normal(r, normal);
    rec.prim = node;
    return true;
}

bool triangle_hit(const rt_triangle &tr, int node, const Ray &r, Interval ray_t, Rec &rec, Ctx &c) { // triangle.h:23
    c.cnt.tri++;
    const float EPSILON = 0.000001; // triangle.h:16,80 (const float)
    V v0 = mk(tr.v0), v1 = mk(tr.v1), v2 = mk(tr.v2);
    V edge1 = v1 - v0;
    V edge2 = v2 - v0;
    V h = cross(r.d, edge2);
    float a = dot(edge1, h);
    if (tr.cull && a < EPSILON) return false;
    if (std::fabs(a) < EPSILON) return false;
    float f = 1.0 / a;
    V s = r.o - v0;
    float u = f * dot(s, h);
    if (u < 0.0 || u > 1.0) return false;
    V q = cross(s, edge1);
    float v = f * dot(r.d, q);
    if (v < 0.0 || u + v > 1.0) return false;
    float t = f * dot(edge2, q);
    if (!ray_t.contains(t)) return false;
    rec.t = t;
    rec.p = r.at(t);
    rec.normal = mk(tr.normal);
    rec.mat = tr.mat;
    rec.set_face_normal(r, rec.normal);
    rec.u = u;
    rec.v = v;
    rec.prim = node;
    return true;
}

bool node_hit(int id, const Ray &r, Interval ray_t, Rec &rec, Ctx &c);

// The constant_medium that takes the fifth uniform (see Ctx): the one whose boundary has the bounding box of largest
// surface area, the lowest node id among equals; -1 if the scene has none.
int primary_medium(const rt_scene_desc *sc) {
    int best = -1;
    double best_area = -1.0;
    for (int i = 0; i < sc->n_nodes; i++) {
        const rt_node &n = sc->nodes[i];
        if (n.kind != RT_NODE_MEDIUM || n.a < 0 || n.a >= sc->n_nodes) continue;
        const double *b = sc->nodes[n.a].bbox;
        const double dx = b[1] - b[0], dy = b[3] - b[2], dz = b[5] - b[4];
        const double area = 2.0 * (dx * dy + dy * dz + dz * dx);
        if (area > best_area) { best_area = area; best = i; }
    }
    return best;
}

bool medium_hit(const rt_node &n, int node, const Ray &r, Interval ray_t, Rec &rec, Ctx &c) { // constant_medium.h:21
    Rec rec1, rec2;
    if (!node_hit(n.a, r, Interval{-infinity, infinity}, rec1, c)) return false;
    if (!node_hit(n.a, r, Interval{rec1.t + 0.0001, infinity}, rec2, c)) return false;
    if (rec1.t < ray_t.min) rec1.t = ray_t.min;
    if (rec2.t > ray_t.max) rec2.t = ray_t.max;
    if (rec1.t >= rec2.t) return false;
    if (rec1.t < 0) rec1.t = 0;
    double ray_length = length(r.d);
    double distance_inside_boundary = (rec2.t - rec1.t) * ray_length;
    double hit_distance = n.p[0] * std::log(c.medium_draw(node));
    if (hit_distance > distance_inside_boundary) return false;
    rec.t = rec1.t + hit_distance / ray_length;
    rec.p = r.at(rec.t);
    rec.normal = mk(1, 0, 0);
    rec.front_face = true;
    rec.mat = n.b;
    rec.prim = node;
    return true;
}

bool node_hit(int id, const Ray &r, Interval ray_t, Rec &rec, Ctx &c) {
    const rt_scene_desc *sc = c.sc;
    const rt_node &n = sc->nodes[id];
    switch (n.kind) {
    case RT_NODE_BVH: { // bvh.h:43
        if (!aabb_hit(n.bbox, r, ray_t, c)) return false;
        const uint32_t dup = c.dup;
        if (n.a == n.b) c.dup = 2 * dup; // span 1: the same child twice (see Ctx)
        bool hit_left = node_hit(n.a, r, ray_t, rec, c);
        if (n.a == n.b) c.dup = 2 * dup + 1;
        bool hit_right = node_hit(n.b, r, Interval{ray_t.min, hit_left ? rec.t : ray_t.max}, rec, c);
        c.dup = dup;
        return hit_left || hit_right;
    }
    case RT_NODE_LIST: { // hittable_list.h:24
        Rec temp_rec;
        bool hit_anything = false;
        double closest_so_far = ray_t.max;
        for (int k = 0; k < n.b; k++) {
            int child = sc->child_index[n.a + k];
            if (node_hit(child, r, Interval{ray_t.min, closest_so_far}, temp_rec, c)) {
                hit_anything = true;
                closest_so_far = temp_rec.t;
                rec = temp_rec;
            }
        }
        return hit_anything;
    }
    case RT_NODE_TRANSLATE: { // hittable.h:53
        V offset = mk(n.p);
        Ray offset_r{r.o - offset, r.d, r.tm};
        if (!node_hit(n.a, offset_r, ray_t, rec, c)) return false;
        rec.p = rec.p + offset;
        return true;
    }
    case RT_NODE_ROTATE_Y: { // hittable.h:110
        double sin_theta = n.p[0], cos_theta = n.p[1];
        V origin = mk((cos_theta * r.o.x) - (sin_theta * r.o.z), r.o.y, (sin_theta * r.o.x) + (cos_theta * r.o.z));
        V direction = mk((cos_theta * r.d.x) - (sin_theta * r.d.z), r.d.y, (sin_theta * r.d.x) + (cos_theta * r.d.z));
        Ray rotated_r{origin, direction, r.tm};
        if (!node_hit(n.a, rotated_r, ray_t, rec, c)) return false;
        rec.p = mk((cos_theta * rec.p.x) + (sin_theta * rec.p.z), rec.p.y, (-sin_theta * rec.p.x) + (cos_theta * rec.p.z));
        rec.normal = mk((cos_theta * rec.normal.x) + (sin_theta * rec.normal.z), rec.normal.y,
                        (-sin_theta * rec.normal.x) + (cos_theta * rec.normal.z));
        return true;
    }
    case RT_NODE_MEDIUM: return medium_hit(n, id, r, ray_t, rec, c);
    case RT_NODE_SPHERE: return sphere_hit(sc->spheres[n.a], id, r, ray_t, rec, c);
    case RT_NODE_QUAD: return quad_hit(sc->quads[n.a], id, r, ray_t, rec, c);
    case RT_NODE_TRIANGLE: return triangle_hit(sc->triangles[n.a], id, r, ray_t, rec, c);
    }
    return false;
}

// ---- hittable::pdf_value / random (hittable.h:35-41 defaults) ---------------------------------------
double node_pdf_value(int id, V origin, V direction, Ctx &c) {
    const rt_scene_desc *sc = c.sc;
    const rt_node &n = sc->nodes[id];
    switch (n.kind) {
    case RT_NODE_LIST: { // hittable_list.h:42
        double weight = 1.0 / n.b;
        double sum = 0.0;
        for (int k = 0; k < n.b; k++) sum += weight * node_pdf_value(sc->child_index[n.a + k], origin, direction, c);
        return sum;
    }
    case RT_NODE_SPHERE: { // sphere.h:62
        const rt_sphere &s = sc->spheres[n.a];
        Rec rec;
        Counters keep = c.cnt;
        bool h = sphere_hit(s, id, Ray{origin, direction, 0}, Interval{0.001, infinity}, rec, c);
        c.cnt = keep; // light-pdf probes are not part of the per-sample work table
        if (!h) return 0;
        double dist_squared = length_squared(mk(s.center) + 0.0 * mk(s.dcenter) - origin); // center.at(0)
        double cos_theta_max = std::sqrt(1 - s.radius * s.radius / dist_squared);
        double solid_angle = 2 * pi * (1 - cos_theta_max);
        return 1 / solid_angle;
    }
    case RT_NODE_QUAD: { // quad.h:75
        const rt_quad &q = sc->quads[n.a];
        Rec rec;
        Counters keep = c.cnt;
        bool h = quad_hit(q, id, Ray{origin, direction, 0}, Interval{0.001, infinity}, rec, c);
        c.cnt = keep;
        if (!h) return 0;
        double distance_squared = rec.t * rec.t * length_squared(direction);
        double cosine = std::fabs(dot(direction, rec.normal) / length(direction));
        return distance_squared / (cosine * q.area);
    }
    default: return 0.0; // hittable.h:35
    }
}

V node_random(int id, V origin, Ctx &c) {
    const rt_scene_desc *sc = c.sc;
    const rt_node &n = sc->nodes[id];
    switch (n.kind) {
    case RT_NODE_LIST: { // hittable_list.h:52
        int int_size = n.b;
        int pick = c.next_int(0, int_size - 1);
        return node_random(sc->child_index[n.a + pick], origin, c);
    }
    case RT_NODE_SPHERE: { // sphere.h:76, 98
        const rt_sphere &s = sc->spheres[n.a];
        V direction = mk(s.center) + 0.0 * mk(s.dcenter) - origin;
        double distance_squared = length_squared(direction);
        Onb uvw(direction);
        double r1 = c.next();
        double r2 = c.next();
        double z = 1 + r2 * (std::sqrt(1 - s.radius * s.radius / distance_squared) - 1);
        double phi = 2 * pi * r1;
        double x = std::cos(phi) * std::sqrt(1 - z * z);
        double y = std::sin(phi) * std::sqrt(1 - z * z);
        return uvw.transform(mk(x, y, z));
    }
    case RT_NODE_QUAD: { // quad.h:86: Q + (random_double()*u) + (random_double()*v)
        const rt_quad &q = sc->quads[n.a];
        // operand evaluation order of `+` is compiler-chosen; g++ 11 -O2 draws the v coefficient first
        // (pinned by the whole-frame fixtures of cornell_box/final_scene/mesh_scene against oracle/_ref).
        double rv = c.next();
        double ru = c.next();
        V p = mk(q.Q) + (ru * mk(q.u)) + (rv * mk(q.v));
        return p - origin;
    }
    default: return mk(1, 0, 0); // hittable.h:39
    }
}

// ---- perlin.h / texture.h ------------------------------------------------------------------------------------
double perlin_noise(const rt_perlin &pn, V p) { // perlin.h:16, 75
    double u = p.x - std::floor(p.x);
    double v = p.y - std::floor(p.y);
    double w = p.z - std::floor(p.z);
    int i = int(std::floor(p.x));
    int j = int(std::floor(p.y));
    int k = int(std::floor(p.z));
    V c[2][2][2];
    for (int di = 0; di < 2; di++)
        for (int dj = 0; dj < 2; dj++)
            for (int dk = 0; dk < 2; dk++)
                c[di][dj][dk] = mk(pn.randvec[pn.perm_x[(i + di) & 255] ^ pn.perm_y[(j + dj) & 255] ^ pn.perm_z[(k + dk) & 255]]);
    double uu = u * u * (3 - 2 * u);
    double vv = v * v * (3 - 2 * v);
    double ww = w * w * (3 - 2 * w);
    double accum = 0.0;
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++)
            for (int d = 0; d < 2; d++) {
                V weight_v = mk(u - a, v - b, w - d);
                accum += (a * uu + (1 - a) * (1 - uu)) * (b * vv + (1 - b) * (1 - vv)) * (d * ww + (1 - d) * (1 - ww)) *
                         dot(c[a][b][d], weight_v);
            }
    return accum;
}

double perlin_turb(const rt_perlin &pn, V p, int depth) { // perlin.h:38
    double accum = 0.0;
    V temp_p = p;
    double weight = 1.0;
    for (int i = 0; i < depth; i++) {
        accum += weight * perlin_noise(pn, temp_p);
        weight *= 0.5;
        temp_p = mk(temp_p.x * 2, temp_p.y * 2, temp_p.z * 2); // vec3 operator*=(double)
    }
    return std::fabs(accum);
}

V texture_value(const rt_scene_desc *sc, int tex, double u, double v, V p) {
    const rt_texture &t = sc->textures[tex];
    switch (t.kind) {
    case RT_TEX_SOLID: return mk(t.color); // texture.h:22
    case RT_TEX_CHECKER: {                 // texture.h:39
        int xInteger = int(std::floor(t.scale * p.x));
        int yInteger = int(std::floor(t.scale * p.y));
        int zInteger = int(std::floor(t.scale * p.z));
        bool isEven = (xInteger + yInteger + zInteger) % 2 == 0;
        return isEven ? texture_value(sc, t.a, u, v, p) : texture_value(sc, t.b, u, v, p);
    }
    case RT_TEX_IMAGE: { // texture.h:60
        if (t.a < 0 || sc->images[t.a].height <= 0) return mk(0, 1, 1);
        const rt_image &im = sc->images[t.a];
        u = Interval{0, 1}.clamp(u);
        v = 1.0 - Interval{0, 1}.clamp(v);
        int i = int(u * im.width);
        int j = int(v * im.height);
        // rtw_image::pixel_data + clamp (stb_image_utils.h:60-85)
        i = i < 0 ? 0 : (i < im.width ? i : im.width - 1);
        j = j < 0 ? 0 : (j < im.height ? j : im.height - 1);
        const uint8_t *pixel = sc->image_bytes + im.offset + (size_t)j * (im.width * 3) + (size_t)i * 3;
        double color_scale = 1.0 / 255.0;
        return mk(color_scale * pixel[0], color_scale * pixel[1], color_scale * pixel[2]);
    }
    case RT_TEX_NOISE: // texture.h:85
        return mk(.5, .5, .5) * (1 + std::sin(t.scale * p.z + 10 * perlin_turb(sc->perlins[t.a], p, 7)));
    }
    return mk(0, 0, 0);
}

// ---- material.h ------------------------------------------------------------------------------------
struct Srec { // material.h:9
    V attenuation;
    int pdf_kind; // 0 none, 1 cosine(normal), 2 sphere
    V pdf_w;
    bool skip_pdf;
    Ray skip_pdf_ray;
};

V mat_emitted(const rt_scene_desc *sc, int mat, const Rec &rec, double u, double v, V p) {
    const rt_material &m = sc->materials[mat];
    if (m.kind != RT_MAT_DIFFUSE_LIGHT) return mk(0, 0, 0); // material.h:22
    if (!rec.front_face) return mk(0, 0, 0);                // material.h:131
    return texture_value(sc, m.tex, u, v, p);
}

double reflectance(double cosine, double refraction_index) { // material.h:115
    double r0 = (1 - refraction_index) / (1 + refraction_index);
    r0 = r0 * r0;
    return r0 + (1 - r0) * std::pow((1 - cosine), 5);
}

bool mat_scatter(const rt_scene_desc *sc, int mat, const Ray &r_in, const Rec &rec, Srec &srec, Ctx &c) {
    const rt_material &m = sc->materials[mat];
    switch (m.kind) {
    case RT_MAT_LAMBERTIAN: // material.h:44
        srec.attenuation = texture_value(sc, m.tex, rec.u, rec.v, rec.p);
        srec.pdf_kind = 1;
        srec.pdf_w = rec.normal;
        srec.skip_pdf = false;
        return true;
    case RT_MAT_METAL: { // material.h:66
        V reflected = reflect(r_in.d, rec.normal);
        V ruv = random_unit_vector(c);
        reflected = unit_vector(reflected) + (m.fuzz * ruv);
        srec.attenuation = mk(m.albedo);
        srec.pdf_kind = 0;
        srec.skip_pdf = true;
        srec.skip_pdf_ray = Ray{rec.p, reflected, r_in.tm};
        return true;
    }
    case RT_MAT_DIELECTRIC: { // material.h:88
        srec.attenuation = mk(1.0, 1.0, 1.0);
        srec.pdf_kind = 0;
        srec.skip_pdf = true;
        double ri = rec.front_face ? (1.0 / m.ior) : m.ior;
        V unit_direction = unit_vector(r_in.d);
        double cos_theta = std::fmin(dot(-unit_direction, rec.normal), 1.0);
        double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
        bool cannot_refract = ri * sin_theta > 1.0;
        V direction;
        if (cannot_refract || reflectance(cos_theta, ri) > c.next())
            direction = reflect(unit_direction, rec.normal);
        else
            direction = refract(unit_direction, rec.normal, ri);
        srec.skip_pdf_ray = Ray{rec.p, direction, r_in.tm};
        return true;
    }
    case RT_MAT_DIFFUSE_LIGHT: return false; // material.h:28
    case RT_MAT_ISOTROPIC:                   // material.h:146
        srec.attenuation = texture_value(sc, m.tex, rec.u, rec.v, rec.p);
        srec.pdf_kind = 2;
        srec.skip_pdf = false;
        return true;
    }
    return false;
}

double mat_scattering_pdf(const rt_scene_desc *sc, int mat, const Rec &rec, const Ray &scattered) {
    const rt_material &m = sc->materials[mat];
    if (m.kind == RT_MAT_LAMBERTIAN) { // material.h:51
        double cos_theta = dot(rec.normal, unit_vector(scattered.d));
        return cos_theta < 0 ? 0 : cos_theta / pi;
    }
    if (m.kind == RT_MAT_ISOTROPIC) return 1 / (4 * pi); // material.h:153
    return 0;                                            // material.h:32
}

// ---- camera.h ------------------------------------------------------------------------------------
struct Cam {
    const rt_camera *c;
    double recip_sqrt_spp;
};

Ray get_ray(const Cam &cam, int i, int j, int s_i, int s_j, Ctx &c) { // camera.h:249
    const rt_camera &k = *cam.c;
    double px = ((s_i + c.next()) * cam.recip_sqrt_spp) - 0.5; // camera.h:269
    double py = ((s_j + c.next()) * cam.recip_sqrt_spp) - 0.5;
    V pixel_sample = mk(k.pixel00_loc) + ((i + px) * mk(k.pixel_delta_u)) + ((j + py) * mk(k.pixel_delta_v));
    V ray_origin;
    if (k.defocus_angle <= 0) ray_origin = mk(k.center);
    else { // camera.h:285
        V p = random_in_unit_disk(c);
        ray_origin = mk(k.center) + (p.x * mk(k.defocus_disk_u)) + (p.y * mk(k.defocus_disk_v));
    }
    V ray_direction = pixel_sample - ray_origin;
    double ray_time = c.next();
    return Ray{ray_origin, ray_direction, ray_time};
}

V ray_color(const Cam &cam, const Ray &r, int depth, Ctx &c) { // camera.h:291
    const rt_scene_desc *sc = c.sc;
    if (depth <= 0) return mk(0, 0, 0);
    c.cnt.rays++;
    c.bounce = cam.c->max_depth - depth;
    c.begin_bounce(c.bounce);
    Rec rec;
    if (!node_hit(sc->world_root, r, Interval{0.001, infinity}, rec, c)) return mk(cam.c->background);
    Srec srec;
    V color_from_emission = mat_emitted(sc, rec.mat, rec, rec.u, rec.v, rec.p);
    if (!mat_scatter(sc, rec.mat, r, rec, srec, c)) return color_from_emission;
    if (srec.skip_pdf) return srec.attenuation * ray_color(cam, srec.skip_pdf_ray, depth - 1, c);

    // mixture_pdf p(hittable_pdf(lights, rec.p), srec.pdf_ptr): pdf.h:69-85
    V dir;
    if (c.next() < 0.5) dir = node_random(sc->lights_root, rec.p, c);
    else if (srec.pdf_kind == 1) dir = Onb(srec.pdf_w).transform(random_cosine_direction(c)); // pdf.h:40
    else dir = random_unit_vector(c);                                                          // pdf.h:25
    Ray scattered{rec.p, dir, r.tm};
    double v0 = node_pdf_value(sc->lights_root, rec.p, scattered.d, c);
    double v1;
    if (srec.pdf_kind == 1) { // pdf.h:35
        double cosine_theta = dot(unit_vector(scattered.d), Onb(srec.pdf_w).axis[2]);
        v1 = std::fmax(0, cosine_theta / pi);
    } else v1 = 1 / (4 * pi); // pdf.h:21
    double pdf_value = 0.5 * v0 + 0.5 * v1;
    double scattering_pdf = mat_scattering_pdf(sc, rec.mat, rec, scattered);
    V sample_color = ray_color(cam, scattered, depth - 1, c);
    V color_from_scatter = (srec.attenuation * scattering_pdf * sample_color) / pdf_value;
    return color_from_emission + color_from_scatter;
}

} // namespace

extern "C" {

// Philox known-answer hook for tests.
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}

// world.hit(r, interval(0.001, infinity), rec) for each ray (camera.h:300).  xi: see rt_trace_batch.
int orc_trace_batch(const rt_scene_desc *sc, const rt_ray *rays, const double *xi, size_t n, rt_hit *out,
                    uint64_t *counters /* optional [5]: aabb,sphere,quad,tri,rays */) {
    Ctx c;
    c.sc = sc;
    c.primary = primary_medium(sc);
    c.mode = RNG_INJECT;
    c.trace_keyed = (xi == nullptr);
    for (size_t k = 0; k < n; k++) {
        Ray r{mk(rays[k].orig), mk(rays[k].dir), rays[k].time};
        c.inject = xi ? xi[k] : 0.0;
        c.ray_lo = (uint32_t)k;
        c.ray_hi = (uint32_t)((uint64_t)k >> 32);
        Rec rec;
        bool h = node_hit(sc->world_root, r, Interval{0.001, infinity}, rec, c);
        c.cnt.rays++;
        rt_hit &o = out[k];
        std::memset(&o, 0, sizeof(o));
        if (!h) {
            o.prim = -1;
            o.t = infinity;
            continue;
        }
        o.prim = rec.prim;
        o.front_face = rec.front_face ? 1 : 0;
        o.t = rec.t;
        o.p[0] = rec.p.x; o.p[1] = rec.p.y; o.p[2] = rec.p.z;
        o.normal[0] = rec.normal.x; o.normal[1] = rec.normal.y; o.normal[2] = rec.normal.z;
        // constant_medium::hit leaves rec.u/rec.v untouched (constant_medium.h:46-51): stale or uninitialised in the
        // reference, reported as 0 here (and by oracle/refgen.cpp and the HIP path).
        bool medium = sc->nodes[rec.prim].kind == RT_NODE_MEDIUM;
        o.u = medium ? 0.0 : rec.u;
        o.v = medium ? 0.0 : rec.v;
    }
    if (counters) {
        counters[0] = c.cnt.aabb; counters[1] = c.cnt.sphere; counters[2] = c.cnt.quad; counters[3] = c.cnt.tri;
        counters[4] = c.cnt.rays;
    }
    return 0;
}

// The pixel/sample loop of camera::render (camera.h:171-186) over linear stratum indices [s_begin, s_end).
// accum: image_height*image_width*3 doubles, the UNSCALED sum of ray_color over the samples.
// rng_mode 0: std::rand in the reference's order (single thread, full sample range -> bit-exact with oracle/_ref);
// rng_mode 1: Philox keyed by (seed, pixel, sample) (OpenMP over rows; same stream protocol as the HIP kernels).
int orc_render(const rt_scene_desc *sc, const rt_camera *cam, uint64_t seed, int s_begin, int s_end, int rng_mode,
               int num_threads, double *accum, uint64_t *counters /* optional [6]: aabb,sphere,quad,tri,rays,samples */) {
    const int W = cam->image_width, H = cam->image_height, n = cam->sqrt_spp;
    if (s_end < 0) s_end = n * n; // RT_ALL_STRATA; s_begin == s_end is an empty range
    Cam cm{cam, 1.0 / n};
    uint64_t tot[6] = {0, 0, 0, 0, 0, 0};
    if (rng_mode == RNG_GLIBC) num_threads = 1;
    if (num_threads < 1) num_threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads) if (num_threads > 1)
    for (int j = 0; j < H; j++) {
        Ctx c;
        c.sc = sc;
        c.primary = primary_medium(sc);
        c.mode = rng_mode;
        c.k0 = (uint32_t)seed;
        c.k1 = (uint32_t)(seed >> 32);
        for (int i = 0; i < W; i++) {
            V pixel_color = mk(0, 0, 0);
            for (int s = s_begin; s < s_end; s++) {
                int s_j = s / n, s_i = s % n; // camera.h:176-177 loop nest: s_j outer, s_i inner
                c.begin_sample((uint32_t)(j * W + i), (uint32_t)s);
                Ray r = get_ray(cm, i, j, s_i, s_j, c);
                V col = ray_color(cm, r, cam->max_depth, c);
                pixel_color = pixel_color + col; // vec3 operator+=
                c.cnt.samples++;
            }
            double *a = accum + 3 * ((size_t)j * W + i);
            a[0] = pixel_color.x; a[1] = pixel_color.y; a[2] = pixel_color.z;
        }
#pragma omp critical
        {
            tot[0] += c.cnt.aabb; tot[1] += c.cnt.sphere; tot[2] += c.cnt.quad; tot[3] += c.cnt.tri;
            tot[4] += c.cnt.rays; tot[5] += c.cnt.samples;
        }
    }
    if (counters) std::memcpy(counters, tot, sizeof(tot));
    return 0;
}

// write_color (color.h:20-45) applied to scale*sum.
void orc_write_color(const double *accum, double scale, size_t n_pixels, int *out) {
    for (size_t k = 0; k < n_pixels; k++) {
        double ch[3] = {scale * accum[3 * k], scale * accum[3 * k + 1], scale * accum[3 * k + 2]}; // t*v (vec3.h:78)
        for (int q = 0; q < 3; q++) {
            double x = ch[q];
            if (x != x) x = 0.0;
            x = (x > 0) ? std::sqrt(x) : 0; // linear_to_gamma color.h:11
            Interval intensity{0.000, 0.999};
            out[3 * k + q] = int(256 * intensity.clamp(x));
        }
    }
}

} // extern "C"
