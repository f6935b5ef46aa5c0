// rt_flatten.h -- host side: turn an rt_scene_desc (the reference's object graph as plain structs) into the threaded
// device layout of rt_scene.h.  Pure C++ (no HIP); runs once per scene upload.
#ifndef RT_FLATTEN_H
#define RT_FLATTEN_H

#include <cmath>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_scene.h"

namespace rtamd {

template <class R> struct Flat {
    std::vector<UnitT<R>> stream; // threaded record stream (rt_scene.h)
    std::vector<UnitT<R>> cold;   // parallel to `stream`: finalisation-only data
    std::vector<XformT<R>> xforms; // [0] = world (unused record)
    std::vector<MatT<R>> mats;
    std::vector<TexT<R>> texs;
    std::vector<ImageT> images;
    std::vector<PerlinT<R>> perlins;
    std::vector<LightT<R>> lights;
    int lights_is_list = 0;
    int n_fused_boxes = 0;
    int n_baked_instances = 0; // translate / rotate_y subtrees of plain spheres emitted in world space (f32 stream)
    int has_gen_medium = 0;    // the stream holds MEDIUM_GEN records (kernels with the sub-stream walk are launched)
};

// conversions double -> R; boxes are rounded outward so that an f32 box always contains the f64 one
inline void cvt(double x, double &o) { o = x; }
inline void cvt(double x, float &o) { o = (float)x; }
inline void cvt_lo(double x, double &o) { o = x; }
inline void cvt_hi(double x, double &o) { o = x; }
inline void cvt_lo(double x, float &o) {
    o = (float)x;
    if ((double)o > x) o = std::nextafterf(o, -INFINITY);
}
inline void cvt_hi(double x, float &o) {
    o = (float)x;
    if ((double)o < x) o = std::nextafterf(o, INFINITY);
}
template <class R> inline void cvt3(const double *s, R *d) {
    cvt(s[0], d[0]);
    cvt(s[1], d[1]);
    cvt(s[2], d[2]);
}

template <class R> class Flattener {
  public:
    // fuse_boxes: replace a hittable_list of six axis-aligned quads forming a closed box (quad.h:103 box()) by one
    // AABOX record.  Used for the f32 stream only; the f64 parity stream keeps the reference's six quads.
    Flattener(const rt_scene_desc &d, Flat<R> &out, std::string &err, bool fuse_boxes)
        : d_(d), f_(out), err_(err), fuse_(fuse_boxes) {}

    int run() {
        int rc = validate();
        if (rc) return rc;
        convert_tables();
        // FLAT float BOX records are widened to what float can resolve at the scale of the scene.  The reference pads flat
        // boxes to 1e-4 (aabb.h:74-81: a wall quad's box); in float, with coordinates of several hundred, that is thinner
        // than one ulp of the slab distances (b * (1/d) - o * (1/d), each part ~1e2: 8e-6 apart at best): the near and
        // far distances of such a box round to the same number and the box -- with the wall in it -- is missed.  A side
        // thinner than 64 ulp of twice the largest world coordinate gets 32 ulp on either end (box_hit also accepts
        // near == far).  Only flat sides: widening EVERY box, even by 4 ulp, lets each ray that leaves a surface re-enter
        // the box of the object it just left (its exit distance then exceeds t_min = 0.001): +1 % frame time measured.
        if constexpr (sizeof(R) == 4) {
            double m = 0;
            for (int a = 0; a < 6; a++) {
                const double v = std::fabs(d_.nodes[d_.world_root].bbox[a]);
                if (std::isfinite(v)) m = std::fmax(m, v);
            }
            box_pad_ = 32.0 * 5.9604644775390625e-08 * 2.0 * m;
            if (const char *e = std::getenv("RT_AMD_BOX_PAD_ULPS")) box_pad_ = std::atof(e) * 5.9604644775390625e-08 * 2.0 * m; // development knob
        }
        f_.xforms.push_back(XformT<R>{}); // context 0 = world space
        rc = emit(d_.world_root, 0, 0);
        if (rc) return rc;
        // end-of-stream sentinel: every skip link that leaves the last subtree lands here, and its header says
        // "traversal finished", so a traversal step never needs a bounds check
        const int end = push_units(2, -1);
        f_.stream[end + 1].w[3] = I(make_hdr(K_READY, 0));
        // boundaries of constant_mediums that are not a static sphere: one sub-stream each, behind the main stream, in the
        // transform context of its medium and with a sentinel of its own (constant_medium.h:24-28 runs boundary->hit twice)
        for (size_t k = 0; k < pending_.size(); k++) {
            const Pending pd = pending_[k];
            f_.stream[pd.record].w[0] = I(pos());
            rc = emit(pd.boundary, pd.ctx, 0);
            if (rc) return rc;
            const int e2 = push_units(2, -1);
            f_.stream[e2 + 1].w[3] = I(make_hdr(K_READY, 0));
        }
        f_.has_gen_medium = pending_.empty() ? 0 : 1;
        return emit_lights();
    }

  private:
    const rt_scene_desc &d_;
    Flat<R> &f_;
    std::string &err_;
    bool fuse_;
    struct Pending { // a MEDIUM_GEN record waiting for its boundary sub-stream
        int record, boundary, ctx;
    };
    std::vector<Pending> pending_;
    double box_pad_ = 0; // float streams: see run()
    static constexpr size_t kMaxUnits = size_t(1) << 25; // positions are byte offsets in an int
    int pos() const { return (int)(f_.stream.size() * sizeof(UnitT<R>)); } // where the next record will stand

    int fail(int code, const std::string &msg) {
        err_ = msg;
        return code;
    }

    int validate() {
        if (d_.abi_version != RT_AMD_ABI_VERSION) return fail(RT_ERR_INVALID, "rt_scene_desc.abi_version mismatch");
        if (d_.n_nodes < 0 || (d_.n_nodes > 0 && !d_.nodes)) return fail(RT_ERR_INVALID, "bad node table");
        if (d_.world_root < 0 || d_.world_root >= d_.n_nodes) return fail(RT_ERR_INVALID, "world_root out of range");
        if (d_.lights_root < 0 || d_.lights_root >= d_.n_nodes) return fail(RT_ERR_INVALID, "lights_root out of range");
        auto tab = [&](int n, const void *p) { return n >= 0 && (n == 0 || p != nullptr); };
        if (!tab(d_.n_child_index, d_.child_index) || !tab(d_.n_spheres, d_.spheres) || !tab(d_.n_quads, d_.quads) ||
            !tab(d_.n_triangles, d_.triangles) || !tab(d_.n_materials, d_.materials) || !tab(d_.n_textures, d_.textures) ||
            !tab(d_.n_images, d_.images) || !tab(d_.n_perlins, d_.perlins) ||
            d_.n_image_bytes < 0 || (d_.n_image_bytes > 0 && !d_.image_bytes))
            return fail(RT_ERR_INVALID, "bad table pointer/count");
        for (int i = 0; i < d_.n_nodes; i++) {
            const rt_node &n = d_.nodes[i];
            auto node_ok = [&](int id) { return id >= 0 && id < d_.n_nodes; };
            auto mat_ok = [&](int m) { return m >= -1 && m < d_.n_materials; };
            bool ok = true;
            switch (n.kind) {
            case RT_NODE_BVH: ok = node_ok(n.a) && node_ok(n.b); break;
            case RT_NODE_LIST:
                ok = n.b >= 0 && n.a >= 0 && (int64_t)n.a + n.b <= d_.n_child_index;
                for (int k = 0; ok && k < n.b; k++) ok = node_ok(d_.child_index[n.a + k]);
                break;
            case RT_NODE_TRANSLATE:
            case RT_NODE_ROTATE_Y: ok = node_ok(n.a); break;
            case RT_NODE_MEDIUM: ok = node_ok(n.a) && n.b >= 0 && n.b < d_.n_materials; break;
            case RT_NODE_SPHERE: ok = n.a >= 0 && n.a < d_.n_spheres && mat_ok(d_.spheres[n.a].mat); break;
            case RT_NODE_QUAD: ok = n.a >= 0 && n.a < d_.n_quads && mat_ok(d_.quads[n.a].mat); break;
            case RT_NODE_TRIANGLE: ok = n.a >= 0 && n.a < d_.n_triangles && mat_ok(d_.triangles[n.a].mat); break;
            default: ok = false;
            }
            if (!ok) return fail(RT_ERR_INVALID, "node " + std::to_string(i) + ": bad kind or index");
        }
        // The graph must be acyclic (shared subtrees are fine: the reference's final_scene lists one boundary sphere twice).
        // Iterative three-colour depth-first search from both roots: a child that is still open when it is met again
        // closes a cycle.
        {
            std::vector<uint8_t> colour((size_t)d_.n_nodes, 0); // 0 new, 1 open, 2 done
            std::vector<std::pair<int, int>> stack;             // (node, next child slot)
            auto n_children = [&](const rt_node &n) {
                switch (n.kind) {
                case RT_NODE_BVH: return 2;
                case RT_NODE_LIST: return (int)n.b;
                case RT_NODE_TRANSLATE:
                case RT_NODE_ROTATE_Y:
                case RT_NODE_MEDIUM: return 1;
                default: return 0;
                }
            };
            auto child = [&](const rt_node &n, int k) {
                if (n.kind == RT_NODE_BVH) return k == 0 ? n.a : n.b;
                if (n.kind == RT_NODE_LIST) return (int)d_.child_index[n.a + k];
                return n.a;
            };
            for (int root : {d_.world_root, d_.lights_root}) {
                if (colour[root]) continue;
                colour[root] = 1;
                stack.push_back({root, 0});
                while (!stack.empty()) {
                    auto &[id, k] = stack.back();
                    const rt_node &n = d_.nodes[id];
                    if (k == n_children(n)) {
                        colour[id] = 2;
                        stack.pop_back();
                        continue;
                    }
                    const int c = child(n, k++);
                    if (colour[c] == 1) return fail(RT_ERR_INVALID, "scene graph has a cycle through node " + std::to_string(c));
                    if (colour[c] == 0) {
                        colour[c] = 1;
                        stack.push_back({c, 0});
                    }
                }
            }
        }
        for (int i = 0; i < d_.n_materials; i++) {
            const rt_material &m = d_.materials[i];
            if (m.kind < RT_MAT_LAMBERTIAN || m.kind > RT_MAT_ISOTROPIC) return fail(RT_ERR_INVALID, "bad material kind");
            bool needs_tex = m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_DIFFUSE_LIGHT || m.kind == RT_MAT_ISOTROPIC;
            if (needs_tex && (m.tex < 0 || m.tex >= d_.n_textures)) return fail(RT_ERR_INVALID, "material texture out of range");
        }
        for (int i = 0; i < d_.n_textures; i++) {
            const rt_texture &t = d_.textures[i];
            bool ok = true;
            switch (t.kind) {
            case RT_TEX_SOLID: break;
            case RT_TEX_CHECKER: ok = t.a >= 0 && t.a < d_.n_textures && t.b >= 0 && t.b < d_.n_textures; break;
            case RT_TEX_IMAGE: ok = t.a >= -1 && t.a < d_.n_images; break;
            case RT_TEX_NOISE: ok = t.a >= 0 && t.a < d_.n_perlins; break;
            default: ok = false;
            }
            if (!ok) return fail(RT_ERR_INVALID, "texture " + std::to_string(i) + ": bad kind or index");
        }
        for (int i = 0; i < d_.n_images; i++) {
            const rt_image &im = d_.images[i];
            if (im.width < 0 || im.height < 0 || im.offset < 0 ||
                im.offset + (int64_t)im.width * im.height * 3 > d_.n_image_bytes)
                return fail(RT_ERR_INVALID, "image out of range");
        }
        return RT_OK;
    }

    bool contains_medium(int id, int rdepth) const {
        if (rdepth > 256) return false;
        const rt_node &n = d_.nodes[id];
        switch (n.kind) {
        case RT_NODE_MEDIUM: return true;
        case RT_NODE_BVH: return contains_medium(n.a, rdepth + 1) || contains_medium(n.b, rdepth + 1);
        case RT_NODE_LIST:
            for (int k = 0; k < n.b; k++)
                if (contains_medium(d_.child_index[n.a + k], rdepth + 1)) return true;
            return false;
        case RT_NODE_TRANSLATE:
        case RT_NODE_ROTATE_Y: return contains_medium(n.a, rdepth + 1);
        default: return false;
        }
    }

    bool tex_needs_uv(int tex, int guard = 0) const {
        if (tex < 0 || guard > 16) return false;
        const rt_texture &t = d_.textures[tex];
        if (t.kind == RT_TEX_IMAGE) return true;
        if (t.kind == RT_TEX_CHECKER) return tex_needs_uv(t.a, guard + 1) || tex_needs_uv(t.b, guard + 1);
        return false;
    }

    static SphereT<R> cvt_sphere(const rt_sphere &s) {
        SphereT<R> o{};
        cvt3(s.center, o.c);
        cvt3(s.dcenter, o.d);
        cvt(s.radius, o.radius);
        o.mat = s.mat;
        return o;
    }
    static QuadT<R> cvt_quad(const rt_quad &q) {
        QuadT<R> o{};
        cvt3(q.normal, o.n);
        cvt(q.D, o.D);
        cvt3(q.Q, o.Q);
        cvt(q.area, o.area);
        cvt3(q.u, o.u);
        cvt3(q.v, o.v);
        cvt3(q.w, o.w);
        o.mat = q.mat;
        return o;
    }

    void convert_tables() {
        for (int i = 0; i < d_.n_materials; i++) {
            const rt_material &m = d_.materials[i];
            MatT<R> o{};
            o.kind = m.kind;
            o.tex = m.tex;
            o.needs_uv = tex_needs_uv(m.tex) ? 1 : 0;
            cvt3(m.albedo, o.albedo);
            cvt(m.fuzz, o.fuzz);
            cvt(m.ior, o.ior);
            f_.mats.push_back(o);
        }
        for (int i = 0; i < d_.n_textures; i++) {
            const rt_texture &t = d_.textures[i];
            TexT<R> o{};
            o.kind = t.kind;
            o.a = t.a;
            o.b = t.b;
            cvt3(t.color, o.color);
            cvt(t.scale, o.scale);
            f_.texs.push_back(o);
        }
        for (int i = 0; i < d_.n_images; i++) f_.images.push_back(ImageT{d_.images[i].width, d_.images[i].height, d_.images[i].offset});
        f_.perlins.resize(d_.n_perlins);
        for (int i = 0; i < d_.n_perlins; i++) {
            const rt_perlin &p = d_.perlins[i];
            PerlinT<R> &o = f_.perlins[i];
            for (int k = 0; k < 256; k++) {
                cvt3(p.randvec[k], o.randvec[k]);
                o.perm_x[k] = p.perm_x[k] & 255;
                o.perm_y[k] = p.perm_y[k] & 255;
                o.perm_z[k] = p.perm_z[k] & 255;
            }
        }
    }

    static R I(int v) { return enc_int(v, R(0)); }
    static R F(double x) {
        R o;
        cvt(x, o);
        return o;
    }

    // append a record of `n` units; returns its unit offset
    int push_units(int n, int node_id) {
        int off = (int)f_.stream.size();
        UnitT<R> z{};
        f_.stream.insert(f_.stream.end(), n, z);
        f_.cold.insert(f_.cold.end(), n, z);
        f_.cold[off].w[3] = I(node_id);
        return off;
    }

    // Is description node `id` a hittable_list of six axis-aligned quads that form a closed box with one material?
    // On success fills lo/hi, the material, face_id[2*axis+side] (description node of that face) and the bit mask of
    // faces whose stored normal points out of the box.
    bool match_box(int id, double lo[3], double hi[3], int &mat, int face_id[6], int &outward_mask) const {
        const rt_node &n = d_.nodes[id];
        if (n.kind != RT_NODE_LIST || n.b != 6) return false;
        for (int a = 0; a < 3; a++) {
            lo[a] = INFINITY;
            hi[a] = -INFINITY;
        }
        mat = -2;
        int axis_of[6];
        for (int k = 0; k < 6; k++) {
            const rt_node &c = d_.nodes[d_.child_index[n.a + k]];
            if (c.kind != RT_NODE_QUAD) return false;
            const rt_quad &q = d_.quads[c.a];
            if (q.mat < 0 || (mat != -2 && q.mat != mat)) return false;
            mat = q.mat;
            int au = -1, av = -1;
            for (int a = 0; a < 3; a++) {
                if (q.u[a] != 0.0) {
                    if (au >= 0) return false;
                    au = a;
                }
                if (q.v[a] != 0.0) {
                    if (av >= 0) return false;
                    av = a;
                }
            }
            if (au < 0 || av < 0 || au == av) return false;
            axis_of[k] = 3 - au - av;
            for (int a = 0; a < 3; a++) {
                const double c0 = q.Q[a], c1 = q.Q[a] + q.u[a], c2 = q.Q[a] + q.v[a], c3 = q.Q[a] + q.u[a] + q.v[a];
                lo[a] = std::fmin(lo[a], std::fmin(std::fmin(c0, c1), std::fmin(c2, c3)));
                hi[a] = std::fmax(hi[a], std::fmax(std::fmax(c0, c1), std::fmax(c2, c3)));
            }
        }
        if (tex_needs_uv(d_.materials[mat].tex)) return false; // rec.u, rec.v of the faces would be needed
        double ext = 0;
        for (int a = 0; a < 3; a++) {
            if (!(hi[a] > lo[a])) return false;
            ext = std::fmax(ext, std::fmax(std::fabs(lo[a]), std::fabs(hi[a])));
        }
        const double tol = 1e-9 * ext;
        for (int f = 0; f < 6; f++) face_id[f] = -1;
        outward_mask = 0;
        for (int k = 0; k < 6; k++) {
            const int cid = d_.child_index[n.a + k];
            const rt_quad &q = d_.quads[d_.nodes[cid].a];
            const int c = axis_of[k];
            int side;
            if (std::fabs(q.Q[c] - lo[c]) <= tol) side = 0;
            else if (std::fabs(q.Q[c] - hi[c]) <= tol) side = 1;
            else return false;
            for (int a = 0; a < 3; a++) { // the face must span the whole box in its two in-plane axes
                if (a == c) continue;
                const double e0 = std::fmin(q.Q[a], q.Q[a] + q.u[a] + q.v[a]), e1 = std::fmax(q.Q[a], q.Q[a] + q.u[a] + q.v[a]);
                if (std::fabs(e0 - lo[a]) > tol || std::fabs(e1 - hi[a]) > tol) return false;
            }
            const int f = 2 * c + side;
            if (face_id[f] >= 0) return false;
            face_id[f] = cid;
            const bool outward = side ? q.normal[c] > 0 : q.normal[c] < 0;
            if (outward) outward_mask |= 1 << f;
        }
        return true;
    }

    // ---- baked instances (f32 stream only) --------------------------------------------------------------------------
    // translate / rotate_y over a subtree that holds nothing but triangles and static spheres whose materials need no
    // (u, v): a sphere under a rigid transform is a sphere, a triangle a triangle, so the subtree is emitted in world space (same tree, same visiting order; BOX
    // records get the bounds of their transformed children) and the ray never has to change frames: no XF_ENTER /
    // XF_EXIT records, no second set of inverse directions.  object -> world: p = Ry(p) + t with
    // Ry(x, y, z) = (c x + s z, y, -s x + c z) (hittable.h:135-145), which translate and rotate_y compose within.
    struct Rigid {
        double c = 1, s = 0, t[3] = {0, 0, 0};
        void apply(const double p[3], double o[3]) const {
            o[0] = c * p[0] + s * p[2] + t[0];
            o[1] = p[1] + t[1];
            o[2] = -s * p[0] + c * p[2] + t[2];
        }
    };
    bool bakeable(int id, int rdepth = 0) const {
        if (rdepth > 256) return false;
        const rt_node &n = d_.nodes[id];
        switch (n.kind) {
        case RT_NODE_TRANSLATE:
        case RT_NODE_ROTATE_Y: return bakeable(n.a, rdepth + 1);
        case RT_NODE_BVH: return bakeable(n.a, rdepth + 1) && (n.b == n.a || bakeable(n.b, rdepth + 1));
        case RT_NODE_LIST:
            for (int k = 0; k < n.b; k++)
                if (!bakeable(d_.child_index[n.a + k], rdepth + 1)) return false;
            return n.b > 0;
        case RT_NODE_SPHERE: {
            const rt_sphere &sp = d_.spheres[n.a];
            return sp.mat >= 0 && sp.dcenter[0] == 0 && sp.dcenter[1] == 0 && sp.dcenter[2] == 0 &&
                   !tex_needs_uv(d_.materials[sp.mat].tex);
        }
        case RT_NODE_TRIANGLE: return d_.triangles[n.a].mat >= 0; // (u, v) of a triangle are barycentric: frame-independent
        default: return false;
        }
    }
    // emits the subtree under transform `x`; lo/hi = its world bounds
    int emit_baked(int id, const Rigid &x, int rdepth, double lo[3], double hi[3]) {
        if (f_.stream.size() > kMaxUnits) return fail(RT_ERR_NOMEM, "flattened scene too large");
        const rt_node &n = d_.nodes[id];
        auto grow = [&](const double l[3], const double h[3]) {
            for (int a = 0; a < 3; a++) {
                lo[a] = std::fmin(lo[a], l[a]);
                hi[a] = std::fmax(hi[a], h[a]);
            }
        };
        for (int a = 0; a < 3; a++) {
            lo[a] = INFINITY;
            hi[a] = -INFINITY;
        }
        switch (n.kind) {
        case RT_NODE_TRANSLATE: {
            Rigid y = x; // x o translate: p -> Ry(p + off) + t
            const double z[3] = {0, 0, 0};
            double r0[3], r1[3];
            Rigid rot = x;
            rot.t[0] = rot.t[1] = rot.t[2] = 0;
            rot.apply(n.p, r0);
            rot.apply(z, r1);
            for (int a = 0; a < 3; a++) y.t[a] = x.t[a] + (r0[a] - r1[a]);
            return emit_baked(n.a, y, rdepth + 1, lo, hi);
        }
        case RT_NODE_ROTATE_Y: { // x o rotate: the angles add
            Rigid y = x;
            const double sn = n.p[0], cs = n.p[1];
            y.c = x.c * cs - x.s * sn;
            y.s = x.s * cs + x.c * sn;
            return emit_baked(n.a, y, rdepth + 1, lo, hi);
        }
        case RT_NODE_BVH: {
            const int self = push_units(2, id);
            double l[3], h[3];
            int rc = emit_baked(n.a, x, rdepth + 1, l, h);
            if (rc) return rc;
            grow(l, h);
            if (n.b != n.a) {
                rc = emit_baked(n.b, x, rdepth + 1, l, h);
                if (rc) return rc;
                grow(l, h);
            }
            R rl[3], rh[3];
            for (int a = 0; a < 3; a++) {
                const double pad = (hi[a] - lo[a]) < 2 * box_pad_ ? box_pad_ : 0.0;
                cvt_lo(lo[a] - pad, rl[a]);
                cvt_hi(hi[a] + pad, rh[a]);
            }
            f_.stream[self] = UnitT<R>{{rl[0], rh[0], rl[1], rh[1]}};
            f_.stream[self + 1] = UnitT<R>{{rl[2], rh[2], I(pos()), I(make_hdr(K_BOX, 0))}};
            return RT_OK;
        }
        case RT_NODE_LIST:
            for (int k = 0; k < n.b; k++) {
                double l[3], h[3];
                int rc = emit_baked(d_.child_index[n.a + k], x, rdepth + 1, l, h);
                if (rc) return rc;
                grow(l, h);
            }
            return RT_OK;
        case RT_NODE_TRIANGLE: {
            const rt_triangle &t = d_.triangles[n.a];
            Rigid rot = x;
            rot.t[0] = rot.t[1] = rot.t[2] = 0;
            double w0[3], w1[3], w2[3], nw[3];
            x.apply(t.v0, w0);
            x.apply(t.v1, w1);
            x.apply(t.v2, w2);
            rot.apply(t.normal, nw);
            const double e1[3] = {w1[0] - w0[0], w1[1] - w0[1], w1[2] - w0[2]};
            const double e2[3] = {w2[0] - w0[0], w2[1] - w0[1], w2[2] - w0[2]};
            const int off = push_units(3, id);
            f_.stream[off] = UnitT<R>{{F(w0[0]), F(w0[1]), F(w0[2]), F(e1[0])}};
            f_.stream[off + 1] = UnitT<R>{{F(e1[1]), F(e1[2]), F(e2[0]), I(make_hdr(K_TRI, 0))}};
            f_.stream[off + 2] = UnitT<R>{{F(e2[1]), F(e2[2]), I(t.mat), I(t.cull)}};
            f_.cold[off] = UnitT<R>{{F(nw[0]), F(nw[1]), F(nw[2]), I(id)}};
            for (int a = 0; a < 3; a++) {
                const double mn = std::fmin(w0[a], std::fmin(w1[a], w2[a])), mx = std::fmax(w0[a], std::fmax(w1[a], w2[a]));
                const double pad = 1e-6 * (std::fabs(mn) + std::fabs(mx)) + 5e-5; // aabb.h:77-80 pads thin boxes by 5e-5 too
                lo[a] = mn - pad;
                hi[a] = mx + pad;
            }
            return RT_OK;
        }
        default: { // RT_NODE_SPHERE (bakeable() admitted nothing else)
            const rt_sphere &sp = d_.spheres[n.a];
            double cw[3];
            x.apply(sp.center, cw);
            const int off = push_units(2, id);
            f_.stream[off] = UnitT<R>{{F(cw[0]), F(cw[1]), F(cw[2]), F(sp.radius)}};
            f_.stream[off + 1] = UnitT<R>{{I(sp.mat), I(0), I(0), I(make_hdr(K_SPHERE, 0))}};
            const double rr = std::fabs(sp.radius) * (1.0 + 1e-6) + 1e-6 * (std::fabs(cw[0]) + std::fabs(cw[1]) + std::fabs(cw[2]));
            for (int a = 0; a < 3; a++) {
                lo[a] = cw[a] - rr;
                hi[a] = cw[a] + rr;
            }
            return RT_OK;
        }
        }
    }

    // Emit the subtree of description node `id` in the reference's visiting order.
    int emit(int id, int ctx, int rdepth) {
        if (rdepth > 256) return fail(RT_ERR_INVALID, "scene graph too deep (cycle?)");
        if (f_.stream.size() > kMaxUnits) return fail(RT_ERR_NOMEM, "flattened scene too large");
        const rt_node &n = d_.nodes[id];
        switch (n.kind) {
        case RT_NODE_BVH: { // bvh.h:43-51: own box, then left, then right
            const int self = push_units(2, id);
            R lo[3], hi[3];
            for (int a = 0; a < 3; a++) {
                const double pad = (n.bbox[2 * a + 1] - n.bbox[2 * a]) < 2 * box_pad_ ? box_pad_ : 0.0;
                cvt_lo(n.bbox[2 * a] - pad, lo[a]);
                cvt_hi(n.bbox[2 * a + 1] + pad, hi[a]);
            }
            f_.stream[self] = UnitT<R>{{lo[0], hi[0], lo[1], hi[1]}};
            f_.stream[self + 1] = UnitT<R>{{lo[2], hi[2], I(0), I(make_hdr(K_BOX, 0))}};
            int rc = emit(n.a, ctx, rdepth + 1);
            if (rc) return rc;
            // span-1 nodes have left == right (bvh.h:30) and the reference visits the child twice.  For geometry the second
            // visit cannot change the result; a constant_medium draws a NEW uniform on it (constant_medium.h:41), i.e. the
            // reference scatters in such a medium at the closer of two free paths -- so that child is emitted twice.
            if (n.b != n.a || contains_medium(n.a, 0)) {
                rc = emit(n.b, ctx, rdepth + 1);
                if (rc) return rc;
            }
            f_.stream[self + 1].w[2] = I(pos());
            return RT_OK;
        }
        case RT_NODE_LIST: { // hittable_list.h:29: children in insertion order, each against the closest hit so far
            double lo[3], hi[3];
            int mat, face_id[6], outward;
            if (fuse_ && match_box(id, lo, hi, mat, face_id, outward)) {
                const int off = push_units(2, id);
                f_.stream[off] = UnitT<R>{{F(lo[0]), F(hi[0]), F(lo[1]), F(hi[1])}};
                f_.stream[off + 1] = UnitT<R>{{F(lo[2]), F(hi[2]), I(mat), I(make_hdr(K_AABOX, outward))}};
                f_.cold[off] = UnitT<R>{{I(face_id[0]), I(face_id[1]), I(face_id[2]), I(face_id[3])}};
                f_.cold[off + 1] = UnitT<R>{{I(face_id[4]), I(face_id[5]), I(0), I(0)}};
                f_.n_fused_boxes++;
                return RT_OK;
            }
            for (int k = 0; k < n.b; k++) {
                int rc = emit(d_.child_index[n.a + k], ctx, rdepth + 1);
                if (rc) return rc;
            }
            return RT_OK;
        }
        case RT_NODE_TRANSLATE:
        case RT_NODE_ROTATE_Y: {
            if (fuse_ && ctx == 0 && bakeable(id)) {
                double lo[3], hi[3];
                f_.n_baked_instances++;
                return emit_baked(id, Rigid{}, rdepth, lo, hi);
            }
            const XformT<R> parent = f_.xforms[ctx];
            const int pdepth = ctx == 0 ? 0 : parent.depth;
            if (pdepth + 1 > MAX_XF_DEPTH)
                return fail(RT_ERR_UNSUPPORTED, "more than 4 nested translate/rotate_y instances");
            XformT<R> x{};
            x.kind = n.kind == RT_NODE_TRANSLATE ? XF_TRANSLATE : XF_ROTATE_Y;
            x.parent = ctx;
            x.depth = pdepth + 1;
            int c = (int)f_.xforms.size();
            for (int k = 0; k < pdepth; k++) x.chain[k] = parent.chain[k];
            x.chain[pdepth] = c;
            if (n.kind == RT_NODE_TRANSLATE) {
                cvt3(n.p, x.p);
            } else {
                cvt(n.p[0], x.p[0]);
                cvt(n.p[1], x.p[1]);
            }
            f_.xforms.push_back(x);
            const int en = push_units(2, id);
            f_.stream[en] = UnitT<R>{{x.p[0], x.p[1], x.p[2], R(0)}};
            f_.stream[en + 1] = UnitT<R>{{I(x.kind), I(c), I(0), I(make_hdr(K_XF_ENTER, 0))}};
            int rc = emit(n.a, c, rdepth + 1);
            if (rc) return rc;
            const int ex = push_units(2, id);
            f_.stream[ex] = UnitT<R>{{x.p[0], x.p[1], x.p[2], R(0)}}; // the float path leaves the frame by the inverse transform
            f_.stream[ex + 1] = UnitT<R>{{I(x.kind), I(ctx), I(0), I(make_hdr(K_XF_EXIT, 0))}};
            return RT_OK;
        }
        case RT_NODE_MEDIUM: {
            const rt_node &b = d_.nodes[n.a];
            const bool static_sphere = b.kind == RT_NODE_SPHERE && d_.spheres[b.a].dcenter[0] == 0 && d_.spheres[b.a].dcenter[1] == 0 &&
                                       d_.spheres[b.a].dcenter[2] == 0;
            if (!static_sphere) { // any other hittable: its records become a sub-stream that the MEDIUM_GEN step walks twice
                if (contains_medium(n.a, 0)) return fail(RT_ERR_UNSUPPORTED, "constant_medium inside the boundary of a constant_medium");
                const int off = push_units(2, id);
                f_.stream[off] = UnitT<R>{{I(0), R(0), R(0), R(0)}};
                f_.stream[off + 1] = UnitT<R>{{F(n.p[0]), I(n.b), I(id), I(make_hdr(K_MEDIUM_GEN, 0))}};
                pending_.push_back(Pending{off, n.a, ctx});
                return RT_OK;
            }
            const rt_sphere &s = d_.spheres[b.a];
            const int off = push_units(2, id);
            f_.stream[off] = UnitT<R>{{F(s.center[0]), F(s.center[1]), F(s.center[2]), F(s.radius)}};
            f_.stream[off + 1] = UnitT<R>{{F(n.p[0]), I(n.b), I(id), I(make_hdr(K_MEDIUM, 0))}};
            return RT_OK;
        }
        case RT_NODE_SPHERE: {
            const rt_sphere &s = d_.spheres[n.a];
            if (s.mat < 0) return fail(RT_ERR_INVALID, "world sphere without material");
            const bool moving = s.dcenter[0] != 0 || s.dcenter[1] != 0 || s.dcenter[2] != 0;
            const int off = push_units(moving ? 3 : 2, id);
            f_.stream[off] = UnitT<R>{{F(s.center[0]), F(s.center[1]), F(s.center[2]), F(s.radius)}};
            f_.stream[off + 1] = UnitT<R>{{I(s.mat), I(0), I(0), I(make_hdr(moving ? K_MSPHERE : K_SPHERE, 0))}};
            if (moving) f_.stream[off + 2] = UnitT<R>{{F(s.dcenter[0]), F(s.dcenter[1]), F(s.dcenter[2]), R(0)}};
            return RT_OK;
        }
        case RT_NODE_QUAD: {
            const rt_quad &q = d_.quads[n.a];
            if (q.mat < 0) return fail(RT_ERR_INVALID, "world quad without material");
            const int off = push_units(5, id);
            f_.stream[off] = UnitT<R>{{F(q.normal[0]), F(q.normal[1]), F(q.normal[2]), F(q.D)}};
            f_.stream[off + 1] = UnitT<R>{{F(q.Q[0]), F(q.Q[1]), F(q.Q[2]), I(make_hdr(K_QUAD, 0))}};
            f_.stream[off + 2] = UnitT<R>{{F(q.u[0]), F(q.u[1]), F(q.u[2]), I(q.mat)}};
            f_.stream[off + 3] = UnitT<R>{{F(q.v[0]), F(q.v[1]), F(q.v[2]), F(q.area)}};
            f_.stream[off + 4] = UnitT<R>{{F(q.w[0]), F(q.w[1]), F(q.w[2]), R(0)}};
            return RT_OK;
        }
        case RT_NODE_TRIANGLE: {
            const rt_triangle &t = d_.triangles[n.a];
            if (t.mat < 0) return fail(RT_ERR_INVALID, "world triangle without material");
            // triangle.h:24-25 computes the edges from the double vertices at every hit(); the same subtraction here
            const double e1[3] = {t.v1[0] - t.v0[0], t.v1[1] - t.v0[1], t.v1[2] - t.v0[2]};
            const double e2[3] = {t.v2[0] - t.v0[0], t.v2[1] - t.v0[1], t.v2[2] - t.v0[2]};
            const int off = push_units(3, id);
            f_.stream[off] = UnitT<R>{{F(t.v0[0]), F(t.v0[1]), F(t.v0[2]), F(e1[0])}};
            f_.stream[off + 1] = UnitT<R>{{F(e1[1]), F(e1[2]), F(e2[0]), I(make_hdr(K_TRI, 0))}};
            f_.stream[off + 2] = UnitT<R>{{F(e2[1]), F(e2[2]), I(t.mat), I(t.cull)}};
            f_.cold[off] = UnitT<R>{{F(t.normal[0]), F(t.normal[1]), F(t.normal[2]), I(id)}};
            return RT_OK;
        }
        }
        return fail(RT_ERR_INVALID, "bad node kind");
    }

    LightT<R> light_of(int id) {
        const rt_node &n = d_.nodes[id];
        LightT<R> L{};
        if (n.kind == RT_NODE_SPHERE) {
            L.kind = LIGHT_SPHERE;
            L.s = cvt_sphere(d_.spheres[n.a]);
        } else if (n.kind == RT_NODE_QUAD) {
            L.kind = LIGHT_QUAD;
            L.q = cvt_quad(d_.quads[n.a]);
        } else {
            L.kind = LIGHT_OTHER; // hittable.h:35-41 defaults: pdf_value 0, random (1,0,0)
        }
        return L;
    }

    int emit_lights() {
        const rt_node &n = d_.nodes[d_.lights_root];
        if (n.kind == RT_NODE_LIST) {
            if (n.b == 0) return fail(RT_ERR_INVALID, "lights list is empty (the reference would index objects[-1])");
            f_.lights_is_list = 1;
            for (int k = 0; k < n.b; k++) {
                int c = d_.child_index[n.a + k];
                if (d_.nodes[c].kind == RT_NODE_LIST)
                    return fail(RT_ERR_UNSUPPORTED, "nested hittable_list inside lights");
                f_.lights.push_back(light_of(c));
            }
        } else {
            f_.lights_is_list = 0;
            f_.lights.push_back(light_of(d_.lights_root));
        }
        return RT_OK;
    }
};

} // namespace rtamd
#endif
