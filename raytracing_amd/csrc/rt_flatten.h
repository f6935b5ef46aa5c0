// rt_flatten.h -- host side: turn an rt_scene_desc (the reference's object graph as plain structs) into the threaded
// device layout of rt_scene.h.  Pure C++ (no HIP); runs once per scene upload.
#ifndef RT_FLATTEN_H
#define RT_FLATTEN_H

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <numeric>
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

// Settings of the float stream's hierarchy build (Flattener::build).  The defaults are the best single setting over the
// reference's five configurations; rt_scene_tune (rt_abi.cpp) times a few others on the caller's camera and keeps the fastest.
struct BuildKnobs {
    double thr2 = 0.8, thrn = 0.95; // a node of 2 / of more leaves gets a BOX record if its area is below this share of its parent's
    int order = 1;                  // which child goes first: see build()
    int bins = 12;
    double axis_min = 0;            // only axes whose centroid extent is at least this share of the widest are split
    double overlap = 0.1;           // weight of the area the two sides of a split share, in the split's cost (0 -> 0.1:
                                    // final_scene 635 -> 623, cornell_box 105 -> 103, mesh_scene 477 -> 471; 1.0: 665)
    uint32_t jitter = 0;            // != 0: seed of a randomised build (see build(): bin grid shifted per node, near-ties broken at random)
    double jitter_eps = 0.03;       //   candidates within this share of the cheapest split count as ties
    BuildKnobs() { // development knobs
        if (const char *e = dev_knob("RT_AMD_BVH_JITTER")) jitter = (uint32_t)std::atoll(e);
        if (const char *e = dev_knob("RT_AMD_BVH_EPS")) jitter_eps = std::atof(e);
        if (const char *e = dev_knob("RT_AMD_BVH_T2")) thr2 = std::atof(e);
        if (const char *e = dev_knob("RT_AMD_BVH_TN")) thrn = std::atof(e);
        if (const char *e = dev_knob("RT_AMD_BVH_ORDER")) order = std::atoi(e);
        if (const char *e = dev_knob("RT_AMD_BVH_AXIS_MIN")) axis_min = std::atof(e);
        if (const char *e = dev_knob("RT_AMD_BVH_OVERLAP")) overlap = std::atof(e);
        if (const char *e = dev_knob("RT_AMD_BVH_BINS")) bins = std::min(32, std::max(2, std::atoi(e)));
    }
};

template <class R> class Flattener {
  public:
    // fuse_boxes: replace a hittable_list of six axis-aligned quads forming a closed box (quad.h:103 box()) by one
    // AABOX record.  Used for the f32 stream only; the f64 parity stream keeps the reference's six quads.
    // own_hierarchy (f32 stream only): the stream's BOX records are not the reference's bvh_nodes but a surface-area-
    // heuristic hierarchy built here over ALL primitives of a frame (see emit_group()).
    Flattener(const rt_scene_desc &d, Flat<R> &out, std::string &err, bool fuse_boxes, bool own_hierarchy = false, const BuildKnobs *knobs = nullptr)
        : d_(d), f_(out), err_(err), fuse_(fuse_boxes), rebuild_(own_hierarchy) {
        if (knobs) knobs_ = *knobs;
    }

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
            if (const char *e = dev_knob("RT_AMD_BOX_PAD_ULPS")) box_pad_ = std::atof(e) * 5.9604644775390625e-08 * 2.0 * m; // development knob
        }
        primary_ = primary_medium();
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

    // For the device build (rt_build.h): the leaf records alone -- what run() would put under its own hierarchy -- written one
    // after the other into f.stream / f.cold (no BOX records above them, no sentinel), with each leaf's place and bounds.  A
    // leaf may be several records: an instance that stays an instance (XF_ENTER, its own hierarchy in the object frame --
    // built here, on the host --, XF_EXIT), a constant_medium with a general boundary (its MEDIUM_GEN record; the boundaries'
    // sub-streams follow ALL leaves, from unit `sub_begin` on).  Links inside such a leaf and inside the sub-streams are
    // positions in THIS leaf stream: the device relocates them when it moves the records (rt_build.hip, emit_kernel).
    struct LeafRecord {
        int off, n;
        double lo[3], hi[3];
    };
    int run_leaves(std::vector<LeafRecord> &leaves, double &box_pad, int &sub_begin) {
        int rc = validate();
        if (rc) return rc;
        convert_tables();
        double m = 0;
        for (int a = 0; a < 6; a++) {
            const double v = std::fabs(d_.nodes[d_.world_root].bbox[a]);
            if (std::isfinite(v)) m = std::fmax(m, v);
        }
        box_pad_ = 32.0 * 5.9604644775390625e-08 * 2.0 * m; // as in run()
        box_pad = box_pad_;
        primary_ = primary_medium();
        f_.xforms.push_back(XformT<R>{});
        std::vector<Leaf> lv;
        rc = collect(d_.world_root, 0, Rigid{}, false, 0, 0, lv);
        if (rc) return rc;
        for (const Leaf &l : lv) {
            LeafRecord r;
            r.off = (int)f_.stream.size();
            rc = emit_leaf(l, 0, 0);
            if (rc) return rc;
            r.n = (int)f_.stream.size() - r.off;
            for (int a = 0; a < 3; a++) {
                r.lo[a] = l.lo[a];
                r.hi[a] = l.hi[a];
            }
            leaves.push_back(r);
        }
        sub_begin = (int)f_.stream.size();
        for (size_t k = 0; k < pending_.size(); k++) { // as in run(): a general medium's boundary, with a sentinel of its own
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
    bool rebuild_;
    int primary_ = -1; // the constant_medium that takes the fifth uniform (rt_kernels.h, Rng)
    int cur_dup_ = 0;  // second-visit path of the span-1 bvh_nodes above the node being emitted (rt_kernels.h, Rng)
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

    // The constant_medium whose draw is the fifth uniform of a generator block (rt_kernels.h, Rng; the same rule as
    // oracle/rt_oracle.cpp): the boundary box of largest surface area, the lowest node id among equals.
    int primary_medium() const {
        int best = -1;
        double best_area = -1.0;
        for (int i = 0; i < d_.n_nodes; i++) {
            const rt_node &n = d_.nodes[i];
            if (n.kind != RT_NODE_MEDIUM) continue;
            const double *b = d_.nodes[n.a].bbox;
            const double dx = b[1] - b[0], dy = b[3] - b[2], dz = b[5] - b[4];
            const double area = 2.0 * (dx * dy + dy * dz + dz * dx);
            if (area > best_area) {
                best_area = area;
                best = i;
            }
        }
        return best;
    }
    int medium_payload(int id) const { return ((id == primary_ && cur_dup_ == 0) ? 1 : 0) | (cur_dup_ << 1); }

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
            o.noise = -1; // (the render kernel evaluates a noise texture by the whole wave: it asks the material, not the texture)
            if ((m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_ISOTROPIC) && m.tex >= 0 && m.tex < d_.n_textures &&
                d_.textures[m.tex].kind == RT_TEX_NOISE)
                o.noise = d_.textures[m.tex].a;
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
    // translate(rotate_y(box())), rotate_y(box()) or translate(box()) -- the way the reference places a box (cpuRenderer.cpp:
    // cornell_box, cornell_smoke) -- with a box that match_box() accepts: one OBOX record (f32 stream).  sn, cs, t: the
    // instance's rotation (identity: 0, 1) and offset (0 if none).
    bool is_obox(int id) const {
        const int k = d_.nodes[id].kind;
        int box;
        double sn, cs, t[3];
        return fuse_ && (k == RT_NODE_TRANSLATE || k == RT_NODE_ROTATE_Y) && !dev_knob("RT_AMD_NO_OBOX") && match_obox(id, box, sn, cs, t);
    }
    bool match_obox(int id, int &box, double &sn, double &cs, double t[3]) const {
        const rt_node *n = &d_.nodes[id];
        sn = 0; cs = 1; t[0] = t[1] = t[2] = 0;
        if (n->kind == RT_NODE_TRANSLATE) {
            for (int a = 0; a < 3; a++) t[a] = n->p[a];
            n = &d_.nodes[n->a];
        }
        if (n->kind == RT_NODE_ROTATE_Y) {
            sn = n->p[0]; cs = n->p[1];
            n = &d_.nodes[n->a];
        }
        if (n->kind != RT_NODE_LIST) return false;
        box = (int)(n - d_.nodes);
        double lo[3], hi[3];
        int mat, face_id[6], outward;
        return match_box(box, lo, hi, mat, face_id, outward);
    }
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
        default: return baked_prim(id, x, lo, hi, true); // triangle or static sphere (bakeable() admitted nothing else)
        }
    }
    // One triangle or static sphere under the rigid transform `x`: its bounds in the outer frame (padded for the rounding
    // of the transform) and, if `do_emit`, its record.
    int baked_prim(int id, const Rigid &x, double lo[3], double hi[3], bool do_emit) {
        const rt_node &n = d_.nodes[id];
        if (n.kind == RT_NODE_TRIANGLE) {
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
            if (do_emit) {
                const int off = push_units(3, id);
                f_.stream[off] = UnitT<R>{{F(w0[0]), F(w0[1]), F(w0[2]), F(e1[0])}};
                f_.stream[off + 1] = UnitT<R>{{F(e1[1]), F(e1[2]), F(e2[0]), I(make_hdr(K_TRI, 0))}};
                f_.stream[off + 2] = UnitT<R>{{F(e2[1]), F(e2[2]), I(t.mat), I(t.cull)}};
                f_.cold[off] = UnitT<R>{{F(nw[0]), F(nw[1]), F(nw[2]), I(id)}};
            }
            for (int a = 0; a < 3; a++) {
                const double mn = std::fmin(w0[a], std::fmin(w1[a], w2[a])), mx = std::fmax(w0[a], std::fmax(w1[a], w2[a]));
                const double pad = 1e-6 * (std::fabs(mn) + std::fabs(mx)) + 5e-5; // aabb.h:77-80 pads thin boxes by 5e-5 too
                lo[a] = mn - pad;
                hi[a] = mx + pad;
            }
            return RT_OK;
        }
        const rt_sphere &sp = d_.spheres[n.a];
        double cw[3];
        x.apply(sp.center, cw);
        if (do_emit) {
            const int off = push_units(2, id);
            f_.stream[off] = UnitT<R>{{F(cw[0]), F(cw[1]), F(cw[2]), F(sp.radius)}};
            f_.stream[off + 1] = UnitT<R>{{I(sp.mat), I(0), I(0), I(make_hdr(K_SPHERE, 0))}};
        }
        const double rr = std::fabs(sp.radius) * (1.0 + 1e-6) + 1e-6 * (std::fabs(cw[0]) + std::fabs(cw[1]) + std::fabs(cw[2]));
        for (int a = 0; a < 3; a++) {
            lo[a] = cw[a] - rr;
            hi[a] = cw[a] + rr;
        }
        return RT_OK;
    }

    // ---- the float stream's own hierarchy ---------------------------------------------------------------------------
    // world.hit() is "the closest hit over all primitives"; which boxes are tested on the way is the reference's choice
    // (bvh.h:15-41: the longest axis of each node's box, the objects sorted along it and cut at the median; the world itself a flat list
    // of eleven objects in final_scene), not part of the result.  The float stream therefore gets a hierarchy built for
    // the walk the kernel does: every primitive of a frame -- whatever bvh_nodes and hittable_lists it sits under, baked
    // instances included -- becomes a leaf, and a binned surface-area-heuristic build decides the BOX records.  An
    // instance that cannot be baked is ONE leaf (XF_ENTER, its own hierarchy in the object frame, XF_EXIT); so is a
    // constant_medium (its draw is keyed by its record, see rt_kernels.h Rng, so the visiting order is free).  A node
    // whose box is nearly as large as its parent's gets no BOX record (its children stand in the parent's sequence).
    // The double stream keeps the reference's own nodes and order: it is the parity instrument.
    struct Leaf {
        int id;      // description node
        int dup;     // cur_dup_ for its emission
        bool baked;  // triangle / static sphere emitted in the outer frame under `x`
        Rigid x;
        double lo[3], hi[3];
    };
    static void b_empty(double lo[3], double hi[3]) {
        for (int a = 0; a < 3; a++) {
            lo[a] = INFINITY;
            hi[a] = -INFINITY;
        }
    }
    static void b_grow(double lo[3], double hi[3], const double l[3], const double h[3]) {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::fmin(lo[a], l[a]);
            hi[a] = std::fmax(hi[a], h[a]);
        }
    }
    static void b_pad_thin(double lo[3], double hi[3]) { // aabb.h:74-81: a side thinner than 0.0001 grows by half of that on either end
        for (int a = 0; a < 3; a++)
            if (hi[a] - lo[a] < 0.0001) {
                lo[a] -= 0.00005;
                hi[a] += 0.00005;
            }
    }
    static double b_area(const double lo[3], const double hi[3]) {
        const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0 && dy >= 0 && dz >= 0)) return 0.0;
        return 2.0 * (dx * dy + dy * dz + dz * dx);
    }
    // bounds of description node `id` in its own frame, from the geometry (the description's bbox fields are the caller's)
    int geom_bounds(int id, double lo[3], double hi[3], int rdepth) {
        if (rdepth > 256) return fail(RT_ERR_INVALID, "scene graph too deep (cycle?)");
        const rt_node &n = d_.nodes[id];
        b_empty(lo, hi);
        double l[3], h[3];
        switch (n.kind) {
        case RT_NODE_BVH: {
            int rc = geom_bounds(n.a, lo, hi, rdepth + 1);
            if (rc || n.b == n.a) return rc;
            rc = geom_bounds(n.b, l, h, rdepth + 1);
            b_grow(lo, hi, l, h);
            return rc;
        }
        case RT_NODE_LIST:
            for (int k = 0; k < n.b; k++) {
                int rc = geom_bounds(d_.child_index[n.a + k], l, h, rdepth + 1);
                if (rc) return rc;
                b_grow(lo, hi, l, h);
            }
            return RT_OK;
        case RT_NODE_TRANSLATE: {
            int rc = geom_bounds(n.a, lo, hi, rdepth + 1);
            for (int a = 0; a < 3; a++) {
                lo[a] += n.p[a];
                hi[a] += n.p[a];
            }
            return rc;
        }
        case RT_NODE_ROTATE_Y: { // hittable.h:86-107: the box of the eight rotated corners
            int rc = geom_bounds(n.a, l, h, rdepth + 1);
            if (rc) return rc;
            if (!(l[0] <= h[0])) return RT_OK; // empty
            const double sn = n.p[0], cs = n.p[1];
            for (int c = 0; c < 8; c++) {
                const double x = (c & 1) ? h[0] : l[0], y = (c & 2) ? h[1] : l[1], z = (c & 4) ? h[2] : l[2];
                const double q[3] = {cs * x + sn * z, y, -sn * x + cs * z};
                b_grow(lo, hi, q, q);
            }
            return RT_OK;
        }
        case RT_NODE_MEDIUM: return geom_bounds(n.a, lo, hi, rdepth + 1);
        case RT_NODE_SPHERE: {
            const rt_sphere &s = d_.spheres[n.a];
            const double r = std::fabs(s.radius);
            for (int a = 0; a < 3; a++) { // sphere.h:14-27: the boxes at time 0 and time 1
                const double c0 = s.center[a], c1 = s.center[a] + s.dcenter[a];
                lo[a] = std::fmin(c0, c1) - r;
                hi[a] = std::fmax(c0, c1) + r;
            }
            return RT_OK;
        }
        case RT_NODE_QUAD: {
            const rt_quad &q = d_.quads[n.a];
            for (int c = 0; c < 4; c++) {
                double v[3];
                for (int a = 0; a < 3; a++) v[a] = q.Q[a] + ((c & 1) ? q.u[a] : 0.0) + ((c & 2) ? q.v[a] : 0.0);
                b_grow(lo, hi, v, v);
            }
            b_pad_thin(lo, hi);
            return RT_OK;
        }
        case RT_NODE_TRIANGLE: {
            const rt_triangle &t = d_.triangles[n.a];
            b_grow(lo, hi, t.v0, t.v0);
            b_grow(lo, hi, t.v1, t.v1);
            b_grow(lo, hi, t.v2, t.v2);
            b_pad_thin(lo, hi);
            return RT_OK;
        }
        }
        return fail(RT_ERR_INVALID, "bad node kind");
    }

    // the leaves below description node `id` (frame of context `ctx`; `x` = accumulated rigid transform of a bakeable subtree)
    int collect(int id, int ctx, const Rigid &x, bool in_baked, int dup, int rdepth, std::vector<Leaf> &out) {
        if (rdepth > 256) return fail(RT_ERR_INVALID, "scene graph too deep (cycle?)");
        if (out.size() > kMaxUnits / 2) return fail(RT_ERR_NOMEM, "flattened scene too large");
        const rt_node &n = d_.nodes[id];
        switch (n.kind) {
        case RT_NODE_BVH: {
            const bool twice = n.b == n.a && !in_baked && contains_medium(n.a, 0); // see emit(): the reference hits a span-1 child twice
            if (twice && dup >= 64) return fail(RT_ERR_UNSUPPORTED, "constant_medium below more than 6 nested span-1 bvh_nodes");
            int rc = collect(n.a, ctx, x, in_baked, twice ? 2 * dup : dup, rdepth + 1, out);
            if (rc) return rc;
            if (n.b != n.a) return collect(n.b, ctx, x, in_baked, dup, rdepth + 1, out);
            if (twice) return collect(n.b, ctx, x, in_baked, 2 * dup + 1, rdepth + 1, out);
            return RT_OK;
        }
        case RT_NODE_LIST: {
            double lo[3], hi[3];
            int mat, face_id[6], outward;
            if (!in_baked && fuse_ && match_box(id, lo, hi, mat, face_id, outward)) {
                Leaf lf{id, dup, false, Rigid{}, {lo[0], lo[1], lo[2]}, {hi[0], hi[1], hi[2]}};
                out.push_back(lf);
                return RT_OK;
            }
            for (int k = 0; k < n.b; k++) {
                int rc = collect(d_.child_index[n.a + k], ctx, x, in_baked, dup, rdepth + 1, out);
                if (rc) return rc;
            }
            return RT_OK;
        }
        case RT_NODE_TRANSLATE:
        case RT_NODE_ROTATE_Y:
            if (in_baked || (ctx == 0 && bakeable(id))) { // compose (see emit_baked) and look through
                Rigid y = x;
                if (n.kind == RT_NODE_TRANSLATE) {
                    const double z[3] = {0, 0, 0};
                    double r0[3], r1[3];
                    Rigid rot = x;
                    rot.t[0] = rot.t[1] = rot.t[2] = 0;
                    rot.apply(n.p, r0);
                    rot.apply(z, r1);
                    for (int a = 0; a < 3; a++) y.t[a] = x.t[a] + (r0[a] - r1[a]);
                } else {
                    const double sn = n.p[0], cs = n.p[1];
                    y.c = x.c * cs - x.s * sn;
                    y.s = x.s * cs + x.c * sn;
                }
                if (!in_baked) f_.n_baked_instances++;
                return collect(n.a, ctx, y, true, dup, rdepth + 1, out);
            }
            [[fallthrough]];
        default: { // a primitive, a constant_medium or an instance that stays an instance: one leaf
            Leaf lf{id, dup, in_baked, x, {0, 0, 0}, {0, 0, 0}};
            int rc = in_baked ? baked_prim(id, x, lf.lo, lf.hi, false) : geom_bounds(id, lf.lo, lf.hi, rdepth);
            if (rc) return rc;
            if (!(lf.lo[0] <= lf.hi[0] && lf.lo[1] <= lf.hi[1] && lf.lo[2] <= lf.hi[2])) {
                if (n.kind == RT_NODE_SPHERE || n.kind == RT_NODE_QUAD || n.kind == RT_NODE_TRIANGLE)
                    return fail(RT_ERR_INVALID, "node " + std::to_string(id) + ": primitive with NaN coordinates");
                return RT_OK; // an empty list behind an instance or a medium: nothing to hit
            }
            out.push_back(lf);
            return RT_OK;
        }
        }
    }

    int emit_leaf(const Leaf &lf, int ctx, int rdepth) {
        const int dup = cur_dup_;
        cur_dup_ = lf.dup;
        double lo[3], hi[3];
        const int rc = lf.baked ? baked_prim(lf.id, lf.x, lo, hi, true) : emit(lf.id, ctx, rdepth + 1);
        cur_dup_ = dup;
        return rc;
    }

    // Measured (ms / frame; final_scene 1080p with the reference's own nodes and order: 850).  What decides is less the
    // number of records a ray visits than WHICH hierarchy the surface-area heuristic happens to produce and in which order
    // its children stand -- the rays that paths send are not the heuristic's uniform ones (three quarters start on a
    // surface), and 64 lanes walking one stream in lockstep care about coherence.  final_scene over the bin count alone:
    // 4 bins 951, 5 674, 6 708, 8 678, 10 684, 11 712, 12 651, 13 692, 14 711, 16 696, 20 813, 24 817, 32 761.  Child
    // order at 16 bins: split order 858, larger box first 814, fewer leaves first 799, larger box area PER LEAF first 792
    // (+ BOX thresholds 0.8 / 0.95: 763; a two-leaf node boxed only below 0.4: 901).  The defaults are the best setting
    // over the reference's five configurations: 12 bins, larger box first -- final_scene 634 (696 at 16 bins / area per
    // leaf), cornell_box 105 (125), mesh_scene 475 (482), first_scene 126 (122), bounce frame 86 (86).  (development knobs:
    // RT_AMD_BVH_BINS / _ORDER / _T2 / _TN; tools/grid_bvh.sh)
    BuildKnobs knobs_;

    // idx[0 .. n): leaves of one node.  parent_area: area of the closest ancestor that has a BOX record.
    int build(const std::vector<Leaf> &lv, int *idx, int n, int ctx, int rdepth, double parent_area, int bdepth) {
        if (f_.stream.size() > kMaxUnits) return fail(RT_ERR_NOMEM, "flattened scene too large");
        if (n == 1) return emit_leaf(lv[idx[0]], ctx, rdepth);
        double lo[3], hi[3], clo[3], chi[3];
        b_empty(lo, hi);
        b_empty(clo, chi);
        for (int k = 0; k < n; k++) {
            const Leaf &l = lv[idx[k]];
            b_grow(lo, hi, l.lo, l.hi);
            const double c[3] = {0.5 * (l.lo[0] + l.hi[0]), 0.5 * (l.lo[1] + l.hi[1]), 0.5 * (l.lo[2] + l.hi[2])};
            b_grow(clo, chi, c, c);
        }
        const double area = b_area(lo, hi);
        const bool boxed = area < (n == 2 ? knobs_.thr2 : knobs_.thrn) * parent_area;
        int self = -1;
        if (boxed) {
            self = push_units(2, -1);
            R rl[3], rh[3];
            for (int a = 0; a < 3; a++) {
                const double pad = (hi[a] - lo[a]) < 2 * box_pad_ ? box_pad_ : 0.0;
                cvt_lo(lo[a] - pad, rl[a]);
                cvt_hi(hi[a] + pad, rh[a]);
            }
            f_.stream[self] = UnitT<R>{{rl[0], rh[0], rl[1], rh[1]}};
            f_.stream[self + 1] = UnitT<R>{{rl[2], rh[2], I(0), I(make_hdr(K_BOX, 0))}};
        }
        // binned surface-area heuristic over the centroids: cost(split) = area(L) * |L| + area(R) * |R|
        int n_left = -1;
        if (n > 2 && bdepth < 64) {
            const int NB = knobs_.bins;
            double best = INFINITY;
            int best_axis = -1, best_bin = -1;
            // randomised build (the tuner's candidates): the bin grid of this node is shifted by a random share of a bin, and
            // among the splits within jitter_eps of the cheapest one is picked at random -- trees of the same quality by the
            // heuristic, which the walk may like better or worse (the landscape is rough: see above)
            uint32_t jr = knobs_.jitter ? knobs_.jitter * 0x9E3779B9u + (uint32_t)n * 0x85EBCA6Bu + (uint32_t)bdepth * 0xC2B2AE35u + (uint32_t)idx[0] : 0u;
            auto jnext = [&]() { jr ^= jr << 13; jr ^= jr >> 17; jr ^= jr << 5; return (jr >> 8) * (1.0 / 16777216.0); };
            const double shift = knobs_.jitter ? jnext() : 0.0; // bins cover [clo - shift * w, ...) with w = ext / (NB - 1)
            struct Cand { double cost; int axis, bin; };
            std::vector<Cand> cands;
            double max_ext = 0;
            for (int a = 0; a < 3; a++)
                if (std::isfinite(chi[a] - clo[a])) max_ext = std::fmax(max_ext, chi[a] - clo[a]);
            for (int a = 0; a < 3; a++) {
                const double ext = chi[a] - clo[a];
                if (!(ext > 0) || !std::isfinite(ext)) continue;
                if (ext < knobs_.axis_min * max_ext) continue; // (a split across a thin extent stacks the two sides on one footprint)
                struct Bin { double lo[3], hi[3]; int n; } bin[32];
                for (int b = 0; b < NB; b++) { b_empty(bin[b].lo, bin[b].hi); bin[b].n = 0; }
                const double sc = knobs_.jitter ? (NB - 1) / ext : NB / ext;
                for (int k = 0; k < n; k++) {
                    const Leaf &l = lv[idx[k]];
                    int b = (int)((0.5 * (l.lo[a] + l.hi[a]) - clo[a]) * sc + shift);
                    b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                    b_grow(bin[b].lo, bin[b].hi, l.lo, l.hi);
                    bin[b].n++;
                }
                double ra[32];
                int rn[32];
                double l3[3], h3[3];
                b_empty(l3, h3);
                int cnt = 0;
                double rlo[32][3], rhi[32][3];
                for (int b = NB - 1; b > 0; b--) {
                    b_grow(l3, h3, bin[b].lo, bin[b].hi);
                    cnt += bin[b].n;
                    ra[b] = b_area(l3, h3);
                    rn[b] = cnt;
                    for (int c = 0; c < 3; c++) { rlo[b][c] = l3[c]; rhi[b][c] = h3[c]; }
                }
                b_empty(l3, h3);
                cnt = 0;
                for (int b = 0; b + 1 < NB; b++) { // left = bins 0..b
                    b_grow(l3, h3, bin[b].lo, bin[b].hi);
                    cnt += bin[b].n;
                    if (cnt == 0 || rn[b + 1] == 0) continue;
                    double cost = b_area(l3, h3) * cnt + ra[b + 1] * rn[b + 1];
                    if (knobs_.overlap > 0) { // a ray through the part the two sides share walks both
                        double ol[3], oh[3];
                        bool any = true;
                        for (int c = 0; c < 3; c++) {
                            ol[c] = std::fmax(l3[c], rlo[b + 1][c]);
                            oh[c] = std::fmin(h3[c], rhi[b + 1][c]);
                            any = any && oh[c] >= ol[c];
                        }
                        if (any) cost += knobs_.overlap * b_area(ol, oh) * (double)(cnt + rn[b + 1]);
                    }
                    if (cost < best) { best = cost; best_axis = a; best_bin = b; }
                    if (knobs_.jitter) cands.push_back(Cand{cost, a, b});
                }
            }
            if (knobs_.jitter && best_axis >= 0) {
                std::vector<Cand> near;
                for (const Cand &c : cands)
                    if (c.cost <= best * (1.0 + knobs_.jitter_eps)) near.push_back(c);
                const Cand &pick = near[std::min(near.size() - 1, (size_t)(jnext() * (double)near.size()))];
                best_axis = pick.axis;
                best_bin = pick.bin;
            }
            if (best_axis >= 0) {
                const int a = best_axis;
                const double sc = knobs_.jitter ? (NB - 1) / (chi[a] - clo[a]) : NB / (chi[a] - clo[a]);
                int *mid = std::stable_partition(idx, idx + n, [&](int k) { // stable: each side keeps the reference's relative order
                    const Leaf &l = lv[k];
                    int b = (int)((0.5 * (l.lo[a] + l.hi[a]) - clo[a]) * sc + shift);
                    b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                    return b <= best_bin;
                });
                n_left = (int)(mid - idx);
            }
        }
        if (n_left <= 0 || n_left >= n) { // no useful split (equal centroids, two leaves, depth guard): halves, along the widest centroid axis
            int a = 0;
            for (int k = 1; k < 3; k++)
                if (chi[k] - clo[k] > chi[a] - clo[a]) a = k;
            n_left = n / 2;
            std::nth_element(idx, idx + n_left, idx + n, [&](int p, int q) {
                const double cp = lv[p].lo[a] + lv[p].hi[a], cq = lv[q].lo[a] + lv[q].hi[a];
                return cp < cq || (cp == cq && p < q);
            });
            std::sort(idx, idx + n_left); // the leaves were collected in the reference's visiting order: keep it inside each side
            std::sort(idx + n_left, idx + n);
        }
        // Which side goes first in the stream (the walk has no stack, so the order is the same for every ray).  0: as split
        // (low centroids), 1 / 2: larger / smaller box, 3 / 4: more / fewer leaves, 5: larger box area per leaf, 6: larger
        // area per tree level, 7 / 8: larger / smaller (share of the box its leaves cover) / (tree levels).
        int first = 0;
        if (knobs_.order != 0) {
            double al[3], ah[3], bl[3], bh[3];
            b_empty(al, ah);
            b_empty(bl, bh);
            for (int k = 0; k < n_left; k++) b_grow(al, ah, lv[idx[k]].lo, lv[idx[k]].hi);
            for (int k = n_left; k < n; k++) b_grow(bl, bh, lv[idx[k]].lo, lv[idx[k]].hi);
            if (knobs_.order <= 2) {
                const bool left_larger = b_area(al, ah) >= b_area(bl, bh);
                first = (knobs_.order == 1) == left_larger ? 0 : 1;
            } else if (knobs_.order <= 4) {
                const bool left_more = n_left >= n - n_left;
                first = (knobs_.order == 3) == left_more ? 0 : 1;
            } else if (knobs_.order == 5) { // larger area per leaf first
                const bool l = b_area(al, ah) * (n - n_left) >= b_area(bl, bh) * n_left;
                first = l ? 0 : 1;
            } else if (knobs_.order >= 7) { // the side more likely to end the search, per unit of work, first
                double sl = 0, sr = 0;
                for (int k = 0; k < n_left; k++) sl += b_area(lv[idx[k]].lo, lv[idx[k]].hi);
                for (int k = n_left; k < n; k++) sr += b_area(lv[idx[k]].lo, lv[idx[k]].hi);
                const double fl = std::fmin(1.0, sl / std::fmax(b_area(al, ah), 1e-300)), fr = std::fmin(1.0, sr / std::fmax(b_area(bl, bh), 1e-300));
                const double cl = 1.0 + std::log2((double)n_left), cr = 1.0 + std::log2((double)(n - n_left));
                first = (fl * cr >= fr * cl) == (knobs_.order == 7) ? 0 : 1;
            } else { // 6: larger area per level of its subtree first
                const double cl = 1.0 + std::log2((double)n_left), cr = 1.0 + std::log2((double)(n - n_left));
                first = b_area(al, ah) * cr >= b_area(bl, bh) * cl ? 0 : 1;
            }
        }
        const double pa = boxed ? area : parent_area;
        for (int side = 0; side < 2; side++) {
            const bool left = (side == 0) == (first == 0);
            int rc = left ? build(lv, idx, n_left, ctx, rdepth, pa, bdepth + 1) : build(lv, idx + n_left, n - n_left, ctx, rdepth, pa, bdepth + 1);
            if (rc) return rc;
        }
        if (boxed) f_.stream[self + 1].w[2] = I(pos());
        return RT_OK;
    }

    // everything below description node `id` (a bvh_node, a hittable_list or a bakeable instance), under a hierarchy of our own
    int emit_group(int id, int ctx, int rdepth) {
        std::vector<Leaf> lv;
        int rc = collect(id, ctx, Rigid{}, false, cur_dup_, rdepth, lv);
        if (rc || lv.empty()) return rc;
        std::vector<int> idx(lv.size());
        std::iota(idx.begin(), idx.end(), 0);
        // the world's own box is hit by nearly every ray (final_scene: the fog's boundary is in it): no BOX record for it
        const bool top = id == d_.world_root && ctx == 0;
        double lo[3], hi[3];
        b_empty(lo, hi);
        for (const Leaf &l : lv) b_grow(lo, hi, l.lo, l.hi);
        return build(lv, idx.data(), (int)lv.size(), ctx, rdepth, top ? b_area(lo, hi) : INFINITY, 0);
    }

    // Emit the subtree of description node `id` in the reference's visiting order.
    int emit(int id, int ctx, int rdepth) {
        if (rdepth > 256) return fail(RT_ERR_INVALID, "scene graph too deep (cycle?)");
        if (f_.stream.size() > kMaxUnits) return fail(RT_ERR_NOMEM, "flattened scene too large");
        const rt_node &n = d_.nodes[id];
        switch (n.kind) {
        case RT_NODE_BVH: { // bvh.h:43-51: own box, then left, then right
            if (rebuild_) return emit_group(id, ctx, rdepth);
            const int self = push_units(2, id);
            R lo[3], hi[3];
            for (int a = 0; a < 3; a++) {
                const double pad = (n.bbox[2 * a + 1] - n.bbox[2 * a]) < 2 * box_pad_ ? box_pad_ : 0.0;
                cvt_lo(n.bbox[2 * a] - pad, lo[a]);
                cvt_hi(n.bbox[2 * a + 1] + pad, hi[a]);
            }
            f_.stream[self] = UnitT<R>{{lo[0], hi[0], lo[1], hi[1]}};
            f_.stream[self + 1] = UnitT<R>{{lo[2], hi[2], I(0), I(make_hdr(K_BOX, 0))}};
            // span-1 nodes have left == right (bvh.h:30) and the reference visits the child twice.  For geometry the second
            // visit cannot change the result; a constant_medium draws a NEW uniform on it (constant_medium.h:41), i.e. the
            // reference scatters in such a medium at the closer of two free paths -- so that child is emitted twice, the
            // copies told apart by cur_dup_ (which keys their draws).
            const bool twice = n.b == n.a && contains_medium(n.a, 0);
            const int dup = cur_dup_;
            if (twice && dup >= 64) return fail(RT_ERR_UNSUPPORTED, "constant_medium below more than 6 nested span-1 bvh_nodes");
            if (twice) cur_dup_ = 2 * dup;
            int rc = emit(n.a, ctx, rdepth + 1);
            if (rc) return rc;
            if (n.b != n.a || twice) {
                if (twice) cur_dup_ = 2 * dup + 1;
                rc = emit(n.b, ctx, rdepth + 1);
                if (rc) return rc;
            }
            cur_dup_ = dup;
            f_.stream[self + 1].w[2] = I(pos());
            return RT_OK;
        }
        case RT_NODE_LIST: { // hittable_list.h:29: children in insertion order, each against the closest hit so far
            double lo[3], hi[3];
            int mat, face_id[6], outward;
            if (fuse_ && match_box(id, lo, hi, mat, face_id, outward)) {
                const int off = push_units(2, id);
                f_.stream[off] = UnitT<R>{{F(lo[0]), F(hi[0]), F(lo[1]), F(hi[1])}};
                f_.stream[off + 1] = UnitT<R>{{F(lo[2]), F(hi[2]), I(mat), I(make_hdr(K_AABOX, outward, ctx))}};
                f_.cold[off] = UnitT<R>{{I(face_id[0]), I(face_id[1]), I(face_id[2]), I(face_id[3])}};
                f_.cold[off + 1] = UnitT<R>{{I(face_id[4]), I(face_id[5]), I(0), I(0)}};
                f_.n_fused_boxes++;
                return RT_OK;
            }
            if (rebuild_) return emit_group(id, ctx, rdepth);
            for (int k = 0; k < n.b; k++) {
                int rc = emit(d_.child_index[n.a + k], ctx, rdepth + 1);
                if (rc) return rc;
            }
            return RT_OK;
        }
        case RT_NODE_TRANSLATE:
        case RT_NODE_ROTATE_Y: {
            if (fuse_ && !dev_knob("RT_AMD_NO_OBOX")) {
                int box;
                double sn, cs, t[3];
                if (match_obox(id, box, sn, cs, t)) {
                    double lo[3], hi[3];
                    int mat, face_id[6], outward;
                    match_box(box, lo, hi, mat, face_id, outward);
                    const int off = push_units(4, id);
                    f_.stream[off] = UnitT<R>{{F(lo[0]), F(hi[0]), F(lo[1]), F(hi[1])}};
                    f_.stream[off + 1] = UnitT<R>{{F(lo[2]), F(hi[2]), I(mat), I(make_hdr(K_OBOX, outward, ctx))}};
                    f_.stream[off + 2] = UnitT<R>{{F(sn), F(cs), F(t[0]), F(t[1])}};
                    f_.stream[off + 3] = UnitT<R>{{F(t[2]), R(0), R(0), R(0)}};
                    f_.cold[off] = UnitT<R>{{I(face_id[0]), I(face_id[1]), I(face_id[2]), I(face_id[3])}};
                    f_.cold[off + 1] = UnitT<R>{{I(face_id[4]), I(face_id[5]), I(0), I(0)}};
                    f_.n_fused_boxes++;
                    return RT_OK;
                }
            }
            if (rebuild_ && ctx == 0 && bakeable(id)) return emit_group(id, ctx, rdepth);
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
            if (c >= (1 << 20)) return fail(RT_ERR_UNSUPPORTED, "more than 2^20 translate / rotate_y instances");
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
                f_.stream[off + 1] = UnitT<R>{{F(n.p[0]), I(n.b), I(id), I(make_hdr(K_MEDIUM_GEN, medium_payload(id), ctx))}};
                pending_.push_back(Pending{off, n.a, ctx});
                return RT_OK;
            }
            const rt_sphere &s = d_.spheres[b.a];
            const int off = push_units(2, id);
            f_.stream[off] = UnitT<R>{{F(s.center[0]), F(s.center[1]), F(s.center[2]), F(s.radius)}};
            f_.stream[off + 1] = UnitT<R>{{F(n.p[0]), I(n.b), I(id), I(make_hdr(K_MEDIUM, medium_payload(id), ctx))}};
            return RT_OK;
        }
        case RT_NODE_SPHERE: {
            const rt_sphere &s = d_.spheres[n.a];
            if (s.mat < 0) return fail(RT_ERR_INVALID, "world sphere without material");
            const bool moving = s.dcenter[0] != 0 || s.dcenter[1] != 0 || s.dcenter[2] != 0;
            const int off = push_units(moving ? 3 : 2, id);
            f_.stream[off] = UnitT<R>{{F(s.center[0]), F(s.center[1]), F(s.center[2]), F(s.radius)}};
            f_.stream[off + 1] = UnitT<R>{{I(s.mat), I(0), I(0), I(make_hdr(moving ? K_MSPHERE : K_SPHERE, 0, ctx))}};
            if (moving) f_.stream[off + 2] = UnitT<R>{{F(s.dcenter[0]), F(s.dcenter[1]), F(s.dcenter[2]), R(0)}};
            return RT_OK;
        }
        case RT_NODE_QUAD: {
            const rt_quad &q = d_.quads[n.a];
            if (q.mat < 0) return fail(RT_ERR_INVALID, "world quad without material");
            if (fuse_ && !dev_knob("RT_AMD_NO_FLAT_QUADS")) {
                // An axis-aligned quad whose material never asks for rec.u / rec.v (walls, ceiling lights) is a box of no
                // thickness: two units instead of five and the AABOX step instead of the general one.  The slab test accepts
                // exactly the points of the plane inside the quad's edges, inclusive (quad.h:63-70: unit.contains); a ray
                // IN the plane never hits (aabox_t: the slab's two distances are 0 or +-1e30 together; quad.h:38 rejects it
                // too).  Both faces of the flat axis name this quad and carry its own normal.
                int au = -1, av = -1;
                bool aligned = true;
                for (int a = 0; a < 3; a++) {
                    if (q.u[a] != 0.0) { if (au >= 0) aligned = false; au = a; }
                    if (q.v[a] != 0.0) { if (av >= 0) aligned = false; av = a; }
                }
                if (aligned && au >= 0 && av >= 0 && au != av && !tex_needs_uv(d_.materials[q.mat].tex)) {
                    const int c = 3 - au - av;
                    double lo[3], hi[3];
                    for (int a = 0; a < 3; a++) {
                        lo[a] = std::fmin(q.Q[a], q.Q[a] + q.u[a] + q.v[a]);
                        hi[a] = std::fmax(q.Q[a], q.Q[a] + q.u[a] + q.v[a]);
                    }
                    lo[c] = hi[c] = q.Q[c];
                    const int outward = q.normal[c] < 0 ? (1 << (2 * c)) : (1 << (2 * c + 1)); // which of the two faces' outward directions is the quad's normal
                    const int off = push_units(2, id);
                    f_.stream[off] = UnitT<R>{{F(lo[0]), F(hi[0]), F(lo[1]), F(hi[1])}};
                    f_.stream[off + 1] = UnitT<R>{{F(lo[2]), F(hi[2]), I(q.mat), I(make_hdr(K_AABOX, outward, ctx))}};
                    f_.cold[off] = UnitT<R>{{I(id), I(id), I(id), I(id)}};
                    f_.cold[off + 1] = UnitT<R>{{I(id), I(id), I(0), I(0)}};
                    return RT_OK;
                }
            }
            const int off = push_units(5, id);
            f_.stream[off] = UnitT<R>{{F(q.normal[0]), F(q.normal[1]), F(q.normal[2]), F(q.D)}};
            f_.stream[off + 1] = UnitT<R>{{F(q.Q[0]), F(q.Q[1]), F(q.Q[2]), I(make_hdr(K_QUAD, 0, ctx))}};
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
            f_.stream[off + 1] = UnitT<R>{{F(e1[1]), F(e1[2]), F(e2[0]), I(make_hdr(K_TRI, 0, ctx))}};
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
