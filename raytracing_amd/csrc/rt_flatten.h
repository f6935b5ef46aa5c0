// rt_flatten.h -- host side: turn an rt_scene_desc (the reference's object graph as plain structs) into the threaded
// device layout of rt_scene.h.  Pure C++ (no HIP); runs once per scene upload.
#ifndef RT_FLATTEN_H
#define RT_FLATTEN_H

#include <cmath>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_scene.h"

namespace rtamd {

template <class R> struct Flat {
    std::vector<NodeT<R>> nodes;
    std::vector<SphereT<R>> spheres;
    std::vector<QuadT<R>> quads;
    std::vector<TriT<R>> tris;
    std::vector<MediumT<R>> media;
    std::vector<XformT<R>> xforms; // [0] = world (unused record)
    std::vector<MatT<R>> mats;
    std::vector<TexT<R>> texs;
    std::vector<ImageT> images;
    std::vector<PerlinT<R>> perlins;
    std::vector<LightT<R>> lights;
    std::vector<int32_t> sphere_ids, quad_ids, tri_ids;
    int lights_is_list = 0;
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
    Flattener(const rt_scene_desc &d, Flat<R> &out, std::string &err) : d_(d), f_(out), err_(err) {}

    int run() {
        int rc = validate();
        if (rc) return rc;
        convert_tables();
        f_.xforms.push_back(XformT<R>{}); // context 0 = world space
        rc = emit(d_.world_root, 0, 0);
        if (rc) return rc;
        return emit_lights();
    }

  private:
    const rt_scene_desc &d_;
    Flat<R> &f_;
    std::string &err_;
    static constexpr size_t kMaxNodes = size_t(1) << 26;

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
        f_.spheres.reserve(d_.n_spheres);
        for (int i = 0; i < d_.n_spheres; i++) f_.spheres.push_back(cvt_sphere(d_.spheres[i]));
        f_.quads.reserve(d_.n_quads);
        for (int i = 0; i < d_.n_quads; i++) f_.quads.push_back(cvt_quad(d_.quads[i]));
        f_.tris.reserve(d_.n_triangles);
        for (int i = 0; i < d_.n_triangles; i++) {
            const rt_triangle &t = d_.triangles[i];
            TriT<R> o{};
            cvt3(t.v0, o.v0);
            cvt3(t.v1, o.v1);
            cvt3(t.v2, o.v2);
            cvt3(t.normal, o.n);
            o.mat = t.mat;
            o.cull = t.cull;
            f_.tris.push_back(o);
        }
        f_.sphere_ids.assign(d_.n_spheres, -1);
        f_.quad_ids.assign(d_.n_quads, -1);
        f_.tri_ids.assign(d_.n_triangles, -1);
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

    int push(int kind, int index, const double *bbox) {
        NodeT<R> n{};
        if (bbox) {
            for (int a = 0; a < 3; a++) {
                cvt_lo(bbox[2 * a], n.bmin[a]);
                cvt_hi(bbox[2 * a + 1], n.bmax[a]);
            }
        }
        n.meta = make_meta(kind, index);
        n.skip = (int)f_.nodes.size() + 1;
        f_.nodes.push_back(n);
        return (int)f_.nodes.size() - 1;
    }

    // Emit the subtree of description node `id` in the reference's visiting order.
    int emit(int id, int ctx, int rdepth) {
        if (rdepth > 256) return fail(RT_ERR_INVALID, "scene graph too deep (cycle?)");
        if (f_.nodes.size() > kMaxNodes) return fail(RT_ERR_NOMEM, "flattened scene too large");
        const rt_node &n = d_.nodes[id];
        switch (n.kind) {
        case RT_NODE_BVH: { // bvh.h:43-51: own box, then left, then right
            int self = push(K_BOX, 0, n.bbox);
            int rc = emit(n.a, ctx, rdepth + 1);
            if (rc) return rc;
            if (n.b != n.a) { // span-1 nodes have left == right (bvh.h:30): the second visit cannot change the result
                rc = emit(n.b, ctx, rdepth + 1);
                if (rc) return rc;
            }
            f_.nodes[self].skip = (int)f_.nodes.size();
            return RT_OK;
        }
        case RT_NODE_LIST: // hittable_list.h:29: children in insertion order, each against the closest hit so far
            for (int k = 0; k < n.b; k++) {
                int rc = emit(d_.child_index[n.a + k], ctx, rdepth + 1);
                if (rc) return rc;
            }
            return RT_OK;
        case RT_NODE_TRANSLATE:
        case RT_NODE_ROTATE_Y: {
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
            push(K_XF_ENTER, c, nullptr);
            int rc = emit(n.a, c, rdepth + 1);
            if (rc) return rc;
            push(K_XF_EXIT, c, nullptr);
            return RT_OK;
        }
        case RT_NODE_MEDIUM: {
            const rt_node &b = d_.nodes[n.a];
            if (b.kind != RT_NODE_SPHERE)
                return fail(RT_ERR_UNSUPPORTED, "constant_medium boundary must be a sphere on the device path");
            MediumT<R> m{};
            cvt(n.p[0], m.neg_inv_density);
            m.sphere = b.a;
            m.mat = n.b;
            m.node_id = id;
            f_.media.push_back(m);
            push(K_MEDIUM, (int)f_.media.size() - 1, nullptr);
            return RT_OK;
        }
        case RT_NODE_SPHERE:
            f_.sphere_ids[n.a] = id;
            if (d_.spheres[n.a].mat < 0) return fail(RT_ERR_INVALID, "world sphere without material");
            push(K_SPHERE, n.a, nullptr);
            return RT_OK;
        case RT_NODE_QUAD:
            f_.quad_ids[n.a] = id;
            if (d_.quads[n.a].mat < 0) return fail(RT_ERR_INVALID, "world quad without material");
            push(K_QUAD, n.a, nullptr);
            return RT_OK;
        case RT_NODE_TRIANGLE:
            f_.tri_ids[n.a] = id;
            if (d_.triangles[n.a].mat < 0) return fail(RT_ERR_INVALID, "world triangle without material");
            push(K_TRI, n.a, nullptr);
            return RT_OK;
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
