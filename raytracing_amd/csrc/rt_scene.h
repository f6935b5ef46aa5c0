// rt_scene.h -- device-resident scene layout (HBM) of the gfx950 path tracer, templated on the arithmetic type.
//
// The reference's object graph (bvh_node / hittable_list / translate / rotate_y / constant_medium / primitives,
// src/cpu/*.h) is flattened on the host into ONE linear "threaded" node array in depth-first pre-order -- exactly
// the order in which the reference's recursion visits objects (bvh.h:43-51 left then right, hittable_list.h:29
// insertion order).  Each node carries a skip link, so traversal needs no stack at all and reproduces the
// reference's closest-hit tie-breaking (strict vs inclusive interval tests, sphere.h:44 / quad.h:41) by construction.
//
//   node i:  kind BOX      -> slab test (aabb.h:36); hit: i+1, miss: skip
//            kind SPHERE.. -> primitive test against [tmin, closest so far]; next: i+1
//            kind XF_ENTER -> transform the ray into object space (hittable.h:55,114); next: i+1
//            kind XF_EXIT  -> restore the parent space's ray; next: i+1
//
// Records are arrays of 16-byte-multiple structs: every lane gathers a different node / primitive, so the unit of
// locality is the record (1-5 dwordx4 loads from one or two cache lines), not a column.
#ifndef RT_SCENE_H
#define RT_SCENE_H

#include <stdint.h>

namespace rtamd {

enum : int {
    K_BOX = 0,
    K_SPHERE = 1,
    K_QUAD = 2,
    K_TRI = 3,
    K_MEDIUM = 4,
    K_XF_ENTER = 5,
    K_XF_EXIT = 6,
};
enum : int { XF_TRANSLATE = 0, XF_ROTATE_Y = 1 };
enum : int { MAX_XF_DEPTH = 4 };
enum : int { LIGHT_OTHER = 0, LIGHT_SPHERE = 1, LIGHT_QUAD = 2 };

inline int make_meta(int kind, int index) { return (index << 4) | kind; }

template <class R> struct alignas(16) NodeT { // f32: 32 B, f64: 64 B
    R bmin[3];
    R bmax[3];
    int32_t skip;
    int32_t meta; // kind | index<<4
};

template <class R> struct alignas(16) SphereT { // f32: 32 B, f64: 64 B
    R c[3];
    R radius;
    R d[3]; // centre velocity (moving sphere: centre = c + time*d, sphere.h:30)
    int32_t mat;
};

template <class R> struct alignas(16) QuadT { // f32: 80 B, f64: 144 B
    R n[3];
    R D;
    R Q[3];
    R area;
    R u[3];
    R v[3];
    R w[3];
    int32_t mat;
};

template <class R> struct alignas(16) TriT { // f32: 64 B, f64: 112 B
    R v0[3];
    R v1[3];
    R v2[3];
    R n[3];
    int32_t mat;
    int32_t cull;
};

template <class R> struct alignas(16) MediumT {
    R neg_inv_density;
    int32_t sphere; // boundary (index into spheres)
    int32_t mat;
    int32_t node_id; // key of the medium's RNG draw
    int32_t pad_;
};

template <class R> struct alignas(16) XformT {
    int32_t kind;
    int32_t parent; // enclosing transform context (0 = world)
    int32_t depth;  // number of transforms from world to here (>=1)
    int32_t chain[MAX_XF_DEPTH]; // context ids from outermost to this one
    int32_t pad_;
    R p[4]; // translate: offset xyz; rotate_y: sin, cos
};

template <class R> struct alignas(16) MatT {
    int32_t kind;
    int32_t tex;
    int32_t needs_uv; // texture graph contains an image texture
    int32_t pad_;
    R albedo[3];
    R fuzz;
    R ior;
    R pad1_;
};

template <class R> struct alignas(16) TexT {
    int32_t kind;
    int32_t a;
    int32_t b;
    int32_t pad_;
    R color[3];
    R scale;
};

struct ImageT {
    int32_t width, height;
    int64_t offset;
};

template <class R> struct PerlinT {
    R randvec[256][3];
    int32_t perm_x[256], perm_y[256], perm_z[256];
};

template <class R> struct alignas(16) LightT {
    int32_t kind;
    int32_t pad_[3];
    SphereT<R> s;
    QuadT<R> q;
};

template <class R> struct CameraT {
    int32_t W, H, sqrt_spp, max_depth;
    R recip_sqrt_spp;
    R defocus_angle;
    R center[3];
    R pixel00[3];
    R du[3];
    R dv[3];
    R ddu[3];
    R ddv[3];
    R background[3];
};

template <class R> struct SceneT { // device pointers; passed to kernels by value
    const NodeT<R> *nodes;
    const SphereT<R> *spheres;
    const QuadT<R> *quads;
    const TriT<R> *tris;
    const MediumT<R> *media;
    const XformT<R> *xforms;
    const MatT<R> *mats;
    const TexT<R> *texs;
    const ImageT *images;
    const uint8_t *image_bytes;
    const PerlinT<R> *perlins;
    const LightT<R> *lights;
    const int32_t *sphere_ids; // table index -> description node id (prim id reported by rt_trace_batch)
    const int32_t *quad_ids;
    const int32_t *tri_ids;
    int32_t n_nodes;
    int32_t n_lights;
    int32_t lights_is_list; // lights_root is a hittable_list (hittable_list.h:52 draws the pick) vs a bare shape
    int32_t pad_;
};

} // namespace rtamd
#endif
