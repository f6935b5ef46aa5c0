// rt_scene.h -- device-resident scene layout of the gfx950 path tracer, templated on the arithmetic type.
//
// The reference's object graph (bvh_node / hittable_list / translate / rotate_y / constant_medium / primitives,
// src/cpu/*.h) is flattened on the host into ONE linear "threaded" record stream in depth-first pre-order -- exactly
// the order in which the reference's recursion visits objects (bvh.h:43-51 left then right, hittable_list.h:29
// insertion order).  Box records carry a skip link, so traversal needs no stack at all and reproduces the
// reference's closest-hit tie-breaking (strict vs inclusive interval tests, sphere.h:44 / quad.h:41) by construction.
//
// The stream is an array of UNITS of four scalars (f32: 16 B = one ds_read_b128 / global_load_dwordx4, f64: 32 B).
// Primitives are stored inline in the stream (no second dependent fetch), every record is >= 2 units and keeps its
// header (kind | payload << 4) in the last scalar of its second unit, so one traversal step always starts with the
// same two unit loads.  A stream that fits is staged into the CU's LDS (160 KiB) by every workgroup (`lds_units` ==
// `n_units`); one that does not (large meshes) is read from global memory through L1/L2 as a whole (`lds_units` == 0).
//
//   record            units  unit0                     unit1                              unit2..
//   BOX     (bvh_node)   2   x.min, x.max, y.min, y.max   z.min, z.max, skip, hdr          (min/max pairs: packed f32 math;
//                                                                                         skip = BYTE offset of the record behind the subtree)
//   SPHERE / MSPHERE     2/3 c.xyz, radius             mat, -, -, hdr                     d.xyz (MSPHERE: centre = c + time*d)
//   AABOX   (box())      2   x.min, x.max, y.min, y.max   z.min, z.max, mat, hdr(payload = 6 face-orientation bits)
//   QUAD                 5   n.xyz, D                  Q.xyz, hdr                         u.xyz, mat | v.xyz, area | w.xyz, -
//   TRI                  3   v0.xyz, e1.x              e1.y, e1.z, e2.x, hdr              e2.y, e2.z, mat, cull
//   MEDIUM               2   c.xyz, radius (boundary)  -1/density, mat, node_id, hdr      (boundary = static sphere, inline)
//   MEDIUM_GEN           2   sub-stream offset, -, -, -  -1/density, mat, node_id, hdr     (any other boundary: box(), instances,
//                                                                                         lists, a bvh ...; its records form a
//                                                                                         sub-stream with a sentinel of its own,
//                                                                                         placed behind the main stream's)
//   XF_ENTER             2   p0, p1, p2, -             xf kind, ctx id, -, hdr            (translate: offset; rotate_y: sin, cos)
//   XF_EXIT              2   p0, p1, p2, -             xf kind, parent ctx id, -, hdr     (same parameters as its XF_ENTER)
//   (end)                2   -                         -, -, -, hdr = K_READY             sentinel behind the last record
//
// AABOX exists in the f32 stream only: a hittable_list of six axis-aligned quads that form a closed box with one
// material (what box() of quad.h:103 builds) is fused into one slab test.  The f64 parity stream keeps the six quads.
//
// `cold` is a parallel array (same unit offsets) that only hit finalisation and rt_trace_batch read:
// cold[off] = { triangle normal xyz | 0, description node id }.
#ifndef RT_SCENE_H
#define RT_SCENE_H

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

namespace rtamd {

// Development knobs (hierarchy-build settings, scheduling constants, layout limits that tools/ sweeps through the
// environment) exist only in a library built with -DRT_DEV_KNOBS (`make variant`, `make prof`); the shipped library never
// reads the environment.
#ifdef RT_DEV_KNOBS
inline const char *dev_knob(const char *name) { return getenv(name); }
#else
inline const char *dev_knob(const char *) { return nullptr; }
#endif

// Record kinds.  0..3 are the kinds that have a code block of their own in the render kernel's wave machine, >= 4 share
// the "other" block; K_READY is the header of the end-of-stream sentinel (traversal finished: shade), K_IDLE never
// occurs in the stream: it marks a lane without work.
enum : int {
    K_BOX = 0,
    K_SPHERE = 1, // static sphere
    K_AABOX = 2,
    K_MEDIUM = 3,
    K_QUAD = 4,
    K_TRI = 5,
    K_XF_ENTER = 6,
    K_XF_EXIT = 7,
    K_MSPHERE = 8, // moving sphere (3 units)
    K_MEDIUM_GEN = 9, // constant_medium whose boundary is not a static sphere: the boundary is a sub-stream of its own
    K_OBOX = 10, // f32 stream: translate(rotate_y(box())) as ONE record (4 units): the AABOX in its own frame + sin, cos, offset
    K_IDLE = 14,
    K_READY = 15,
};
enum : int { XF_TRANSLATE = 0, XF_ROTATE_Y = 1 };
enum : int { MAX_XF_DEPTH = 4 };
enum : int { LIGHT_OTHER = 0, LIGHT_SPHERE = 1, LIGHT_QUAD = 2 };

template <class R> struct alignas(4 * sizeof(R)) UnitT {
    R w[4];
};

// integers live in scalar slots by bit pattern (never touched by arithmetic)
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_HD __host__ __device__ __forceinline__
#elif defined(__HIPCC__)
#define RT_HD __host__ __device__ inline
#else
#define RT_HD inline
#endif
RT_HD float enc_int(int32_t v, float) {
    float f;
    __builtin_memcpy(&f, &v, 4);
    return f;
}
RT_HD double enc_int(int32_t v, double) {
    int64_t x = v;
    double f;
    __builtin_memcpy(&f, &x, 8);
    return f;
}
RT_HD int32_t dec_int(float f) {
    int32_t v;
    __builtin_memcpy(&v, &f, 4);
    return v;
}
RT_HD int32_t dec_int(double f) {
    int64_t v;
    __builtin_memcpy(&v, &f, 8);
    return (int32_t)v;
}
// record header: kind (4 bits) | kind-specific payload (8 bits: AABOX outward-normal mask, MEDIUM draw selector) | transform
// context of the record (20 bits; a leaf's frame is a property of its place in the stream, so the traversal carries no
// "current context" and no "context of the best hit": finalisation reads it here).  A BOX header is the integer 0.
RT_HD int make_hdr(int kind, int payload, int ctx = 0) { return (int)(((unsigned)ctx << 12) | ((unsigned)(payload & 255) << 4) | (unsigned)kind); }
RT_HD int hdr_payload(int h) { return (h >> 4) & 255; }
RT_HD int hdr_ctx(int h) { return (int)((unsigned)h >> 12); }

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
    int32_t noise;    // lambertian / isotropic whose texture is a noise_texture: index of its perlin table, else -1
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

template <class R> struct SphereT { // light shapes only (world spheres live in the stream)
    R c[3];
    R radius;
    R d[3];
    int32_t mat;
};

template <class R> struct QuadT { // light shapes only
    R n[3];
    R D;
    R Q[3];
    R area;
    R u[3];
    R v[3];
    R w[3];
    int32_t mat;
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
    const UnitT<R> *stream;
    const UnitT<R> *cold;
    const XformT<R> *xforms;
    const MatT<R> *mats;
    const TexT<R> *texs;
    const ImageT *images;
    const uint8_t *image_bytes;
    const PerlinT<R> *perlins;
    const LightT<R> *lights;
    // the shading tables (xforms .. lights) are one contiguous device range [tab, tab + tab_bytes); when tab_in_lds is
    // set the kernels stage it into LDS behind the stream and shade from there
    const uint8_t *tab;
    int32_t tab_bytes;
    int32_t tab_in_lds;
    int32_t n_units;
    int32_t lds_units; // units of `stream` that the kernels stage into LDS: all of them or none
    int32_t n_lights;
    int32_t lights_is_list; // lights_root is a hittable_list (hittable_list.h:52 draws the pick) vs a bare shape
    int32_t has_gen_medium; // MEDIUM_GEN records present: launch the kernels that can walk a boundary sub-stream
};

// LDS map of a workgroup (all offsets are compile-time constants, so no register holds them):
//   [0, stream_off)            per wave: item_slots accumulator slots (items in flight) x tile_px pixels x RGB cells of 8
//                              bytes (f32 kernels: 64-bit fixed point, so that a pixel's sum does not depend on the order
//                              of the adds; f64 kernels: double), then the wave's item state (4 ints per slot + 16 + the
//                              slots' NaN flags)
//   [stream_off, ...)          the record stream (when it fits: whole or not at all), then (when they fit) the shading tables
#ifndef RT_ITEM_SLOTS
#define RT_ITEM_SLOTS 16
#endif
#ifndef RT_TILE_H // an item's pixel tile is 8 wide and RT_TILE_H (8, 4, 2 or 1) high
#define RT_TILE_H 1
#endif
#ifndef RT_F32_BLOCK
#define RT_F32_BLOCK 1024
#endif
template <class R> struct LdsLayout {
    static constexpr int block = sizeof(R) == 4 ? RT_F32_BLOCK : 512; // f32: 16 waves/CU, <= 128 VGPRs; f64 (parity path): 8 waves
    static constexpr int waves = block / 64;
    static constexpr int item_slots = RT_ITEM_SLOTS;                           // items in flight per wave
    static constexpr int tile_h = RT_TILE_H, tile_px = 8 * tile_h, tile_shift = tile_h == 8 ? 6 : tile_h == 4 ? 5 : tile_h == 2 ? 4 : 3;
    static constexpr int slot_acc = tile_px * 3;                               // accumulator cells per slot
    static constexpr int wave_acc = item_slots * slot_acc;                     // accumulator cells per wave
    static constexpr int nan_words = (slot_acc + 31) / 32;                     // per slot: one "a NaN sample was added" bit per cell
    static constexpr int state_ints = 4 * item_slots + 16 + nan_words * item_slots; // per slot: i0, j0, chunk, valid; the current item; NaN bits
    static constexpr int wave_stride = wave_acc * 8 + state_ints * 4;          // bytes per wave
    static_assert((tile_h == 8 || tile_h == 4 || tile_h == 2 || tile_h == 1) && item_slots >= 2 && item_slots <= 16, "tile / slot geometry");
    static_assert(wave_stride % 8 == 0, "accumulator cells are 8-byte aligned");
    static constexpr int wave_off = 0;
    static constexpr int stream_off = wave_off + waves * wave_stride;
    static constexpr int total_bytes = sizeof(R) == 4 ? 160 * 1024 - 512 : 160 * 1024 - 512 - 16 * 1024;
    static constexpr int scene_bytes = total_bytes - stream_off;               // stream + tables
    static_assert(stream_off % 16 == 0 && stream_off < 65536 - 64, "ds_read immediate offset");
};

} // namespace rtamd
#endif
