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
#include <stddef.h>
#include <type_traits>
#include "rt_scene.h"

namespace rtamd {

#define RT_DEV __device__ __forceinline__

// ------------------------------------------------------------------------------------------------ scalar helpers
RT_DEV float m_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); } // v_sqrt_f32 (1 ulp); sqrtf() adds denormal range scaling: 6 instructions
RT_DEV double m_sqrt(double x) { return sqrt(x); }
RT_DEV float m_sin(float x) { return sinf(x); }
RT_DEV double m_sin(double x) { return sin(x); }
RT_DEV float m_cos(float x) { return cosf(x); }
RT_DEV double m_cos(double x) { return cos(x); }
// acos / atan2 of the float path (sphere.h:89 get_sphere_uv is their only caller: texture coordinates of an image-textured
// sphere).  The library routines are ~70 instructions each and run in every shading trip in which one lane hit such a
// sphere (final_scene's globe: three trips in four); these are 13 and 18.  acos: Abramowitz & Stegun 4.4.46,
// sqrt(1 - |x|) * P7(|x|), |error| <= 2e-8 + float rounding (2.3e-7 measured); atan2: odd minimax polynomial of degree 11
// on [0, 1] after the usual octant reduction, |error| <= 2e-6 rad = 3e-7 of a turn -- a third of a thousandth of
// earthmap.jpg's 1024 texels.  (The double path calls libm, as the reference does.)
RT_DEV float m_acos(float x) {
    const float ax = fabsf(x);
    float p = -0.0012624911f;
    p = fmaf(p, ax, 0.0066700901f);
    p = fmaf(p, ax, -0.0170881256f);
    p = fmaf(p, ax, 0.0308918810f);
    p = fmaf(p, ax, -0.0501743046f);
    p = fmaf(p, ax, 0.0889789874f);
    p = fmaf(p, ax, -0.2145988016f);
    p = fmaf(p, ax, 1.5707963050f);
    const float r = __builtin_amdgcn_sqrtf(1.0f - ax) * p;
    return x < 0.0f ? 3.14159265358979323846f - r : r;
}
RT_DEV double m_acos(double x) { return acos(x); }
RT_DEV float m_atan2(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float t = fminf(ax, ay) * __builtin_amdgcn_rcpf(fmaxf(fmaxf(ax, ay), 1e-30f)); // (0, 0) -> 0, as atan2f
    const float q = t * t;
    float p = -0.01172120f;
    p = fmaf(p, q, 0.05265332f);
    p = fmaf(p, q, -0.11643287f);
    p = fmaf(p, q, 0.19354346f);
    p = fmaf(p, q, -0.33262347f);
    p = fmaf(p, q, 0.99997726f);
    float r = p * t;
    r = ay > ax ? 1.57079632679489661923f - r : r;
    r = x < 0.0f ? 3.14159265358979323846f - r : r;
    return y < 0.0f ? -r : r;
}
RT_DEV double m_atan2(double y, double x) { return atan2(y, x); }
// v_log_f32 (log2, ~1 ulp) * ln 2: ~1e-6 relative on (0,1), enough for a free-path length.  (__logf would add a denormal
// rescale and an extended-precision correction: ten more instructions per medium test; the argument is a 24-bit uniform.)
RT_DEV float m_log(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994531f; }
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

// sin / cos of 2*pi*x.  v_sin_f32 / v_cos_f32 take their argument in revolutions: one instruction each.
RT_DEV void sincos_rev(float x, float &s, float &c) {
    s = __builtin_amdgcn_sinf(x);
    c = __builtin_amdgcn_cosf(x);
}
RT_DEV void sincos_rev(double x, double &s, double &c) {
    const double phi = 2 * 3.1415926535897932385 * x; // 2*pi*r1 (vec3.h:146, sphere.h:104)
    s = sin(phi);
    c = cos(phi);
}
RT_DEV float sin_wide(float x) { // sin of an unreduced argument (noise texture): reduce in revolutions, then v_sin_f32
    float r = x * 0.15915494309189535f;
    return __builtin_amdgcn_sinf(r - floorf(r));
}
RT_DEV double sin_wide(double x) { return sin(x); }

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
// t * v + w, each component ONE expression: the float translation unit is built -ffp-contract=on, where the front end fuses
// a * b + c within an expression and nothing else is ever fused -- every inlined copy of a routine then rounds alike
// (see sphere_t) -- so the multiply-adds worth having are written as such.  (The double unit never fuses: reference order.)
template <class R> RT_DEV V3<R> madd(R t, V3<R> v, V3<R> w) { return mk<R>(t * v.x + w.x, t * v.y + w.y, t * v.z + w.z); }
// Division: the double path divides (reference arithmetic); the float path multiplies by v_rcp_f32 (1 ulp) -- hipcc's float
// division is a 5-6 instruction sequence (frexp / rcp / ldexp range scaling) even without correct rounding.
template <class R> RT_DEV R m_rcp(R x) {
    if constexpr (sizeof(R) == 4) return __builtin_amdgcn_rcpf(x);
    else return R(1) / x;
}
template <class R> RT_DEV R m_div(R a, R b) {
    if constexpr (sizeof(R) == 4) return a * __builtin_amdgcn_rcpf(b);
    else return a / b;
}
template <class R> RT_DEV V3<R> divide(V3<R> v, R t) { return scale<R>(m_rcp(t), v); }                      // vec3.h:86
template <class R> RT_DEV R dot(V3<R> u, V3<R> v) { return u.x * v.x + u.y * v.y + u.z * v.z; }             // vec3.h:90
template <class R> RT_DEV V3<R> cross(V3<R> u, V3<R> v) {                                                   // vec3.h:96
    return mk<R>(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
template <class R> RT_DEV R len2(V3<R> v) { return v.x * v.x + v.y * v.y + v.z * v.z; }                     // vec3.h:41
template <class R> RT_DEV R len(V3<R> v) { return m_sqrt(len2(v)); }
template <class R> RT_DEV V3<R> unit(V3<R> v) { // vec3.h:102
    if constexpr (sizeof(R) == 4) return scale<R>(__builtin_amdgcn_rsqf(len2(v)), v);
    else return divide<R>(v, len(v));
}
template <class R> RT_DEV V3<R> reflect(V3<R> v, V3<R> n) {                                                   // vec3.h:131
    if constexpr (sizeof(R) == 4) return madd<R>(R(-2) * dot(v, n), n, v);
    else return v - scale<R>(R(2) * dot(v, n), n);
}
template <class R> RT_DEV V3<R> refract(V3<R> uv, V3<R> n, R etai_over_etat) {                              // vec3.h:135
    R cos_theta = m_min(dot(-uv, n), R(1));
    V3<R> r_out_perp = scale<R>(etai_over_etat, sizeof(R) == 4 ? madd<R>(cos_theta, n, uv) : uv + scale<R>(cos_theta, n));
    V3<R> r_out_parallel = scale<R>(-m_sqrt(m_abs(R(1) - len2(r_out_perp))), n);
    return r_out_perp + r_out_parallel;
}

template <class R> struct RayT { // ray.h:7 (direction is NOT normalised)
    V3<R> o, d;
    R tm;
};
template <class R> RT_DEV V3<R> ray_at(const RayT<R> &r, R t) { // ray.h:22
    if constexpr (sizeof(R) == 4) return madd<R>(t, r.d, r.o);
    else return r.o + scale<R>(t, r.d);
}

// ------------------------------------------------------------------------------------------------ RNG
// Philox4x32-7, keyed by the 64-bit seed; counter = (block, sample, pixel, stream).  Identical to
// oracle/rt_oracle.cpp's so that f64 renders can be compared path by path.
RT_DEV void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t &o0,
                          uint32_t &o1, uint32_t &o2, uint32_t &o3) {
#ifndef RT_PHILOX_ROUNDS // seven: the fewest rounds Salmon et al. report as Crush-resistant for Philox4x32 (ten: 9 % of the frame)
#define RT_PHILOX_ROUNDS 7
#endif
#pragma unroll
    for (int r = 0; r < RT_PHILOX_ROUNDS; r++) {
        // one v_mad_u64_u32 per product (same issue cost as v_mul_lo_u32 or v_mul_hi_u32 alone)
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0;
        uint32_t h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
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

// Stream protocol (shared with oracle/rt_oracle.cpp): counter = (block, sample, pixel, stream); stream 0 = the camera
// ray's draws (get_ray), 0x80000000|b = the draws of ray_color invocation b of the path (scatter + mixture pdf),
// 1+b (+ dup << 16) = constant_medium draws keyed by node id -- except the scene's PRIMARY medium (the one with the
// largest boundary box, chosen at upload: final_scene's fog, which draws on every ray -- a generator call per ray for it
// was 3 % of the frame): it takes the fifth uniform (fifth()) of the block that started the ray's predecessor (camera
// block for ray 0, bounce b-1's block 0 otherwise), which the lane carries in `med_xi`.  Which uniform a medium gets is
// fixed by its record (header payload: bit 0 = takes the fifth uniform, bits 1.. = dup, the second-visit path of span-1
// bvh_nodes above it), not by the order in which the ray meets the scene's media: the float stream's hierarchy is not
// the reference's.  begin*() computes block 0 unconditionally -- a wave-uniform point, so the common case (<= 4 draws
// per bounce) never runs the generator under a divergent branch.
template <class R> struct Rng {
    uint32_t k0, k1, pixel, sample, stream, block;
    uint32_t b0, b1, b2, b3;
    int have;
    // the fifth uniform of the block just generated: the low bytes of words 0..2 (24 bits no float uniform uses)
    RT_DEV R fifth() const { return R((int)(((b0 & 255u) << 16) | ((b1 & 255u) << 8) | (b2 & 255u))) * R(1.0 / 16777216.0); }
    RT_DEV void refill() {
        philox4x32(block, sample, pixel, stream, k0, k1, b0, b1, b2, b3);
        block++;
        have = 4;
    }
    RT_DEV void begin_stream(uint32_t px, uint32_t s, uint32_t st) { pixel = px; sample = s; stream = st; block = 0; refill(); }
    RT_DEV void begin(uint32_t px, uint32_t s) { pixel = px; sample = s; stream = 0u; block = 0; refill(); }
    RT_DEV void begin_bounce(uint32_t px, uint32_t s, int bounce) {
        pixel = px; sample = s; stream = 0x80000000u | (uint32_t)bounce; block = 0; refill();
    }
    RT_DEV uint32_t next_word() {
        if (have == 0) refill();
        uint32_t x = b0;
        b0 = b1; b1 = b2; b2 = b3;
        have--;
        return x;
    }
    RT_DEV R next() { return u01<R>(next_word()); } // utils.h:33 random_double()
    // Rejection loops draw their candidates from a xorshift32 sub-sequence seeded by one word of the stream (see
    // oracle/rt_oracle.cpp Ctx::sub_begin): a loop never needs a second generator block, whatever it rejects.
    RT_DEV uint32_t sub_begin() {
        uint32_t s = next_word();
        return s == 0u ? 0x9e3779b9u : s;
    }
    static RT_DEV R sub_next(uint32_t &s, R mn, R mx) {
        s ^= s << 13;
        s ^= s >> 17;
        s ^= s << 5;
        return mn + (mx - mn) * u01<R>(s);
    }
    RT_DEV R next(R mn, R mx) { return mn + (mx - mn) * next(); } // utils.h:38
};
// constant_medium.h:41's draw for a medium that does not take the fifth uniform itself: a hash (lowbias32) of the same 24
// bits keyed by the medium's node id and visit path (oracle/rt_oracle.cpp Ctx::medium_hash).  `fifth` is that uniform.
template <class R> RT_DEV R medium_draw(R fifth, int node_id, int dup) {
    uint32_t x = (uint32_t)(fifth * R(16777216.0)) ^ ((uint32_t)node_id * 0x9E3779B9u) ^ ((uint32_t)dup * 0x85EBCA6Bu);
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return u01<R>(x);
}

// ------------------------------------------------------------------------------------------------ scene memory
// The record stream lives in LDS (ALL_LDS: it fits, staged by stage_scene()) or in global memory, as a whole either way:
// no per-load address-space branch.
extern __shared__ __attribute__((aligned(16))) unsigned char rt_smem[];
// The kernels here declare no static __shared__ data, so the dynamic array starts at LDS address 0 (checked at kernel entry:
// lds_origin_check) and an offset into rt_smem IS the LDS address.  Addressing a record as rt_smem + offset costs one v_add of
// the array's link-time address per fetch -- `v_add_u32 v8, 0, v86` in front of every ds_read of the BOX step --; as an absolute
// LDS pointer the offset goes straight into the instruction.  (RT_LDS_ABS=0: the array's symbol, as before.)
#ifndef RT_LDS_ABS
#define RT_LDS_ABS 1
#endif
template <class T> RT_DEV const T *lds_at(unsigned off) {
#if RT_LDS_ABS
    return (const T *)(__attribute__((address_space(3))) const T *)(uintptr_t)off;
#else
    return reinterpret_cast<const T *>(rt_smem + off);
#endif
}
RT_DEV void lds_origin_check() {
#if RT_LDS_ABS
    if (__builtin_amdgcn_groupstaticsize() != 0) __builtin_trap();
#endif
}

template <class R> RT_DEV int ldi(R x) { return dec_int(x); }

// A unit always moves as whole 16-byte vectors (ds_read_b128 / global_load_dwordx4), never as scalar pieces.
typedef float rt_f4 __attribute__((ext_vector_type(4)));
typedef double rt_d2 __attribute__((ext_vector_type(2)));
typedef float rt_f2 __attribute__((ext_vector_type(2)));
RT_DEV UnitT<float> ld_unit(const UnitT<float> *p) {
    const rt_f4 v = *reinterpret_cast<const rt_f4 *>(p);
    return UnitT<float>{{v.x, v.y, v.z, v.w}};
}
RT_DEV UnitT<double> ld_unit(const UnitT<double> *p) {
    const rt_d2 a = reinterpret_cast<const rt_d2 *>(p)[0], b = reinterpret_cast<const rt_d2 *>(p)[1];
    return UnitT<double>{{a.x, a.y, b.x, b.y}};
}

// Positions in the stream are BYTE offsets (cursor, skip links, the winning record): a unit's LDS address is then the
// position plus a constant, with no shift in the hot BOX step.
template <class R> struct Pos {
    static constexpr int unit = (int)sizeof(UnitT<R>);
};
template <class R, bool ALL_LDS> struct Mem {
    const UnitT<R> *glob;
    int lds_bytes;
    RT_DEV UnitT<R> ld(int pos) const {
        if constexpr (ALL_LDS) {
            return ld_unit(lds_at<UnitT<R>>((unsigned)(LdsLayout<R>::stream_off + pos)));
        } else {
            // a stream that does not fit is read from global memory as a whole (its cache lines live in L2 / the vector L1):
            // a resident PREFIX with a per-load address-space branch measured slower than no prefix at all (mesh_scene 444 vs
            // 406 ms; final_scene forced to 98 % resident: 756 against 555 fully resident), and LDS then holds the tables
            return ld_unit(reinterpret_cast<const UnitT<R> *>(reinterpret_cast<const unsigned char *>(glob) + pos));
        }
    }
};
template <class R> RT_DEV UnitT<R> cold_at(const UnitT<R> *cold, int pos) {
    return *reinterpret_cast<const UnitT<R> *>(reinterpret_cast<const unsigned char *>(cold) + pos);
}

// What the kernels see of the scene's shading tables.  TAB_LDS: the tables were staged into LDS behind the stream and are
// addressed as offsets into rt_smem, so the lookups compile to ds_read (a pointer that may be LDS or global would make
// every lookup a flat load).  Otherwise: offsets from the device pointer `tab`.
template <class R, bool TAB_LDS> struct SceneV {
    const UnitT<R> *cold;
    const uint8_t *image_bytes;
    const uint8_t *tab;
    uint32_t o_xforms, o_mats, o_texs, o_images, o_perlins, o_lights;
    int32_t n_lights, lights_is_list;
    template <class T> RT_DEV const T *at(uint32_t off) const {
        if constexpr (TAB_LDS) return lds_at<T>(off);
        else return reinterpret_cast<const T *>(tab + off);
    }
    RT_DEV const XformT<R> &xform(int i) const { return at<XformT<R>>(o_xforms)[i]; }
    RT_DEV const MatT<R> &mat(int i) const { return at<MatT<R>>(o_mats)[i]; }
    RT_DEV const TexT<R> &tex(int i) const { return at<TexT<R>>(o_texs)[i]; }
    RT_DEV const ImageT &image(int i) const { return at<ImageT>(o_images)[i]; }
    RT_DEV const PerlinT<R> &perlin(int i) const { return at<PerlinT<R>>(o_perlins)[i]; }
    RT_DEV const LightT<R> &light(int i) const { return at<LightT<R>>(o_lights)[i]; }
};

// Stage the stream (if it is to live there) and, for TAB_LDS, the shading tables into LDS; returns the kernels' view of the tables.
template <class R, bool TAB_LDS> RT_DEV SceneV<R, TAB_LDS> stage_scene(const SceneT<R> &sc) {
    UnitT<R> *l = reinterpret_cast<UnitT<R> *>(rt_smem + LdsLayout<R>::stream_off);
    for (int k = threadIdx.x; k < sc.lds_units; k += blockDim.x) l[k] = sc.stream[k];
    uint32_t base = 0;
    if constexpr (TAB_LDS) {
        base = (uint32_t)LdsLayout<R>::stream_off + (uint32_t)sc.lds_units * (uint32_t)sizeof(UnitT<R>);
        const rt_f4 *src = reinterpret_cast<const rt_f4 *>(sc.tab);
        rt_f4 *d4 = reinterpret_cast<rt_f4 *>(rt_smem + base);
        for (int k = threadIdx.x; k < sc.tab_bytes / 16; k += blockDim.x) d4[k] = src[k];
    }
    SceneV<R, TAB_LDS> v;
    v.cold = sc.cold;
    v.image_bytes = sc.image_bytes;
    v.tab = sc.tab;
    auto off = [&](const void *p) { return base + (uint32_t)(reinterpret_cast<const uint8_t *>(p) - sc.tab); };
    v.o_xforms = off(sc.xforms);
    v.o_mats = off(sc.mats);
    v.o_texs = off(sc.texs);
    v.o_images = off(sc.images);
    v.o_perlins = off(sc.perlins);
    v.o_lights = off(sc.lights);
    v.n_lights = sc.n_lights;
    v.lights_is_list = sc.lights_is_list;
    __syncthreads();
    return v;
}

// ------------------------------------------------------------------------------------------------ intersection
// aabb::hit (aabb.h:36) over the interval (tmin, tmax).  The double path keeps the reference's statement order; the
// float path computes the same slab distances and folds them with min/max.
template <class R>
RT_DEV bool box_hit(const UnitT<R> &u0, const UnitT<R> &u1, V3<R> o, V3<R> inv, V3<R> noi, R tmin, R tmax) {
    const R bx0 = u0.w[0], bx1 = u0.w[1], by0 = u0.w[2], by1 = u0.w[3], bz0 = u1.w[0], bz1 = u1.w[1];
    if constexpr (sizeof(R) == 4) {
        // (min, max) pairs sit in adjacent registers: one v_pk_fma_f32 does both slab planes of an axis, (b - o) * inv
        // as b * inv + (-o * inv).  inv is finite (inv_dir_trav): a ray parallel to the slab gets +-1e30-sized distances with
        // the signs of b - o, i.e. the axis passes exactly when the origin lies inside the slab, as in the reference.
        const rt_f2 tx = rt_f2{bx0, bx1} * inv.x + noi.x, ty = rt_f2{by0, by1} * inv.y + noi.y, tz = rt_f2{bz0, bz1} * inv.z + noi.z;
        // (Calibration, round 3: two / four dummy vector instructions here -- +3.2 % / +6.4 % of the kernel's instructions, off
        // the step's dependent chain -- cost +1.6 % / +1.9 % of the frame; taking the one v_add per record fetch out of the
        // chain, 2.1 % of the instructions (lds_at), gained 0.3 %.  The vector unit is busy 96 % of the time AND the
        // step's chain -- two LDS reads, fma, min, max3, compare, select, the next address -- is what a wave waits on:
        // fewer instructions alone no longer buy time.)
        R tn = m_max(m_max(m_min(tx.x, tx.y), m_min(ty.x, ty.y)), m_max(m_min(tz.x, tz.y), tmin));
        // (tmax is the closest hit so far, a value the compiler cannot prove canonical: through fminf it would first be
        // quieted by a v_max_f32 x, x; the instruction itself needs no such thing)
        float far_z;
        asm("v_min_f32 %0, %1, %2" : "=v"(far_z) : "v"(m_max(tz.x, tz.y)), "v"(tmax));
        R tf = m_min(m_min(m_max(tx.x, tx.y), m_max(ty.x, ty.y)), far_z);
        return tf >= tn; // (a box thinner than the float resolution of its distances still counts: rt_flatten.h pads, too)
    } else {
        {
            R t0 = (bx0 - o.x) * inv.x, t1 = (bx1 - o.x) * inv.x;
            if (t0 < t1) { if (t0 > tmin) tmin = t0; if (t1 < tmax) tmax = t1; }
            else { if (t1 > tmin) tmin = t1; if (t0 < tmax) tmax = t0; }
            if (tmax <= tmin) return false;
        }
        {
            R t0 = (by0 - o.y) * inv.y, t1 = (by1 - o.y) * inv.y;
            if (t0 < t1) { if (t0 > tmin) tmin = t0; if (t1 < tmax) tmax = t1; }
            else { if (t1 > tmin) tmin = t1; if (t0 < tmax) tmax = t0; }
            if (tmax <= tmin) return false;
        }
        {
            R t0 = (bz0 - o.z) * inv.z, t1 = (bz1 - o.z) * inv.z;
            if (t0 < t1) { if (t0 > tmin) tmin = t0; if (t1 < tmax) tmax = t1; }
            else { if (t1 > tmin) tmin = t1; if (t0 < tmax) tmax = t0; }
            if (tmax <= tmin) return false;
        }
        return true;
    }
}

// Fused box() primitive (f32 stream only): the six quads of quad.h:103 as one slab test.  The ray meets the box
// surface at the slab entry and exit distances; quad::hit accepts tmin <= t <= tmax (interval::contains, quad.h:41).
template <class R>
RT_DEV bool aabox_t(const UnitT<R> &u0, const UnitT<R> &u1, V3<R> o, V3<R> inv, R tmin, R tmax, R &t_out) {
    R tx0 = (u0.w[0] - o.x) * inv.x, tx1 = (u0.w[1] - o.x) * inv.x;
    R ty0 = (u0.w[2] - o.y) * inv.y, ty1 = (u0.w[3] - o.y) * inv.y;
    R tz0 = (u1.w[0] - o.z) * inv.z, tz1 = (u1.w[1] - o.z) * inv.z;
    R tn = m_max(m_max(m_min(tx0, tx1), m_min(ty0, ty1)), m_min(tz0, tz1));
    R tf = m_min(m_min(m_max(tx0, tx1), m_max(ty0, ty1)), m_max(tz0, tz1));
    if (!(tf >= tn)) return false;
    R t = tn >= tmin ? tn : tf;
    if (!(tmin <= t && t <= tmax)) return false;
    t_out = t;
    return true;
}

// Float arithmetic with the fused multiply-adds WRITTEN OUT and no others formed (contract off): under -ffp-contract=fast
// the compiler fuses each inlined COPY of a routine as it sees fit, and the same test then rounds differently in the
// first and in the repeated step of a block's loop -- c = |oc|^2 - r^2 came out as fma in one copy and mul + sub in the
// other, one ulp apart, times the cancellation of first_scene's r = 1000 ground.  Which copy a lane runs depends on the
// wave's schedule, so frames did not repeat (values one float ulp apart, now and then a different ray count).
RT_DEV float fdot3(V3<float> a, V3<float> b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }

// (Tried in round 3: the traversal walking rays of UNIT direction -- a = d.d = 1 takes a reciprocal and four multiplies out of
// every sphere / medium test and the square root and division out of the medium's path length -- with camera.h:300's 0.001
// and constant_medium.h:27's 0.0001 scaled by |d| per ray.  The per-ray interval minimum is one more register of loop-carried
// state and the normalisation runs in every shading trip: final_scene 517 -> 523 ms, first_scene 113.6 -> 117.1.  Not kept.)
template <class R> RT_DEV bool sphere_t(V3<R> cc, R radius, V3<R> ro, V3<R> rd, R tmin, R tmax, R &t_out) { // sphere.h:29
    if constexpr (sizeof(R) == 4) {
#pragma clang fp contract(off)
        // Same quadratic, conditioned for float: h*h - a*c cancels catastrophically when |oc| >> radius (first_scene's
        // r=1 light 1000 units away, the r=1000 ground, final_scene's r=5000 fog).  disc = a*(r^2 - |oc - (h/a)d|^2) and
        // the roots c/q, q/a with q = h + sign(h)*sqrt(disc) are algebraically identical and lose no digits.
        const V3<R> oc = cc - ro;
        const R a = fdot3(rd, rd);
        const R h = fdot3(rd, oc);
        const R c = __builtin_fmaf(-radius, radius, fdot3(oc, oc));
        const R inva = m_rcp(a);
        const R hia = h * inva;
        const V3<R> perp = mk<R>(__builtin_fmaf(-hia, rd.x, oc.x), __builtin_fmaf(-hia, rd.y, oc.y), __builtin_fmaf(-hia, rd.z, oc.z));
        const R disc = a * __builtin_fmaf(radius, radius, -fdot3(perp, perp));
        if (disc < R(0)) return false;
        const R sq = m_sqrt(disc);
        const R q = h + (h < R(0) ? -sq : sq);
        R ta = m_div(c, q), tb = q * inva;
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
        V3<R> oc = cc - ro;
        R a = len2(rd);
        R h = dot(rd, oc);
        R c = len2(oc) - radius * radius;
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
template <class R> RT_DEV bool sphere_t(const SphereT<R> &s, const RayT<R> &r, R tmin, R tmax, R &t_out) {
    return sphere_t(ld3(s.c) + scale<R>(r.tm, ld3(s.d)), s.radius, r.o, r.d, tmin, tmax, t_out);
}

template <class R>
RT_DEV bool quad_t(V3<R> n, R D, V3<R> Q, V3<R> qu, V3<R> qv, V3<R> w, V3<R> ro, V3<R> rd, R tmin, R tmax, R &t_out, R &alpha,
                   R &beta) { // quad.h:32
    R denom = dot(n, rd);
    if (m_abs(denom) < R(1e-8)) return false;
    R t = m_div(D - dot(n, ro), denom);
    if (!(tmin <= t && t <= tmax)) return false; // interval::contains
    V3<R> ph = (ro + scale<R>(t, rd)) - Q;
    alpha = dot(w, cross(ph, qv));
    beta = dot(w, cross(qu, ph));
    if (!(R(0) <= alpha && alpha <= R(1)) || !(R(0) <= beta && beta <= R(1))) return false; // quad.h:62
    t_out = t;
    return true;
}
template <class R>
RT_DEV bool quad_t(const QuadT<R> &q, const RayT<R> &r, R tmin, R tmax, R &t_out, R &alpha, R &beta) {
    return quad_t(ld3(q.n), q.D, ld3(q.Q), ld3(q.u), ld3(q.v), ld3(q.w), r.o, r.d, tmin, tmax, t_out, alpha, beta);
}

template <class R>
RT_DEV bool tri_t(V3<R> v0, V3<R> edge1, V3<R> edge2, int cull, V3<R> ro, V3<R> rd, R tmin, R tmax, R &t_out, float &u_out,
                  float &v_out) { // triangle.h:23
    const float EPSILON = 0.000001; // triangle.h:16 (const float): a, f, u, v, t are float in the reference
    V3<R> h = cross(rd, edge2);
    float a = (float)dot(edge1, h);
    if (cull && a < EPSILON) return false;
    if (fabsf(a) < EPSILON) return false;
    float f = (float)(R(1.0) / R(a));
    V3<R> s = ro - v0;
    float u = (float)(R(f) * dot(s, h));
    if (u < 0.0f || u > 1.0f) return false;
    V3<R> q = cross(s, edge1);
    float v = (float)(R(f) * dot(rd, q));
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = (float)(R(f) * dot(edge2, q));
    if (!(tmin <= R(t) && R(t) <= tmax)) return false;
    t_out = R(t);
    u_out = u;
    v_out = v;
    return true;
}

// constant_medium.h:30-44: the boundary is entered at t1 and left at t2 (both hits found); clip to the ray's interval, draw
// the free path.  The uniform is drawn only where the reference draws it (line 41).
template <class R, class XiFn>
RT_DEV bool medium_span(R t1, R t2, R neg_inv_density, int node_id, int sel, V3<R> rd, R tmin, R tmax, XiFn xi_fn, R &t_out) {
    if (t1 < tmin) t1 = tmin;
    if (t2 > tmax) t2 = tmax;
    if (t1 >= t2) return false;
    if (t1 < R(0)) t1 = R(0);
    R ray_length = len(rd);
    R distance_inside_boundary = (t2 - t1) * ray_length;
    R hit_distance = neg_inv_density * m_log(xi_fn(node_id, sel)); // sel = the record's header payload (see Rng)
    if (hit_distance > distance_inside_boundary) return false;
    t_out = t1 + m_div(hit_distance, ray_length);
    return true;
}

// constant_medium.h:21 (boundary is a static sphere)
template <class R, class XiFn>
RT_DEV bool medium_t(V3<R> cc, R radius, R neg_inv_density, int node_id, int sel, V3<R> ro, V3<R> rd, R tmin, R tmax,
                     XiFn xi_fn, R &t_out) {
    R t1, t2;
    if constexpr (sizeof(R) == 4) {
        // both boundary hits from ONE solve of the quadratic (same conditioned form as sphere_t): over the whole line the
        // first hit is the smaller root; the second search, over (t1 + 0.0001, inf), can only return the larger one
        V3<R> oc = cc - ro;
        R a = len2(rd), h = dot(rd, oc), c = len2(oc) - radius * radius;
        R inva = m_rcp(a);
        V3<R> perp = oc - scale<R>(h * inva, rd);
        R disc = a * (radius * radius - len2(perp));
        if (disc < R(0)) return false;
        R sq = m_sqrt(disc);
        R q = h + (h < R(0) ? -sq : sq);
        R ta = m_div(c, q), tb = q * inva;
        t1 = m_min(ta, tb);
        t2 = m_max(ta, tb);
        if (q == R(0)) { t1 = R(0); t2 = R(0); }
        if (!(t1 + R(0.0001) < t2)) return false;
    } else {
        if (!sphere_t(cc, radius, ro, rd, -Const<R>::inf, Const<R>::inf, t1)) return false;
        if (!sphere_t(cc, radius, ro, rd, t1 + R(0.0001), Const<R>::inf, t2)) return false;
    }
    return medium_span(t1, t2, neg_inv_density, node_id, sel, rd, tmin, tmax, xi_fn, t_out);
}

template <class R> RT_DEV void xf_apply(int kind, R p0, R p1, R p2, V3<R> &o, V3<R> &d) {
    if (kind == XF_TRANSLATE) { // hittable.h:55
        o = o - mk<R>(p0, p1, p2);
    } else { // hittable.h:114-124
        R s = p0, c = p1;
        o = mk<R>((c * o.x) - (s * o.z), o.y, (s * o.x) + (c * o.z));
        d = mk<R>((c * d.x) - (s * d.z), d.y, (s * d.x) + (c * d.z));
    }
}

template <class R, class SC> RT_DEV RayT<R> ray_in_ctx(const SC &sc, const RayT<R> &world, int ctx) {
    RayT<R> r = world;
    if (ctx != 0) {
        const int depth = sc.xform(ctx).depth;
        for (int k = 0; k < depth; k++) {
            const XformT<R> x = sc.xform(sc.xform(ctx).chain[k]);
            xf_apply(x.kind, x.p[0], x.p[1], x.p[2], r.o, r.d);
        }
    }
    return r;
}

template <class R> RT_DEV V3<R> inv_dir(V3<R> d) { return mk<R>(m_rcp(d.x), m_rcp(d.y), m_rcp(d.z)); } // aabb.h:42
// The inverse direction the TRAVERSAL keeps.  double: the reference's 1.0 / d (+-inf for a zero component; (b - o) * inf has
// the sign of b - o, aabb.h:42-53).  float: the slab distances are evaluated as b * inv + (-o * inv), where an infinite inv
// turns one plane of the axis into inf - inf = NaN and the slab into a reject although the origin lies inside it; |inv| is
// therefore kept finite (one v_med3_f32 per component per ray): b * 1e30 - o * 1e30 keeps the sign of b - o for every
// coordinate below 3e8, which is all the slab test needs of a ray parallel to the slab.
template <class R> RT_DEV V3<R> inv_dir_trav(V3<R> d) {
    V3<R> inv = inv_dir(d);
    if constexpr (sizeof(R) == 4) {
        inv.x = __builtin_amdgcn_fmed3f(inv.x, -1e30f, 1e30f);
        inv.y = __builtin_amdgcn_fmed3f(inv.y, -1e30f, 1e30f);
        inv.z = __builtin_amdgcn_fmed3f(inv.z, -1e30f, 1e30f);
    }
    return inv;
}

// Per-ray traversal state of world.hit(r, interval(tmin, inf), rec): a cursor into the threaded stream (pre-order ==
// the reference's recursion order), the ray in the current transform context and the closest hit so far.
template <class R> struct Trav {
    V3<R> o, d, inv; // ray in the space of context `ctx`
    V3<R> noi;       // float path: -o * inv, so that a slab distance is one fma (b * inv + noi)
    R best_t;
    int best_off; // position (byte offset) of the winning record, -1 = miss
    int i; // cursor (byte offset of the record)
    // the record at the cursor
    UnitT<R> u0, u1;
    int hdr;
};

// Move the cursor.  No bounds check: the stream ends in a sentinel record whose header is K_READY ("traversal
// finished"), and every skip link that leaves the last subtree points at it.
template <class R, bool ALL_LDS> RT_DEV void trav_goto(const Mem<R, ALL_LDS> &mem, Trav<R> &tv, int i) {
    tv.i = i;
    tv.u0 = mem.ld(i);
    tv.u1 = mem.ld(i + Pos<R>::unit);
    tv.hdr = ldi(tv.u1.w[3]);
}

template <class R, bool ALL_LDS>
RT_DEV void trav_begin(const Mem<R, ALL_LDS> &mem, Trav<R> &tv, const RayT<R> &world) {
    tv.o = world.o;
    tv.d = world.d;
    tv.inv = inv_dir_trav(world.d);
    if constexpr (sizeof(R) == 4) tv.noi = -(tv.o * tv.inv);
    tv.best_t = Const<R>::inf;
    tv.best_off = -1;
    trav_goto(mem, tv, 0);
}

// One BOX record: descend (next record) or take the skip link.
template <class R, bool ALL_LDS>
RT_DEV void trav_box_step(const Mem<R, ALL_LDS> &mem, Trav<R> &tv, R tmin) {
    const bool hit = box_hit(tv.u0, tv.u1, tv.o, tv.inv, tv.noi, tmin, tv.best_t);
    trav_goto(mem, tv, hit ? tv.i + 2 * Pos<R>::unit : ldi(tv.u1.w[2]));
}

// One non-BOX record.  GM: the stream may hold MEDIUM_GEN records (a constant_medium whose boundary is a sub-stream).
template <bool GM, class R, bool ALL_LDS, class SC, class MediumXi>
RT_DEV void trav_leaf_step(const Mem<R, ALL_LDS> &mem, const SC &sc, Trav<R> &tv, const RayT<R> &world, R tmin,
                           MediumXi medium_xi) {
    const int kind = tv.hdr & 15;
    constexpr int U = Pos<R>::unit;
    const int i = tv.i;
    int next = i + 2 * U;
    if (kind == K_SPHERE || kind == K_MSPHERE) {
        V3<R> cc = mk<R>(tv.u0.w[0], tv.u0.w[1], tv.u0.w[2]);
        if (kind == K_MSPHERE) { // moving sphere: centre = c + time*d (sphere.h:30)
            const UnitT<R> u2 = mem.ld(i + 2 * U);
            cc = cc + scale<R>(world.tm, mk<R>(u2.w[0], u2.w[1], u2.w[2]));
            next = i + 3 * U;
        } else if constexpr (sizeof(R) == 8) {
            cc = cc + scale<R>(world.tm, mk<R>(R(0), R(0), R(0))); // the reference always evaluates center.at(time)
        }
        R t;
        if (sphere_t(cc, tv.u0.w[3], tv.o, tv.d, tmin, tv.best_t, t)) { tv.best_t = t; tv.best_off = i; }
    } else if (kind == K_AABOX) {
        R t;
        if (aabox_t(tv.u0, tv.u1, tv.o, tv.inv, tmin, tv.best_t, t)) { tv.best_t = t; tv.best_off = i; }
    } else if (kind == K_QUAD) {
        const UnitT<R> u2 = mem.ld(i + 2 * U), u3 = mem.ld(i + 3 * U), u4 = mem.ld(i + 4 * U);
        R t, al, be;
        if (quad_t(mk<R>(tv.u0.w[0], tv.u0.w[1], tv.u0.w[2]), tv.u0.w[3], mk<R>(tv.u1.w[0], tv.u1.w[1], tv.u1.w[2]),
                   mk<R>(u2.w[0], u2.w[1], u2.w[2]), mk<R>(u3.w[0], u3.w[1], u3.w[2]), mk<R>(u4.w[0], u4.w[1], u4.w[2]), tv.o, tv.d,
                   tmin, tv.best_t, t, al, be)) {
            tv.best_t = t; tv.best_off = i;
        }
        next = i + 5 * U;
    } else if (kind == K_TRI) {
        const UnitT<R> u2 = mem.ld(i + 2 * U);
        R t;
        float u, v;
        if (tri_t(mk<R>(tv.u0.w[0], tv.u0.w[1], tv.u0.w[2]), mk<R>(tv.u0.w[3], tv.u1.w[0], tv.u1.w[1]),
                  mk<R>(tv.u1.w[2], u2.w[0], u2.w[1]), ldi(u2.w[3]), tv.o, tv.d, tmin, tv.best_t, t, u, v)) {
            tv.best_t = t; tv.best_off = i;
        }
        next = i + 3 * U;
    } else if (kind == K_MEDIUM) {
        R t;
        if (medium_t(mk<R>(tv.u0.w[0], tv.u0.w[1], tv.u0.w[2]), tv.u0.w[3], tv.u1.w[0], ldi(tv.u1.w[2]), hdr_payload(tv.hdr), tv.o, tv.d,
                     tmin, tv.best_t, medium_xi, t)) {
            tv.best_t = t; tv.best_off = i;
        }
    } else if (GM && kind == K_MEDIUM_GEN) {
        if constexpr (GM) {
            // constant_medium.h:24-28: boundary->hit(r, universe, rec1), then boundary->hit(r, (rec1.t + 0.0001, inf), rec2) --
            // two closest-hit walks of the boundary's sub-stream (its own records, transforms and sentinel), starting in
            // this medium's frame.  Only the distances are used.  (A medium inside a boundary is refused at upload, so
            // the walk never nests.)
            const int sub_off = ldi(tv.u0.w[0]);
            auto boundary_hit = [&](R lo, R &t_hit) {
                Trav<R> s;
                s.o = tv.o; s.d = tv.d; s.inv = tv.inv; s.noi = tv.noi;
                s.best_t = Const<R>::inf; s.best_off = -1;
                trav_goto(mem, s, sub_off);
                while ((s.hdr & 15) != K_READY) {
                    if ((s.hdr & 15) == K_BOX) trav_box_step(mem, s, lo);
                    else trav_leaf_step<false>(mem, sc, s, world, lo, medium_xi);
                }
                t_hit = s.best_t;
                return s.best_off >= 0;
            };
            R t1, t2, t;
            if (boundary_hit(-Const<R>::inf, t1) && boundary_hit(t1 + R(0.0001), t2) &&
                medium_span(t1, t2, tv.u1.w[0], ldi(tv.u1.w[2]), hdr_payload(tv.hdr), tv.d, tmin, tv.best_t, medium_xi, t)) {
                tv.best_t = t; tv.best_off = i;
            }
        }
    } else if (kind == K_OBOX) {
        // translate(rotate_y(box())): hittable.h:53-65 moves the ray by -offset, hittable.h:110-124 turns it, quad.h:103's
        // six quads are one slab test in that frame; t is the same number in both frames (rigid transforms, d not normalised)
        const UnitT<R> u2 = mem.ld(i + 2 * U), u3 = mem.ld(i + 3 * U);
        V3<R> o = tv.o, d = tv.d;
        xf_apply(XF_TRANSLATE, u2.w[2], u2.w[3], u3.w[0], o, d);
        xf_apply(XF_ROTATE_Y, u2.w[0], u2.w[1], R(0), o, d);
        R t;
        if (aabox_t(tv.u0, tv.u1, o, inv_dir_trav(d), tmin, tv.best_t, t)) { tv.best_t = t; tv.best_off = i; }
        next = i + 4 * U;
    } else if (kind == K_XF_ENTER) {
        xf_apply(ldi(tv.u1.w[0]), tv.u0.w[0], tv.u0.w[1], tv.u0.w[2], tv.o, tv.d);
        tv.inv = inv_dir_trav(tv.d);
        if constexpr (sizeof(R) == 4) tv.noi = -(tv.o * tv.inv);
    } else if (kind == K_XF_EXIT) {
        if constexpr (sizeof(R) == 4) {
            // leave the frame by the inverse transform: the world-space ray need not stay in registers during traversal
            if (ldi(tv.u1.w[0]) == XF_TRANSLATE) {
                tv.o = tv.o + mk<R>(tv.u0.w[0], tv.u0.w[1], tv.u0.w[2]);
            } else {
                const R s = tv.u0.w[0], c = tv.u0.w[1];
                tv.o = mk<R>((c * tv.o.x) + (s * tv.o.z), tv.o.y, (c * tv.o.z) - (s * tv.o.x));
                tv.d = mk<R>((c * tv.d.x) + (s * tv.d.z), tv.d.y, (c * tv.d.z) - (s * tv.d.x));
            }
            tv.inv = inv_dir_trav(tv.d);
            tv.noi = -(tv.o * tv.inv);
        } else { // the reference's outer ray is untouched by the instance (hittable.h:53-65): restore it exactly
            const RayT<R> r = ray_in_ctx(sc, world, ldi(tv.u1.w[1])); // the frame outside the instance
            tv.o = r.o;
            tv.d = r.d;
            tv.inv = inv_dir(r.d);
        }
    }
    trav_goto(mem, tv, next);
}

template <class R> struct Hit { // hittable.h:10 hit_record
    V3<R> p, n;
    R t, u, v;
    int mat;
    int kind, off;
    int prim; // description node id of the primitive
    bool front;
};

template <class R> RT_DEV void sphere_uv(V3<R> p, R &u, R &v) { // sphere.h:89
    R ny = -p.y;
    if constexpr (sizeof(R) == 4) ny = m_max(R(-1), m_min(R(1), ny)); // (p-c)/r can exceed 1 by an ulp in float
    R theta = m_acos(ny);
    R phi = m_atan2(-p.z, p.x) + Const<R>::pi;
    u = m_div(phi, R(2) * Const<R>::pi);
    v = m_div(theta, Const<R>::pi);
}

// Re-derive the full hit record of the winning primitive (same arithmetic as the reference's success path).
template <class R, bool ALL_LDS, class SC>
RT_DEV Hit<R> finalize(const Mem<R, ALL_LDS> &mem, const SC &sc, const RayT<R> &world, R best_t, int best_off, bool want_all) {
    Hit<R> h;
    h.t = best_t;
    h.off = best_off;
    h.u = R(0);
    h.v = R(0);
    h.prim = -1;
    constexpr int U = Pos<R>::unit;
    const UnitT<R> u0 = mem.ld(best_off), u1 = mem.ld(best_off + U);
    const int hdr = ldi(u1.w[3]);
    h.kind = hdr & 15;
    const int best_ctx = hdr_ctx(hdr);
    const RayT<R> r = ray_in_ctx(sc, world, best_ctx);
    h.p = ray_at(r, best_t);
    if (h.kind == K_SPHERE || h.kind == K_MSPHERE) {
        V3<R> cc = mk<R>(u0.w[0], u0.w[1], u0.w[2]);
        if (h.kind == K_MSPHERE) {
            const UnitT<R> u2 = mem.ld(best_off + 2 * U);
            cc = cc + scale<R>(r.tm, mk<R>(u2.w[0], u2.w[1], u2.w[2]));
        } else if constexpr (sizeof(R) == 8) {
            cc = cc + scale<R>(r.tm, mk<R>(R(0), R(0), R(0)));
        }
        V3<R> outward = divide<R>(h.p - cc, u0.w[3]);
        h.front = dot(r.d, outward) < R(0);
        h.n = h.front ? outward : -outward;
        h.mat = ldi(u1.w[0]);
        if (want_all || sc.mat(h.mat).needs_uv) sphere_uv(outward, h.u, h.v);
        if (want_all) h.prim = ldi(cold_at(sc.cold, best_off).w[3]);
    } else if (h.kind == K_AABOX || h.kind == K_OBOX) {
        RayT<R> rb = r; // the ray in the box's frame
        UnitT<R> u2{}, u3{};
        if (h.kind == K_OBOX) {
            u2 = mem.ld(best_off + 2 * U);
            u3 = mem.ld(best_off + 3 * U);
            xf_apply(XF_TRANSLATE, u2.w[2], u2.w[3], u3.w[0], rb.o, rb.d);
            xf_apply(XF_ROTATE_Y, u2.w[0], u2.w[1], R(0), rb.o, rb.d);
            h.p = ray_at(rb, best_t);
        }
        // which face: the slab plane whose distance is the winning t
        // (tried: the plane whose coordinate is closest to the hit POINT's -- six differences instead of six subtract-multiply-
        // compare probes and three reciprocals: final_scene 510.6 -> 509.2 ms, and 197 more of the 6 694 rays aimed at box edges
        // and corners name the neighbouring face (all ties); the distance names the reference's face there: kept)
        const V3<R> inv = inv_dir(rb.d);
        int face = 0;
        R err = Const<R>::inf;
        auto probe = [&](R plane, R ro, R ri, int f) {
            R e = m_abs((plane - ro) * ri - best_t);
            if (e < err) { err = e; face = f; }
        };
        probe(u0.w[0], rb.o.x, inv.x, 0);
        probe(u0.w[1], rb.o.x, inv.x, 1);
        probe(u0.w[2], rb.o.y, inv.y, 2);
        probe(u0.w[3], rb.o.y, inv.y, 3);
        probe(u1.w[0], rb.o.z, inv.z, 4);
        probe(u1.w[1], rb.o.z, inv.z, 5);
        // a box of no thickness is an axis-aligned quad (rt_flatten.h): its face is the flat axis', whatever edge the hit lies on
        if (u0.w[0] == u0.w[1]) face = 0;
        if (u0.w[2] == u0.w[3]) face = 2;
        if (u1.w[0] == u1.w[1]) face = 4;
        const int axis = face >> 1;
        const R sgn = (face & 1) ? R(1) : R(-1); // outward direction of this face
        V3<R> outward = mk<R>(axis == 0 ? sgn : R(0), axis == 1 ? sgn : R(0), axis == 2 ? sgn : R(0));
        const bool stored_outward = (hdr_payload(hdr) >> face) & 1;
        const V3<R> n = stored_outward ? outward : -outward; // the quad's own normal (quad.h:15)
        h.front = dot(rb.d, n) < R(0);
        h.n = h.front ? n : -n;
        h.mat = ldi(u1.w[2]);
        if (want_all) {
            const UnitT<R> c = cold_at(sc.cold, best_off + (face >> 2) * U);
            const int k = face & 3;
            h.prim = ldi(k == 0 ? c.w[0] : (k == 1 ? c.w[1] : (k == 2 ? c.w[2] : c.w[3])));
        }
        if (h.kind == K_OBOX) { // back to the frame the record stands in: rotate (hittable.h:135-145), then the offset (hittable.h:62)
            const R sn = u2.w[0], cs = u2.w[1];
            h.p = mk<R>((cs * h.p.x) + (sn * h.p.z), h.p.y, (-sn * h.p.x) + (cs * h.p.z)) + mk<R>(u2.w[2], u2.w[3], u3.w[0]);
            h.n = mk<R>((cs * h.n.x) + (sn * h.n.z), h.n.y, (-sn * h.n.x) + (cs * h.n.z));
        }
    } else if (h.kind == K_QUAD) {
        const UnitT<R> u2 = mem.ld(best_off + 2 * U), u3 = mem.ld(best_off + 3 * U), u4 = mem.ld(best_off + 4 * U);
        V3<R> ph = h.p - mk<R>(u1.w[0], u1.w[1], u1.w[2]);
        V3<R> w = mk<R>(u4.w[0], u4.w[1], u4.w[2]);
        h.u = dot(w, cross(ph, mk<R>(u3.w[0], u3.w[1], u3.w[2])));
        h.v = dot(w, cross(mk<R>(u2.w[0], u2.w[1], u2.w[2]), ph));
        V3<R> n = mk<R>(u0.w[0], u0.w[1], u0.w[2]);
        h.front = dot(r.d, n) < R(0);
        h.n = h.front ? n : -n;
        h.mat = ldi(u2.w[3]);
        if (want_all) h.prim = ldi(cold_at(sc.cold, best_off).w[3]);
    } else if (h.kind == K_TRI) {
        const UnitT<R> u2 = mem.ld(best_off + 2 * U);
        R t;
        float u = 0.f, v = 0.f;
        tri_t(mk<R>(u0.w[0], u0.w[1], u0.w[2]), mk<R>(u0.w[3], u1.w[0], u1.w[1]), mk<R>(u1.w[2], u2.w[0], u2.w[1]), ldi(u2.w[3]),
              r.o, r.d, -Const<R>::inf, Const<R>::inf, t, u, v);
        h.u = R(u);
        h.v = R(v);
        const UnitT<R> c = cold_at(sc.cold, best_off);
        V3<R> n = mk<R>(c.w[0], c.w[1], c.w[2]);
        h.front = dot(r.d, n) < R(0);
        h.n = h.front ? n : -n;
        h.mat = ldi(u2.w[2]);
        h.prim = ldi(c.w[3]);
    } else { // K_MEDIUM, constant_medium.h:46-51
        h.n = mk<R>(R(1), R(0), R(0));
        h.front = true;
        h.mat = ldi(u1.w[1]);
        h.prim = ldi(u1.w[2]);
    }
    if (best_ctx != 0) { // leave object space, innermost transform first (hittable.h:62, 135-145)
        const XformT<R> *X = &sc.xform(best_ctx);
        for (int k = X->depth - 1; k >= 0; k--) {
            const XformT<R> x = sc.xform(X->chain[k]);
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
#pragma unroll 1
    for (int i = 0; i < depth; i++) {
        accum += weight * perlin_noise(pn, p);
        weight *= R(0.5);
        p = mk<R>(p.x * R(2), p.y * R(2), p.z * R(2));
    }
    return m_abs(accum);
}

// perlin_turb for ONE point by the whole wave (float kernels; all 64 lanes must be active).  A noise texture is rare on
// screen -- final_scene's marble sphere covers a percent of the image -- so of the 45 lanes of a shading trip one or two
// need it, and the 7 octaves x 8 lattice corners they would walk alone (900 instructions, run by 3 of 4 trips: 4 % of
// the frame) are 56 independent terms: lane l evaluates corner (l & 7) of octave (l >> 3), a butterfly sum adds them.
// The sum's order differs from perlin.h:38-47 (float rounding only; the double path keeps the serial loop).
RT_DEV float perlin_turb_wave(const PerlinT<float> &pn, V3<float> p, int depth, int lane) {
    float total = 0.0f;
    const int a = (lane >> 2) & 1, b = (lane >> 1) & 1, d = lane & 1;
    for (int base = 0; base < depth; base += 8) {
        const int o = base + (lane >> 3);
        const float up = __builtin_ldexpf(1.0f, o), weight = __builtin_ldexpf(1.0f, -o); // p * 2^o, 0.5^o: exact, as repeated doubling is
        const V3<float> q = mk<float>(p.x * up, p.y * up, p.z * up);
        const float fx = m_floor(q.x), fy = m_floor(q.y), fz = m_floor(q.z);
        const float u = q.x - fx, v = q.y - fy, w = q.z - fz;
        const int i = int(fx), j = int(fy), k = int(fz);
        const float uu = u * u * (3.0f - 2.0f * u), vv = v * v * (3.0f - 2.0f * v), ww = w * w * (3.0f - 2.0f * w);
        const int g = pn.perm_x[(i + a) & 255] ^ pn.perm_y[(j + b) & 255] ^ pn.perm_z[(k + d) & 255];
        const V3<float> c = ld3(pn.randvec[g]);
        const V3<float> wv = mk<float>(u - a, v - b, w - d);
        const float term = (a ? uu : 1.0f - uu) * (b ? vv : 1.0f - vv) * (d ? ww : 1.0f - ww) * dot(c, wv);
        if (o < depth) total += weight * term;
    }
    // sum over the wave: four DPP steps leave every lane of a row of 16 with its row's sum, four readlanes add the rows
    auto dpp_add = [&](auto ctl) {
        total += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, total), decltype(ctl)::value, 0xf, 0xf, false));
    };
    dpp_add(std::integral_constant<int, 0xb1>{});  // quad_perm [1,0,3,2]
    dpp_add(std::integral_constant<int, 0x4e>{});  // quad_perm [2,3,0,1]
    dpp_add(std::integral_constant<int, 0x124>{}); // row_ror:4
    dpp_add(std::integral_constant<int, 0x128>{}); // row_ror:8
    const int ti = __builtin_bit_cast(int, total);
    const float sum = (__builtin_bit_cast(float, __builtin_amdgcn_readlane(ti, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(ti, 16))) +
                      (__builtin_bit_cast(float, __builtin_amdgcn_readlane(ti, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(ti, 48)));
    return m_abs(sum);
}
// texture.h:85 noise_texture::value given turb(p, 7)
template <class R> RT_DEV V3<R> noise_color(const TexT<R> &t, V3<R> p, R turb) {
    R s = R(1) + sin_wide(t.scale * p.z + R(10) * turb);
    return mk<R>(R(.5) * s, R(.5) * s, R(.5) * s);
}

template <class R, class SC> RT_DEV V3<R> tex_value(const SC &sc, int tex, R u, R v, V3<R> p) {
    TexT<R> t = sc.tex(tex);
    for (int guard = 0; guard < 16 && t.kind == RT_TEX_CHECKER; guard++) { // texture.h:39
        int xi = int(m_floor(t.scale * p.x));
        int yi = int(m_floor(t.scale * p.y));
        int zi = int(m_floor(t.scale * p.z));
        bool even = (xi + yi + zi) % 2 == 0;
        t = sc.tex(even ? t.a : t.b);
    }
    if (t.kind == RT_TEX_SOLID) return ld3(t.color); // texture.h:22
    // (-DRT_EXP_NO_IMAGE / -DRT_EXP_NO_NOISE: measurement builds that stub a texture kind out -- what does it cost?)
#ifndef RT_EXP_NO_IMAGE
    if (t.kind == RT_TEX_IMAGE) {                    // texture.h:60
        if (t.a < 0) return mk<R>(R(0), R(1), R(1));
        const ImageT im = sc.image(t.a);
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
#endif
#ifdef RT_EXP_NO_NOISE
    if (t.kind == RT_TEX_NOISE) return mk<R>(R(.5), R(.5), R(.5));
#endif
#ifdef RT_EXP_NO_IMAGE
    if (t.kind == RT_TEX_IMAGE) return mk<R>(R(.5), R(.5), R(.5));
#endif
    if (t.kind == RT_TEX_NOISE) { // texture.h:85
        return noise_color(t, p, perlin_turb(sc.perlin(t.a), p, 7));
    }
    return mk<R>(R(0), R(0), R(0));
}

// ------------------------------------------------------------------------------------------------ sampling
template <class R> struct Onb { // onb.h:5
    V3<R> u, v, w;
    // unit_n: `n` is a hit record's normal -- unit length by construction (sphere: (p - c) / r; stored quad / triangle /
    // box-face normals; rotations keep lengths) -- so the float path skips onb.h:8's re-normalisation (7 instructions)
    RT_DEV explicit Onb(V3<R> n, bool unit_n = false) {
        w = (sizeof(R) == 4 && unit_n) ? n : unit(n);
        V3<R> a = (m_abs(w.x) > R(0.9)) ? mk<R>(R(0), R(1), R(0)) : mk<R>(R(1), R(0), R(0));
        v = unit(cross(w, a));
        u = cross(w, v);
    }
    RT_DEV V3<R> transform(V3<R> q) const {
        if constexpr (sizeof(R) == 4) return madd<R>(q.z, w, madd<R>(q.y, v, scale<R>(q.x, u)));
        else return scale<R>(q.x, u) + scale<R>(q.y, v) + scale<R>(q.z, w);
    }
};

// Draw orders below follow oracle/rt_oracle.cpp (which is pinned to the compiled reference): vec3::random(-1,1)
// consumes z, y, x; random_in_unit_disk y, x; quad::random v then u.
// vec3.h:114 random_unit_vector, vec3.h:106 random_in_unit_disk: rejection loops in the reference; under the counter-based
// generator (protocol v4, oracle/rt_oracle.cpp Ctx) ONE word of the stream seeds a xorshift32 sub-sequence of two uniforms
// and the distribution is sampled directly -- the same distribution, without a loop that all 64 lanes of a wave run
// until the unluckiest one accepts (protocol v3 ran the loops on the sub-sequence: 75-150 instructions per shading trip).
template <class R> RT_DEV V3<R> random_unit_vector(Rng<R> &g) {
    uint32_t s = g.sub_begin();
    const R u1 = Rng<R>::sub_next(s, R(0), R(1));
    const R u2 = Rng<R>::sub_next(s, R(0), R(1));
    const R z = R(1) - R(2) * u1;
    const R r = m_sqrt(m_max(R(0), R(1) - z * z));
    R sp, cp;
    sincos_rev(u2, sp, cp);
    return mk<R>(r * cp, r * sp, z);
}
template <class R> RT_DEV void random_in_unit_disk(Rng<R> &g, R &x, R &y) {
    uint32_t s = g.sub_begin();
    const R u1 = Rng<R>::sub_next(s, R(0), R(1));
    const R u2 = Rng<R>::sub_next(s, R(0), R(1));
    const R r = m_sqrt(u1);
    R sp, cp;
    sincos_rev(u2, sp, cp);
    x = r * cp;
    y = r * sp;
}

template <class R> RT_DEV V3<R> random_cosine_direction(Rng<R> &g) { // vec3.h:142
    R r1 = g.next();
    R r2 = g.next();
    R sr = m_sqrt(r2);
    R sp, cp;
    sincos_rev(r1, sp, cp); // phi = 2*pi*r1
    return mk<R>(cp * sr, sp * sr, m_sqrt(R(1) - r2));
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
        R x = m_div(L.s.radius * L.s.radius, dist_squared);
        if constexpr (sizeof(R) == 4) {
            // an origin ON the sphere (the floor point a sphere rests on) lands a few ulp inside it in float: x = 1 + eps
            // would give sqrt(<0).  Genuinely interior origins (x well above 1) keep the reference's NaN (sphere.h:70).
            if (x > R(1) && x < R(1.0001)) x = R(1);
        }
        R cos_theta_max = m_sqrt(R(1) - x);
        R one_minus_cos = R(1) - cos_theta_max;
        if constexpr (sizeof(R) == 4) one_minus_cos = m_div(x, R(1) + cos_theta_max); // same value, no cancellation in float
        R solid_angle = R(2) * Const<R>::pi * one_minus_cos;
        return m_rcp(solid_angle);
    }
    if (L.kind == LIGHT_QUAD) { // quad.h:75
        R t, al, be;
        if (skip_test) {
            V3<R> n = ld3(L.q.n);
            R denom = dot(n, dir);
            if (m_abs(denom) < R(1e-8)) return R(0);
            t = m_div(L.q.D - dot(n, origin), denom);
            if (!(t >= R(0.001))) return R(0);
        } else {
            if (!quad_t(L.q, r, R(0.001), Const<R>::inf, t, al, be)) return R(0);
        }
        R distance_squared = t * t * len2(dir);
        R cosine = m_abs(m_div(dot(dir, ld3(L.q.n)), len(dir))); // |.| makes the face flip of rec.normal irrelevant
        return m_div(distance_squared, cosine * L.q.area);
    }
    return R(0); // hittable.h:35
}

// `picked` = index of the light that generated `dir`, or -1 when the direction came from the BRDF sampler.
template <class R, class SC> RT_DEV R light_pdf(const SC &sc, V3<R> origin, V3<R> dir, int picked) {
    if (!sc.lights_is_list) return light_pdf_one(sc.light(0), origin, dir, picked == 0);
    R weight = m_rcp(R(sc.n_lights)); // hittable_list.h:42
    R sum = R(0);
    for (int k = 0; k < sc.n_lights; k++) sum += weight * light_pdf_one(sc.light(k), origin, dir, picked == k);
    return sum;
}

template <class R, class SC> RT_DEV V3<R> light_random(const SC &sc, V3<R> origin, Rng<R> &g, int &pick) {
    pick = 0;
    if (sc.lights_is_list) { // hittable_list.h:52: random_int(0, n-1) = int(0 + n*xi)
        pick = int(R(0) + (R(sc.n_lights) - R(0)) * g.next());
        if (pick >= sc.n_lights) pick = sc.n_lights - 1; // f32 rounding guard (xi < 1 always in f64)
    }
    const LightT<R> &L = sc.light(pick);
    if (L.kind == LIGHT_SPHERE) { // sphere.h:76, 98
        V3<R> direction = ld3(L.s.c) - origin;
        R distance_squared = len2(direction);
        Onb<R> uvw(direction);
        R r1 = g.next();
        R r2 = g.next();
        R x = m_div(L.s.radius * L.s.radius, distance_squared);
        if constexpr (sizeof(R) == 4) {
            if (x > R(1) && x < R(1.0001)) x = R(1); // see light_pdf_one
        }
        R cm = m_sqrt(R(1) - x);
        R z = R(1) + r2 * (cm - R(1));
        R sp, cp;
        sincos_rev(r1, sp, cp); // phi = 2*pi*r1
        R s = m_sqrt(R(1) - z * z);
        if constexpr (sizeof(R) == 4) { // 1-z = r2*x/(1+cm) exactly; sqrt(1-z^2) = sqrt((1-z)(1+z)) keeps the small cone angle
            R omz = m_div(r2 * x, R(1) + cm);
            z = R(1) - omz;
            s = m_sqrt(omz * (R(1) + z));
        }
        return uvw.transform(mk<R>(cp * s, sp * s, z));
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
        random_in_unit_disk(g, dx, dy);
        origin = ld3(cam.center) + scale<R>(dx, ld3(cam.ddu)) + scale<R>(dy, ld3(cam.ddv));
    }
    RayT<R> r;
    r.o = origin;
    r.d = pixel_sample - origin;
    r.tm = g.next();
    return r;
}

#ifdef RT_PROFILE_ACTIONS
// leaf-trip sub-blocks: cycles, executions and lanes of the AABOX / MEDIUM / other code (actions 2..4 of the profile)
#define RT_SUB_BEGIN()                                                                                                 \
    unsigned long long t_sub = 0;                                                                                      \
    if (leaf_trip) {                                                                                                   \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                    \
        t_sub = __builtin_amdgcn_s_memtime();                                                                          \
    }
#define RT_SUB_END(k, mask)                                                                                            \
    if (leaf_trip) {                                                                                                   \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                    \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                                    \
        if ((mask) != 0ull) { sub_cyc[k] += t_ - t_sub; sub_trips[k] += 1; sub_lanes[k] += popc64(mask); }             \
        t_sub = t_;                                                                                                    \
    }
#else
#define RT_SUB_BEGIN()
#define RT_SUB_END(k, mask)
#endif

// The camera's vectors are needed by the few lanes per shading trip that start a camera ray (and the background by the
// lanes whose ray missed), yet as ordinary kernel arguments their 23 words would be live in SGPRs across the whole wave
// loop -- the kernel is over its SGPR budget, so values spilled to VGPR lanes come back through v_readlane + s_nop pairs
// on the vector unit.  The float kernels therefore re-read them from the kernel-argument segment AT USE: two scalar
// loads (scalar cache hits) into registers that live for one block.  KA_CAM = byte offset of cam_arg in that segment.
#ifndef RT_CAM_AT_USE
#define RT_CAM_AT_USE 1
#endif
typedef float rt_f16 __attribute__((ext_vector_type(16)));
template <int OFF> RT_DEV void kernarg_load_camera(CameraT<float> &c) {
    static_assert(offsetof(CameraT<float>, recip_sqrt_spp) == 16 && offsetof(CameraT<float>, ddv) == 84 && sizeof(CameraT<float>) == 108, "camera layout");
    rt_f16 a;
    rt_f4 b;
    asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx4 %1, %2, %4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b)
                 : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "i"(OFF + 16), "i"(OFF + 80)
                 : "memory");
    c.recip_sqrt_spp = a[0]; c.defocus_angle = a[1];
    c.center[0] = a[2]; c.center[1] = a[3]; c.center[2] = a[4];
    c.pixel00[0] = a[5]; c.pixel00[1] = a[6]; c.pixel00[2] = a[7];
    c.du[0] = a[8]; c.du[1] = a[9]; c.du[2] = a[10];
    c.dv[0] = a[11]; c.dv[1] = a[12]; c.dv[2] = a[13];
    c.ddu[0] = a[14]; c.ddu[1] = a[15]; c.ddu[2] = b[0];
    c.ddv[0] = b[1]; c.ddv[1] = b[2]; c.ddv[2] = b[3];
}
template <int OFF> RT_DEV V3<float> kernarg_load_background() {
    static_assert(offsetof(CameraT<float>, background) == 96, "camera layout");
    rt_f4 b;
    asm volatile("s_load_dwordx4 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(b) : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "i"(OFF + 92) : "memory");
    return mk<float>(b[1], b[2], b[3]);
}

#ifdef RT_PROFILE_SHADE
// stages of the shading trip: cycles and lanes of hit records / dealing / noise / generator / camera rays / scatter / trav_begin
#define RT_SH_BEGIN()                                                                                                  \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                        \
    unsigned long long t_sh = __builtin_amdgcn_s_memtime();
#define RT_SH_END(k, pred)                                                                                             \
    {                                                                                                                  \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                    \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                                    \
        sh_cyc[k] += t_ - t_sh; sh_lanes[k] += popc64(lane_mask(pred)); sh_trips += (k) == 0;                          \
        t_sh = t_;                                                                                                     \
    }
#else
#define RT_SH_BEGIN()
#define RT_SH_END(k, pred)
#endif

// The part of camera::ray_color (camera.h:291-326) that follows world.hit(), written iteratively: radiance
// L += beta * (emitted | background), beta *= attenuation * scattering_pdf / pdf -- in two halves, so that the render
// kernel can deal new units to the lanes whose paths end BEFORE it runs the generator: one generator block then serves
// the scattering lanes and the new camera rays alike.
// shade_hit(): the hit record; returns false when the path ends here (no draw is needed for that: a miss, or a light --
// every hit that scatters emits nothing (material.h:28-31), so the recursion's sum of emissions has exactly one non-zero
// term, beta * (emitted | background), which is L).
template <class R, bool ALL_LDS, class SC>
RT_DEV bool shade_hit(const Mem<R, ALL_LDS> &mem, const SC &sc, const V3<R> &background, const RayT<R> &ray, R best_t, int best_off,
                      const V3<R> &beta, V3<R> &L, Hit<R> &h, int &mat_noise) {
    mat_noise = -1;
    if (best_off < 0) { // camera.h:300
        L = beta * background;
        return false;
    }
    h = finalize(mem, sc, ray, best_t, best_off, false);
    const MatT<R> m = sc.mat(h.mat);
    mat_noise = m.noise; // (for the caller's noise-texture step: one word of the record that is read here anyway)
    if (m.kind == RT_MAT_DIFFUSE_LIGHT) { // material.h:129
        V3<R> emitted = mk<R>(R(0), R(0), R(0));
        if (h.front) emitted = tex_value(sc, m.tex, h.u, h.v, h.p);
        L = beta * emitted;
        return false; // material.h:28: no scatter
    }
    return true;
}

// shade_scatter(): material::scatter and the mixture pdf for a hit that shade_hit() let through; the next ray and beta.
// has_turb: the material's texture is a noise_texture and `turb` its perlin turbulence at h.p (perlin_turb_wave()).
template <class R, class SC>
RT_DEV void shade_scatter(const SC &sc, RayT<R> &ray, const Hit<R> &h, V3<R> &beta, Rng<R> &g, bool has_turb, R turb) {
    const MatT<R> m = sc.mat(h.mat);
    if (m.kind == RT_MAT_METAL) { // material.h:66
        V3<R> reflected = reflect(ray.d, h.n);
        V3<R> ruv = random_unit_vector(g);
        reflected = unit(reflected) + scale<R>(m.fuzz, ruv);
        beta = beta * ld3(m.albedo);
        ray.o = h.p;
        ray.d = reflected;
        return;
    }
    if (m.kind == RT_MAT_DIELECTRIC) { // material.h:88
        R ri = h.front ? m_rcp(m.ior) : m.ior;
        V3<R> ud = unit(ray.d);
        R cos_theta = m_min(dot(-ud, h.n), R(1.0));
        R sin_theta = m_sqrt(R(1.0) - cos_theta * cos_theta);
        bool cannot_refract = ri * sin_theta > R(1.0);
        bool do_reflect = cannot_refract;
        if (!do_reflect) {
            R r0 = m_div(R(1) - ri, R(1) + ri);
            r0 = r0 * r0;
            R refl = r0 + (R(1) - r0) * m_pow5(R(1) - cos_theta);
            do_reflect = refl > g.next();
        }
        V3<R> dir = do_reflect ? reflect(ud, h.n) : refract(ud, h.n, ri);
        // attenuation (1,1,1): beta unchanged up to a multiplication by one
        ray.o = h.p;
        ray.d = dir;
        return;
    }
    // lambertian / isotropic: mixture of light sampling and BRDF sampling (camera.h:313-325, pdf.h:69-85)
    const bool lambert = (m.kind == RT_MAT_LAMBERTIAN);
    V3<R> atten;
    if (has_turb) atten = noise_color(sc.tex(m.tex), h.p, turb);
    else atten = tex_value(sc, m.tex, h.u, h.v, h.p);
    V3<R> dir;
    int picked = -1;
    if (g.next() < R(0.5)) dir = light_random(sc, h.p, g, picked);
    else if (lambert) dir = Onb<R>(h.n, true).transform(random_cosine_direction(g)); // pdf.h:40
    else dir = random_unit_vector(g);                                          // pdf.h:25
    R v0 = light_pdf(sc, h.p, dir, picked);
    R v1, spdf;
    if (lambert) {
        if constexpr (sizeof(R) == 4) { // |h.n| = 1: the two cosines below are one number
            const R c = dot(h.n, unit(dir)) * R(1.0 / 3.14159265358979323846);
            v1 = m_max(R(0), c);
            spdf = v1;
        } else {
            R cosine_theta = dot(unit(dir), unit(h.n)); // cosine_pdf::value pdf.h:35 (onb re-normalises w)
            v1 = m_max(R(0), m_div(cosine_theta, Const<R>::pi));
            R cos_theta = dot(h.n, unit(dir)); // lambertian::scattering_pdf material.h:51
            spdf = cos_theta < R(0) ? R(0) : m_div(cos_theta, Const<R>::pi);
        }
    } else {
        v1 = R(1) / (R(4) * Const<R>::pi);   // pdf.h:21
        spdf = R(1) / (R(4) * Const<R>::pi); // material.h:153
    }
    R pdf_value = R(0.5) * v0 + R(0.5) * v1;
    R wgt = m_div(spdf, pdf_value);
    beta = beta * scale<R>(wgt, atten);
    ray.o = h.p;
    ray.d = dir;
}

// LDS accumulation of a finished path (wave scope: only this wave touches its accumulators).
//   double kernels: ds_add_f64 on double cells (rounding ~1e-16: the float result practically never depends on the order).
//   float kernels : 64-bit FIXED-POINT cells, ds_add_u64.  Integer adds commute, so a pixel's sum does not depend on
//   which wave picked the item up or on the order in which the wave's lanes finish their paths: frames repeat bit for
//   bit, and a frame equals the sum of its stratum ranges up to the final conversion to float.  Unit 2^-24 (6e-8); a
//   sample is clamped to [0, 2^20] first -- a sample above sqrt_spp^2 saturates its pixel to 255 by itself whatever else
//   is added (write_color clamps at 0.999), so the 8-bit image is the reference's for up to 2^20 samples per pixel --
//   and a NaN sample (which makes the reference's pixel NaN, i.e. black: color.h:26-28) sets the cell's NaN flag.
enum : int { RT_FX_SHIFT = 24 };
#define RT_FX_CLAMP 1048576.0f
struct I64Out { // what a float kernel stores per cell when several sample chunks are summed afterwards
    static constexpr long long nan_cell = 0x7fffffffffffffffll;
    static constexpr unsigned long long nan_bit = 1ull << 63; // the shared frame (out_i64 == 2): a NaN sample sets the cell's top bit
};
template <class R> struct CellOf { using T = double; };
template <> struct CellOf<float> { using T = unsigned long long; };
RT_DEV void lds_add(double *p, double v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
RT_DEV void lds_add(unsigned long long *p, unsigned long long v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
RT_DEV unsigned long long to_fixed(float v) { // v in [0, 2^20], exact: a float has 24 significant bits
    const float f = v * 16777216.0f;                                   // v * 2^24 < 2^45
    const unsigned hi = (unsigned)(f * 2.3283064365386963e-10f);       // floor(f / 2^32)
    const unsigned lo = (unsigned)(fmaf(-(float)hi, 4294967296.0f, f) + 0.5f); // the remainder, rounded to nearest
    return ((unsigned long long)hi << 32) + lo;
}

// Wave-uniform helpers that stay on the scalar unit: the lane mask of a predicate without the int round trip of
// __ballot(), and its population count as two 32-bit s_bcnt1 (a 64-bit count is compared on the VALU).
RT_DEV unsigned long long lane_mask(bool p) { return __builtin_amdgcn_ballot_w64(p); }
RT_DEV int popc64(unsigned long long m) { return __builtin_popcount((unsigned)m) + __builtin_popcount((unsigned)(m >> 32)); }
RT_DEV unsigned smax_u32(unsigned a, unsigned b) { // the compiler's own choice for a uniform max chain is v_max3_u32 + readfirstlane
    unsigned r;
    asm("s_max_u32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc");
    return r;
}

enum : int { RT_MAX_SCHED_CHUNKS = 24 };
// Scheduling constants of the render kernel's wave machine (tuned on final_scene 1080p, tools/sweep.py history in
// DESIGN.md; compile-time so that they cost no SGPRs): weights of the waiting-lane counts when a block is chosen,
// lanes that must wait before the shading block runs, BOX steps per trip while enough lanes stay at BOX records.
#ifndef RT_W_BOX
#define RT_W_BOX 2
#endif
#ifndef RT_W_LEAF
#define RT_W_LEAF 3
#endif
#ifndef RT_SHADE_T
#define RT_SHADE_T 40
#endif
#ifndef RT_BOX_REPEAT
#define RT_BOX_REPEAT 24
#endif
#ifndef RT_WAVE_NOISE
#define RT_WAVE_NOISE 1
#endif
// rounds of the leaf trip (the leaves of a node stand next to each other in the stream), lanes that must stand at a leaf
// again for another round.  1 round: 600 ms; 2 rounds / 8 lanes: 592; 2 / 16: 598; 2 / 24: 604; 3 / 8: 589; 4 / 12: 593.
#ifndef RT_LEAF_ROUNDS
#define RT_LEAF_ROUNDS 2
#endif
#ifndef RT_LEAF_AGAIN_MIN
#define RT_LEAF_AGAIN_MIN 8
#endif
#ifndef RT_BOX_REPEAT_MIN
#define RT_BOX_REPEAT_MIN 8
#endif
struct RenderArgs {
    uint32_t k0, k1;
    int32_t s_begin, s_end;
    int32_t chunk_size;      // > 0: equal chunks of this many samples; 0: the boundaries in chunk_start[]
    int32_t tiles_x, tiles_y;
    int32_t n_chunks;
    int32_t out_i64;         // float kernels: 0 `out` receives floats; 1 64-bit fixed-point cells, one partial frame per chunk (summed
                             // afterwards); 2 ONE zeroed frame of such cells that every chunk's items add to with 64-bit atomics
    int32_t shade_t;         // lanes that must wait before the shading block runs (RT_SHADE_T unless the launcher knows better)
    int32_t chunk_start[RT_MAX_SCHED_CHUNKS + 1];        // decreasing-size schedule (chunk c = [chunk_start[c], chunk_start[c+1]))
};
enum : int { A_BOX = 0, A_LEAF = 1, A_SHADE = 5, A_COUNT = 6, A_DEAL = 7 };

// Persistent megakernel.  A workgroup stages the scene stream into LDS once, then each of its waves pulls work items
// (8 x RT_TILE_H pixel tile x chunk of the stratified sample range) from a global cursor and deals the item's (pixel,
// sample) units to whichever lanes are free.  The wave is a state machine: at any time a lane stands at a BOX record,
// at a leaf record (sphere / AABOX / medium / one of the rarer kinds), waits to be shaded, or is idle; each trip of the
// loop runs ONE of three blocks -- BOX steps, one step of every leaf kind that has lanes, or shading + dealing --
// chosen wave-uniformly, so the hot BOX step is never serialised behind leaf or shading code that only a few lanes
// need (64-lane divergence is the limiter of this workload: see DESIGN.md, profiles/).
// per-wave item state (LdsLayout::state_ints ints behind the wave's accumulators):
// slot k: ws[4k .. 4k+3] = tile origin i0, j0, chunk, valid; then the dealing item's slot, first sample and unit count
enum : int { WS_SLOT_I0 = 0, WS_SLOT_J0 = 1, WS_SLOT_CHUNK = 2, WS_SLOT_VALID = 3, WS_CUR_SLOT = 4 * RT_ITEM_SLOTS, WS_CUR_S0 = 4 * RT_ITEM_SLOTS + 1,
              WS_N_UNITS = 4 * RT_ITEM_SLOTS + 2, WS_NAN = 4 * RT_ITEM_SLOTS + 16 /* nan_words ints per slot */ };

template <class R, bool ALL_LDS, bool TAB_LDS, bool GEN_MEDIUM, int BLOCK>
__global__ void __launch_bounds__(BLOCK) render_kernel(SceneT<R> sc_arg, CameraT<R> cam_arg, RenderArgs a_arg, void *__restrict__ out_,
                                                       unsigned long long *__restrict__ counters) {
    using Cell = typename CellOf<R>::T;
    float *const out = reinterpret_cast<float *>(out_);
    long long *const out64 = reinterpret_cast<long long *>(out_);
    static_assert(BLOCK == LdsLayout<R>::block, "LDS layout");
    lds_origin_check();
    // Kernel arguments arrive as 8- and 16-dword scalar loads; left alone, the register allocator spills and reloads such a
    // tuple WHOLE (16 v_readlane for one pointer).  Passing every field through an empty asm makes each its own value.
    SceneV<R, TAB_LDS> sc = stage_scene<R, TAB_LDS>(sc_arg);
    CameraT<R> cam = cam_arg;
    const RenderArgs &a = a_arg;
#define RT_OWN(x) asm volatile("" : "+s"(x))
    RT_OWN(sc.cold); RT_OWN(sc.image_bytes); RT_OWN(sc.tab); RT_OWN(sc.o_xforms); RT_OWN(sc.o_mats); RT_OWN(sc.o_texs);
    RT_OWN(sc.o_images); RT_OWN(sc.o_perlins); RT_OWN(sc.o_lights); RT_OWN(sc.n_lights); RT_OWN(sc.lights_is_list);
    RT_OWN(cam.W); RT_OWN(cam.H); RT_OWN(cam.sqrt_spp); RT_OWN(cam.max_depth);
    constexpr bool cam_at_use = sizeof(R) == 4 && RT_CAM_AT_USE; // (see kernarg_load_camera)
    constexpr int KA_CAM = (int)((sizeof(SceneT<R>) + alignof(CameraT<R>) - 1) / alignof(CameraT<R>) * alignof(CameraT<R>));
    if constexpr (!cam_at_use) {
        RT_OWN(cam.recip_sqrt_spp); RT_OWN(cam.defocus_angle);
        for (int k = 0; k < 3; k++) {
            RT_OWN(cam.center[k]); RT_OWN(cam.pixel00[k]); RT_OWN(cam.du[k]); RT_OWN(cam.dv[k]); RT_OWN(cam.ddu[k]); RT_OWN(cam.ddv[k]);
            RT_OWN(cam.background[k]);
        }
    }
#undef RT_OWN
    const Mem<R, ALL_LDS> mem{sc_arg.stream, sc_arg.lds_units * Pos<R>::unit};
    const int lane = threadIdx.x & 63;
    const R tmin = R(0.001);
    // this wave's accumulators (item_slots slots = items in flight, x tile_px pixels x RGB) and item state
    Cell *const acc = reinterpret_cast<Cell *>(rt_smem + LdsLayout<R>::wave_off + (threadIdx.x >> 6) * LdsLayout<R>::wave_stride);
    int *const ws = reinterpret_cast<int *>(acc + LdsLayout<R>::wave_acc);
    unsigned rays = 0;
#ifdef RT_PROFILE_TAIL
    const unsigned long long t_tail0 = __builtin_amdgcn_s_memrealtime(); // device-wide constant-rate clock
    unsigned long long t_dry = 0, dry_trips = 0, dry_lane_trips = 0; // when this wave found the item cursor exhausted; its trips from then on
#endif
#ifdef RT_PROFILE_ACTIONS
    unsigned long long prof_cyc[A_COUNT] = {}, prof_trips[A_COUNT] = {}, prof_lanes[A_COUNT] = {};
    const unsigned long long t_wave0 = __builtin_amdgcn_s_memtime();
    unsigned long long prof_idle = 0, prof_ready = 0, prof_all = 0, prof_tail = 0, prof_box_steps = 0, prof_box_lane_steps = 0;
    unsigned long long sub_cyc[A_COUNT] = {}, sub_trips[A_COUNT] = {}, sub_lanes[A_COUNT] = {};
#endif
#ifdef RT_PROFILE_SHADE
    unsigned long long sh_cyc[8] = {}, sh_lanes[8] = {}, sh_trips = 0;
#endif
    // Work: an item = (8 x RT_TILE_H pixel tile, chunk of the sample range) = TPX x samples UNITS, unit u = sample s0 + u / TPX
    // of the tile's pixel u % TPX.  Lanes do not own pixels: a lane whose path has ended takes the next unit (wave-level dealing:
    // rank among the asking lanes + the wave's cursor), so a pixel with short paths never leaves its lane idle while a
    // neighbour grinds through long ones; when the item has no units left the wave fetches the next item and keeps
    // dealing while the old items' last paths finish (NS items in flight, one accumulator slot each).  The RNG is keyed
    // by (pixel, sample), so the paths do not depend on who traces what; a finished path is added to its item's
    // per-pixel accumulator in LDS (float kernels: 64-bit fixed point, so the sums do not depend on the order of the adds).
    constexpr int NS = LdsLayout<R>::item_slots;
    constexpr int TPX = LdsLayout<R>::tile_px, TSH = LdsLayout<R>::tile_shift; // pixels per item, log2 of it
    for (int k = lane; k < LdsLayout<R>::wave_acc; k += 64) acc[k] = Cell(0);
    for (int k = lane; k < LdsLayout<R>::state_ints; k += 64) ws[k] = 0;
    RayT<R> ray;
    V3<R> beta;
    Trav<R> tv;
    tv.i = 0;
    tv.hdr = K_IDLE;
    R med_xi = R(0);    // the uniform of the scene's primary constant_medium on this ray (see Rng)
    int bounce = 0;     // index of the ray this lane traces within its path (0 = camera ray)
    int smp = 0;        // the unit this lane works on: sample index,
    int pp = 0;         //   pixel within the tile | accumulator slot << 6,
    uint32_t pixel = 0; //   pixel within the image (row-major)
    auto wsl = [&](int k) { return __builtin_amdgcn_readfirstlane(ws[k]); };
    int units_left = 0; // of the current item; -1: the item cursor is exhausted (wave-uniform)
#ifndef RT_DEAL_SGPR
#define RT_DEAL_SGPR 1
#endif
#if RT_DEAL_SGPR
    // the item being dealt: slot, first sample, unit count, tile origin -- five wave-uniform values in scalar registers (they
    // lived in LDS while the camera's 23 words crowded the scalar file: five ds_read + v_readfirstlane per shading trip)
    int d_slot = 0, d_s0 = 0, d_units = 0, d_i0 = 0, d_j0 = 0;
#endif

    // every path of an item has ended: lane p stores pixel p's sums and clears the accumulator for its next item
    auto flush = [&](int slot, int i0, int j0, int chunk) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int cell0 = (lane & (TPX - 1)) * 3;
        Cell *const b = acc + slot * LdsLayout<R>::slot_acc + cell0;
        const int i = i0 + (lane & 7), j = j0 + ((lane & (TPX - 1)) >> 3);
        if (lane < TPX && i < cam.W && j < cam.H) {
            const size_t o = ((size_t)chunk * cam.H * cam.W + (size_t)j * cam.W + i) * 3;
            if constexpr (sizeof(R) == 4) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const int bit = cell0 + c;
                    const bool nan = (ws[WS_NAN + slot * LdsLayout<R>::nan_words + (bit >> 5)] >> (bit & 31)) & 1;
                    if (a.out_i64 == 2) { // ONE 64-bit frame for all sample chunks: integer adds commute, so the frame is the same
                        unsigned long long *cell = reinterpret_cast<unsigned long long *>(out64) + (o - (size_t)chunk * cam.H * cam.W * 3 + c);
                        if (nan) (void)__hip_atomic_fetch_or(cell, I64Out::nan_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else if (b[c] != 0ull) (void)__hip_atomic_fetch_add(cell, (unsigned long long)b[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else if (a.out_i64) out64[o + c] = nan ? I64Out::nan_cell : (long long)b[c];
                    else out[o + c] = nan ? __builtin_nanf("") : (float)((double)b[c] * (1.0 / 16777216.0));
                }
            } else {
                out[o] = (float)b[0]; out[o + 1] = (float)b[1]; out[o + 2] = (float)b[2];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // the NaN words are read above by all lanes, cleared below by some
        if (lane < TPX) { b[0] = Cell(0); b[1] = Cell(0); b[2] = Cell(0); }
        if constexpr (sizeof(R) == 4) {
            if (lane < LdsLayout<R>::nan_words) ws[WS_NAN + slot * LdsLayout<R>::nan_words + lane] = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    };

    for (;;) {
        {
            // lane state = low bits of tv.hdr: the kind of the record at the cursor, K_READY (traversal finished, waiting to
            // be shaded) or K_IDLE (no unit)
            const int kind = tv.hdr & 15;
            // (a moving sphere is a sphere whose centre takes one more fetch and three fmas: it shares the sphere code)
            const bool is_box = kind == K_BOX, is_sph = kind == K_SPHERE || kind == K_MSPHERE, is_aab = kind == K_AABOX, is_med = kind == K_MEDIUM;
            const bool is_oth = kind >= K_QUAD && kind < K_IDLE && kind != K_MSPHERE; // quad, triangle, instance boundary, general medium
            const bool ready = kind == K_READY;
            const unsigned long long mb = lane_mask(is_box), ms = lane_mask(is_sph), ma = lane_mask(is_aab), mm = lane_mask(is_med),
                                     mo = lane_mask(is_oth), mr = lane_mask(ready);
            if (!(mb | ms | ma | mm | mo | mr) && units_left < 0) break; // nothing in flight, no items left: the wave is done
#ifdef RT_PROFILE_TAIL
            if (units_left < 0) { dry_trips += 1; dry_lane_trips += popc64(mb | ms | ma | mm | mo | mr); }
#endif
            {
            // Run ONE block per trip: shading once enough lanes wait for it, else the record kind with the largest weighted
            // number of waiting lanes (score = count * weight, block id in the low bits -> one max chain).
            const int nr = popc64(mr);
            int action = (mb | ms | ma | mm | mo | mr) ? A_SHADE : A_DEAL;
            // (ONE leaf trip serves every lane that stands at a non-BOX record, each kind's code under its own mask: four
            // blocks with 14 - 35 lanes each cost four trips' worth of selection and loop-join overhead; merged: 903 -> 884 ms)
            const unsigned long long ml = ms | ma | mm | mo;
            if (nr < a.shade_t) {
                const unsigned sb = ((unsigned)(popc64(mb) * RT_W_BOX) << 3) | A_BOX, sl = ((unsigned)(popc64(ml) * RT_W_LEAF) << 3) | A_LEAF;
                const unsigned best = smax_u32(sb, sl);
                if (best >= 8u) action = (int)(best & 7u);
            }
            const bool leaf_trip = action == A_LEAF;
#ifdef RT_PROFILE_ACTIONS
            const unsigned long long t_a = __builtin_amdgcn_s_memtime();
            prof_idle += 64 - popc64(mb | ms | ma | mm | mo | mr);
            if (units_left <= 0) prof_tail += 1;
            prof_ready += nr;
            prof_all += 1;
            const int n_act = action == A_BOX ? popc64(mb) : action == A_LEAF ? popc64(ml) : nr;
#endif
            // (Tried: d.d and its reciprocal kept with the ray instead of recomputed by every sphere and medium test -- two more
            // registers of loop-carried state, 711 -> 721 ms; spheres and sphere-bounded media sharing one solve of the
            // quadratic in one block -> 736 ms.  The short separate blocks below are what the scheduler handles best.)
            // the leaf trip: one step of every kind that has lanes, as ONE routine under `if (leaf_trip)` (four blocks each behind
            // its own `if (leaf_trip && mask)`: 622 -> 600 ms for the same instructions -- the code the compiler lays round
            // the loop's joins decides), run up to RT_LEAF_ROUNDS times per trip.  (A second step PER KIND for lanes that meet
            // the same kind again measured slower.)
            // (Tried in round 3: each kind's block only says (hit, t, next), the closest hit updated and the cursor moved ONCE
            // behind the four -- the copies hipcc lays at the joins of four blocks that each define the twelve registers of
            // traversal state are a sixth of the kernel's static vector instructions: 3 814 -> 3 588 static VALU, 603 -> 460
            // copies, and 514 -> 538 ms.  And the reverse: each block moving the cursor BEFORE its test, so that the test covers
            // the next record's two LDS reads: 514 -> 518 ms.)
            auto leaf_round = [&](const int kind, const bool is_sph, const bool is_aab, const bool is_med, const bool is_oth,
                                  const unsigned long long ms, const unsigned long long ma, const unsigned long long mm,
                                  const unsigned long long mo) {
                if (leaf_trip && ms != 0ull) {
                    if (is_sph) {
                        V3<R> cc = mk<R>(tv.u0.w[0], tv.u0.w[1], tv.u0.w[2]);
                        int next = tv.i + 2 * Pos<R>::unit;
                        if (kind == K_MSPHERE) { // centre = c + time * d (sphere.h:30)
                            const UnitT<R> u2 = mem.ld(next);
                            cc = cc + scale<R>(ray.tm, mk<R>(u2.w[0], u2.w[1], u2.w[2]));
                            next += Pos<R>::unit;
                        } else if constexpr (sizeof(R) == 8) {
                            cc = cc + scale<R>(ray.tm, mk<R>(R(0), R(0), R(0))); // the reference always evaluates center.at(time)
                        }
                        R t;
                        if (sphere_t(cc, tv.u0.w[3], tv.o, tv.d, tmin, tv.best_t, t)) { tv.best_t = t; tv.best_off = tv.i; }
                        trav_goto(mem, tv, next);
                    }
                }
                RT_SUB_BEGIN();
                if (leaf_trip && ma != 0ull) {
                    if (is_aab) {
                        R t;
                        if (aabox_t(tv.u0, tv.u1, tv.o, tv.inv, tmin, tv.best_t, t)) { tv.best_t = t; tv.best_off = tv.i; }
                        trav_goto(mem, tv, tv.i + 2 * Pos<R>::unit);
                    }
                }
                RT_SUB_END(2, ma);
                if (leaf_trip && mm != 0ull) {
                    if (is_med) {
                        auto mxi = [&](int node_id, int sel) {
                            R x = med_xi;
                            if (sel != 1) x = medium_draw<R>(med_xi, node_id, sel >> 1);
                            return x;
                        };
                        R t;
                        if (medium_t(mk<R>(tv.u0.w[0], tv.u0.w[1], tv.u0.w[2]), tv.u0.w[3], tv.u1.w[0], ldi(tv.u1.w[2]), hdr_payload(tv.hdr), tv.o, tv.d,
                                     tmin, tv.best_t, mxi, t)) {
                            tv.best_t = t; tv.best_off = tv.i;
                        }
                        trav_goto(mem, tv, tv.i + 2 * Pos<R>::unit);
                    }
                }
                RT_SUB_END(3, mm);
                // (letting these few lanes -- final_scene: the light's quad, 4 per trip on average -- wait until 8 / 16 of them
                // stand here: 762 -> 782 / 794 ms; a waiting lane is a lost lane)
                if (leaf_trip && mo != 0ull) {
                    if (is_oth) {
                        auto mxi = [&](int node_id, int sel) {
                            R x = med_xi;
                            if (sel != 1) x = medium_draw<R>(med_xi, node_id, sel >> 1);
                            return x;
                        };
                        trav_leaf_step<GEN_MEDIUM>(mem, sc, tv, ray, tmin, mxi);
                    }
                }
                RT_SUB_END(4, mo);
            };
            // (tried: s_setprio per block -- BOX trips at priority 3: 512 -> 529 ms; leaf trips at 3: 525; shading trips at 3: 518.6 --
            // the SIMD's own rotation between its four waves is the better arbiter)
            if (action == A_BOX) {
                // a few BOX steps in a row while enough lanes stay at BOX records (amortises the block selection)
                // (a BOX header is the integer 0: kind 0, no payload; one compare per step serves both the next step's lane
                // mask and the repeat decision)
                bool at_box = is_box;
                for (int rep = 0; rep < RT_BOX_REPEAT; rep++) {
#ifdef RT_PROFILE_ACTIONS
                    prof_box_steps += 1; prof_box_lane_steps += popc64(lane_mask(at_box));
#endif
                    if (at_box) trav_box_step(mem, tv, tmin);
                    int h = tv.hdr;
                    at_box = h == make_hdr(K_BOX, 0);
                    // (tried: going on while NO lane waits for anything else -- a wave's last long paths draining at the end of the
                    // kernel take 2 300 mostly one-step trips on a 1/8 range -- 2 300 -> 2 030 trips, the drain no shorter, frame +0.5 %)
                    if (popc64(lane_mask(at_box)) < RT_BOX_REPEAT_MIN) break;
                }
            }
            // (the three blocks as an if / else-if chain instead of three ifs in a row: 591 -> 688 ms; a BOX trip that leaves
            // >= 16 / 24 / 32 lanes at leaf records going straight on with a leaf round: 565 -> 587 / 589 / 591 ms)
            if (leaf_trip) {
                leaf_round(kind, is_sph, is_aab, is_med, is_oth, ms, ma, mm, mo);
#if RT_LEAF_ROUNDS > 1
                // the leaves of a node stand next to each other in the stream: a second round, in the same trip, for the lanes
                // that stand at a leaf again
                for (int round = 1; round < RT_LEAF_ROUNDS; round++) {
                    const int k2 = tv.hdr & 15;
                    const bool s2 = k2 == K_SPHERE || k2 == K_MSPHERE, a2 = k2 == K_AABOX, m2 = k2 == K_MEDIUM;
                    const bool o2 = k2 >= K_QUAD && k2 < K_IDLE && k2 != K_MSPHERE;
                    const unsigned long long ms2 = lane_mask(s2), ma2 = lane_mask(a2), mm2 = lane_mask(m2), mo2 = lane_mask(o2);
                    if (popc64(ms2 | ma2 | mm2 | mo2) < RT_LEAF_AGAIN_MIN) break;
                    leaf_round(k2, s2, a2, m2, o2, ms2, ma2, mm2, mo2);
                }
#endif
            }
            // The shading trip (and the trip that only deals, when nothing is in flight), in three steps:
            //   1. every lane whose traversal has finished gets its hit record; a path that ends here -- a miss, a light --
            //      needs no draw: its sample is added and the lane is free;
            //   2. free lanes are dealt their next unit;
            //   3. ONE generator block per lane that now starts a ray -- bounce b's block for the lanes that scatter, the
            //      camera block for the new units -- then material::scatter / get_ray, and one trav_begin for both.
            // (Dealing behind the scattering code, with a generator call and a trav_begin of its own: 739 -> see DESIGN.md)
            if (action == A_SHADE || action == A_DEAL) {
                // a finished path: add the sample to its pixel's accumulator; the lane asks for its next unit
                auto add_sample = [&](const V3<R> &L) {
                    Cell *const px = acc + pp * 3;
                    if constexpr (sizeof(R) == 4) {
                        const float lv[3] = {L.x, L.y, L.z};
                        if ((pp >> TSH) >= NS) {
                            // an ORPHAN: the accumulator slot of this path's item was taken over while the path was still under way
                            // (see the dealing step); its sample goes straight to the shared 64-bit frame -- integer adds commute
                            unsigned long long *cell = reinterpret_cast<unsigned long long *>(out64) + (size_t)pixel * 3;
#pragma unroll
                            for (int c = 0; c < 3; c++) {
                                if (lv[c] != lv[c]) (void)__hip_atomic_fetch_or(cell + c, I64Out::nan_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                else (void)__hip_atomic_fetch_add(cell + c, to_fixed(__builtin_amdgcn_fmed3f(lv[c], 0.0f, RT_FX_CLAMP)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                            tv.hdr = K_IDLE;
                            return;
                        }
                        // the usual sample: three finite values in [0, 256) -> v * 2^24 fits 32 bits, one fma + one
                        // conversion per channel.  (Which branch a sample takes depends on the sample alone.)
                        const float sum = (L.x + L.y) + L.z; // NaN if any channel is
                        if (sum < 256.0f && fminf(fminf(L.x, L.y), L.z) >= 0.0f) {
#pragma unroll
                            for (int c = 0; c < 3; c++) lds_add(px + c, (unsigned long long)(unsigned)fmaf(lv[c], 16777216.0f, 0.5f));
                        } else {
#pragma unroll
                            for (int c = 0; c < 3; c++) {
                                if (lv[c] != lv[c]) { // NaN: flag the cell (0/0 weights, a light sphere seen from inside)
                                    const int bit = (pp & (TPX - 1)) * 3 + c;
                                    (void)__hip_atomic_fetch_or(&ws[WS_NAN + (pp >> TSH) * LdsLayout<R>::nan_words + (bit >> 5)], 1 << (bit & 31),
                                                                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                                } else {
                                    lds_add(px + c, to_fixed(__builtin_amdgcn_fmed3f(lv[c], 0.0f, RT_FX_CLAMP)));
                                }
                            }
                        }
                    } else {
                        lds_add(px, L.x); lds_add(px + 1, L.y); lds_add(px + 2, L.z);
                    }
                    tv.hdr = K_IDLE;
                };
                // ---- 1
                RT_SH_BEGIN();
                Hit<R> h;
                h.p = mk<R>(R(0), R(0), R(0));
                h.mat = 0;
                bool scatter = false;
                int mat_noise = -1;
                if (ready) {
                    if constexpr (sizeof(R) == 4) { // traversal ended in the world frame: tv holds the world-space ray
                        ray.o = tv.o;
                        ray.d = tv.d;
                    }
                    V3<R> L, background;
                    if constexpr (cam_at_use) background = kernarg_load_background<KA_CAM>();
                    else background = ld3(cam.background);
                    scatter = shade_hit(mem, sc, background, ray, tv.best_t, tv.best_off, beta, L, h, mat_noise);
                    if (!scatter) add_sample(L);
                }
                RT_SH_END(0, ready);
                // ---- 2
                bool fresh = false; // this lane was dealt a unit in this trip: pixel (fi, fj), sample smp
                int fi = 0, fj = 0;
                const int kind0 = tv.hdr & 15;
                const unsigned long long midle = lane_mask(kind0 == K_IDLE);
                // (tried: up to two / three items per trip -- an item that runs out in the middle of a trip followed by the next one
                // in the SAME trip, so that the lanes that asked in vain do not sit out the trips up to the next shading trip, 5.6 of
                // 64 lanes on a 1/8 range: the loop round this step costs more than those lanes bring, 514 -> 532 / 528 ms, 1/8
                // range 68.7 -> 70.4 / 70.0 ms; only in the kernels that read the stream from global memory, where a quarter of the trips
                // run after an item's last unit was dealt: mesh_scene 299 -> 302-311 / 300-308 ms)
                if (midle && units_left >= 0) {
                    if (units_left == 0) {
                        // the current item has no units left: take an accumulator slot none of whose paths is still in flight (the
                        // first one in ring order: a slot that waits for one long path does not hold up the others), store the item
                        // that used it and fetch a new item into it
                        unsigned free_slots = (1u << NS) - 1u;
#pragma unroll
                        for (int k = 0; k < NS; k++)
                            if (lane_mask(kind0 != K_IDLE && (pp >> TSH) == k)) free_slots &= ~(1u << k);
#if RT_DEAL_SGPR
                        const int first = d_slot + 1; // 1 .. NS
#else
                        const int first = wsl(WS_CUR_SLOT) + 1; // 1 .. NS
#endif
                        const unsigned rot = ((free_slots >> first) | (free_slots << (NS - first))) & ((1u << NS) - 1u);
                        // No slot without a path in flight (short items -- an 8-GPU rank's share of the samples -- and a few
                        // 50-bounce paths use them up: 6 of 64 lanes sat idle per trip on a 1/8 range): when the chunks share one
                        // 64-bit frame, the OLDEST slot is taken over anyway -- its sums so far are flushed now, and the paths of
                        // its item that are still under way become orphans that add their sample to the frame themselves.
                        const bool evict = !rot && a.out_i64 == 2;
                        if (rot || evict) {
                            int t = first + (rot ? __builtin_ctz(rot) : 0);
                            t = t >= NS ? t - NS : t;
                            if (evict && kind0 != K_IDLE && (pp >> TSH) == t) pp = (pp & (TPX - 1)) | (NS << TSH);
                            const bool t_valid = wsl(4 * t + WS_SLOT_VALID) != 0;
                            if (t_valid) flush(t, wsl(4 * t + WS_SLOT_I0), wsl(4 * t + WS_SLOT_J0), wsl(4 * t + WS_SLOT_CHUNK));
                            int item = 0;
                            if (lane == 0) item = (int)atomicAdd(&counters[1], 1ull);
                            item = __builtin_amdgcn_readfirstlane(item);
                            const int n_tiles = a.tiles_x * a.tiles_y;
                            const bool got = item < n_tiles * a.n_chunks; // the cursor is finite: every wave sees its end
                            const int chunk = item / n_tiles, tile = item - chunk * n_tiles;
                            const int tj = tile / a.tiles_x, ti = tile - tj * a.tiles_x;
                            int s0, s1;
                            if (a.chunk_size > 0) {
                                s0 = a.s_begin + chunk * a.chunk_size;
                                s1 = s0 + a.chunk_size;
                                if (s1 > a.s_end) s1 = a.s_end;
                            } else {
                                s0 = a.chunk_start[got ? chunk : 0];
                                s1 = a.chunk_start[got ? chunk + 1 : 0];
                            }
                            const int n_new = got && cam.max_depth > 0 ? (s1 - s0) << TSH : 0; // max_depth <= 0: every sample is black (camera.h:294)
                            if (lane == 0) {
                                ws[4 * t + WS_SLOT_I0] = ti * 8; ws[4 * t + WS_SLOT_J0] = tj * LdsLayout<R>::tile_h; ws[4 * t + WS_SLOT_CHUNK] = chunk;
                                ws[4 * t + WS_SLOT_VALID] = got ? 1 : 0;
#if !RT_DEAL_SGPR
                                ws[WS_CUR_SLOT] = t; ws[WS_CUR_S0] = s0; ws[WS_N_UNITS] = n_new;
#endif
                            }
#if RT_DEAL_SGPR
                            d_slot = t; d_s0 = s0; d_units = n_new; d_i0 = ti * 8; d_j0 = tj * LdsLayout<R>::tile_h;
#endif
                            units_left = got ? n_new : -1;
#ifdef RT_PROFILE_TAIL
                            if (!got && t_dry == 0) t_dry = __builtin_amdgcn_s_memrealtime();
#endif
                        }
                    }
                    if (units_left > 0) {
                        // deal units to the idle lanes: rank among the asking lanes + the item's cursor
#if RT_DEAL_SGPR
                        const int cur_slot = d_slot, cur_s0 = d_s0, n_item_units = d_units, cur_i0 = d_i0, cur_j0 = d_j0;
#else
                        const int cur_slot = wsl(WS_CUR_SLOT), cur_s0 = wsl(WS_CUR_S0), n_item_units = wsl(WS_N_UNITS);
                        const int cur_i0 = wsl(4 * cur_slot + WS_SLOT_I0), cur_j0 = wsl(4 * cur_slot + WS_SLOT_J0);
#endif
                        const int slot6 = cur_slot << TSH;
                        int next_unit = n_item_units - units_left;
                        bool need = kind0 == K_IDLE;
                        unsigned long long m = midle;
                        do {
                            const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                            const int un = next_unit + rank;
                            if (need && un < n_item_units) {
                                const int p = un & (TPX - 1);
                                const int i = cur_i0 + (p & 7), j = cur_j0 + (p >> 3);
                                if (i < cam.W && j < cam.H) { // (units of pixels beyond the image edge are skipped)
                                    need = false;
                                    fresh = true;
                                    fi = i;
                                    fj = j;
                                    pp = p | slot6;
                                    pixel = (uint32_t)(j * cam.W + i);
                                    smp = cur_s0 + (un >> TSH);
                                }
                            }
                            next_unit += popc64(m);
                            m = lane_mask(need);
                        } while (m && next_unit < n_item_units);
                        units_left = next_unit < n_item_units ? n_item_units - next_unit : 0;
                    }
                }
                RT_SH_END(1, fresh);
                // ---- 3
                // (noise textures first, by the whole wave for one lane at a time: see perlin_turb_wave)
                bool has_turb = false;
                R turb = R(0);
                if constexpr (sizeof(R) == 4 && RT_WAVE_NOISE) {
                    const int noise_idx = scatter ? mat_noise : -1; // (one word of the material record shade_hit read)
                    unsigned long long mn = lane_mask(noise_idx >= 0);
                    while (mn) { // wave-uniform
                        const int src = __builtin_ctzll(mn);
                        mn &= mn - 1;
                        auto from = [&](float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src)); };
                        const V3<float> q = mk<float>(from((float)h.p.x), from((float)h.p.y), from((float)h.p.z));
                        const float tw = perlin_turb_wave(sc.perlin(__builtin_amdgcn_readlane(noise_idx, src)), q, 7, lane);
                        if (lane == src) {
                            turb = tw;
                            has_turb = true;
                        }
                    }
                }
                RT_SH_END(2, has_turb);
                if (scatter || fresh) {
                    if (fresh) bounce = 0;
                    Rng<R> g;
                    g.k0 = a.k0;
                    g.k1 = a.k1;
                    // (the key schedule is uniform and loop-invariant: hoisted, its fourteen words are spilled to VGPR lanes and
                    // come back as v_readlane + s_nop pairs on the vector unit; as values of THIS trip they are s_add on the scalar unit)
                    asm volatile("" : "+s"(g.k0), "+s"(g.k1));
                    g.begin_stream(pixel, (uint32_t)smp, fresh ? 0u : (0x80000000u | (uint32_t)bounce));
                    med_xi = g.fifth(); // for the ray that starts now
                    RT_SH_END(3, true);
                    if (fresh) {
                        CameraT<R> camr = cam;
                        if constexpr (cam_at_use) kernarg_load_camera<KA_CAM>(camr);
                        int sj = (int)((float)smp * (float)camr.recip_sqrt_spp); // s / sqrt_spp, s % sqrt_spp without an integer division
                        int si = smp - sj * cam.sqrt_spp;
                        if (si < 0) { sj--; si += cam.sqrt_spp; }
                        if (si >= cam.sqrt_spp) { sj++; si -= cam.sqrt_spp; }
                        ray = get_ray(camr, fi, fj, si, sj, g);
                        beta = mk<R>(R(1), R(1), R(1));
                        RT_SH_END(4, true);
                    } else {
                        RT_SH_END(4, false);
                        shade_scatter(sc, ray, h, beta, g, has_turb, turb);
                        bounce++;
                        if (bounce == cam.max_depth) { // depth exhausted: the recursion returns 0 here (camera.h:294-295)
                            add_sample(beta * mk<R>(R(0), R(0), R(0)));
                            scatter = false; // (the lane gets its next unit in the next shading trip)
                        }
                        RT_SH_END(5, true);
                    }
                    if (fresh || scatter) {
                        rays++;
                        trav_begin(mem, tv, ray);
                        // the stream's first record, when it is a sphere-bounded medium (final_scene's fog stands there): every
                        // ray that starts now stands at it -- one step here, by all of them (47 lanes), instead of a leaf trip's
                        // (34): final_scene 565 -> 554 ms.  (The same for a sphere in that place -- first_scene's ground --
                        // measured nothing: +-1 %; so did 1 / 2 / 4 BOX steps behind it for the rays that start: 552 / 555 / 558.
                        // A loop over up to eight leading medium / AABOX records -- a room's walls: mesh_scene 308 -> 302 ms,
                        // cornell_box unchanged, final_scene 556 -> 562.)
                        if constexpr (sizeof(R) == 4) {
                            if ((tv.hdr & 15) == K_MEDIUM) {
                                auto mxi = [&](int node_id, int sel) {
                                    R x = med_xi;
                                    if (sel != 1) x = medium_draw<R>(med_xi, node_id, sel >> 1);
                                    return x;
                                };
                                R t;
                                if (medium_t(mk<R>(tv.u0.w[0], tv.u0.w[1], tv.u0.w[2]), tv.u0.w[3], tv.u1.w[0], ldi(tv.u1.w[2]), hdr_payload(tv.hdr), tv.o, tv.d,
                                             tmin, tv.best_t, mxi, t)) {
                                    tv.best_t = t; tv.best_off = tv.i;
                                }
                                trav_goto(mem, tv, tv.i + 2 * Pos<R>::unit);
                            }
                        }
                        RT_SH_END(6, true);
                    }
                }
            }
#ifdef RT_PROFILE_ACTIONS
            { // diagnostic build only: cycles, trips and active lanes per action -> counters[4 + 3*action ...]
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_a;
#pragma unroll
                for (int k = 0; k < A_COUNT; k++)
                    if (action == k) { prof_cyc[k] += dt; prof_trips[k] += 1; prof_lanes[k] += (unsigned long long)n_act; }
            }
#endif
            }
        }
    }
    for (int k = 0; k < NS; k++)
        if (wsl(4 * k + WS_SLOT_VALID)) flush(k, wsl(4 * k + WS_SLOT_I0), wsl(4 * k + WS_SLOT_J0), wsl(4 * k + WS_SLOT_CHUNK));
    // wave-level ray count -> one atomic per wave
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_xor(rays, off);
    if (lane == 0) atomicAdd(&counters[0], (unsigned long long)rays);
#ifdef RT_PROFILE_TAIL
    if (lane == 0) { // diagnostic build: how long did this wave have work, relative to the kernel's span?
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        atomicMin(&counters[37], t_tail0);
        atomicMax(&counters[38], t1);
        atomicAdd(&counters[39], t1 - t_tail0);
        atomicMax(&counters[36], t1 - t_tail0);
        atomicMax(&counters[35], ~(t1 - t_tail0)); // maximum of the complement = minimum duration (the counter starts at zero)
        atomicAdd(&counters[58], t1 - (t_dry ? t_dry : t1)); // time this wave ran after it found the cursor exhausted
        atomicMax(&counters[59], t1 - (t_dry ? t_dry : t1));
        atomicAdd(&counters[60], dry_trips);
        atomicAdd(&counters[61], dry_lane_trips);
    }
#endif
#ifdef RT_PROFILE_SHADE
    if (lane == 0) {
        for (int k = 0; k < 7; k++) {
            atomicAdd(&counters[40 + 2 * k], sh_cyc[k]);
            atomicAdd(&counters[41 + 2 * k], sh_lanes[k]);
        }
        atomicAdd(&counters[54], sh_trips);
    }
#endif
#ifdef RT_PROFILE_ACTIONS
    if (lane == 0) {
        for (int k = 0; k < A_COUNT; k++) {
            atomicAdd(&counters[4 + 3 * k], prof_cyc[k] + sub_cyc[k]);
            atomicAdd(&counters[5 + 3 * k], prof_trips[k] + sub_trips[k]);
            atomicAdd(&counters[6 + 3 * k], prof_lanes[k] + sub_lanes[k]);
        }
        atomicAdd(&counters[3], __builtin_amdgcn_s_memtime() - t_wave0);
        atomicAdd(&counters[33], prof_idle);
        atomicAdd(&counters[34], prof_ready);
        atomicAdd(&counters[35], prof_all);
        atomicAdd(&counters[36], prof_tail);
        atomicAdd(&counters[56], prof_box_steps);
        atomicAdd(&counters[57], prof_box_lane_steps);
    }
#endif
}

struct TraceArgs {
    uint32_t k0, k1;
    int has_xi;
};

// rt_trace_batch: closest hit of each ray over [0.001, inf) (camera.h:300).
template <class R, bool ALL_LDS, bool TAB_LDS, bool GEN_MEDIUM, int BLOCK>
__global__ void __launch_bounds__(BLOCK) trace_kernel(SceneT<R> sc_arg, const rt_ray *__restrict__ rays, const double *__restrict__ xi,
                                                      size_t n, rt_hit *__restrict__ out, TraceArgs a) {
    lds_origin_check();
    const SceneV<R, TAB_LDS> sc = stage_scene<R, TAB_LDS>(sc_arg);
    const Mem<R, ALL_LDS> mem{sc_arg.stream, sc_arg.lds_units * Pos<R>::unit};
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) {
        RayT<R> r;
        r.o = mk<R>(R(rays[k].orig[0]), R(rays[k].orig[1]), R(rays[k].orig[2]));
        r.d = mk<R>(R(rays[k].dir[0]), R(rays[k].dir[1]), R(rays[k].dir[2]));
        r.tm = R(rays[k].time);
        const R xik = a.has_xi ? R(xi[k]) : R(0);
        auto mxi = [&](int node_id, int sel) {
            if (a.has_xi) return xik;
            uint32_t o0, o1, o2, o3;
            philox4x32((uint32_t)node_id, (uint32_t)k, (uint32_t)((uint64_t)k >> 32), 0x7fffffffu - (uint32_t)(sel >> 1), a.k0, a.k1, o0, o1, o2, o3);
            return u01<R>(o0);
        };
        Trav<R> tv;
        trav_begin(mem, tv, r);
        while ((tv.hdr & 15) != K_READY) {
            if ((tv.hdr & 15) == K_BOX) trav_box_step(mem, tv, R(0.001));
            else trav_leaf_step<GEN_MEDIUM>(mem, sc, tv, r, R(0.001), mxi);
        }
        rt_hit o;
        o.prim = -1;
        o.front_face = 0;
        o.t = (double)Const<double>::inf;
        o.p[0] = o.p[1] = o.p[2] = 0;
        o.normal[0] = o.normal[1] = o.normal[2] = 0;
        o.u = o.v = 0;
        if (tv.best_off >= 0) {
            Hit<R> h = finalize(mem, sc, r, tv.best_t, tv.best_off, true);
            o.prim = h.prim;
            o.front_face = h.front ? 1 : 0;
            o.t = (double)h.t;
            o.p[0] = h.p.x; o.p[1] = h.p.y; o.p[2] = h.p.z;
            o.normal[0] = h.n.x; o.normal[1] = h.n.y; o.normal[2] = h.n.z;
            o.u = (double)h.u;
            o.v = (double)h.v;
        }
        out[k] = o;
    }
}

} // namespace rtamd
#endif
