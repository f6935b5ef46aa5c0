// rt_abi.cpp -- implementation of the C ABI declared in include/rt_amd.h (host side, HIP runtime API).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_build.h"
#include "rt_flatten.h"
#include "rt_launch.h"
#include "rt_scene.h"

using namespace rtamd;

// ---- the scratch of rt_render_static (the entry point that allocates nothing) --------------------------------------------
// Device memory beyond the caller's inputs and outputs that one render needs: the kernels' counters (ray count, work-item
// cursor) and the per-chunk partial frames that reduce_chunks_kernel sums.  rt_render_device keeps both in hipMalloc'ed
// buffers of the context; rt_render_static keeps them in this file-scope array: the counters in its first
// RT_STATIC_COUNTER_BYTES, the partial frames behind them.  Every GPU the library runs on holds its own copy of the array
// at its own address until the process ends.  In host code the NAME below is only a stand-in for hipGetSymbolAddress --
// rt_init asks for the address once per context, with the context's device current -- and everything else (kernel
// arguments, hipMemsetAsync) uses that pointer, never the name.
enum : int { RT_N_COUNTERS = 64 }; // [0] rays, [1] the work-item cursor, [3..] diagnostic builds
enum : size_t { RT_STATIC_COUNTER_BYTES = 4096, RT_STATIC_SCRATCH_BYTES = (size_t)64 << 20 };
static __device__ __align__(256) unsigned char scratch[RT_STATIC_SCRATCH_BYTES];

struct rt_ctx {
    int device = 0;
    std::string err;
    void *partial = nullptr; // per-chunk partial frames [n_chunks][H][W][3]: 64-bit fixed-point cells (f32) or floats (f64)
    size_t partial_bytes = 0;
    float *accum = nullptr; // used by the host-buffer convenience entry points rt_render / rt_render_multi
    size_t accum_bytes = 0;
    float *stage = nullptr; // rt_render_multi: where the root device receives a peer's buffer
    size_t stage_bytes = 0;
    uint8_t *rgb = nullptr;
    size_t rgb_bytes = 0;
    unsigned long long *counter = nullptr;
    unsigned char *static_scratch = nullptr; // this device's copy of `scratch` (hipGetSymbolAddress in rt_init)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t xstream = nullptr; // rt_render_multi: this device's side of the exchange (created on first use)
    int n_cus = 256;
    // development knobs (environment RT_AMD_LDS_UNITS / RT_AMD_CHUNKS); -1 / 0 = library default
    int lds_limit_units = -1;
    int min_chunk = 0, chunk_divisor = 0, max_chunk = 0, chunk_pct = 0, frame_mode = 0;
};

template <class R> struct DeviceScene {
    void *blob = nullptr;
    size_t bytes = 0;
    SceneT<R> sc{};
    int n_fused_boxes = 0;
};

// A private copy of the caller's description: the caller owns (and may free) the arrays it passed to rt_scene_upload, and
// the f64 parity stream is only built when a call first asks for RT_PREC_F64.
struct DescCopy {
    rt_scene_desc d{};
    std::vector<rt_node> nodes;
    std::vector<int32_t> child_index;
    std::vector<rt_sphere> spheres;
    std::vector<rt_quad> quads;
    std::vector<rt_triangle> triangles;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<rt_image> images;
    std::vector<uint8_t> image_bytes;
    std::vector<rt_perlin> perlins;
    void assign(const rt_scene_desc &src) {
        d = src;
        auto cp = [](auto &v, const auto *p, int64_t n) {
            v.assign(p, p + (n > 0 && p ? n : 0));
            return v.empty() ? nullptr : v.data();
        };
        d.nodes = cp(nodes, src.nodes, src.n_nodes);
        d.child_index = cp(child_index, src.child_index, src.n_child_index);
        d.spheres = cp(spheres, src.spheres, src.n_spheres);
        d.quads = cp(quads, src.quads, src.n_quads);
        d.triangles = cp(triangles, src.triangles, src.n_triangles);
        d.materials = cp(materials, src.materials, src.n_materials);
        d.textures = cp(textures, src.textures, src.n_textures);
        d.images = cp(images, src.images, src.n_images);
        d.image_bytes = cp(image_bytes, src.image_bytes, src.n_image_bytes);
        d.perlins = cp(perlins, src.perlins, src.n_perlins);
    }
};

struct rt_scene {
    rt_ctx *ctx = nullptr;
    DeviceScene<float> f32;
    DeviceScene<double> f64; // built on first use (RT_PREC_F64 is the parity path, never the production one)
    bool have_f64 = false;
    DescCopy desc;
    int n_nodes_desc = 0;
};

namespace {

int set_err(rt_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg;
    return code;
}

#define HIP_TRY(ctx, call)                                                                                             \
    do {                                                                                                               \
        hipError_t e_ = (call);                                                                                        \
        if (e_ != hipSuccess)                                                                                          \
            return set_err(ctx, RT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));                       \
    } while (0)

struct BlobBuilder {
    std::vector<uint8_t> host;
    template <class T> size_t add(const std::vector<T> &v) {
        size_t off = (host.size() + 255) & ~size_t(255);
        host.resize(off + std::max<size_t>(v.size() * sizeof(T), 16));
        if (!v.empty()) std::memcpy(host.data() + off, v.data(), v.size() * sizeof(T));
        return off;
    }
    size_t add_bytes(const uint8_t *p, size_t n) {
        size_t off = (host.size() + 255) & ~size_t(255);
        host.resize(off + std::max<size_t>(n, 16));
        if (n) std::memcpy(host.data() + off, p, n);
        return off;
    }
};

double wall_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// device_build (float stream only): the hierarchy is built on the device from the leaf records (rt_build.h)
template <class R> int upload_one(rt_ctx *ctx, const rt_scene_desc *desc, DeviceScene<R> &ds, bool device_build = false, const BuildKnobs *knobs = nullptr) {
    const bool timing = dev_knob("RT_AMD_TIMING") != nullptr; // development aid: where an upload's time goes
    const double t0 = wall_ms();
    Flat<R> f;
    std::string err;
    // RT_AMD_REF_ORDER (development knob): the float stream keeps the reference's bvh_nodes and visiting order
    Flattener<R> fl(*desc, f, err, /*fuse_boxes=*/sizeof(R) == 4, /*own_hierarchy=*/sizeof(R) == 4 && !dev_knob("RT_AMD_REF_ORDER"), knobs);
    std::vector<typename Flattener<R>::LeafRecord> leaves;
    std::vector<UnitT<R>> leaf_units, leaf_cold;
    double box_pad = 0;
    int sub_begin = 0; // device build: where the boundary sub-streams of general media start in the leaf stream
    size_t built_units = 0; // device build: units the device wrote (tree, sentinel, sub-streams)
    int rc = device_build ? fl.run_leaves(leaves, box_pad, sub_begin) : fl.run();
    if (rc) return set_err(ctx, rc, err);
    DeviceBuildKnobs dk; // device build: the host build's settings (the tuner's candidate, if any)
    size_t leaves_units = 0;
    if (device_build) {
        if (leaves.empty()) return set_err(ctx, RT_ERR_UNSUPPORTED, "device build: the world holds no primitive");
        const BuildKnobs hk = knobs ? *knobs : BuildKnobs();
        dk.bins = hk.bins; dk.order = hk.order; dk.thr2 = hk.thr2; dk.thrn = hk.thrn; dk.overlap = hk.overlap;
        for (const auto &l : leaves) leaves_units += (size_t)l.n;
        leaf_units.swap(f.stream);
        leaf_cold.swap(f.cold);
        const size_t sub_units = leaf_units.size() - (size_t)sub_begin;
        const size_t total = built_stream_units(leaves_units, leaves.size()) + sub_units; // an upper bound
        if (total > (size_t(1) << 25)) return set_err(ctx, RT_ERR_NOMEM, "flattened scene too large");
        f.stream.assign(total, UnitT<R>{}); // the device writes these two ranges
        f.cold.assign(total, UnitT<R>{});
    }
    BlobBuilder b;
    size_t o_stream = b.add(f.stream), o_cold = b.add(f.cold);
    size_t o_bytes = b.add_bytes(desc->image_bytes, (size_t)desc->n_image_bytes);
    // the shading tables: one contiguous range (staged into LDS as a whole when it fits)
    size_t o_xf = b.add(f.xforms), o_mats = b.add(f.mats), o_texs = b.add(f.texs);
    size_t o_img = b.add(f.images), o_perlin = b.add(f.perlins), o_lights = b.add(f.lights);
    b.host.resize((b.host.size() + 255) & ~size_t(255));
    const size_t tab_bytes = b.host.size() - o_xf;
    ds.bytes = b.host.size();
    const double t1 = wall_ms();
    HIP_TRY(ctx, hipMalloc(&ds.blob, ds.bytes));
    const double t2 = wall_ms();
    HIP_TRY(ctx, hipMemcpy(ds.blob, b.host.data(), ds.bytes, hipMemcpyHostToDevice));
    if (timing)
        std::fprintf(stderr, "[rt_amd timing] upload<%s>: flatten + blob %.2f ms, hipMalloc %.2f ms, hipMemcpy of %zu bytes %.2f ms\n",
                     sizeof(R) == 4 ? "f32" : "f64", t1 - t0, t2 - t1, ds.bytes, wall_ms() - t2);
    uint8_t *base = (uint8_t *)ds.blob;
    if constexpr (sizeof(R) == 4) {
        if (device_build) {
            const double t3 = wall_ms();
            std::vector<BuildLeaf> bl(leaves.size());
            for (size_t k = 0; k < leaves.size(); k++) {
                bl[k].off = leaves[k].off;
                bl[k].n = leaves[k].n;
                for (int a = 0; a < 3; a++) {
                    bl[k].lo[a] = leaves[k].lo[a];
                    bl[k].hi[a] = leaves[k].hi[a];
                }
            }
            BlobBuilder in; // what the build reads: leaf table, leaf stream, its cold units
            const size_t o_bl = in.add(bl), o_lu = in.add(leaf_units), o_lc = in.add(leaf_cold);
            void *d_in = nullptr;
            HIP_TRY(ctx, hipMalloc(&d_in, in.host.size()));
            hipError_t e = hipMemcpy(d_in, in.host.data(), in.host.size(), hipMemcpyHostToDevice);
            int built = 0;
            if (e == hipSuccess)
                e = build_stream_on_device((const BuildLeaf *)((uint8_t *)d_in + o_bl), (int)bl.size(), (const UnitT<float> *)((uint8_t *)d_in + o_lu),
                                           (const UnitT<float> *)((uint8_t *)d_in + o_lc), sub_begin, (int)(leaf_units.size() - (size_t)sub_begin),
                                           (UnitT<float> *)(base + o_stream), (UnitT<float> *)(base + o_cold), box_pad, dk, &built);
            (void)hipFree(d_in);
            HIP_TRY(ctx, e);
            built_units = (size_t)built;
            if (built_units > f.stream.size()) return set_err(ctx, RT_ERR_HIP, "device build: the stream came out longer than its bound");
            if (timing) std::fprintf(stderr, "[rt_amd timing] device build of %zu leaves: %.2f ms\n", bl.size(), wall_ms() - t3);
        }
    }
    SceneT<R> &s = ds.sc;
    s.stream = (const UnitT<R> *)(base + o_stream);
    s.cold = (const UnitT<R> *)(base + o_cold);
    s.xforms = (const XformT<R> *)(base + o_xf);
    s.mats = (const MatT<R> *)(base + o_mats);
    s.texs = (const TexT<R> *)(base + o_texs);
    s.images = (const ImageT *)(base + o_img);
    s.image_bytes = base + o_bytes;
    s.perlins = (const PerlinT<R> *)(base + o_perlin);
    s.lights = (const LightT<R> *)(base + o_lights);
    s.n_units = (int)(built_units ? built_units : f.stream.size());
    int cap = LaunchCfg<R>::lds_bytes / (int)sizeof(UnitT<R>);
    if (ctx->lds_limit_units >= 0 && ctx->lds_limit_units < cap) cap = ctx->lds_limit_units;
    // the stream is staged into LDS whole or not at all (rt_kernels.h, Mem); the shading tables whenever they fit behind it
    s.lds_units = s.n_units <= cap ? s.n_units : 0;
    s.tab = base + o_xf;
    s.tab_bytes = (int32_t)tab_bytes;
    s.tab_in_lds = ((size_t)s.lds_units * sizeof(UnitT<R>) + tab_bytes <= (size_t)LaunchCfg<R>::lds_bytes && !dev_knob("RT_AMD_NO_TAB_LDS")) ? 1 : 0;
    s.n_lights = (int)f.lights.size();
    s.lights_is_list = f.lights_is_list;
    s.has_gen_medium = f.has_gen_medium;
    ds.n_fused_boxes = f.n_fused_boxes;
    return RT_OK;
}

template <class R> CameraT<R> convert_camera(const rt_camera &c) {
    CameraT<R> k{};
    k.W = c.image_width;
    k.H = c.image_height;
    k.sqrt_spp = c.sqrt_spp;
    k.max_depth = c.max_depth;
    k.recip_sqrt_spp = (R)(1.0 / c.sqrt_spp); // camera.h:216
    k.defocus_angle = (R)c.defocus_angle;
    for (int a = 0; a < 3; a++) {
        k.center[a] = (R)c.center[a];
        k.pixel00[a] = (R)c.pixel00_loc[a];
        k.du[a] = (R)c.pixel_delta_u[a];
        k.dv[a] = (R)c.pixel_delta_v[a];
        k.ddu[a] = (R)c.defocus_disk_u[a];
        k.ddv[a] = (R)c.defocus_disk_v[a];
        k.background[a] = (R)c.background[a];
    }
    return k;
}

int ensure(rt_ctx *ctx, void **p, size_t *have, size_t want) {
    if (*have >= want) return RT_OK;
    if (*p) HIP_TRY(ctx, hipFree(*p));
    *p = nullptr;
    *have = 0;
    HIP_TRY(ctx, hipMalloc(p, want));
    *have = want;
    return RT_OK;
}

// The f64 (parity) stream of a scene, built and uploaded the first time a call asks for RT_PREC_F64.
int need_f64(rt_ctx *ctx, const rt_scene *scene_) {
    rt_scene *scene = const_cast<rt_scene *>(scene_);
    if (scene->have_f64) return RT_OK;
    int rc = upload_one<double>(ctx, &scene->desc.d, scene->f64);
    if (rc == RT_OK) scene->have_f64 = true;
    return rc;
}

// What one render launch does, worked out on the host before anything is enqueued.
struct RenderPlan {
    RenderLaunch rl{};
    int n_chunks = 0;
    int n_samples = 0;
    size_t n_px = 0;
};

// Argument checks + work decomposition.  A work item = (8 x RT_TILE_H pixel tile, chunk of the sample range); waves pull items
// from a cursor in chunk-major order.  Default: a DECREASING-size schedule (each chunk takes a half / a third of what is left,
// down to a minimum), so the bulk of the samples runs in long items (lanes regenerate paths inside an item; a short item
// wastes lanes while its last paths finish) and the kernel ends on short ones (the end-of-kernel imbalance is one short item
// per wave).  params->sample_chunks > 0 forces that many equal chunks instead.  max_chunks caps the number of chunks (the
// partial frames must fit where the caller keeps them).
// (scene: for the chunk schedule -- a stream read from global memory wants longer middle chunks; may be null)
int plan_render(rt_ctx *ctx, const rt_scene *scene, const rt_camera *cam, const rt_render_params *params, int max_chunks, RenderPlan &pl) {
    if (cam->image_width <= 0 || cam->image_height <= 0 || cam->sqrt_spp <= 0 || cam->max_depth < 0)
        return set_err(ctx, RT_ERR_INVALID, "render: bad camera");
    if ((int64_t)cam->image_width * cam->image_height > (int64_t)1 << 28) return set_err(ctx, RT_ERR_INVALID, "render: image too large");
    if (cam->sqrt_spp > 46340) return set_err(ctx, RT_ERR_INVALID, "render: sqrt_spp^2 does not fit 32 bits");
    if (params->precision != RT_PREC_F32 && params->precision != RT_PREC_F64) return set_err(ctx, RT_ERR_INVALID, "bad precision");
    const int n_strata = cam->sqrt_spp * cam->sqrt_spp;
    const int s_begin = params->s_begin, s_end = params->s_end < 0 ? n_strata : params->s_end;
    if (s_begin < 0 || s_end > n_strata || s_begin > s_end) return set_err(ctx, RT_ERR_INVALID, "bad sample range");
    pl.n_px = (size_t)cam->image_width * cam->image_height;
    pl.n_samples = s_end - s_begin;
    RenderLaunch &rl = pl.rl;
    rl.k0 = (uint32_t)params->seed;
    rl.k1 = (uint32_t)(params->seed >> 32);
    rl.s_begin = s_begin;
    rl.s_end = s_end;
    rl.n_cus = ctx->n_cus;
    pl.n_chunks = 0;
    if (pl.n_samples == 0) return RT_OK;
    const int n_samples = pl.n_samples;
    const int tiles = ((cam->image_width + 7) / 8) * ((cam->image_height + RT_TILE_H - 1) / RT_TILE_H);
    int n_chunks = params->sample_chunks;
    const int wave_slots = ctx->n_cus * 16;
    if (n_chunks <= 0 && tiles < 2 * wave_slots) { // small image: equal chunks, enough of them to give every wave ~32 items
        n_chunks = (wave_slots * 32 + tiles - 1) / tiles;
        int most = n_samples / 24;
        if (most < 1) most = 1;
        if (n_chunks > most) n_chunks = most;
    }
    if (max_chunks < 1) max_chunks = 1;
    const int sched_cap = max_chunks < RT_MAX_SCHED_CHUNKS_HOST ? max_chunks : RT_MAX_SCHED_CHUNKS_HOST;
    if (n_chunks > 0) {
        if (n_chunks > n_samples) n_chunks = n_samples;
        if (n_chunks > 65535) n_chunks = 65535;
        if (n_chunks > max_chunks) n_chunks = max_chunks;
        rl.chunk_size = (n_samples + n_chunks - 1) / n_chunks;
        n_chunks = (n_samples + rl.chunk_size - 1) / rl.chunk_size;
    } else {
        const int min_chunk = ctx->min_chunk > 0 ? ctx->min_chunk : (n_samples >= 360 ? 32 : 16);
        const int divisor = ctx->chunk_divisor > 1 ? ctx->chunk_divisor : (n_samples >= 360 ? 3 : 2); // (round 2's schedule: a third / a half of what is left)
        // round 3: each chunk takes 85 % of what is left: 961 strata are 816 + 123 + 22 instead of nine chunks, an 8-GPU rank's 120
        // are 102 + 18 -- a third of the partial frames to write and sum; the kernel still ends on its shortest items.  (Whole
        // frame: a third / a half / three quarters / 85 % / 90 % of what is left: 517 / 519 / 513.0 / 512.5 / 512.2 ms; a 1/8
        // range at a half / two thirds / 85 % / one chunk: 70.3 / 68.6 / 67.8 / 69.8 ms; tools/ab.py with RT_AMD_CHUNKS.)
        // A stream that is read from global memory (mesh_scene) takes three quarters: its 484 strata as 411 + 41 + 32 cost 324-338
        // ms against 297-304 as 363 + 91 + 30 -- short items cost such a stream more, so fewer samples should sit in them.
        const bool resident = !scene || scene->f32.sc.lds_units > 0;
        const int pct = ctx->chunk_pct > 0 ? ctx->chunk_pct : (ctx->chunk_divisor > 1 ? 0 : (resident ? 85 : 75));
        rl.chunk_size = 0;
        n_chunks = 0;
        int at = s_begin;
        rl.chunk_start[0] = at;
        while (at < s_end) {
            int left = s_end - at, take = pct > 0 ? (int)((int64_t)left * pct / 100) : left / divisor;
            if (ctx->max_chunk > 0 && take > ctx->max_chunk) take = ctx->max_chunk;
            if (take < min_chunk) take = min_chunk;
            // (what would be left is too short for a chunk of its own: it joins this one -- unless that makes the LAST chunk a long
            // one: mesh_scene's 484 strata as 411 + 73 took 327-338 ms against 301-304 as 363 + 91 + 30; the kernel has to end on
            // short items)
            if (take < left && left - take < min_chunk / 2) take = left > 2 * min_chunk ? left - min_chunk : left;
            if (take > left || n_chunks == sched_cap - 1) take = left;
            at += take;
            rl.chunk_start[++n_chunks] = at;
        }
    }
    rl.n_chunks = n_chunks;
    pl.n_chunks = n_chunks;
    return RT_OK;
}

// Enqueue one planned render on `stream` of the context's device (current on entry): counters cleared, the path-tracing
// kernel into `partial` (n_chunks > 1) or straight into d_accum, then the fixed-order sum of the partial frames.  Nothing
// here allocates or synchronises.

int enqueue_render(rt_ctx *ctx, const rt_scene *scene, const rt_camera *cam, int precision, const RenderPlan &pl_, void *partial,
                   unsigned long long *counters, float *d_accum, hipStream_t stream, bool shared_frame_wanted) {
    RenderPlan pl = pl_;
    // Several sample chunks, float kernels: 64-bit fixed-point cells, summed exactly (the frame does not depend on the cut
    // into chunks or on who flushes when).  Two layouts:
    //   a partial frame per chunk, written with plain stores and summed by one pass afterwards (rt_render_device: the
    //   decreasing schedule is 3-4 chunks at 1080p, i.e. 3-4 x 24 B / pixel written and read once);
    //   ONE zeroed frame that every chunk's work items add to with 64-bit atomics (rt_render_static: it fits the static
    //   scratch whatever the number of chunks).  Same time; the atomics reach memory as 32-byte requests -- measured 1.23 GB
    //   of L2 write requests per 1080p launch against 0.51 GB for nine partial frames -- so it is not the default.
    // double kernels (parity path): a float partial frame per chunk, summed in double in a fixed order.
    const bool shared_frame = precision == RT_PREC_F32 && pl.n_chunks > 1 && shared_frame_wanted;
    const bool cells_i64 = precision == RT_PREC_F32 && pl.n_chunks > 1;
    pl.rl.out_i64 = shared_frame ? 2 : (cells_i64 ? 1 : 0);
    if (pl.n_samples == 0) {
        HIP_TRY(ctx, hipMemsetAsync(d_accum, 0, pl.n_px * 3 * sizeof(float), stream));
        HIP_TRY(ctx, hipMemsetAsync(counters, 0, RT_N_COUNTERS * sizeof(unsigned long long), stream));
        return RT_OK;
    }
    void *target = pl.n_chunks > 1 ? partial : (void *)d_accum;
    HIP_TRY(ctx, hipMemsetAsync(counters, 0, RT_N_COUNTERS * sizeof(unsigned long long), stream));
    HIP_TRY(ctx, hipMemsetAsync(counters + 37, 0xff, sizeof(unsigned long long), stream)); // [37]: a minimum (diagnostic builds)
    if (shared_frame) HIP_TRY(ctx, hipMemsetAsync(partial, 0, pl.n_px * 3 * 8, stream));
    if (precision == RT_PREC_F64)
        HIP_TRY(ctx, launch_render<double>(scene->f64.sc, convert_camera<double>(*cam), pl.rl, target, counters, stream));
    else
        HIP_TRY(ctx, launch_render<float>(scene->f32.sc, convert_camera<float>(*cam), pl.rl, target, counters, stream));
    if (pl.n_chunks > 1) HIP_TRY(ctx, launch_reduce_chunks(partial, cells_i64, d_accum, pl.n_px * 3, shared_frame ? 1 : pl.n_chunks, stream));
    return RT_OK;
}

// bytes of scratch a planned render needs beyond d_accum
size_t partial_bytes_of(const RenderPlan &pl, int precision, bool shared_frame) {
    if (pl.n_chunks <= 1) return 0;
    if (precision == RT_PREC_F32) return (shared_frame ? 1 : (size_t)pl.n_chunks) * pl.n_px * 3 * 8;
    return (size_t)pl.n_chunks * pl.n_px * 3 * 4;
}

void print_profile(rt_ctx *ctx, const unsigned long long *c) { // diagnostic builds (-DRT_PROFILE_*) fill counters[3..39]
    if (dev_knob("RT_AMD_PROFILE_ACTIONS")) {
        const char *nm[6] = {"box", "leaf", "leaf/aabox", "leaf/medium", "leaf/other", "shade+deal"};
        for (int k = 0; k < 6; k++)
            std::fprintf(stderr, "[rt_amd profile] %-11s cycles %.4g (%.1f%% of wave time) trips %.4g cyc/trip %.1f lanes/trip %.1f\n",
                         nm[k], (double)c[4 + 3 * k], 100.0 * (double)c[4 + 3 * k] / (double)(c[3] ? c[3] : 1), (double)c[5 + 3 * k],
                         (double)c[4 + 3 * k] / (double)(c[5 + 3 * k] ? c[5 + 3 * k] : 1),
                         (double)c[6 + 3 * k] / (double)(c[5 + 3 * k] ? c[5 + 3 * k] : 1));
        std::fprintf(stderr, "[rt_amd profile] box steps %.4g (%.1f per trip), lanes per step %.1f, lane-steps %.4g\n", (double)c[56],
                     (double)c[56] / (double)(c[5] ? c[5] : 1), (double)c[57] / (double)(c[56] ? c[56] : 1), (double)c[57]);
        std::fprintf(stderr, "[rt_amd profile] per trip: %.1f lanes idle (item tail), %.1f lanes waiting for shading, of 64; %.1f%% of trips after the item's last unit was dealt\n",
                     (double)c[33] / (double)(c[35] ? c[35] : 1), (double)c[34] / (double)(c[35] ? c[35] : 1), 100.0 * (double)c[36] / (double)(c[35] ? c[35] : 1));
    }
    if (dev_knob("RT_AMD_PROFILE_SHADE")) { // builds with -DRT_PROFILE_SHADE: the stages of the shading trip
        const char *nm[7] = {"hit records", "dealing", "noise (wave)", "generator", "camera rays", "scatter", "trav_begin+fog"};
        const double trips = (double)(c[54] ? c[54] : 1);
        for (int k = 0; k < 7; k++)
            std::fprintf(stderr, "[rt_amd profile] shading/%-14s cycles/trip %.1f lanes/trip %.1f\n", nm[k], (double)c[40 + 2 * k] / trips,
                         (double)c[41 + 2 * k] / trips);
    }
    if (dev_knob("RT_AMD_PROFILE_TAIL")) { // builds with -DRT_PROFILE_TAIL: wave end times
        const double span = (double)(c[38] - c[37]), n_waves = (double)ctx->n_cus * 16.0;
        std::fprintf(stderr, "[rt_amd profile] waves busy on average %.1f%% of the kernel's span (first start to last end); wave run time min %.1f%% max %.1f%% of the span\n",
                     span > 0 ? 100.0 * (double)c[39] / n_waves / span : 0.0, span > 0 ? 100.0 * (double)(~c[35]) / span : 0.0,
                     span > 0 ? 100.0 * (double)c[36] / span : 0.0);
        std::fprintf(stderr, "[rt_amd profile] after the item cursor ran dry a wave kept running for %.1f%% of the span on average (max %.1f%%): %.0f trips per wave with %.1f lanes active\n",
                     span > 0 ? 100.0 * (double)c[58] / n_waves / span : 0.0, span > 0 ? 100.0 * (double)c[59] / span : 0.0, (double)c[60] / n_waves,
                     (double)c[61] / (double)(c[60] ? c[60] : 1));
    }
}

// After the work enqueued between ev0 and ev1 has finished: fill `stats` from the events and the kernel's counters.
int collect_stats(rt_ctx *ctx, const RenderPlan &pl, const unsigned long long *counters, rt_stats *stats) {
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    unsigned long long c[RT_N_COUNTERS];
    HIP_TRY(ctx, hipMemcpy(c, counters, sizeof(c), hipMemcpyDeviceToHost));
    if (pl.n_samples > 0) print_profile(ctx, c);
    stats->samples = (uint64_t)pl.n_px * (uint64_t)pl.n_samples;
    stats->rays = pl.n_samples > 0 ? c[0] : 0;
    stats->kernel_ms = pl.n_samples > 0 ? ms : 0.0;
    stats->sample_chunks = (uint32_t)pl.n_chunks;
    stats->reserved = 0;
    return RT_OK;
}

} // namespace

extern "C" {

int rt_abi_sizeof(const char *name) {
#define S(T)                                                                                                           \
    if (!std::strcmp(name, #T)) return (int)sizeof(T)
    S(rt_node);
    S(rt_sphere);
    S(rt_quad);
    S(rt_triangle);
    S(rt_material);
    S(rt_texture);
    S(rt_image);
    S(rt_perlin);
    S(rt_scene_desc);
    S(rt_camera);
    S(rt_render_params);
    S(rt_ray);
    S(rt_hit);
    S(rt_stats);
#undef S
    return -1;
}

int rt_init(int device, rt_ctx **out) {
    if (!out) return RT_ERR_INVALID;
    *out = nullptr;
    rt_ctx *ctx = new (std::nothrow) rt_ctx;
    if (!ctx) return RT_ERR_NOMEM;
    ctx->device = device;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0 || device < 0 || device >= n) {
        std::fprintf(stderr, "rt_init: no usable HIP device %d (count %d): %s\n", device, n, hipGetErrorString(e));
        delete ctx;
        return RT_ERR_HIP;
    }
    if (hipSetDevice(device) != hipSuccess || hipMalloc((void **)&ctx->counter, RT_N_COUNTERS * sizeof(unsigned long long)) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        rt_shutdown(ctx);
        return RT_ERR_HIP;
    }
    // The runtime sets up its host-to-device copy path (staging buffers, blit kernels) on the FIRST copy of the process:
    // 15 ms, 150 ms in the first process on a freshly booted box.  Paid here, with the other one-off costs of creating a
    // context, instead of inside the first frame's scene upload.
    {
        const unsigned long long zeros[RT_N_COUNTERS] = {};
        if (hipMemcpy(ctx->counter, zeros, sizeof(zeros), hipMemcpyHostToDevice) != hipSuccess) {
            rt_shutdown(ctx);
            return RT_ERR_HIP;
        }
    }
    // this device's copy of the static scratch array (rt_render_static); the device is current here
    e = hipGetSymbolAddress((void **)&ctx->static_scratch, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ctx->static_scratch) {
        std::fprintf(stderr, "rt_init: hipGetSymbolAddress(scratch) on device %d: %s\n", device, hipGetErrorString(e));
        rt_shutdown(ctx);
        return RT_ERR_HIP;
    }
    // ... and a copy of the size of a scene blob takes another path (pinned staging chunks, set up on its first use: 6 ms
    // of the first upload, whatever its size).  The destination is a buffer of this context's own, freed again: never the
    // static scratch, which a render of another context on this device may be using.
    {
        const std::vector<unsigned char> warm((size_t)2 << 20, 0);
        void *d_warm = nullptr;
        if (hipMalloc(&d_warm, warm.size()) != hipSuccess || hipMemcpy(d_warm, warm.data(), warm.size(), hipMemcpyHostToDevice) != hipSuccess) {
            if (d_warm) (void)hipFree(d_warm);
            rt_shutdown(ctx);
            return RT_ERR_HIP;
        }
        (void)hipFree(d_warm);
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->n_cus = prop.multiProcessorCount;
    if (const char *e1 = dev_knob("RT_AMD_LDS_UNITS")) ctx->lds_limit_units = std::atoi(e1);
    if (const char *e5 = dev_knob("RT_AMD_CHUNKS")) std::sscanf(e5, "%d,%d,%d,%d", &ctx->min_chunk, &ctx->chunk_divisor, &ctx->max_chunk, &ctx->chunk_pct);
    if (const char *e6 = dev_knob("RT_AMD_FRAME_MODE")) ctx->frame_mode = std::atoi(e6); // 1: a partial frame per chunk, 2: one shared frame (atomics)
    *out = ctx;
    return RT_OK;
}

void rt_shutdown(rt_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->partial) (void)hipFree(ctx->partial);
    if (ctx->accum) (void)hipFree(ctx->accum);
    if (ctx->stage) (void)hipFree(ctx->stage);
    if (ctx->rgb) (void)hipFree(ctx->rgb);
    if (ctx->counter) (void)hipFree(ctx->counter);
    if (ctx->xstream) (void)hipStreamDestroy(ctx->xstream);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    delete ctx;
}

const char *rt_last_error(const rt_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

static int scene_upload(rt_ctx *ctx, const rt_scene_desc *desc, rt_scene **out, bool device_build) {
    if (!ctx || !desc || !out) return RT_ERR_INVALID;
    *out = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    rt_scene *s = new (std::nothrow) rt_scene;
    if (!s) return set_err(ctx, RT_ERR_NOMEM, "out of host memory");
    s->ctx = ctx;
    s->n_nodes_desc = desc->n_nodes;
    int rc = upload_one<float>(ctx, desc, s->f32, device_build); // validates `desc` before anything reads through its pointers
    if (rc != RT_OK) {
        rt_scene_free(s);
        return rc;
    }
    s->desc.assign(*desc);
    *out = s;
    return RT_OK;
}

int rt_scene_upload(rt_ctx *ctx, const rt_scene_desc *desc, rt_scene **out) { return scene_upload(ctx, desc, out, false); }
int rt_scene_upload_device_build(rt_ctx *ctx, const rt_scene_desc *desc, rt_scene **out) { return scene_upload(ctx, desc, out, true); }

void rt_scene_free(rt_scene *scene) {
    if (!scene) return;
    if (scene->ctx) (void)hipSetDevice(scene->ctx->device);
    if (scene->f32.blob) (void)hipFree(scene->f32.blob);
    if (scene->f64.blob) (void)hipFree(scene->f64.blob);
    delete scene;
}

int rt_scene_info(const rt_scene *scene, int32_t *out /* [4]: flat nodes, lights, f32 bytes, f64 bytes (0 until first used) */) {
    if (!scene || !out) return RT_ERR_INVALID;
    out[0] = scene->f32.sc.n_units;
    out[1] = scene->f32.sc.n_lights;
    out[2] = (int32_t)scene->f32.bytes;
    out[3] = (int32_t)scene->f64.bytes;
    return RT_OK;
}

// The candidate settings of rt_scene_tune: the default first (a candidate must beat it by 2 % to replace it), then the
// neighbours that won on one or another of the reference's scenes in round 2's sweeps (bins 8 ... 16, larger box / larger box
// area per leaf first).
static const struct { int bins, order; } kTuneCandidates[RT_TUNE_CANDIDATES] = {{12, 1}, {8, 1}, {16, 1}, {10, 1}, {12, 5}, {16, 5}};

int rt_scene_tune(rt_ctx *ctx, rt_scene *scene, const rt_camera *cam, int probe_strata, float *ms_out, int32_t *chosen) {
    if (!ctx || !scene || !cam) return set_err(ctx, RT_ERR_INVALID, "rt_scene_tune: null argument");
    if (scene->ctx != ctx) return set_err(ctx, RT_ERR_INVALID, "rt_scene_tune: the scene belongs to another context");
    if (cam->image_width <= 0 || cam->image_height <= 0 || cam->sqrt_spp <= 0) return set_err(ctx, RT_ERR_INVALID, "rt_scene_tune: bad camera");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int n_strata = cam->sqrt_spp * cam->sqrt_spp;
    const size_t n_px = (size_t)cam->image_width * cam->image_height;
    if (probe_strata <= 0) { // enough samples for the differences to show above a launch's fixed cost: ~60 M, at least 16 strata
        probe_strata = (int)std::min<size_t>((size_t)n_strata, std::max<size_t>(16, (size_t)60000000 / n_px));
    }
    if (probe_strata > n_strata) probe_strata = n_strata;
    int rc = ensure(ctx, (void **)&ctx->accum, &ctx->accum_bytes, n_px * 3 * sizeof(float));
    if (rc) return rc;
    rt_render_params p{};
    p.seed = 0x74756e65ull; // any: the probe's image is thrown away
    p.s_begin = 0;
    p.s_end = probe_strata;
    p.precision = RT_PREC_F32;
    auto probe = [&](float &ms) {
        ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            rt_stats st{};
            int r = rt_render_device(ctx, scene, cam, &p, ctx->accum, nullptr, &st);
            if (r) return r;
            ms = rep == 0 ? (float)st.kernel_ms : std::min(ms, (float)st.kernel_ms);
        }
        return (int)RT_OK;
    };
    int best_k = 0;
    float best_ms = 0;
    if ((rc = probe(best_ms)) != RT_OK) return rc; // candidate 0 is the stream the scene was uploaded with
    if (ms_out) ms_out[0] = best_ms;
    for (int k = 1; k < RT_TUNE_CANDIDATES; k++) {
        BuildKnobs kn;
        kn.bins = kTuneCandidates[k].bins;
        kn.order = kTuneCandidates[k].order;
        DeviceScene<float> cand;
        rc = upload_one<float>(ctx, &scene->desc.d, cand, false, &kn);
        if (rc != RT_OK) {
            if (cand.blob) (void)hipFree(cand.blob);
            return rc;
        }
        std::swap(scene->f32, cand); // the scene renders from the candidate now; `cand` holds what it had
        float ms = 0;
        rc = probe(ms);
        if (ms_out) ms_out[k] = ms;
        if (rc == RT_OK && ms < 0.98f * best_ms) { // keep the candidate
            best_ms = ms;
            best_k = k;
        } else {
            std::swap(scene->f32, cand); // back to the incumbent
        }
        if (cand.blob) (void)hipFree(cand.blob);
        if (rc != RT_OK) return rc;
    }
    if (chosen) *chosen = best_k;
    return RT_OK;
}

int rt_scene_flatten_stream(const rt_scene_desc *desc, int reference_order, void *units, int32_t *node_ids, size_t cap_units,
                            size_t *n_units) {
    if (!desc || !n_units || (cap_units > 0 && !units)) return RT_ERR_INVALID;
    std::string err;
    Flat<float> f;
    int rc = Flattener<float>(*desc, f, err, true, !reference_order).run();
    if (rc) return rc;
    *n_units = f.stream.size();
    const size_t n = std::min(cap_units, f.stream.size());
    if (n) std::memcpy(units, f.stream.data(), n * sizeof(UnitT<float>));
    if (node_ids)
        for (size_t k = 0; k < n; k++) node_ids[k] = dec_int(f.cold[k].w[3]);
    return RT_OK;
}

int rt_scene_download_stream(const rt_scene *scene, void *units, int32_t *node_ids, size_t cap_units, size_t *n_units) {
    if (!scene || !n_units || (cap_units > 0 && !units)) return RT_ERR_INVALID;
    rt_ctx *ctx = scene->ctx;
    const size_t total = (size_t)scene->f32.sc.n_units;
    *n_units = total;
    const size_t n = std::min(cap_units, total);
    if (n == 0) return RT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpy(units, scene->f32.sc.stream, n * sizeof(UnitT<float>), hipMemcpyDeviceToHost));
    if (node_ids) {
        std::vector<UnitT<float>> cold(n);
        HIP_TRY(ctx, hipMemcpy(cold.data(), scene->f32.sc.cold, n * sizeof(UnitT<float>), hipMemcpyDeviceToHost));
        for (size_t k = 0; k < n; k++) node_ids[k] = dec_int(cold[k].w[3]);
    }
    return RT_OK;
}

int rt_scene_flatten_info(const rt_scene_desc *desc, int reference_order, int32_t *out) {
    if (!desc || !out) return RT_ERR_INVALID;
    std::string err;
    Flat<float> f32;
    int rc = Flattener<float>(*desc, f32, err, true, !reference_order).run();
    if (rc) return rc;
    Flat<double> f64;
    rc = Flattener<double>(*desc, f64, err, false).run();
    if (rc) return rc;
    out[0] = (int32_t)f32.stream.size();
    out[1] = (int32_t)f64.stream.size();
    out[2] = f32.n_fused_boxes;
    out[3] = (int32_t)f32.lights.size();
    out[4] = f32.n_baked_instances;
    return RT_OK;
}

int rt_trace_batch(rt_ctx *ctx, const rt_scene *scene, const rt_ray *rays, const double *xi, size_t n, rt_hit *out,
                   int precision) {
    if (!ctx || !scene || (n && (!rays || !out))) return set_err(ctx, RT_ERR_INVALID, "rt_trace_batch: null argument");
    if (precision != RT_PREC_F32 && precision != RT_PREC_F64) return set_err(ctx, RT_ERR_INVALID, "bad precision");
    if (n == 0) return RT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (precision == RT_PREC_F64) {
        int rc64 = need_f64(ctx, scene);
        if (rc64) return rc64;
    }
    rt_ray *d_rays = nullptr;
    double *d_xi = nullptr;
    rt_hit *d_out = nullptr;
    int rc = RT_OK;
    hipError_t e = hipMalloc((void **)&d_rays, n * sizeof(rt_ray));
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, n * sizeof(rt_hit));
    if (e == hipSuccess && xi) e = hipMalloc((void **)&d_xi, n * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(d_rays, rays, n * sizeof(rt_ray), hipMemcpyHostToDevice);
    if (e == hipSuccess && xi) e = hipMemcpy(d_xi, xi, n * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        if (precision == RT_PREC_F64) e = launch_trace<double>(scene->f64.sc, d_rays, d_xi, n, d_out, 0u, 0u, ctx->n_cus, nullptr);
        else e = launch_trace<float>(scene->f32.sc, d_rays, d_xi, n, d_out, 0u, 0u, ctx->n_cus, nullptr);
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, d_out, n * sizeof(rt_hit), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = set_err(ctx, RT_ERR_HIP, std::string("rt_trace_batch: ") + hipGetErrorString(e));
    if (d_rays) (void)hipFree(d_rays);
    if (d_xi) (void)hipFree(d_xi);
    if (d_out) (void)hipFree(d_out);
    return rc;
}

int rt_render_device(rt_ctx *ctx, const rt_scene *scene, const rt_camera *cam, const rt_render_params *params,
                     float *d_accum, void *stream_, rt_stats *stats) {
    if (!ctx || !scene || !cam || !params || !d_accum) return set_err(ctx, RT_ERR_INVALID, "rt_render_device: null argument");
    RenderPlan pl;
    int rc = plan_render(ctx, scene, cam, params, 65535, pl);
    if (rc) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (params->precision == RT_PREC_F64 && (rc = need_f64(ctx, scene)) != RT_OK) return rc;
    if (pl.n_chunks > 1) {
        rc = ensure(ctx, &ctx->partial, &ctx->partial_bytes, partial_bytes_of(pl, params->precision, ctx->frame_mode == 2));
        if (rc) return rc;
    }
    if (stats) HIP_TRY(ctx, hipEventRecord(ctx->ev0, stream));
    rc = enqueue_render(ctx, scene, cam, params->precision, pl, ctx->partial, ctx->counter, d_accum, stream, ctx->frame_mode == 2);
    if (rc) return rc;
    if (stats) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev1, stream));
        return collect_stats(ctx, pl, ctx->counter, stats);
    }
    return RT_OK;
}

int rt_render_static(rt_ctx *ctx, const rt_scene *scene, const rt_camera *cam, const rt_render_params *params, float *d_accum,
                     void *stream_) {
    if (!ctx || !scene || !cam || !params || !d_accum) return set_err(ctx, RT_ERR_INVALID, "rt_render_static: null argument");
    if (scene->ctx != ctx) return set_err(ctx, RT_ERR_INVALID, "rt_render_static: the scene belongs to another context");
    if (cam->image_width <= 0 || cam->image_height <= 0) return set_err(ctx, RT_ERR_INVALID, "render: bad camera");
    // The scratch of a render with several sample chunks lives in the static array: float kernels need ONE frame of 64-bit
    // cells (W * H * 24 bytes: 1920 x 1080 is 49.8 MB and fits), double kernels a float frame per chunk.  What does not fit
    // renders as ONE chunk, straight into d_accum (no scratch beyond the counters; the kernel then ends on long work items
    // instead of the decreasing schedule's short ones: a few percent of a frame).
    const size_t cells = (size_t)cam->image_width * (size_t)cam->image_height * 3;
    const size_t room = RT_STATIC_SCRATCH_BYTES - RT_STATIC_COUNTER_BYTES;
    const size_t max_chunks = params->precision == RT_PREC_F64 ? room / (cells * 4) : (cells * 8 <= room ? 65535 : 1);
    if (params->precision == RT_PREC_F64 && !scene->have_f64)
        return set_err(ctx, RT_ERR_INVALID, "rt_render_static: the f64 stream of this scene is built (and allocated) by the first "
                                            "RT_PREC_F64 call of rt_render_device / rt_trace_batch; make that call first");
    RenderPlan pl;
    int rc = plan_render(ctx, scene, cam, params, (int)std::min<size_t>(max_chunks, 65535), pl); // (0 or 1: a single chunk)
    if (rc) return rc;
    if (partial_bytes_of(pl, params->precision, true) > room) // (cannot happen: the plan honours its cap)
        return set_err(ctx, RT_ERR_INVALID, "rt_render_static: " + std::to_string(partial_bytes_of(pl, params->precision, true)) +
                                                " bytes of scratch do not fit the static array (" + std::to_string(room) + " bytes): use rt_render_device");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // every use of the array's memory goes through the per-device pointer taken in rt_init
    unsigned long long *counters = reinterpret_cast<unsigned long long *>(ctx->static_scratch);
    void *partial = ctx->static_scratch + RT_STATIC_COUNTER_BYTES;
    return enqueue_render(ctx, scene, cam, params->precision, pl, partial, counters, d_accum, (hipStream_t)stream_, /*shared_frame_wanted=*/true);
}

int rt_tonemap_device(rt_ctx *ctx, const float *d_accum, double scale, size_t n_pixels, uint8_t *d_rgb8, void *stream) {
    if (!ctx || !d_accum || !d_rgb8) return set_err(ctx, RT_ERR_INVALID, "rt_tonemap_device: null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_tonemap(d_accum, scale, n_pixels * 3, d_rgb8, (hipStream_t)stream));
    return RT_OK;
}

// accum (device, on ctx's device) -> the caller's host buffers
static int finish_to_host(rt_ctx *ctx, const rt_camera *cam, float *accum_rgb, uint8_t *rgb8) {
    const size_t n_px = (size_t)cam->image_width * cam->image_height;
    if (accum_rgb) HIP_TRY(ctx, hipMemcpy(accum_rgb, ctx->accum, n_px * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (rgb8) {
        int rc = ensure(ctx, (void **)&ctx->rgb, &ctx->rgb_bytes, n_px * 3);
        if (rc) return rc;
        const double scale = 1.0 / ((double)cam->sqrt_spp * cam->sqrt_spp); // camera.h:215 pixel_samples_scale
        rc = rt_tonemap_device(ctx, ctx->accum, scale, n_px, ctx->rgb, nullptr);
        if (rc) return rc;
        HIP_TRY(ctx, hipDeviceSynchronize());
        HIP_TRY(ctx, hipMemcpy(rgb8, ctx->rgb, n_px * 3, hipMemcpyDeviceToHost));
    }
    return RT_OK;
}

int rt_render(rt_ctx *ctx, const rt_scene *scene, const rt_camera *cam, const rt_render_params *params, float *accum_rgb,
              uint8_t *rgb8, rt_stats *stats) {
    if (!ctx || !scene || !cam || !params) return set_err(ctx, RT_ERR_INVALID, "rt_render: null argument");
    if (cam->image_width <= 0 || cam->image_height <= 0 || cam->sqrt_spp <= 0)
        return set_err(ctx, RT_ERR_INVALID, "rt_render: bad camera");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n_px = (size_t)cam->image_width * cam->image_height;
    int rc = ensure(ctx, (void **)&ctx->accum, &ctx->accum_bytes, n_px * 3 * sizeof(float));
    if (rc) return rc;
    rt_stats st{};
    rc = rt_render_device(ctx, scene, cam, params, ctx->accum, nullptr, &st);
    if (rc) return rc;
    if (stats) *stats = st;
    return finish_to_host(ctx, cam, accum_rgb, rgb8);
}

int rt_render_multi(rt_ctx *const *ctxs, const rt_scene *const *scenes, int n_devices, const rt_camera *cam,
                    const rt_render_params *params, float *accum_rgb, uint8_t *rgb8, rt_stats *stats) {
    if (!ctxs || !scenes || n_devices < 1 || !ctxs[0]) return RT_ERR_INVALID;
    rt_ctx *root = ctxs[0];
    if (!cam || !params) return set_err(root, RT_ERR_INVALID, "rt_render_multi: null argument");
    if (n_devices > 64) return set_err(root, RT_ERR_INVALID, "rt_render_multi: more than 64 contexts");
    for (int k = 0; k < n_devices; k++)
        if (!ctxs[k] || !scenes[k] || scenes[k]->ctx != ctxs[k])
            return set_err(root, RT_ERR_INVALID, "rt_render_multi: scenes[k] must have been uploaded through ctxs[k]");
    for (int k = 0; k < n_devices; k++)
        for (int j = 0; j < k; j++)
            if (ctxs[j] == ctxs[k]) // a context's buffers, counters and events serve one launch at a time
                return set_err(root, RT_ERR_INVALID, "rt_render_multi: the same context is listed twice (create one per entry)");
    RenderPlan whole;
    int rc = plan_render(root, scenes[0], cam, params, 65535, whole); // validates the camera and the range once
    if (rc) return rc;
    const int s_begin = whole.rl.s_begin, n_samples = whole.n_samples;
    const size_t n_px = whole.n_px, bytes = n_px * 3 * sizeof(float);
    // enqueue every device's share: strata [s_begin + k*n/G, s_begin + (k+1)*n/G); nothing below waits until all are running
    std::vector<RenderPlan> plans((size_t)n_devices);
    auto fail_from = [&](int k, int code) { // report a context's error through the root context
        if (k != 0) root->err = "device " + std::to_string(ctxs[k]->device) + ": " + ctxs[k]->err;
        return code;
    };
    for (int k = 0; k < n_devices; k++) {
        rt_ctx *c = ctxs[k];
        rt_render_params p = *params;
        p.s_begin = s_begin + (int)(((int64_t)k * n_samples) / n_devices);
        p.s_end = s_begin + (int)(((int64_t)(k + 1) * n_samples) / n_devices);
        if ((rc = plan_render(c, scenes[k], cam, &p, 65535, plans[k])) != RT_OK) return fail_from(k, rc);
        if (hipSetDevice(c->device) != hipSuccess) return fail_from(k, set_err(c, RT_ERR_HIP, "hipSetDevice failed"));
        if (params->precision == RT_PREC_F64 && (rc = need_f64(c, scenes[k])) != RT_OK) return fail_from(k, rc);
        if ((rc = ensure(c, (void **)&c->accum, &c->accum_bytes, bytes)) != RT_OK) return fail_from(k, rc);
        if (plans[k].n_chunks > 1 && (rc = ensure(c, &c->partial, &c->partial_bytes, partial_bytes_of(plans[k], params->precision, false))) != RT_OK)
            return fail_from(k, rc);
        if (hipEventRecord(c->ev0, nullptr) != hipSuccess) return fail_from(k, set_err(c, RT_ERR_HIP, "hipEventRecord failed"));
        if ((rc = enqueue_render(c, scenes[k], cam, params->precision, plans[k], c->partial, c->counter, c->accum, nullptr, false)) != RT_OK)
            return fail_from(k, rc);
        if (hipEventRecord(c->ev1, nullptr) != hipSuccess) return fail_from(k, set_err(c, RT_ERR_HIP, "hipEventRecord failed"));
    }
    rt_stats total{};
    for (int k = 0; k < n_devices; k++) {
        rt_ctx *c = ctxs[k];
        rt_stats st{};
        if (hipSetDevice(c->device) != hipSuccess) return fail_from(k, set_err(c, RT_ERR_HIP, "hipSetDevice failed"));
        if ((rc = collect_stats(c, plans[k], c->counter, &st)) != RT_OK) return fail_from(k, rc);
        total.samples += st.samples;
        total.rays += st.rays;
        if (st.kernel_ms > total.kernel_ms) total.kernel_ms = st.kernel_ms; // the devices run side by side
        if (st.sample_chunks > total.sample_chunks) total.sample_chunks = st.sample_chunks;
    }
    // The exchange step: a binary tree over the contexts in a FIXED order (the frame repeats): at stride 1, 2, 4 ... context k
    // takes the sums of context k + stride -- a direct peer copy over xGMI into its staging buffer, then one add kernel, both
    // on its own exchange stream -- so the pairs of a level run side by side on different devices and links, and 8 GPUs
    // take 3 levels instead of 7 copies in a row on the root.  (Every render above has finished: collect_stats waited.)
    for (int stride = 1; stride < n_devices; stride *= 2) {
        for (int k = 0; k + stride < n_devices; k += 2 * stride) {
            rt_ctx *rcv = ctxs[k], *snd = ctxs[k + stride];
            if (hipSetDevice(rcv->device) != hipSuccess) return fail_from(k, set_err(rcv, RT_ERR_HIP, "hipSetDevice failed"));
            if (!rcv->xstream && hipStreamCreateWithFlags(&rcv->xstream, hipStreamNonBlocking) != hipSuccess)
                return fail_from(k, set_err(rcv, RT_ERR_HIP, "hipStreamCreateWithFlags failed"));
            if (snd->device != rcv->device) { // direct access to the peer's memory; harmless if it was enabled before
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, rcv->device, snd->device) == hipSuccess && can) {
                    const hipError_t pe = hipDeviceEnablePeerAccess(snd->device, 0);
                    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError(); // the copy below still works, staged by the runtime
                    else (void)hipGetLastError();
                }
            }
            if ((rc = ensure(rcv, (void **)&rcv->stage, &rcv->stage_bytes, bytes)) != RT_OK) return fail_from(k, rc);
            hipError_t e = hipMemcpyPeerAsync(rcv->stage, rcv->device, snd->accum, snd->device, bytes, rcv->xstream);
            if (e == hipSuccess) e = launch_add_into(rcv->accum, rcv->stage, n_px * 3, rcv->xstream);
            if (e != hipSuccess) return fail_from(k, set_err(rcv, RT_ERR_HIP, std::string("rt_render_multi exchange: ") + hipGetErrorString(e)));
        }
        for (int k = 0; k + stride < n_devices; k += 2 * stride) { // the next level reads what this one wrote
            rt_ctx *rcv = ctxs[k];
            if (hipSetDevice(rcv->device) != hipSuccess || hipStreamSynchronize(rcv->xstream) != hipSuccess)
                return fail_from(k, set_err(rcv, RT_ERR_HIP, "rt_render_multi exchange: stream synchronisation failed"));
        }
    }
    HIP_TRY(root, hipSetDevice(root->device));
    HIP_TRY(root, hipDeviceSynchronize());
    if (stats) *stats = total;
    return finish_to_host(root, cam, accum_rgb, rgb8);
}

} // extern "C"
