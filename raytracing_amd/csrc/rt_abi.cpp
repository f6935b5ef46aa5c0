// rt_abi.cpp -- implementation of the C ABI declared in include/rt_amd.h (host side, HIP runtime API).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_flatten.h"
#include "rt_launch.h"
#include "rt_scene.h"

using namespace rtamd;

struct rt_ctx {
    int device = 0;
    std::string err;
    float *partial = nullptr; // per-chunk accumulation buffers [n_chunks][H][W][3]
    size_t partial_bytes = 0;
    float *accum = nullptr; // used by the host-buffer convenience entry point rt_render
    size_t accum_bytes = 0;
    uint8_t *rgb = nullptr;
    size_t rgb_bytes = 0;
    unsigned long long *counter = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int n_cus = 256;
    // development knobs (environment RT_AMD_LDS_UNITS / RT_AMD_CHUNKS); -1 / 0 = library default
    int lds_limit_units = -1;
    int min_chunk = 0, chunk_divisor = 0, max_chunk = 0;
};

template <class R> struct DeviceScene {
    void *blob = nullptr;
    size_t bytes = 0;
    SceneT<R> sc{};
    int n_fused_boxes = 0;
};

struct rt_scene {
    rt_ctx *ctx = nullptr;
    DeviceScene<float> f32;
    DeviceScene<double> f64;
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

template <class R> int upload_one(rt_ctx *ctx, const rt_scene_desc *desc, DeviceScene<R> &ds) {
    Flat<R> f;
    std::string err;
    Flattener<R> fl(*desc, f, err, /*fuse_boxes=*/sizeof(R) == 4);
    int rc = fl.run();
    if (rc) return set_err(ctx, rc, err);
    BlobBuilder b;
    size_t o_stream = b.add(f.stream), o_cold = b.add(f.cold);
    size_t o_bytes = b.add_bytes(desc->image_bytes, (size_t)desc->n_image_bytes);
    // the shading tables: one contiguous range (staged into LDS as a whole when it fits)
    size_t o_xf = b.add(f.xforms), o_mats = b.add(f.mats), o_texs = b.add(f.texs);
    size_t o_img = b.add(f.images), o_perlin = b.add(f.perlins), o_lights = b.add(f.lights);
    b.host.resize((b.host.size() + 255) & ~size_t(255));
    const size_t tab_bytes = b.host.size() - o_xf;
    ds.bytes = b.host.size();
    HIP_TRY(ctx, hipMalloc(&ds.blob, ds.bytes));
    HIP_TRY(ctx, hipMemcpy(ds.blob, b.host.data(), ds.bytes, hipMemcpyHostToDevice));
    uint8_t *base = (uint8_t *)ds.blob;
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
    s.n_units = (int)f.stream.size();
    int cap = LaunchCfg<R>::lds_bytes / (int)sizeof(UnitT<R>);
    if (ctx->lds_limit_units >= 0 && ctx->lds_limit_units < cap) cap = ctx->lds_limit_units;
    s.lds_units = s.n_units < cap ? s.n_units : cap;
    s.tab = base + o_xf;
    s.tab_bytes = (int32_t)tab_bytes;
    s.tab_in_lds = (s.lds_units == s.n_units && ctx->lds_limit_units < 0 &&
                    (size_t)s.n_units * sizeof(UnitT<R>) + tab_bytes <= (size_t)LaunchCfg<R>::lds_bytes) ? 1 : 0;
    s.n_lights = (int)f.lights.size();
    s.lights_is_list = f.lights_is_list;
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
    if (hipSetDevice(device) != hipSuccess || hipMalloc((void **)&ctx->counter, 40 * sizeof(unsigned long long)) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        delete ctx;
        return RT_ERR_HIP;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->n_cus = prop.multiProcessorCount;
    if (const char *e1 = std::getenv("RT_AMD_LDS_UNITS")) ctx->lds_limit_units = std::atoi(e1);
    if (const char *e5 = std::getenv("RT_AMD_CHUNKS")) std::sscanf(e5, "%d,%d,%d", &ctx->min_chunk, &ctx->chunk_divisor, &ctx->max_chunk);
    *out = ctx;
    return RT_OK;
}

void rt_shutdown(rt_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->partial) (void)hipFree(ctx->partial);
    if (ctx->accum) (void)hipFree(ctx->accum);
    if (ctx->rgb) (void)hipFree(ctx->rgb);
    if (ctx->counter) (void)hipFree(ctx->counter);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    delete ctx;
}

const char *rt_last_error(const rt_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int rt_scene_upload(rt_ctx *ctx, const rt_scene_desc *desc, rt_scene **out) {
    if (!ctx || !desc || !out) return RT_ERR_INVALID;
    *out = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    rt_scene *s = new (std::nothrow) rt_scene;
    if (!s) return set_err(ctx, RT_ERR_NOMEM, "out of host memory");
    s->ctx = ctx;
    s->n_nodes_desc = desc->n_nodes;
    int rc = upload_one<float>(ctx, desc, s->f32);
    if (rc == RT_OK) rc = upload_one<double>(ctx, desc, s->f64);
    if (rc != RT_OK) {
        rt_scene_free(s);
        return rc;
    }
    *out = s;
    return RT_OK;
}

void rt_scene_free(rt_scene *scene) {
    if (!scene) return;
    if (scene->ctx) (void)hipSetDevice(scene->ctx->device);
    if (scene->f32.blob) (void)hipFree(scene->f32.blob);
    if (scene->f64.blob) (void)hipFree(scene->f64.blob);
    delete scene;
}

int rt_scene_info(const rt_scene *scene, int32_t *out /* [4]: flat nodes, lights, f32 bytes, f64 bytes */) {
    if (!scene || !out) return RT_ERR_INVALID;
    out[0] = scene->f32.sc.n_units;
    out[1] = scene->f32.sc.n_lights;
    out[2] = (int32_t)scene->f32.bytes;
    out[3] = (int32_t)scene->f64.bytes;
    return RT_OK;
}

int rt_scene_flatten_info(const rt_scene_desc *desc, int32_t *out) {
    if (!desc || !out) return RT_ERR_INVALID;
    std::string err;
    Flat<float> f32;
    int rc = Flattener<float>(*desc, f32, err, true).run();
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
    if (cam->image_width <= 0 || cam->image_height <= 0 || cam->sqrt_spp <= 0 || cam->max_depth < 0)
        return set_err(ctx, RT_ERR_INVALID, "rt_render_device: bad camera");
    if ((int64_t)cam->image_width * cam->image_height > (int64_t)1 << 28)
        return set_err(ctx, RT_ERR_INVALID, "rt_render_device: image too large");
    if (params->precision != RT_PREC_F32 && params->precision != RT_PREC_F64)
        return set_err(ctx, RT_ERR_INVALID, "bad precision");
    const int n_strata = cam->sqrt_spp * cam->sqrt_spp;
    int s_begin = params->s_begin, s_end = params->s_end <= 0 ? n_strata : params->s_end;
    if (s_begin < 0 || s_end > n_strata || s_begin > s_end) return set_err(ctx, RT_ERR_INVALID, "bad sample range");
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n_px = (size_t)cam->image_width * cam->image_height;
    const int n_samples = s_end - s_begin;
    if (n_samples == 0) {
        HIP_TRY(ctx, hipMemsetAsync(d_accum, 0, n_px * 3 * sizeof(float), stream));
        if (stats) *stats = rt_stats{0, 0, 0.0};
        return RT_OK;
    }
    // Work decomposition: a work item = (8 x RT_TILE_H pixel tile, chunk of the sample range); waves pull items from a cursor in
    // chunk-major order.  Default: a DECREASING-size schedule (each chunk takes a half / a third of what is left, down to a minimum),
    // so the bulk of the samples runs in long items (lanes regenerate paths inside an item; a short item wastes lanes while
    // its last paths finish) and the kernel ends on short ones (the end-of-kernel imbalance is one short item per wave).
    // params->sample_chunks > 0 forces that many equal chunks instead.
    const int tiles = ((cam->image_width + 7) / 8) * ((cam->image_height + RT_TILE_H - 1) / RT_TILE_H);
    RenderLaunch rl{};
    int n_chunks = params->sample_chunks;
    const int wave_slots = ctx->n_cus * 16;
    if (n_chunks <= 0 && tiles < 2 * wave_slots) { // small image: equal chunks, enough of them to give every wave ~32 items
        n_chunks = (wave_slots * 32 + tiles - 1) / tiles;
        int max_chunks = n_samples / 24;
        if (max_chunks < 1) max_chunks = 1;
        if (n_chunks > max_chunks) n_chunks = max_chunks;
    }
    if (n_chunks > 0) {
        if (n_chunks > n_samples) n_chunks = n_samples;
        if (n_chunks > 65535) n_chunks = 65535;
        rl.chunk_size = (n_samples + n_chunks - 1) / n_chunks;
        n_chunks = (n_samples + rl.chunk_size - 1) / rl.chunk_size;
    } else {
        const int min_chunk = ctx->min_chunk > 0 ? ctx->min_chunk : (n_samples >= 360 ? 32 : 16);
        const int divisor = ctx->chunk_divisor > 1 ? ctx->chunk_divisor : (n_samples >= 360 ? 3 : 2); // measured on final_scene 1080p
        rl.chunk_size = 0;
        n_chunks = 0;
        int at = s_begin;
        rl.chunk_start[0] = at;
        while (at < s_end) {
            int left = s_end - at, take = left / divisor;
            if (ctx->max_chunk > 0 && take > ctx->max_chunk) take = ctx->max_chunk;
            if (take < min_chunk) take = min_chunk;
            if (take > left || left - take < min_chunk / 2 || n_chunks == 23) take = left;
            at += take;
            rl.chunk_start[++n_chunks] = at;
        }
    }
    float *target = d_accum;
    if (n_chunks > 1) {
        int rc = ensure(ctx, (void **)&ctx->partial, &ctx->partial_bytes, (size_t)n_chunks * n_px * 3 * sizeof(float));
        if (rc) return rc;
        target = ctx->partial;
    }
    rl.k0 = (uint32_t)params->seed;
    rl.k1 = (uint32_t)(params->seed >> 32);
    rl.s_begin = s_begin;
    rl.s_end = s_end;
    rl.n_chunks = n_chunks;
    rl.n_cus = ctx->n_cus;
    HIP_TRY(ctx, hipMemsetAsync(ctx->counter, 0, 40 * sizeof(unsigned long long), stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->counter + 37, 0xff, sizeof(unsigned long long), stream)); // [37]: a minimum (diagnostic builds)
    if (stats) HIP_TRY(ctx, hipEventRecord(ctx->ev0, stream));
    if (params->precision == RT_PREC_F64)
        HIP_TRY(ctx, launch_render<double>(scene->f64.sc, convert_camera<double>(*cam), rl, target, ctx->counter, stream));
    else
        HIP_TRY(ctx, launch_render<float>(scene->f32.sc, convert_camera<float>(*cam), rl, target, ctx->counter, stream));
    if (n_chunks > 1) HIP_TRY(ctx, launch_reduce_chunks(ctx->partial, d_accum, n_px * 3, n_chunks, stream));
    if (stats) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev1, stream));
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
        float ms = 0;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        unsigned long long rays = 0;
        HIP_TRY(ctx, hipMemcpy(&rays, ctx->counter, sizeof(rays), hipMemcpyDeviceToHost));
        if (std::getenv("RT_AMD_PROFILE_ACTIONS")) { // diagnostic builds (-DRT_PROFILE_ACTIONS) fill counters[4..13]
            unsigned long long c[40];
            HIP_TRY(ctx, hipMemcpy(c, ctx->counter, sizeof(c), hipMemcpyDeviceToHost));
            const char *nm[6] = {"box", "sphere", "aabox", "medium", "other", "shade"};
            for (int k = 0; k < 6; k++)
                std::fprintf(stderr, "[rt_amd profile] %-6s cycles %.4g (%.1f%% of wave time) trips %.4g cyc/trip %.1f lanes/trip %.1f\n",
                             nm[k], (double)c[4 + 3 * k], 100.0 * (double)c[4 + 3 * k] / (double)(c[3] ? c[3] : 1), (double)c[5 + 3 * k],
                             (double)c[4 + 3 * k] / (double)(c[5 + 3 * k] ? c[5 + 3 * k] : 1),
                             (double)c[6 + 3 * k] / (double)(c[5 + 3 * k] ? c[5 + 3 * k] : 1));
        }
        if (std::getenv("RT_AMD_PROFILE_ACTIONS")) {
            unsigned long long c[40];
            HIP_TRY(ctx, hipMemcpy(c, ctx->counter, sizeof(c), hipMemcpyDeviceToHost));
            const char *nm[9] = {"rng begin", "finalize", "material+emit", "metal", "dielectric", "texture", "sample dir", "pdfs+weight", "tail (trav_begin / next_path)"};
            for (int k = 0; k < 9; k++)
                std::fprintf(stderr, "[rt_amd profile]   shade/%-30s %.1f%% of wave time\n", nm[k], 100.0 * (double)c[24 + k] / (double)(c[3] ? c[3] : 1));
            std::fprintf(stderr, "[rt_amd profile] per trip: %.1f lanes idle (item tail), %.1f lanes waiting for shading, of 64; %.1f%% of trips after the item's last unit was dealt\n",
                         (double)c[33] / (double)(c[35] ? c[35] : 1), (double)c[34] / (double)(c[35] ? c[35] : 1), 100.0 * (double)c[36] / (double)(c[35] ? c[35] : 1));
        }
        if (std::getenv("RT_AMD_PROFILE_TAIL")) { // builds with -DRT_PROFILE_TAIL: wave end times
            unsigned long long c[40];
            HIP_TRY(ctx, hipMemcpy(c, ctx->counter, sizeof(c), hipMemcpyDeviceToHost));
            const double span = (double)(c[38] - c[37]), n_waves = (double)ctx->n_cus * 16.0;
            std::fprintf(stderr, "[rt_amd profile] waves busy on average %.1f%% of the kernel's span (first start to last end); wave run time min %.1f%% max %.1f%% of the span\n",
                         span > 0 ? 100.0 * (double)c[39] / n_waves / span : 0.0, span > 0 ? 100.0 * (double)(~c[35]) / span : 0.0,
                         span > 0 ? 100.0 * (double)c[36] / span : 0.0);
        }
        stats->samples = (uint64_t)n_px * (uint64_t)n_samples;
        stats->rays = rays;
        stats->kernel_ms = ms;
    }
    return RT_OK;
}

int rt_tonemap_device(rt_ctx *ctx, const float *d_accum, double scale, size_t n_pixels, uint8_t *d_rgb8, void *stream) {
    if (!ctx || !d_accum || !d_rgb8) return set_err(ctx, RT_ERR_INVALID, "rt_tonemap_device: null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_tonemap(d_accum, scale, n_pixels * 3, d_rgb8, (hipStream_t)stream));
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
    if (accum_rgb) HIP_TRY(ctx, hipMemcpy(accum_rgb, ctx->accum, n_px * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (rgb8) {
        rc = ensure(ctx, (void **)&ctx->rgb, &ctx->rgb_bytes, n_px * 3);
        if (rc) return rc;
        const double scale = 1.0 / ((double)cam->sqrt_spp * cam->sqrt_spp); // camera.h:215 pixel_samples_scale
        rc = rt_tonemap_device(ctx, ctx->accum, scale, n_px, ctx->rgb, nullptr);
        if (rc) return rc;
        HIP_TRY(ctx, hipDeviceSynchronize());
        HIP_TRY(ctx, hipMemcpy(rgb8, ctx->rgb, n_px * 3, hipMemcpyDeviceToHost));
    }
    return RT_OK;
}

} // extern "C"
