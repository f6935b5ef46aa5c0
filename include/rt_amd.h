/*
 * rt_amd.h -- C ABI of the MI355X (gfx950) path-tracing hot path.
 *
 * The reference (Jiaqi1song/RayTracing) has no FFI/plugin layer: its boundary is the C++ source-level
 * scene API plus one call, camera::render(world, lights) (src/cpu/camera.h:157).  The host-side mirror
 * of that API (include/rtamd/) flattens the object graph into the plain structs below and calls the
 * entry points declared here.  Everything is plain C: fixed-width scalars, pointers and sizes; no C++
 * or torch types cross this boundary.
 *
 * Conventions: every call returns RT_OK (0) or a negative status and never exits the process
 * (the reference exit()s: color.h:76-79).  The caller owns every host buffer it passes; the library owns
 * device memory behind rt_scene / rt_ctx.  One rt_ctx is used from one host thread at a time.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_AMD_ABI_VERSION 3 /* 2: rt_render_params.s_end < 0 (not <= 0) means "all strata"; rt_render_multi.  3: rt_scene_flatten_info / _stream take
                                 reference_order; rt_scene_tune */

/* ---- status codes ---------------------------------------------------------------------------- */
enum {
    RT_OK = 0,
    RT_ERR_INVALID = -1,     /* bad argument / malformed scene description              */
    RT_ERR_HIP = -2,         /* a HIP runtime call failed (see rt_last_error)            */
    RT_ERR_UNSUPPORTED = -3, /* legal reference scene graph the device path cannot take  */
    RT_ERR_NOMEM = -4
};

/* ---- flat scene description -------------------------------------------------------------------
 * A scene is the reference's hittable graph written as an array of nodes.  Node ids are the
 * "primitive ids" reported by rt_trace_batch.  All geometry is IEEE double exactly as the reference
 * constructors compute it, so that the f64 parity path reproduces reference arithmetic bit for bit.
 */
enum rt_node_kind {
    RT_NODE_BVH = 0,       /* bvh_node            src/cpu/bvh.h:11      a=left node, b=right node            */
    RT_NODE_LIST = 1,      /* hittable_list       hittable_list.h:10    a=first slot in child_index, b=count  */
    RT_NODE_TRANSLATE = 2, /* translate           hittable.h:45         a=child node, p[0..2]=offset          */
    RT_NODE_ROTATE_Y = 3,  /* rotate_y            hittable.h:76         a=child node, p[0]=sin, p[1]=cos      */
    RT_NODE_MEDIUM = 4,    /* constant_medium     constant_medium.h:9   a=boundary node, b=material (isotropic), p[0]=-1/density */
    RT_NODE_SPHERE = 5,    /* sphere              sphere.h:8            a=index into spheres                  */
    RT_NODE_QUAD = 6,      /* quad                quad.h:8              a=index into quads                    */
    RT_NODE_TRIANGLE = 7   /* triangle            triangle.h:13         a=index into triangles                */
};

typedef struct rt_node {
    int32_t kind;
    int32_t a;
    int32_t b;
    int32_t reserved;
    double bbox[6]; /* bounding_box(): x.min,x.max,y.min,y.max,z.min,z.max */
    double p[4];
} rt_node; /* 96 bytes */

typedef struct rt_sphere { /* sphere.h:11-27: center is a ray (center1, center2-center1) */
    double center[3];
    double dcenter[3];
    double radius;
    int32_t mat; /* -1 = null material (legal for light shapes, cpuRenderer.cpp:133-135) */
    int32_t reserved;
} rt_sphere; /* 64 bytes */

typedef struct rt_quad { /* quad.h:10-21; derived fields stored as the ctor computed them */
    double Q[3];
    double u[3];
    double v[3];
    double w[3];
    double normal[3];
    double D;
    double area;
    int32_t mat;
    int32_t reserved;
} rt_quad; /* 144 bytes */

typedef struct rt_triangle { /* triangle.h:15-21,82-86 */
    double v0[3];
    double v1[3];
    double v2[3];
    double normal[3];
    int32_t mat;
    int32_t cull; /* backCulling flag (build_mesh passes false: triangle.h:108) */
} rt_triangle; /* 104 bytes */

enum rt_material_kind {
    RT_MAT_LAMBERTIAN = 0,    /* material.h:39  tex                     */
    RT_MAT_METAL = 1,         /* material.h:62  albedo, fuzz (<=1)      */
    RT_MAT_DIELECTRIC = 2,    /* material.h:84  ior                     */
    RT_MAT_DIFFUSE_LIGHT = 3, /* material.h:124 tex                     */
    RT_MAT_ISOTROPIC = 4      /* material.h:141 tex                     */
};

typedef struct rt_material {
    int32_t kind;
    int32_t tex; /* texture id or -1 */
    double albedo[3];
    double fuzz;
    double ior;
} rt_material; /* 48 bytes */

enum rt_texture_kind {
    RT_TEX_SOLID = 0,   /* texture.h:16  color                                            */
    RT_TEX_CHECKER = 1, /* texture.h:31  scale = inv_scale, a = even texture, b = odd     */
    RT_TEX_IMAGE = 2,   /* texture.h:56  a = image id (or -1: no data -> cyan)            */
    RT_TEX_NOISE = 3    /* texture.h:81  scale, a = perlin id                             */
};

typedef struct rt_texture {
    int32_t kind;
    int32_t a;
    int32_t b;
    int32_t reserved;
    double color[3];
    double scale;
} rt_texture; /* 48 bytes */

typedef struct rt_image { /* rtw_image bdata: linearised RGB8, row-major (stb_image_utils.h:95-109) */
    int32_t width;
    int32_t height;
    int64_t offset; /* byte offset into image_bytes */
} rt_image;

typedef struct rt_perlin { /* perlin.h:52-57 */
    double randvec[256][3];
    int32_t perm_x[256];
    int32_t perm_y[256];
    int32_t perm_z[256];
} rt_perlin;

typedef struct rt_scene_desc {
    int32_t abi_version; /* RT_AMD_ABI_VERSION */
    int32_t world_root;  /* node id of `world`  (camera::render arg 1) */
    int32_t lights_root; /* node id of `lights` (camera::render arg 2) */
    int32_t n_nodes;
    const rt_node *nodes;
    int32_t n_child_index;
    const int32_t *child_index;
    int32_t n_spheres;
    const rt_sphere *spheres;
    int32_t n_quads;
    const rt_quad *quads;
    int32_t n_triangles;
    const rt_triangle *triangles;
    int32_t n_materials;
    const rt_material *materials;
    int32_t n_textures;
    const rt_texture *textures;
    int32_t n_images;
    const rt_image *images;
    int64_t n_image_bytes;
    const uint8_t *image_bytes;
    int32_t n_perlins;
    const rt_perlin *perlins;
} rt_scene_desc;

/* Camera frame exactly as camera::initialize() derives it (camera.h:205-247). */
typedef struct rt_camera {
    int32_t image_width;
    int32_t image_height;
    int32_t sqrt_spp;  /* int(sqrt(samples_per_pixel)), camera.h:214 */
    int32_t max_depth;
    double center[3];
    double pixel00_loc[3];
    double pixel_delta_u[3];
    double pixel_delta_v[3];
    double defocus_disk_u[3];
    double defocus_disk_v[3];
    double defocus_angle;
    double background[3];
} rt_camera;

enum rt_precision {
    RT_PREC_F32 = 0, /* production arithmetic */
    RT_PREC_F64 = 1  /* reference arithmetic (double, no FMA contraction): the parity path */
};

#define RT_ALL_STRATA (-1)
typedef struct rt_render_params {
    uint64_t seed;      /* counter-based RNG seed; results depend on (seed,pixel,sample) only   */
    int32_t s_begin;    /* linear stratum index range [s_begin,s_end) of sqrt_spp^2 samples:     */
    int32_t s_end;      /*   s = s_j*sqrt_spp + s_i (camera.h:176-181); s_end < 0 (RT_ALL_STRATA) means  */
                        /*   sqrt_spp^2; s_begin == s_end is an EMPTY range (a rank with no strata)  */
    int32_t precision;  /* rt_precision */
    int32_t sample_chunks; /* 0 = auto; >0 forces the number of sample chunks per pixel tile     */
} rt_render_params;

typedef struct rt_ray {
    double orig[3];
    double dir[3]; /* NOT normalised (ray.h:7-30) */
    double time;
} rt_ray; /* 56 bytes */

typedef struct rt_hit {
    int32_t prim;       /* node id of the primitive (or medium) that won, -1 = miss */
    int32_t front_face;
    double t;
    double p[3];
    double normal[3];
    double u;
    double v;
} rt_hit; /* 80 bytes */

typedef struct rt_stats {
    uint64_t samples;   /* camera paths traced                                   */
    uint64_t rays;      /* ray_color invocations that traced a ray (R x samples) */
    double kernel_ms;   /* device time of the render kernels (hipEvent)          */
    uint32_t sample_chunks; /* sample chunks the launch was cut into: each wrote a partial frame (1: straight into the caller's frame) */
    uint32_t reserved;
} rt_stats;

typedef struct rt_ctx rt_ctx;
typedef struct rt_scene rt_scene;

/* Create a context on HIP device `device` (one process per GPU; multi-GPU exchange is done by the
 * caller with RCCL on the device buffers this library fills).  No reference counterpart. */
int rt_init(int device, rt_ctx **out);
void rt_shutdown(rt_ctx *ctx);
const char *rt_last_error(const rt_ctx *ctx);

/* Layout self-check for bindings: sizeof() of the named struct of this header ("rt_node", ...), -1 if unknown. */
int rt_abi_sizeof(const char *name);

/* Flatten + upload a scene: replaces the construction of the reference's in-memory graph
 * (cpuRenderer.cpp:86-401 build it; bvh_node ctor bvh.h:15-41 is run by the host API before this). */
int rt_scene_upload(rt_ctx *ctx, const rt_scene_desc *desc, rt_scene **out);
/* rt_scene_upload with the float stream's bounding-volume hierarchy built ON THE DEVICE instead of on the host: for scenes
 * that are rebuilt per frame.  The device runs the host's own build (top-down binned surface-area heuristic, one kernel
 * launch per tree level, boxes fitted bottom-up, the threaded stream written in place) with the host's arithmetic: the
 * stream is the one rt_scene_upload makes (boundary sub-streams of general media may stand in another order behind the
 * sentinel).  Replaces bvh_node's constructor (bvh.h:15-41: longest axis of the node's box, std::sort, median split; the
 * reference's CUDA tree builds on the device as well, src/cuda/bvh.h:140-188) for the production (f32) path; closest hits
 * do not depend on the hierarchy.  Every scene rt_scene_upload takes is taken (RT_ERR_UNSUPPORTED only for an empty world). */
int rt_scene_upload_device_build(rt_ctx *ctx, const rt_scene_desc *desc, rt_scene **out);
void rt_scene_free(rt_scene *scene);
/* out[4] = { threaded device nodes, lights, f32 device bytes, f64 device bytes (0 until a call first asks for
 * RT_PREC_F64: the parity stream is built lazily) }. */
int rt_scene_info(const rt_scene *scene, int32_t *out);

/* Upload-time probe (optional; no reference counterpart -- the reference builds one hierarchy, bvh.h:15-41, and lives with it).
 * Which hierarchy the float path walks fastest depends on the rays the camera's paths actually send, and the landscape over
 * the build's settings is rough (15-30 % between neighbouring bin counts).  This renders the first
 * `probe_strata` strata (<= 0: chosen so that a probe traces ~60 M samples) of `cam`'s frame with the scene's current stream
 * and with RT_TUNE_CANDIDATES - 1 other settings of the build (centroid bins 8 ... 16, child order), twice each, and leaves the
 * scene with the fastest stream (a candidate must beat the incumbent by 2 %).  Closest hits do not depend on the hierarchy.
 * ms_out (optional, RT_TUNE_CANDIDATES floats): the probe times, [0] = the stream the scene came with; chosen (optional): index
 * of the winner.  Cost: ~(flatten + upload + 2 probes) per candidate -- 0.2-0.3 s at 1080p: for scenes rendered more than once. */
#define RT_TUNE_CANDIDATES 6
int rt_scene_tune(rt_ctx *ctx, rt_scene *scene, const rt_camera *cam, int probe_strata, float *ms_out, int32_t *chosen);

/* Host-only (no GPU, no context): run the flattener on `desc` and report what rt_scene_upload would put on the device:
 * out[0] = record-stream units of the f32 (production) stream, out[1] = units of the f64 (parity) stream,
 * out[2] = box() lists fused into one AABOX record in the f32 stream (quad.h:103), out[3] = lights,
 * out[4] = translate / rotate_y instances of plain spheres emitted in world space in the f32 stream (hittable.h:45,76).
 * `out` holds 5 values.  reference_order != 0: the f32 stream as it would be with the reference's own bvh_nodes and visiting
 * order instead of the flattener's hierarchy (what the structure tests compare the production stream against).
 * Returns RT_OK or the status rt_scene_upload would return for this description.  No reference counterpart. */
int rt_scene_flatten_info(const rt_scene_desc *desc, int reference_order, int32_t *out);

/* Host-only (no GPU, no context): the f32 (production) record stream rt_scene_upload would put on the device, for
 * structure tests and traversal statistics: 16-byte units of four 32-bit words (layout: raytracing_amd/csrc/rt_scene.h).
 * Copies at most `cap_units` units into `units` (may be NULL with cap_units 0) and the description node id of the
 * record that starts at each unit (or -1) into `node_ids` (may be NULL); *n_units = size of the whole stream.
 * reference_order: as for rt_scene_flatten_info.  No reference counterpart. */
int rt_scene_flatten_stream(const rt_scene_desc *desc, int reference_order, void *units, int32_t *node_ids, size_t cap_units,
                            size_t *n_units);
/* The same view of an UPLOADED scene's f32 stream, copied back from the device (the stream of
 * rt_scene_upload_device_build exists nowhere else).  No reference counterpart. */
int rt_scene_download_stream(const rt_scene *scene, void *units, int32_t *node_ids, size_t cap_units, size_t *n_units);

/* Closest hit of each ray in [0.001, inf): replaces world.hit(r, interval(0.001, infinity), rec)
 * (camera.h:300 -> hittable_list.h:24 / bvh.h:43).  `xi` (optional, one per ray) is the uniform every
 * constant_medium on that ray consumes (constant_medium.h:41); NULL = counter-based RNG keyed by ray index.
 * Host pointers in, host pointers out. */
int rt_trace_batch(rt_ctx *ctx, const rt_scene *scene, const rt_ray *rays, const double *xi, size_t n,
                   rt_hit *out, int precision);

/* One frame (or one sample range of it): replaces the pixel/sample loop of camera::render
 * (camera.h:171-186).  `d_accum` is a DEVICE pointer to image_height*image_width*3 floats that receives the
 * UNNORMALISED per-pixel radiance sum of samples [s_begin,s_end) (overwritten, not added).
 * `stream` is a hipStream_t (NULL = default stream).  Asynchronous when stats == NULL. */
int rt_render_device(rt_ctx *ctx, const rt_scene *scene, const rt_camera *cam, const rt_render_params *params,
                     float *d_accum, void *stream, rt_stats *stats);

/* rt_render_device for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for code that runs under
 * HIP graph capture, tight serving loops and harnesses that forbid hipMalloc.  Same inputs, same output (d_accum), same
 * kernels, asynchronous on `stream`; it calls no hipMalloc / hipFree.  What rt_render_device keeps in buffers it
 * allocates -- the kernels' counters and the per-chunk partial frames -- lives in a file-scope __device__ array of
 * 64 MiB that each GPU the library runs on holds until the process ends (4 KiB of counters + the frame scratch).
 * Limit: the scratch of a render cut into several sample chunks must fit 64 MiB - 4 KiB.  At RT_PREC_F32 that is ONE frame
 * of 64-bit cells the chunks add to atomically (rt_render_device writes a partial frame per chunk instead) (image_width * image_height * 24 bytes: 1920x1080 is 49.8 MB and fits, up to
 * 2 796 032 pixels); at RT_PREC_F64 a float frame per chunk (the number of chunks is capped to what fits).  A frame too large
 * for that renders as ONE chunk straight into d_accum, whatever its size (cost: the kernel ends on long work items instead
 * of the decreasing schedule's short ones, a few percent of a frame).
 * RT_PREC_F64 only after an allocating call has built the scene's f64 stream (else RT_ERR_INVALID with a message).
 * The array is shared by every call on that GPU and keeps the last call's contents (the entry point clears what must
 * start at zero): two calls that overlap on one GPU -- other streams, other contexts on the device -- race (rt_init of
 * another context on the device does not touch the array).  No stats:
 * reading the ray counter back would synchronise. */
int rt_render_static(rt_ctx *ctx, const rt_scene *scene, const rt_camera *cam, const rt_render_params *params,
                     float *d_accum, void *stream);

/* write_color (color.h:20-45) on a summed buffer: out = int(256*clamp(sqrt(scale*sum),0,0.999)), NaN -> 0.
 * d_accum: device floats [n_pixels*3]; d_rgb8: device bytes [n_pixels*3]. */
int rt_tonemap_device(rt_ctx *ctx, const float *d_accum, double scale, size_t n_pixels, uint8_t *d_rgb8,
                      void *stream);

/* Convenience for C/C++ callers (what the host-API camera::render uses): render all of [s_begin,s_end) on
 * the context's device and return host buffers.  accum_rgb (optional) gets the unnormalised float sums,
 * rgb8 (optional) the write_color bytes scaled by 1/sqrt_spp^2. */
int rt_render(rt_ctx *ctx, const rt_scene *scene, const rt_camera *cam, const rt_render_params *params,
              float *accum_rgb, uint8_t *rgb8, rt_stats *stats);

/* One frame on SEVERAL GPUs from one process (what `rtamd_render --gpus N` and camera::render under RTAMD_DEVICES use;
 * one-process-per-GPU callers use rt_render_device + an RCCL reduce instead).  ctxs[k] / scenes[k]: one context per device
 * and the same scene uploaded through each (two contexts may share a device).  Context k traces strata
 * [s_begin + k*n/G, s_begin + (k+1)*n/G) of the n = s_end - s_begin requested -- the RNG is keyed by (seed, pixel, stratum),
 * so the image does not depend on G (the reference has no multi-GPU code) -- all devices run side by side, then
 * the float sums are added up a fixed binary tree over the contexts (stride 1, 2, 4 ...: context k takes context k + stride's
 * buffer by a direct peer copy over xGMI and adds it, the pairs of a level concurrently), ending on ctxs[0]'s device.  Outputs as rt_render; stats->kernel_ms
 * is the slowest device's.  Errors of any context are reported through rt_last_error(ctxs[0]). */
int rt_render_multi(rt_ctx *const *ctxs, const rt_scene *const *scenes, int n_devices, const rt_camera *cam,
                    const rt_render_params *params, float *accum_rgb, uint8_t *rgb8, rt_stats *stats);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
