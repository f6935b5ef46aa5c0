// rt_launch.h -- host-callable launchers of the kernels in rt_kernels.h (one translation unit per precision).
#ifndef RT_LAUNCH_H
#define RT_LAUNCH_H
#include <hip/hip_runtime.h>

#include "../../include/rt_amd.h"
#include "rt_scene.h"

namespace rtamd {

enum : int { RT_MAX_SCHED_CHUNKS_HOST = 24 }; // == RT_MAX_SCHED_CHUNKS of rt_kernels.h (static_assert in rt_launch_impl.h)
struct RenderLaunch {
    uint32_t k0, k1;
    int32_t s_begin, s_end, chunk_size, n_chunks;
    int32_t out_i64; // float kernels only: 1 = 64-bit fixed-point cells, a partial frame per chunk; 2 = one shared frame, atomic adds
    int32_t chunk_start[25]; // used when chunk_size == 0 (n_chunks <= 24)
    int32_t n_cus;
};

// Workgroup size and LDS budget of the kernels of precision R: see LdsLayout (rt_scene.h).  lds_bytes is what the
// scene (stream + shading tables) may take; every launch asks for stream_off + the scene's bytes.
template <class R> struct LaunchCfg {
    static constexpr int block = LdsLayout<R>::block;
    static constexpr int lds_bytes = LdsLayout<R>::scene_bytes;
    static constexpr int lds_total = LdsLayout<R>::total_bytes;
    static constexpr int stream_off = LdsLayout<R>::stream_off;
    static constexpr int blocks_per_cu = 1;
};

// counters: device pointer to two 64-bit words, both zero at launch: [0] rays traced (out), [1] work-item cursor
template <class R>
hipError_t launch_render(const SceneT<R> &sc, const CameraT<R> &cam, const RenderLaunch &rl, void *out,
                         unsigned long long *counters, hipStream_t stream);
template <class R>
hipError_t launch_trace(const SceneT<R> &sc, const rt_ray *d_rays, const double *d_xi, size_t n, rt_hit *d_out,
                        uint32_t k0, uint32_t k1, int n_cus, hipStream_t stream);
// partial frames -> the caller's float sums.  cells_i64: the partial frames are the float kernels' 64-bit fixed-point cells
// (summed exactly, so the result does not depend on how the samples were cut into chunks), else floats (double kernels).
hipError_t launch_reduce_chunks(const void *partial, bool cells_i64, float *out, size_t n, int n_chunks, hipStream_t stream);
hipError_t launch_tonemap(const float *accum, double scale, size_t n, uint8_t *rgb, hipStream_t stream);
hipError_t launch_add_into(float *dst, const float *src, size_t n, hipStream_t stream); // dst[k] += src[k]

} // namespace rtamd
#endif
