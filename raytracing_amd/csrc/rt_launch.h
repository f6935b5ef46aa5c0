// rt_launch.h -- host-callable launchers of the kernels in rt_kernels.h (one translation unit per precision).
#ifndef RT_LAUNCH_H
#define RT_LAUNCH_H
#include <hip/hip_runtime.h>

#include "../../include/rt_amd.h"
#include "rt_scene.h"

namespace rtamd {

struct RenderLaunch {
    uint32_t k0, k1;
    int32_t s_begin, s_end, chunk_size, n_chunks;
    int32_t chunk_start[25]; // used when chunk_size == 0 (n_chunks <= 24)
    int32_t n_cus;
};

// LDS the kernels of precision R may use for the scene (record stream + shading tables), and their workgroup size.
//   f32: one 1024-thread workgroup per CU owns (almost) the whole 160 KiB -> 16 waves/CU, <= 128 VGPRs
//   f64: the parity path; one 512-thread workgroup per CU (<= 256 VGPRs), 128 KiB
// The render kernel adds accum_bytes behind the scene: per wave 64 pixels x RGB of path sums.
template <class R> struct LaunchCfg;
template <> struct LaunchCfg<float> {
    static constexpr int block = 1024;
    static constexpr int accum_bytes = block * 3 * (int)sizeof(float);
    static constexpr int lds_bytes = 160 * 1024 - 512 - accum_bytes;
    static constexpr int blocks_per_cu = 1;
};
template <> struct LaunchCfg<double> {
    static constexpr int block = 512;
    static constexpr int accum_bytes = block * 3 * (int)sizeof(double);
    static constexpr int lds_bytes = 128 * 1024;
    static constexpr int blocks_per_cu = 1;
};

// counters: device pointer to two 64-bit words, both zero at launch: [0] rays traced (out), [1] work-item cursor
template <class R>
hipError_t launch_render(const SceneT<R> &sc, const CameraT<R> &cam, const RenderLaunch &rl, float *out,
                         unsigned long long *counters, hipStream_t stream);
template <class R>
hipError_t launch_trace(const SceneT<R> &sc, const rt_ray *d_rays, const double *d_xi, size_t n, rt_hit *d_out,
                        uint32_t k0, uint32_t k1, int n_cus, hipStream_t stream);
hipError_t launch_reduce_chunks(const float *partial, float *out, size_t n, int n_chunks, hipStream_t stream);
hipError_t launch_tonemap(const float *accum, double scale, size_t n, uint8_t *rgb, hipStream_t stream);

} // namespace rtamd
#endif
