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
};

template <class R>
hipError_t launch_render(const SceneT<R> &sc, const CameraT<R> &cam, const RenderLaunch &rl, float *out,
                         unsigned long long *ray_counter, hipStream_t stream);
template <class R>
hipError_t launch_trace(const SceneT<R> &sc, const rt_ray *d_rays, const double *d_xi, size_t n, rt_hit *d_out,
                        uint32_t k0, uint32_t k1, hipStream_t stream);
hipError_t launch_reduce_chunks(const float *partial, float *out, size_t n, int n_chunks, hipStream_t stream);
hipError_t launch_tonemap(const float *accum, double scale, size_t n, uint8_t *rgb, hipStream_t stream);

} // namespace rtamd
#endif
