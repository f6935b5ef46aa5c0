// rt_launch_impl.h -- shared body of the per-precision launcher translation units.
#ifndef RT_LAUNCH_IMPL_H
#define RT_LAUNCH_IMPL_H
#include "rt_kernels.h"
#include "rt_launch.h"

namespace rtamd {

template <class R>
hipError_t launch_render(const SceneT<R> &sc, const CameraT<R> &cam, const RenderLaunch &rl, float *out,
                         unsigned long long *ray_counter, hipStream_t stream) {
    RenderArgs a;
    a.k0 = rl.k0;
    a.k1 = rl.k1;
    a.s_begin = rl.s_begin;
    a.s_end = rl.s_end;
    a.chunk_size = rl.chunk_size;
    a.n_chunks = rl.n_chunks;
    a.tiles_x = (cam.W + 7) / 8;
    a.tiles_y = (cam.H + 7) / 8;
    const int tiles = a.tiles_x * a.tiles_y;
    dim3 grid((tiles + 3) / 4, rl.n_chunks, 1), block(256, 1, 1);
    hipLaunchKernelGGL(render_kernel<R>, grid, block, 0, stream, sc, cam, a, out, ray_counter);
    return hipGetLastError();
}

template <class R>
hipError_t launch_trace(const SceneT<R> &sc, const rt_ray *d_rays, const double *d_xi, size_t n, rt_hit *d_out,
                        uint32_t k0, uint32_t k1, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    TraceArgs a;
    a.k0 = k0;
    a.k1 = k1;
    a.has_xi = d_xi != nullptr;
    dim3 grid((unsigned)((n + 255) / 256), 1, 1), block(256, 1, 1);
    hipLaunchKernelGGL(trace_kernel<R>, grid, block, 0, stream, sc, d_rays, d_xi, n, d_out, a);
    return hipGetLastError();
}

} // namespace rtamd
#endif
