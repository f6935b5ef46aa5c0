// Production precision (float) instantiation of the path-tracing kernels + the precision-independent kernels.
#include "rt_launch_impl.h"

namespace rtamd {

// Sum the per-chunk partial buffers (fixed order -> run-to-run deterministic) into the caller's accumulation buffer.
__global__ void reduce_chunks_kernel(const float *__restrict__ partial, float *__restrict__ out, size_t n, int n_chunks) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double s = 0;
    for (int c = 0; c < n_chunks; c++) s += (double)partial[(size_t)c * n + k];
    out[k] = (float)s;
}

// write_color (color.h:20-45) on scale * sum.
__global__ void tonemap_kernel(const float *__restrict__ accum, double scale, size_t n, uint8_t *__restrict__ rgb) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double x = scale * (double)accum[k];
    if (x != x) x = 0.0;
    x = (x > 0) ? sqrt(x) : 0.0;
    if (x < 0.000) x = 0.000;
    if (x > 0.999) x = 0.999;
    rgb[k] = (uint8_t)int(256 * x);
}

// rt_render_multi's exchange step: a peer device's sums, copied next to the root's, are added onto them.
__global__ void add_into_kernel(float *__restrict__ dst, const float *__restrict__ src, size_t n) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) dst[k] += src[k];
}

// The float kernels' partial frames: 64-bit fixed-point cells (unit 2^-24), summed exactly; a NaN-flagged cell in any chunk
// makes the pixel NaN, as one NaN sample does in the reference's sum.
__global__ void reduce_chunks_i64_kernel(const long long *__restrict__ partial, float *__restrict__ out, size_t n, int n_chunks) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    long long s = 0;
    bool nan = false;
    for (int c = 0; c < n_chunks; c++) {
        const long long v = partial[(size_t)c * n + k];
        nan = nan || v == I64Out::nan_cell || v < 0; // (v < 0: the NaN bit of the shared frame, I64Out::nan_bit)
        s += v;
    }
    out[k] = nan ? __builtin_nanf("") : (float)((double)s * (1.0 / 16777216.0));
}

template hipError_t launch_render<float>(const SceneT<float> &, const CameraT<float> &, const RenderLaunch &, void *,
                                         unsigned long long *, hipStream_t);
template hipError_t launch_trace<float>(const SceneT<float> &, const rt_ray *, const double *, size_t, rt_hit *, uint32_t,
                                        uint32_t, int, hipStream_t);

hipError_t launch_reduce_chunks(const void *partial, bool cells_i64, float *out, size_t n, int n_chunks, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (cells_i64) hipLaunchKernelGGL(reduce_chunks_i64_kernel, grid, block, 0, stream, (const long long *)partial, out, n, n_chunks);
    else hipLaunchKernelGGL(reduce_chunks_kernel, grid, block, 0, stream, (const float *)partial, out, n, n_chunks);
    return hipGetLastError();
}

hipError_t launch_add_into(float *dst, const float *src, size_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(add_into_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dst, src, n);
    return hipGetLastError();
}

hipError_t launch_tonemap(const float *accum, double scale, size_t n, uint8_t *rgb, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(tonemap_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, accum, scale, n, rgb);
    return hipGetLastError();
}

} // namespace rtamd
