// rt_launch_impl.h -- shared body of the per-precision launcher translation units.
#ifndef RT_LAUNCH_IMPL_H
#define RT_LAUNCH_IMPL_H
#include <cstdlib>
#include "rt_kernels.h"
#include "rt_launch.h"

namespace rtamd {
static_assert(RT_MAX_SCHED_CHUNKS_HOST == RT_MAX_SCHED_CHUNKS, "chunk schedule capacity");

// A kernel that asks for more than 64 KiB of dynamic LDS must say so once.
template <class K> hipError_t allow_lds(K kernel, int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

template <class R>
hipError_t launch_render(const SceneT<R> &sc, const CameraT<R> &cam, const RenderLaunch &rl, void *out,
                         unsigned long long *counters, hipStream_t stream) {
    using C = LaunchCfg<R>;
    RenderArgs a;
    a.k0 = rl.k0;
    a.k1 = rl.k1;
    a.s_begin = rl.s_begin;
    a.s_end = rl.s_end;
    a.chunk_size = rl.chunk_size;
    a.n_chunks = rl.n_chunks;
    a.out_i64 = sizeof(R) == 4 ? rl.out_i64 : 0;
    // A stream of a few records (cornell_box: 14, no BOX among them) is walked by all lanes in step: the shading trips are
    // what costs, and fuller ones pay (cornell_box 600 x 600 @1000: 83.6 ms at 40 waiting lanes, 79.2 at 52, 76.9 at 56,
    // 76.8 at 60; every larger scene is slower above 44: first_scene 120 -> 154 at 56, final_scene 562 -> 657).
    // (a stream read from global memory: mesh_scene 314 ms at 40, 308-309 at 28 ... 36)
    a.shade_t = sc.n_units <= 64 ? 56 : (sc.lds_units != sc.n_units ? 32 : RT_SHADE_T);
    if (const char *e = dev_knob("RT_AMD_SHADE_T")) a.shade_t = std::atoi(e); // development knob
    for (int k = 0; k <= RT_MAX_SCHED_CHUNKS; k++) a.chunk_start[k] = rl.chunk_start[k];
    if (rl.chunk_size == 0 && (rl.n_chunks < 1 || rl.n_chunks > RT_MAX_SCHED_CHUNKS)) return hipErrorInvalidValue;
    a.tiles_x = (cam.W + 7) / 8;
    a.tiles_y = (cam.H + LdsLayout<R>::tile_h - 1) / LdsLayout<R>::tile_h;
    const size_t lds = (size_t)sc.lds_units * sizeof(UnitT<R>) + (sc.tab_in_lds ? (size_t)sc.tab_bytes : 0);
    if (sc.lds_units < 0 || sc.lds_units > sc.n_units || lds > (size_t)C::lds_bytes || (sc.tab_bytes & 15)) return hipErrorInvalidValue;
    const long long items = (long long)a.tiles_x * a.tiles_y * rl.n_chunks;
    const int waves_per_block = C::block / 64;
    long long want = (items + waves_per_block - 1) / waves_per_block;
    long long cap = (long long)rl.n_cus * C::blocks_per_cu;
    dim3 grid((unsigned)(want < cap ? (want < 1 ? 1 : want) : cap), 1, 1), block(C::block, 1, 1);
    hipError_t e;
    auto go = [&](auto k) {
        hipError_t e2 = allow_lds(k, C::lds_total);
        if (e2 != hipSuccess) return e2;
        hipLaunchKernelGGL(k, grid, block, lds + C::stream_off, stream, sc, cam, a, out, counters);
        return hipSuccess;
    };
    // (the kernels that can walk a constant_medium's boundary sub-stream are separate instantiations: that walk keeps a
    // second traversal state alive, which the common scenes' kernels should not pay for in registers)
    const bool all = sc.lds_units == sc.n_units, tab = sc.tab_in_lds != 0;
    if (sc.has_gen_medium) {
        if (all && tab) e = go(render_kernel<R, true, true, true, C::block>);
        else if (all) e = go(render_kernel<R, true, false, true, C::block>);
        else if (tab) e = go(render_kernel<R, false, true, true, C::block>);
        else e = go(render_kernel<R, false, false, true, C::block>);
    } else {
        if (all && tab) e = go(render_kernel<R, true, true, false, C::block>);
        else if (all) e = go(render_kernel<R, true, false, false, C::block>);
        else if (tab) e = go(render_kernel<R, false, true, false, C::block>);
        else e = go(render_kernel<R, false, false, false, C::block>);
    }
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

template <class R>
hipError_t launch_trace(const SceneT<R> &sc, const rt_ray *d_rays, const double *d_xi, size_t n, rt_hit *d_out,
                        uint32_t k0, uint32_t k1, int n_cus, hipStream_t stream) {
    using C = LaunchCfg<R>;
    if (n == 0) return hipSuccess;
    TraceArgs a;
    a.k0 = k0;
    a.k1 = k1;
    a.has_xi = d_xi != nullptr;
    const size_t lds = (size_t)sc.lds_units * sizeof(UnitT<R>) + (sc.tab_in_lds ? (size_t)sc.tab_bytes : 0);
    if (sc.lds_units < 0 || sc.lds_units > sc.n_units || lds > (size_t)C::lds_bytes || (sc.tab_bytes & 15)) return hipErrorInvalidValue;
    size_t want = (n + C::block - 1) / C::block, cap = (size_t)n_cus * C::blocks_per_cu;
    dim3 grid((unsigned)(want < cap ? want : cap), 1, 1), block(C::block, 1, 1);
    hipError_t e;
    auto go = [&](auto k) {
        hipError_t e2 = allow_lds(k, C::lds_total);
        if (e2 != hipSuccess) return e2;
        hipLaunchKernelGGL(k, grid, block, lds + C::stream_off, stream, sc, d_rays, d_xi, n, d_out, a);
        return hipSuccess;
    };
    const bool all = sc.lds_units == sc.n_units, tab = sc.tab_in_lds != 0;
    if (sc.has_gen_medium) {
        if (all && tab) e = go(trace_kernel<R, true, true, true, C::block>);
        else if (all) e = go(trace_kernel<R, true, false, true, C::block>);
        else if (tab) e = go(trace_kernel<R, false, true, true, C::block>);
        else e = go(trace_kernel<R, false, false, true, C::block>);
    } else {
        if (all && tab) e = go(trace_kernel<R, true, true, false, C::block>);
        else if (all) e = go(trace_kernel<R, true, false, false, C::block>);
        else if (tab) e = go(trace_kernel<R, false, true, false, C::block>);
        else e = go(trace_kernel<R, false, false, false, C::block>);
    }
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

} // namespace rtamd
#endif
