// Does a VALU instruction cost less when part of the wave is masked off?  (gfx950, 4 waves per SIMD)
// The same unrolled instruction stream under exec = all 64 lanes / the low 32 / the low 16 / the even lanes / one lane.
// build: hipcc --offload-arch=gfx950 -O3 -o exec_half exec_half.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int KIND> __global__ void __launch_bounds__(1024) k(float *out, int iters, float seed, int mode) {
    const int lane = threadIdx.x & 63;
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const bool on = mode == 0 ? true : mode == 1 ? lane < 32 : mode == 2 ? lane < 16 : mode == 3 ? (lane & 1) == 0 : lane == 0;
    if (on) {
        for (int it = 0; it < iters; it++) {
            if (KIND == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %1\n v_fma_f32 %1, %1, %8, %2\n v_fma_f32 %2, %2, %8, %3\n v_fma_f32 %3, %3, %8, %4\n v_fma_f32 %4, %4, %8, %5\n v_fma_f32 %5, %5, %8, %6\n v_fma_f32 %6, %6, %8, %7\n v_fma_f32 %7, %7, %8, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));) }
            if (KIND == 1) { REP8(asm volatile("v_min_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_min_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_max_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));) }
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int KIND> void run(const char *name, float *d, int blocks) {
    const int iters = 20000;
    const char *modes[5] = {"all 64 lanes", "lanes 0-31", "lanes 0-15", "even lanes", "lane 0"};
    for (int mode = 0; mode < 5; mode++) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k<KIND><<<blocks, 1024>>>(d, 100, 1.0001f, mode);
        hipEventRecord(e0);
        k<KIND><<<blocks, 1024>>>(d, iters, 1.0001f, mode);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double per_simd = 4.0 * iters * 64.0;
        printf("%-10s %-14s %8.2f ms  %.2f clocks/instr at 2.4 GHz\n", name, modes[mode], ms, 2.4 * ms * 1e6 / per_simd);
    }
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int blocks = p.multiProcessorCount;
    float *d; hipMalloc(&d, (size_t)blocks * 1024 * 4);
    run<0>("v_fma_f32", d, blocks);
    run<1>("v_min/max", d, blocks);
    hipFree(d);
    return 0;
}
