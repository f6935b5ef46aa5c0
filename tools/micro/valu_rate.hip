// VALU issue-rate probe for gfx950: wave-instructions per SIMD per ns for a few instruction kinds, 4 waves per SIMD
// (1024-thread blocks, one per CU), so that the render kernel's VALU counts can be turned into a time budget.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int KIND> __global__ void __launch_bounds__(1024) k(float *out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
    const f2 m = {seed, seed};
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
    unsigned long long w0 = i0, w1 = i1;
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) { REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));) }
        if (KIND == 1) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m));) }
        if (KIND == 2) { REP8(asm volatile("v_min_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_min_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_max_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));) }
        if (KIND == 3) { REP8(asm volatile("v_max3_f32 %0, %0, %8, %1\n v_max3_f32 %1, %1, %8, %2\n v_max3_f32 %2, %2, %8, %3\n v_max3_f32 %3, %3, %8, %4\n v_max3_f32 %4, %4, %8, %5\n v_max3_f32 %5, %5, %8, %6\n v_max3_f32 %6, %6, %8, %7\n v_max3_f32 %7, %7, %8, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));) }
        if (KIND == 4) { REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (KIND == 5) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");) }
        if (KIND == 6) { REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (KIND == 7) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %1\n v_pk_fma_f32 %1, %1, %8, %2\n v_pk_fma_f32 %2, %2, %8, %3\n v_pk_fma_f32 %3, %3, %8, %4\n v_pk_fma_f32 %4, %4, %8, %5\n v_pk_fma_f32 %5, %5, %8, %6\n v_pk_fma_f32 %6, %6, %8, %7\n v_pk_fma_f32 %7, %7, %8, %0" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m));) }
        if (KIND == 8) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %1\n v_fma_f32 %1, %1, %8, %2\n v_fma_f32 %2, %2, %8, %3\n v_fma_f32 %3, %3, %8, %4\n v_fma_f32 %4, %4, %8, %5\n v_fma_f32 %5, %5, %8, %6\n v_fma_f32 %6, %6, %8, %7\n v_fma_f32 %7, %7, %8, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));) }
        if (KIND == 9) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n v_mul_lo_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(iters));) }
        if (KIND == 11) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %2, %4, 0\n v_mad_u64_u32 %1, vcc, %3, %4, 0\n v_mad_u64_u32 %0, vcc, %2, %4, 0\n v_mad_u64_u32 %1, vcc, %3, %4, 0\n v_mad_u64_u32 %0, vcc, %2, %4, 0\n v_mad_u64_u32 %1, vcc, %3, %4, 0\n v_mad_u64_u32 %0, vcc, %2, %4, 0\n v_mad_u64_u32 %1, vcc, %3, %4, 0" : "+v"(w0), "+v"(w1) : "v"(i0), "v"(i1), "v"(iters) : "vcc");) }
        if (KIND == 12) { REP8(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(iters));) }
        if (KIND == 13) { REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]\n v_cndmask_b32_e64 %2, %2, %3, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]\n v_cndmask_b32_e64 %4, %4, %5, s[20:21]\n v_cndmask_b32_e64 %5, %5, %6, s[20:21]\n v_cndmask_b32_e64 %6, %6, %7, s[20:21]\n v_cndmask_b32_e64 %7, %7, %0, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "s20", "s21");) }
        if (KIND == 14) { REP8(asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_gt_f32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n v_cmp_gt_f32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc\n v_cmp_gt_f32 vcc, %5, %4\n v_cndmask_b32 %7, %7, %6, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");) }
        if (KIND == 15) { REP8(asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cmp_gt_f32 vcc, %2, %3\n v_cmp_gt_f32 vcc, %4, %5\n v_cmp_gt_f32 vcc, %6, %7\n v_cmp_gt_f32 vcc, %1, %0\n v_cmp_gt_f32 vcc, %3, %2\n v_cmp_gt_f32 vcc, %5, %4\n v_cmp_gt_f32 vcc, %7, %6" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");) }
        if (KIND == 10) { REP8(asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 3\n v_readlane_b32 s22, %2, 3\n v_readlane_b32 s23, %3, 3\n v_readlane_b32 s20, %4, 3\n v_readlane_b32 s21, %5, 3\n v_readlane_b32 s22, %6, 3\n v_readlane_b32 s23, %7, 3" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "s20", "s21", "s22", "s23");) }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.x + p6.x + p7.x + i0 + i1 + i2 + i3 + (int)w0 + (int)(w1 >> 32);
}
template <int KIND> void run(const char *name, float *d, int blocks) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, 1024>>>(d, 100, 1.0001f);
    hipEventRecord(e0);
    k<KIND><<<blocks, 1024>>>(d, iters, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves x iters x 64 instructions
    double per_simd = 4.0 * iters * 64.0;
    printf("%-14s %8.2f ms  %.3f wave-instr/ns per SIMD  (%.2f clocks/instr at 2.4 GHz)\n", name, ms, per_simd / (ms * 1e6), 2.4 * ms * 1e6 / per_simd);
}
int main() {
    int dev = 0; hipDeviceProp_t p; hipGetDeviceProperties(&p, dev);
    int blocks = p.multiProcessorCount;
    float *d; hipMalloc(&d, (size_t)blocks * 1024 * 4);
    printf("%s, %d CUs, clock %d MHz\n", p.name, blocks, p.clockRate / 1000);
    run<0>("v_mul_f32", d, blocks); run<8>("v_fma_f32", d, blocks); run<1>("v_pk_mul_f32", d, blocks); run<7>("v_pk_fma_f32", d, blocks);
    run<2>("v_min/max_f32", d, blocks); run<3>("v_max3_f32", d, blocks); run<4>("v_mov_b32", d, blocks); run<5>("v_cndmask_b32", d, blocks);
    run<6>("v_rcp_f32", d, blocks); run<9>("v_mul_lo/hi_u32", d, blocks); run<10>("v_readlane_b32", d, blocks); run<11>("v_mad_u64_u32", d, blocks); run<12>("v_xor/add_u32", d, blocks); run<13>("v_cndmask_e64", d, blocks); run<14>("v_cmp+v_cndmask", d, blocks); run<15>("v_cmp_gt_f32", d, blocks);
    hipFree(d);
    return 0;
}
