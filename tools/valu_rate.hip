// valu_rate.hip -- measurement aid: sustained issue rate of a few vector instructions on gfx950 at full occupancy
// (8 waves per SIMD).  Settles whether a wave64 VALU instruction occupies a SIMD for 2 or 4 cycles and what the packed
// forms cost.   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o gpurun_out/valu_rate && gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N_INNER 256
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    unsigned u0 = threadIdx.x, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7, u4 = u0 * 11, u5 = u0 * 13, u6 = u0 * 17, u7 = u0 * 19;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < N_INNER / 8; j++) {
            if (OP == 0) {   // v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                             "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 1) {   // v_pk_fma_f32
                asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n"
                             "v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
            } else if (OP == 2) {   // v_mad_u32_u24
                asm volatile("v_mad_u32_u24 %0, %0, %0, %0\n v_mad_u32_u24 %1, %1, %1, %1\n v_mad_u32_u24 %2, %2, %2, %2\n v_mad_u32_u24 %3, %3, %3, %3\n"
                             "v_mad_u32_u24 %4, %4, %4, %4\n v_mad_u32_u24 %5, %5, %5, %5\n v_mad_u32_u24 %6, %6, %6, %6\n v_mad_u32_u24 %7, %7, %7, %7"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));
            } else if (OP == 3) {   // v_xor_b32
                asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %4\n"
                             "v_xor_b32 %4, %4, %5\n v_xor_b32 %5, %5, %6\n v_xor_b32 %6, %6, %7\n v_xor_b32 %7, %7, %0"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));
            } else if (OP == 4) {   // v_perm_b32
                asm volatile("v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %1, %1, %2, %3\n v_perm_b32 %2, %2, %3, %4\n v_perm_b32 %3, %3, %4, %5\n"
                             "v_perm_b32 %4, %4, %5, %6\n v_perm_b32 %5, %5, %6, %7\n v_perm_b32 %6, %6, %7, %0\n v_perm_b32 %7, %7, %0, %1"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));
            } else if (OP == 5) {   // v_dot2_u32_u16
                asm volatile("v_dot2_u32_u16 %0, %0, %1, %2\n v_dot2_u32_u16 %1, %1, %2, %3\n v_dot2_u32_u16 %2, %2, %3, %4\n v_dot2_u32_u16 %3, %3, %4, %5\n"
                             "v_dot2_u32_u16 %4, %4, %5, %6\n v_dot2_u32_u16 %5, %5, %6, %7\n v_dot2_u32_u16 %6, %6, %7, %0\n v_dot2_u32_u16 %7, %7, %0, %1"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));
            } else if (OP == 6) {   // v_pk_mul_lo_u16
                asm volatile("v_pk_mul_lo_u16 %0, %0, %1\n v_pk_mul_lo_u16 %1, %1, %2\n v_pk_mul_lo_u16 %2, %2, %3\n v_pk_mul_lo_u16 %3, %3, %4\n"
                             "v_pk_mul_lo_u16 %4, %4, %5\n v_pk_mul_lo_u16 %5, %5, %6\n v_pk_mul_lo_u16 %6, %6, %7\n v_pk_mul_lo_u16 %7, %7, %0"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));
            } else if (OP == 7) {   // v_min_f32
                asm volatile("v_min_f32 %0, %0, %1\n v_min_f32 %1, %1, %2\n v_min_f32 %2, %2, %3\n v_min_f32 %3, %3, %4\n"
                             "v_min_f32 %4, %4, %5\n v_min_f32 %5, %5, %6\n v_min_f32 %6, %6, %7\n v_min_f32 %7, %7, %0"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 8) {   // v_med3_f32
                asm volatile("v_med3_f32 %0, %0, %1, %2\n v_med3_f32 %1, %1, %2, %3\n v_med3_f32 %2, %2, %3, %4\n v_med3_f32 %3, %3, %4, %5\n"
                             "v_med3_f32 %4, %4, %5, %6\n v_med3_f32 %5, %5, %6, %7\n v_med3_f32 %6, %6, %7, %0\n v_med3_f32 %7, %7, %0, %1"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 9) {   // v_pk_min_i16 (packed 16-bit integer min)
                asm volatile("v_pk_min_i16 %0, %0, %1\n v_pk_min_i16 %1, %1, %2\n v_pk_min_i16 %2, %2, %3\n v_pk_min_i16 %3, %3, %4\n"
                             "v_pk_min_i16 %4, %4, %5\n v_pk_min_i16 %5, %5, %6\n v_pk_min_i16 %6, %6, %7\n v_pk_min_i16 %7, %7, %0"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y +
                                          (float)(u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7);
}
template <int OP>
void run(const char* name, float* d, int blocks) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 10);
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * iters * N_INNER;       // wave-instructions
    // per SIMD: blocks*4 waves over 1024 SIMDs
    printf("%-18s %8.3f ms  %7.1f G wave-instr/s  => %.2f ns per wave-instr per SIMD (x clock GHz = cycles)\n", name, ms, winstr / ms / 1e6,
           ms * 1e6 / (winstr / 1024.0));
}
int main() {
    float* d; hipMalloc(&d, 256 * 2048 * 4 * 4);
    const int blocks = 256 * 8;   // 8 blocks of 4 waves per CU = 8 waves per SIMD
    run<0>("v_fma_f32", d, blocks); run<1>("v_pk_fma_f32", d, blocks); run<2>("v_mad_u32_u24", d, blocks); run<3>("v_xor_b32", d, blocks);
    run<4>("v_perm_b32", d, blocks); run<5>("v_dot2_u32_u16", d, blocks); run<6>("v_pk_mul_lo_u16", d, blocks); run<7>("v_min_f32", d, blocks);
    run<8>("v_med3_f32", d, blocks); run<9>("v_pk_min_i16", d, blocks);
    const int b1 = 256;           // one block per CU = one wave per SIMD
    printf("one wave per SIMD:\n");
    run<0>("v_fma_f32", d, b1); run<1>("v_pk_fma_f32", d, b1); run<2>("v_mad_u32_u24", d, b1);
    return 0;
}
