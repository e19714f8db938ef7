// Micro-benchmark: issue rate of the integer VALU instructions the requant epilogue is made of.
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 2048
template <int OP>
__global__ void k(int *out, int a, int b) {
    int x0 = threadIdx.x, x1 = a, x2 = b, x3 = a ^ b, x4 = a + 7, x5 = b - 3, x6 = 11, x7 = 13;
    long long y0 = a, y1 = b, y2 = 3, y3 = 5;
#pragma unroll 1
    for (int i = 0; i < ITERS; ++i) {
        if (OP == 0) {  // v_add_u32 x8
            asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                         "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (OP == 1) {  // v_ashrrev_i32 x8
            asm volatile("v_ashrrev_i32 %0, 1, %0\n v_ashrrev_i32 %1, 1, %1\n v_ashrrev_i32 %2, 1, %2\n v_ashrrev_i32 %3, 1, %3\n"
                         "v_ashrrev_i32 %4, 1, %4\n v_ashrrev_i32 %5, 1, %5\n v_ashrrev_i32 %6, 1, %6\n v_ashrrev_i32 %7, 1, %7"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        } else if (OP == 2) {  // v_mad_i64_i32 x4 (independent)
            asm volatile("v_mad_i64_i32 %0, s[10:11], %4, %5, %0\n v_mad_i64_i32 %1, s[10:11], %4, %5, %1\n"
                         "v_mad_i64_i32 %2, s[10:11], %4, %5, %2\n v_mad_i64_i32 %3, s[10:11], %4, %5, %3"
                         : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(x0), "v"(x1) : "s10", "s11");
        } else if (OP == 3) {  // v_mul_hi_i32 x8
            asm volatile("v_mul_hi_i32 %0, %0, %8\n v_mul_hi_i32 %1, %1, %8\n v_mul_hi_i32 %2, %2, %8\n v_mul_hi_i32 %3, %3, %8\n"
                         "v_mul_hi_i32 %4, %4, %8\n v_mul_hi_i32 %5, %5, %8\n v_mul_hi_i32 %6, %6, %8\n v_mul_hi_i32 %7, %7, %8"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (OP == 4) {  // v_max_i32 x8
            asm volatile("v_max_i32 %0, %0, %8\n v_max_i32 %1, %1, %8\n v_max_i32 %2, %2, %8\n v_max_i32 %3, %3, %8\n"
                         "v_max_i32 %4, %4, %8\n v_max_i32 %5, %5, %8\n v_max_i32 %6, %6, %8\n v_max_i32 %7, %7, %8"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (OP == 5) {  // v_fma_f32 x8 (reference: known 2 cycles per wave64)
            asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                         "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (OP == 6) {  // v_cvt_pk_i16_i32 x8
            asm volatile("v_cvt_pk_i16_i32 %0, %0, %8\n v_cvt_pk_i16_i32 %1, %1, %8\n v_cvt_pk_i16_i32 %2, %2, %8\n v_cvt_pk_i16_i32 %3, %3, %8\n"
                         "v_cvt_pk_i16_i32 %4, %4, %8\n v_cvt_pk_i16_i32 %5, %5, %8\n v_cvt_pk_i16_i32 %6, %6, %8\n v_cvt_pk_i16_i32 %7, %7, %8"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (OP == 7) {  // v_mul_lo_u32 x8
            asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                         "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (int)(y0 + y1 + y2 + y3);
}
template <int OP>
void run(const char *name, int per_iter) {
    int *d;
    hipMalloc(&d, 256 * 8 * 1024 * sizeof(int));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256 * 8;  // 8 blocks of 256 threads per CU: 8 waves per SIMD
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3, 5);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3, 5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = blocks * 4.0 / 1024.0;
    const double insts = (double)ITERS * per_iter * waves_per_simd;  // wave-instructions per SIMD
    printf("%-18s %.3f ms  -> %.2f ns per wave-instruction per SIMD (= %.2f cycles @2.4GHz)\n", name, ms,
           ms * 1e6 / insts, ms * 1e6 / insts * 2.4);
    hipFree(d);
}
int main() {
    run<5>("v_fma_f32", 8);
    run<0>("v_add_u32", 8);
    run<1>("v_ashrrev_i32", 8);
    run<4>("v_max_i32", 8);
    run<6>("v_cvt_pk_i16_i32", 8);
    run<2>("v_mad_i64_i32", 4);
    run<3>("v_mul_hi_i32", 8);
    run<7>("v_mul_lo_u32", 8);
    return 0;
}
