// Micro-benchmark (round 4): does the ORDER of the residual epilogue's instructions matter, and what do its 12 instructions per output
// cost?  The requantisation arithmetic is the largest block of a compute wave's time in the expand-class kernels
// (profiles/r04_pair_kernels.md) and hipcc emits each output's chain  mad -> ashr -> mad -> ashr -> add -> max -> mad -> ashr
// back to back (two chains interleaved at most).  Here the same 11 instructions per output run (a) chain by chain, (b) stage by stage over 4
// outputs, (c) stage by stage over 8 outputs, with 1-4 waves per SIMD.
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench/requant_ilp.hip -o /tmp/requant_ilp && /tmp/requant_ilp
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 1024

#define MAD(t, x, m, c) "v_mad_i64_i32 " t ", s[10:11], " x ", " m ", " c "\n"
#define ASHR(d, s, hi) "v_ashrrev_i32 " d ", " s ", " hi "\n"

template <int ORDER>
__global__ __launch_bounds__(256) void k(int *out, int a, int b) {
    // 8 accumulators, per-channel (m, s, C) in VGPRs, packed residual words, scalar tables in VGPR pairs
    int acc[8], m[8], s[8], o[8];
    long long C[8], t[8], u[8];
    int r[4];
    long long Cb = ((long long)a << 32) | 5, Cq = ((long long)b << 32) | 7;
    int mi = a * 977 + 13, mq = b * 1031 + 7, si = 3, sq = 9;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        acc[i] = threadIdx.x * (i + 3) + a, m[i] = a * (i + 11) + 1000003, s[i] = (i & 3) + 1, C[i] = ((long long)(i + 1) << 33) + b;
        o[i] = 0;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = threadIdx.x * 65537 + i;
    int sink = 0;
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
#define STAGE_A(i) asm volatile(MAD("%0", "%1", "%2", "%3") : "=&v"(t[i]) : "v"(acc[i]), "v"(m[i]), "v"(C[i]) : "s10", "s11");
#define STAGE_X0(i) asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n" : "=v"(o[i]) : "v"(si), "v"(r[(i) >> 1]));
#define STAGE_X1(i) asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n" : "=v"(o[i]) : "v"(si), "v"(r[(i) >> 1]));
#define STAGE_B(i) asm volatile("v_ashrrev_i32 %0, %1, %2\n" : "=v"(acc[i]) : "v"(s[i]), "v"((int)(t[i] >> 32)));
#define STAGE_C(i) asm volatile(MAD("%0", "%1", "%2", "%3") : "=&v"(u[i]) : "v"(o[i]), "v"(mi), "v"(Cb) : "s10", "s11");
#define STAGE_D(i) asm volatile("v_ashrrev_i32 %0, %1, %2\n" : "=v"(o[i]) : "v"(si), "v"((int)(u[i] >> 32)));
#define STAGE_E(i) asm volatile("v_add_u32 %0, %1, %2\n v_max_i32 %0, 0, %0\n" : "=&v"(o[i]) : "v"(acc[i]), "v"(o[i]));
#define STAGE_F(i) asm volatile(MAD("%0", "%1", "%2", "%3") : "=&v"(t[i]) : "v"(o[i]), "v"(mq), "v"(Cq) : "s10", "s11");
#define STAGE_G(i) asm volatile("v_ashrrev_i32 %0, %1, %2\n" : "=v"(acc[i]) : "v"(sq), "v"((int)(t[i] >> 32)));
#define XSEL(i) if ((i) & 1) { STAGE_X1(i) } else { STAGE_X0(i) }
#define CHAIN(i) STAGE_A(i) XSEL(i) STAGE_B(i) STAGE_C(i) STAGE_D(i) STAGE_E(i) STAGE_F(i) STAGE_G(i)
#define ALL4(ST, b) ST(b + 0) ST(b + 1) ST(b + 2) ST(b + 3)
        if (ORDER == 0) {
            CHAIN(0) CHAIN(1) CHAIN(2) CHAIN(3) CHAIN(4) CHAIN(5) CHAIN(6) CHAIN(7)
        } else if (ORDER == 1) {
            ALL4(STAGE_A, 0) XSEL(0) XSEL(1) XSEL(2) XSEL(3) ALL4(STAGE_B, 0) ALL4(STAGE_C, 0) ALL4(STAGE_D, 0) ALL4(STAGE_E, 0) ALL4(STAGE_F, 0) ALL4(STAGE_G, 0)
            ALL4(STAGE_A, 4) XSEL(4) XSEL(5) XSEL(6) XSEL(7) ALL4(STAGE_B, 4) ALL4(STAGE_C, 4) ALL4(STAGE_D, 4) ALL4(STAGE_E, 4) ALL4(STAGE_F, 4) ALL4(STAGE_G, 4)
        } else {
            ALL4(STAGE_A, 0) ALL4(STAGE_A, 4) XSEL(0) XSEL(1) XSEL(2) XSEL(3) XSEL(4) XSEL(5) XSEL(6) XSEL(7) ALL4(STAGE_B, 0) ALL4(STAGE_B, 4)
            ALL4(STAGE_C, 0) ALL4(STAGE_C, 4) ALL4(STAGE_D, 0) ALL4(STAGE_D, 4) ALL4(STAGE_E, 0) ALL4(STAGE_E, 4) ALL4(STAGE_F, 0) ALL4(STAGE_F, 4)
            ALL4(STAGE_G, 0) ALL4(STAGE_G, 4)
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) sink ^= acc[i];
    }
    out[blockIdx.x * 256 + threadIdx.x] = sink;
}

template <int ORDER>
void run(const char *name, int waves_per_simd) {
    int *d;
    hipMalloc(&d, 256 * 4 * 256 * sizeof(int));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd;   // 256-thread blocks: one wave per SIMD each
    hipLaunchKernelGGL(k<ORDER>, dim3(blocks), dim3(256), 0, 0, d, 3, 5);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<ORDER>, dim3(blocks), dim3(256), 0, 0, d, 3, 5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double outputs = (double)ITERS * 8 * waves_per_simd;   // outputs per lane per SIMD
    printf("%-34s %d waves/SIMD: %.3f ms -> %.1f ns per output per SIMD (= %.1f cycles @2.1 GHz; 12 instructions each)\n", name, waves_per_simd, ms,
           ms * 1e6 / outputs, ms * 1e6 / outputs * 2.1);
    hipFree(d);
}

int main() {
    for (int w = 1; w <= 4; ++w) {
        run<0>("chain by chain (hipcc's order)", w);
        run<1>("stage by stage over 4 outputs", w);
        run<2>("stage by stage over 8 outputs", w);
    }
    return 0;
}
