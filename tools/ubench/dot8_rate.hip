// Micro-benchmark for the W4A4 design question (north_star: "wavefront DPP/LDS dot-products for the W4A4 layers"):
//   (a) v_dot8_u32_u4 on the VALU - unsigned activations x biased weights (w + 8), corrected by 8 * sum(a):
//       8 MACs per lane per instruction;
//   (b) what the kernels do instead: nibbles unpacked to int8 in registers (weights as value * 16) feeding
//       v_mfma_i32_32x32x32_i8 - per 16 packed bytes of each operand, 2 MFMAs (2 x 32768 MACs per wave).
// Reports MAC/clk/CU of both at 1, 2 and 4 waves per SIMD on all 256 CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void k_dot8(unsigned *out, int iters) {
    unsigned a[8], w[8], acc[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i, w[i] = threadIdx.x * 40503u + 7 * i, acc[i] = 0;
    for (int it = 0; it < iters; it += 8) {   // 8 independent accumulator chains, all register indices static
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_udot8(a[i], w[(i + r) & 7], acc[i], false);
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__device__ __forceinline__ v4i unpack_u(unsigned x0, unsigned x1) {
    v4i r = {(int)(x0 & 0x0F0F0F0Fu), (int)((x0 >> 4) & 0x0F0F0F0Fu), (int)(x1 & 0x0F0F0F0Fu), (int)((x1 >> 4) & 0x0F0F0F0Fu)};
    return r;
}
__device__ __forceinline__ v4i unpack_s16(unsigned x0, unsigned x1) {
    v4i r = {(int)((x0 << 4) & 0xF0F0F0F0u), (int)(x0 & 0xF0F0F0F0u), (int)((x1 << 4) & 0xF0F0F0F0u), (int)(x1 & 0xF0F0F0F0u)};
    return r;
}
// one wave: 2 x 2 tiles of 32 x 32 (64 pixels x 64 channels), per K = 64 nibble step: 2 packed A + 2 packed W fragments
// (16 B each) -> unpack (4 x 2 unpacks) -> 8 MFMAs
__global__ void k_mfma_nib(int *out, int iters) {
    v16i acc[2][2];
    for (int c = 0; c < 2; ++c) for (int q = 0; q < 2; ++q) for (int r = 0; r < 16; ++r) acc[c][q][r] = 0;
    v4i pa[2], pw[2];
    for (int i = 0; i < 2; ++i) {
        pa[i] = v4i{(int)threadIdx.x, (int)threadIdx.x * 3, i, 7};
        pw[i] = v4i{(int)threadIdx.x * 5, i, 11, (int)threadIdx.x};
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            v4i a8[2], w8[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a8[i] = unpack_u((unsigned)pa[i][2 * hf] + it, (unsigned)pa[i][2 * hf + 1]);
                w8[i] = unpack_s16((unsigned)pw[i][2 * hf] + it, (unsigned)pw[i][2 * hf + 1]);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[c][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w8[c], a8[q], acc[c][q], 0, 0, 0);
        }
    }
    int s = 0;
    for (int c = 0; c < 2; ++c) for (int q = 0; q < 2; ++q) for (int r = 0; r < 16; ++r) s += acc[c][q][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    unsigned *out; hipMalloc(&out, 256 * 1024 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps : {1, 2, 4}) {
        float ms;
        hipLaunchKernelGGL(k_dot8, dim3(256), dim3(256 * wps), 0, 0, out, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_dot8, dim3(256), dim3(256 * wps), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        double macs = 256.0 * 256 * wps * 8.0 * 8 * iters;  // threads x 8 chains x 8 MACs
        printf("v_dot8_u32_u4      waves/SIMD %d: %7.3f ms  %6.1f TMAC/s  (%6.1f MAC/clk/CU at 2.4 GHz; needs + 1 dot8 per 8 activations for the 8*sum(a) correction)\n",
               wps, ms, macs / ms / 1e9, macs / (ms * 1e-3) / 256 / 2.4e9);
        hipLaunchKernelGGL(k_mfma_nib, dim3(256), dim3(256 * wps), 0, 0, (int *)out, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mfma_nib, dim3(256), dim3(256 * wps), 0, 0, (int *)out, iters / 10);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        macs = 256.0 * 4 * wps * 8.0 * 32768 * (iters / 10);  // waves x 8 MFMAs x 32768 MACs
        printf("unpack + mfma i8   waves/SIMD %d: %7.3f ms  %6.1f TMAC/s  (%6.1f MAC/clk/CU at 2.4 GHz)\n", wps, ms, macs / ms / 1e9,
               macs / (ms * 1e-3) / 256 / 2.4e9);
    }
    return 0;
}
