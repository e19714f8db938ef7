// Round 5: the classifier (QuantLinear, quant_modules.py:79-130: fc(x_int) with integer weights and bias, the int32 result times
// fl(S_fc[c] * S_a)) as its own small kernel.  Through the general conv kernel the layer was ONE 64 x 64 tile per 64 outputs walking
// K = 2048 in 32 barrier-separated steps: 16 workgroups for 14-22 us at the very end of each chain, when nothing else is left to overlap
// with (profiles/r05_c_kernel_trace.md).  Here K is split over the 4 waves of a workgroup, a workgroup owns 32 outputs x 32 images, and
// the operands go global -> registers -> MFMA with no LDS staging and no barrier in the loop (both operands are tiny and L2-resident:
// 2 MiB of weights, 128 KiB of activations); the four partial sums meet once in LDS.
//   out[n][o] = (float)(sum_k q[n][k] * w[o][k] + bias[o]) * fscale[o]          (exactly the DEQUANT epilogue of conv_igemm.hip)
#include <stdio.h>

#include "common.h"

namespace {

__global__ __launch_bounds__(256) void fc_dequant_kernel(const int8_t *__restrict__ q, const int8_t *__restrict__ w,
                                                         const int32_t *__restrict__ bias, const float *__restrict__ fscale,
                                                         float *__restrict__ out, int N, int K, int ldo, int n_valid) {
    __shared__ int red[4][16][64];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, l31 = lane & 31, h = lane >> 5;
    const int o0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int n = min(n0 + l31, N - 1);   // rows behind the batch repeat the last image; their sums are never stored
    const int kq = K >> 2;                // this wave's share of K (launcher: K % 128 == 0)
    // MFMA row i of the weight operand is output cperm(i): lane half h then owns outputs 16 h .. 16 h + 15 of the tile (common.h)
    const v4i *wp = reinterpret_cast<const v4i *>(w + (size_t)(o0 + cperm(l31)) * K + wave * kq + h * 16);
    const v4i *qp = reinterpret_cast<const v4i *>(q + (size_t)n * K + wave * kq + h * 16);
    v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int steps = kq >> 5;            // 32 K bytes per MFMA: two 16-byte vectors apart in this lane's row
    for (int s0 = 0; s0 < steps; s0 += 8) {   // eight steps' operands requested before the first MFMA: the loop is a latency chain otherwise
        v4i a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int s = min(s0 + j, steps - 1);
            a[j] = wp[2 * s], b[j] = qp[2 * s];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (s0 + j < steps) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[j], b[j], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    // wave v finishes registers 4 v .. 4 v + 3 of every lane: image n0 + l31, outputs o0 + 16 h + 4 v + j
    if (n0 + l31 >= N) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 4 * wave + j, o = o0 + h * 16 + r;
        if (o >= n_valid) continue;
        const int v = red[0][r][lane] + red[1][r][lane] + red[2][r][lane] + red[3][r][lane] + bias[o];
        out[(size_t)(n0 + l31) * ldo + o] = (float)v * fscale[o];
    }
}

}  // namespace

extern "C" int hawq_fc_dequant_ok(int32_t N, int32_t K, int32_t Nout_p) {
    return N > 0 && K >= 128 && K % 128 == 0 && Nout_p > 0 && Nout_p % 32 == 0;
}

extern "C" int hawq_fc_dequant(const int8_t *q, const int8_t *wgt, const int32_t *bias, const float *fscale, float *out_f32,
                               int32_t N, int32_t K, int32_t Nout_p, int32_t n_valid, int32_t ldo, void *stream) {
    HAWQ_REQUIRE(q && wgt && bias && fscale && out_f32, "hawq_fc_dequant: null pointer");
    HAWQ_REQUIRE(hawq_fc_dequant_ok(N, K, Nout_p), "hawq_fc_dequant: N=%d K=%d Nout_p=%d (K %% 128 == 0, Nout_p %% 32 == 0)", N, K, Nout_p);
    HAWQ_REQUIRE(n_valid > 0 && n_valid <= Nout_p && ldo >= n_valid, "hawq_fc_dequant: n_valid=%d ldo=%d", n_valid, ldo);
    HAWQ_REQUIRE(((reinterpret_cast<size_t>(q) | reinterpret_cast<size_t>(wgt)) & 15) == 0, "hawq_fc_dequant: operands must be 16-byte aligned");
    const int ot = (n_valid + 31) / 32;   // tiles of 32 outputs that hold a valid one
    hipLaunchKernelGGL(fc_dequant_kernel, dim3(ot, (N + 31) / 32), dim3(256), 0, (hipStream_t)stream, q, wgt, bias, fscale, out_f32, N, K,
                       ldo, n_valid);
    HAWQ_CHECK_HIP(hipGetLastError());
    return 0;
}
