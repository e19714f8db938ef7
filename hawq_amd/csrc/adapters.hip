// Module-compatible adapters (gfx950): the reference's modules exchange fp32 NCHW tensors that
// hold integer*scale.  These kernels convert between that convention and the integer NHWC
// tensors of the conv kernels, and implement QuantAct's fixed-point paths directly on fp32
// data so that a Q_ResNet can also be stepped module by module (e.g. for range calibration).
//   hawq_f32_nchw_to_q_nhwc     quant_modules.py:489-490, 125-126   (x_int = x / S_a)
//   hawq_acc_nhwc_to_f32_nchw   quant_modules.py:491-494
//   hawq_fixedpoint_f32         quant_utils.py:363-456
//   hawq_fakequant_f32          quant_utils.py:73-97, 237-258, 281-308
//   hawq_avgpool_f32            quant_modules.py:596-602, quant_utils.py:334-337
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void f32_to_q_kernel(const float *__restrict__ x, void *__restrict__ out, int N,
                                                       int C, int H, int W, int Cpad, int bits, float scale) {
    const int groups = Cpad >> 3;  // 8 channels per thread
    const long long HWl = (long long)H * W;
    const long long total = (long long)N * HWl * groups;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long pix = i % (N * HWl);  // pixel fastest: coalesced reads of each plane
        const int g = (int)(i / (N * HWl));
        const int n = (int)(pix / HWl);
        const long long hw = pix - n * HWl;
        int q[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = g * 8 + k;
            float v = 0.f;
            if (c < C) v = rintf(__fdiv_rn(x[((long long)n * C + c) * HWl + hw], scale));
            q[k] = (int)v;
        }
        const long long elem = pix * Cpad + g * 8;
        if (bits == 8) {
            v2i o;
            o.x = (int)pack4_i8(q[0], q[1], q[2], q[3]);
            o.y = (int)pack4_i8(q[4], q[5], q[6], q[7]);
            *reinterpret_cast<v2i *>((int8_t *)out + elem) = o;
        } else {
            *reinterpret_cast<uint32_t *>((uint8_t *)out + (elem >> 1)) = pack8_u4(q);
        }
    }
}

__global__ __launch_bounds__(256) void acc_to_f32_kernel(const int32_t *__restrict__ acc, float *__restrict__ y, int N,
                                                         int C, int H, int W, int Cpad,
                                                         const float *__restrict__ fscale) {
    const long long HWl = (long long)H * W;
    const long long total = (long long)N * C * HWl;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long hw = i % HWl;
        const long long r = i / HWl;
        const int c = (int)(r % C);
        const int n = (int)(r / C);
        y[i] = __fmul_rn((float)acc[((long long)n * HWl + hw) * Cpad + c], fscale[c]);
    }
}

__device__ __forceinline__ int to_int(const float z, const float s_a, const float s_w) {
    return (int)rintf(__fdiv_rn(__fdiv_rn(z, s_a), s_w));  // torch.round(z / S_a / S_w)
}

__global__ __launch_bounds__(256) void fixedpoint_kernel(const float *__restrict__ z, float *__restrict__ y, int N,
                                                         int C, int HW, float s_a, const float *__restrict__ s_w,
                                                         const int32_t *__restrict__ m, const int32_t *__restrict__ e,
                                                         int per_ch, const float *__restrict__ ident, float s_ida,
                                                         const float *__restrict__ s_idw,
                                                         const int32_t *__restrict__ m_id,
                                                         const int32_t *__restrict__ e_id, int per_ch_id, float s_out,
                                                         int do_clamp, int q_lo, int q_hi) {
    const long long total = (long long)N * C * HW;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)((i / HW) % C);
        const int k = per_ch == 1 ? 0 : c;
        int q;
        if (ident == nullptr) {
            q = dyadic_rne(to_int(z[i], s_a, s_w[k]), m[k], e[k]);
        } else {
            const int k1 = per_ch_id == 1 ? 0 : c;
            const float idv = ident[i];
            const int wx = to_int(idv, s_ida, s_idw[k1]);
            const int wy = to_int(__fsub_rn(z[i], idv), s_a, s_w[k]);  // wy = z - identity in binary32
            q = dyadic_rne(wx, m_id[k1], e_id[k1]) + dyadic_rne(wy, m[k], e[k]);
        }
        if (do_clamp) q = clampi(q, q_lo, q_hi);
        y[i] = __fmul_rn((float)q, s_out);
    }
}

__global__ __launch_bounds__(256) void fakequant_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                        long long n, float inv_scale, float scale, int lo, int hi) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float r = rintf(__fmul_rn(inv_scale, x[i]));
        r = fminf(fmaxf(r, (float)lo), (float)hi);
        y[i] = __fmul_rn(r, scale);
    }
}

__global__ __launch_bounds__(256) void avgpool_f32_kernel(const float *__restrict__ x, float *__restrict__ y, int NC,
                                                          int HW, float scale) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= NC) return;
    long long s = 0;
    for (int k = 0; k < HW; ++k) s += (long long)rintf(__fdiv_rn(x[(long long)i * HW + k], scale));
    const long long p = (100 * s + HW) / (100ll * HW);
    y[i] = __fmul_rn((float)p, scale);
}

inline int grid_for(long long work_items) {
    long long g = (work_items + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" int hawq_f32_nchw_to_q_nhwc(const float *x, void *out, int32_t N, int32_t C, int32_t H, int32_t W,
                                       int32_t Cpad, int32_t bits, float scale, void *stream) {
    HAWQ_REQUIRE(x && out, "hawq_f32_nchw_to_q_nhwc: null pointer");
    HAWQ_REQUIRE(Cpad >= C && Cpad % 8 == 0 && (bits == 8 || bits == 4), "hawq_f32_nchw_to_q_nhwc: bad Cpad/bits");
    hipLaunchKernelGGL(f32_to_q_kernel, dim3(grid_for((long long)N * H * W * (Cpad / 8))), dim3(256), 0,
                       (hipStream_t)stream, x, out, N, C, H, W, Cpad, bits, scale);
    HAWQ_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int hawq_acc_nhwc_to_f32_nchw(const int32_t *acc, float *y, int32_t N, int32_t C, int32_t H, int32_t W,
                                         int32_t Cpad, const float *fscale, void *stream) {
    HAWQ_REQUIRE(acc && y && fscale, "hawq_acc_nhwc_to_f32_nchw: null pointer");
    hipLaunchKernelGGL(acc_to_f32_kernel, dim3(grid_for((long long)N * C * H * W)), dim3(256), 0, (hipStream_t)stream,
                       acc, y, N, C, H, W, Cpad, fscale);
    HAWQ_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int hawq_fixedpoint_f32(const float *z, float *y, int32_t N, int32_t C, int32_t HW, float s_a,
                                   const float *s_w, const int32_t *m, const int32_t *e, int32_t per_ch,
                                   const float *ident, float s_ida, const float *s_idw, const int32_t *m_id,
                                   const int32_t *e_id, int32_t per_ch_id, float s_out, int32_t do_clamp,
                                   int32_t q_lo, int32_t q_hi, void *stream) {
    HAWQ_REQUIRE(z && y && s_w && m && e, "hawq_fixedpoint_f32: null pointer");
    HAWQ_REQUIRE(per_ch == 1 || per_ch == C, "hawq_fixedpoint_f32: per_ch must be 1 or C");
    HAWQ_REQUIRE(!ident || (s_idw && m_id && e_id && (per_ch_id == 1 || per_ch_id == C)),
                 "hawq_fixedpoint_f32: identity tables missing");
    hipLaunchKernelGGL(fixedpoint_kernel, dim3(grid_for((long long)N * C * HW)), dim3(256), 0, (hipStream_t)stream, z,
                       y, N, C, HW, s_a, s_w, m, e, per_ch, ident, s_ida, s_idw, m_id, e_id, per_ch_id, s_out,
                       do_clamp, q_lo, q_hi);
    HAWQ_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int hawq_fakequant_f32(const float *x, float *y, int64_t n, float inv_scale, float scale, int32_t lo,
                                  int32_t hi, void *stream) {
    HAWQ_REQUIRE(x && y && n > 0, "hawq_fakequant_f32: bad arguments");
    hipLaunchKernelGGL(fakequant_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, (long long)n,
                       inv_scale, scale, lo, hi);
    HAWQ_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int hawq_avgpool_f32(const float *x, float *y, int32_t NC, int32_t HW, float scale, void *stream) {
    HAWQ_REQUIRE(x && y && NC > 0 && HW > 0, "hawq_avgpool_f32: bad arguments");
    hipLaunchKernelGGL(avgpool_f32_kernel, dim3((NC + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, y, NC, HW,
                       scale);
    HAWQ_CHECK_HIP(hipGetLastError());
    return 0;
}


// ------------------------------------------------------------------------------------------------------------------
// Grouped / depthwise integer convolution (MobileNetV2's 3x3 depthwise layers, q_mobilenetv2.py; F.conv2d(..., groups)
// in quant_modules.py:489-494 / 727-736): exact int32 accumulators for the module-compatible path.  One thread per output
// element; with one input channel per group (depthwise) neighbouring threads read neighbouring bytes of the NHWC input,
// and 9 MACs per output leave the layer bound by its own bytes.  Not an MFMA shape: K per output is KH*KW*Cin/groups.
namespace {
__global__ void grouped_conv_kernel(const int8_t *__restrict__ in, const int8_t *__restrict__ wgt, const int32_t *__restrict__ bias,
                                    int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int groups,
                                    int Ho, int Wo, int32_t *__restrict__ out) {
    const long long total = (long long)N * Ho * Wo * Cout;
    const int cg_in = Cin / groups, cg_out = Cout / groups;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(idx % Cout);
        const long long m = idx / Cout;
        const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho), n = (int)(m / ((long long)Wo * Ho));
        const int g = co / cg_out;
        int acc = bias ? bias[co] : 0;
        const int8_t *wp = wgt + (size_t)co * KH * KW * cg_in;
        for (int kh = 0; kh < KH; ++kh) {
            const int iy = oy * stride - pad + kh;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int kw = 0; kw < KW; ++kw) {
                const int ix = ox * stride - pad + kw;
                if ((unsigned)ix >= (unsigned)W) continue;
                const int8_t *ip = in + (((size_t)n * H + iy) * W + ix) * Cin + (size_t)g * cg_in;
                const int8_t *wq = wp + (kh * KW + kw) * cg_in;
                for (int ci = 0; ci < cg_in; ++ci) acc += (int)ip[ci] * (int)wq[ci];
            }
        }
        out[idx] = acc;
    }
}
}  // namespace

extern "C" int hawq_conv2d_grouped(const int8_t *in, const int8_t *wgt, const int32_t *bias, int32_t N, int32_t H, int32_t W,
                                   int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad, int32_t groups,
                                   int32_t *out_acc, void *stream) {
    HAWQ_REQUIRE(in && wgt && out_acc, "hawq_conv2d_grouped: null pointer");
    HAWQ_REQUIRE(groups >= 1 && Cin > 0 && Cout > 0 && Cin % groups == 0 && Cout % groups == 0, "hawq_conv2d_grouped: groups=%d must divide Cin=%d and Cout=%d",
                 groups, Cin, Cout);
    HAWQ_REQUIRE(KH > 0 && KW > 0 && stride > 0 && pad >= 0 && N > 0 && H > 0 && W > 0, "hawq_conv2d_grouped: bad geometry");
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    HAWQ_REQUIRE(Ho > 0 && Wo > 0, "hawq_conv2d_grouped: empty output");
    const long long total = (long long)N * Ho * Wo * Cout;
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(grouped_conv_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, wgt, bias, N, H, W, Cin, Cout, KH, KW, stride,
                       pad, groups, Ho, Wo, out_acc);
    HAWQ_CHECK_HIP(hipGetLastError());
    return 0;
}


// ------------------------------------------------------------------------------------------------------------------
// Depthwise 3x3 convolution (MobileNetV2's conv2, q_mobilenetv2.py:46-48; F.conv2d(groups = C), quant_modules.py:489-494):
// int8 NHWC in, weights tap-major [3][3][C] (so that the 4 channels a thread owns are one dword per tap), exact int32 NHWC
// accumulators and / or the requantised int8 tensor out.
//
// The layer is 9 MACs per output: what it costs is bytes and VALU issue, so the kernel is built around both.
//   * A thread owns 4 consecutive channels of DW_PX horizontally adjacent output pixels and walks a BAND of output rows with a
//     three-row register window of packed input dwords: every input row is fetched once per band (the band's two halo rows
//     aside) instead of once per output row it feeds.  The first version mapped every output row to another workgroup; the
//     rows of one image then landed on different XCDs and each of their L2s fetched the same input row again (3.2 TB/s of
//     fabric traffic for 0.9 TB/s of tensor bytes on the 112 x 112 layers).
//   * Consecutive lanes own consecutive channel groups, so each load / store instruction of a wave covers whole 64..256-byte
//     runs of the NHWC rows; all loads of an input row are issued back to back with clamped addresses (no branches), border
//     zeros come from a select.
//   * One MAC is ONE instruction and the packed operands are never unpacked: v_dot4_i32_i8 of the packed input dword with a
//     copy of the tap's weight dword that keeps only byte j adds x_j * w_j to accumulator j.  The 36 masked weight dwords are
//     loop-invariant registers.
namespace {
constexpr int DW_PX = 4;
constexpr int DW_MAX_BAND = 8;

// REQ: conv2 -> ReLU -> QuantAct fused (exact dyadic_rne per channel table), int8 out; `out` (int32) then optional.
// FASTQ (round 4, hawq_depthwise3x3_requant_fast): 1 / 2 = the fast requant contract (2: with the exact-tie correction) against fused
// constants [C][4] (bias folded in) passed through `mult`: 3-4 instructions per output instead of dyadic_rne's ~20 - the requants were
// more than half of this kernel's instructions on the 7 x 7 maps.
template <bool REQ, int S, int FASTQ = 0>
__global__ __launch_bounds__(256) void depthwise3x3_kernel(const int8_t *__restrict__ in, const int8_t *__restrict__ w9c, const int32_t *__restrict__ bias,
                                                           int N, int H, int W, int C, int Cv, int Ho, int Wo, int band, int32_t *__restrict__ out,
                                                           const int32_t *__restrict__ mult, const int32_t *__restrict__ expo, int relu, int q_lo, int q_hi,
                                                           int8_t *__restrict__ out_q) {
    constexpr int NC = (DW_PX - 1) * S + 3;   // input columns under DW_PX outputs
    // Cv <= C: channels >= Cv are padding (zero weights / bias / multipliers): only the ceil(Cv / 4) real groups are computed,
    // each thread also writes the zeros of the padding groups cg + i * cgv of its pixels
    const int cgs = C >> 2, cgv = (Cv + 3) >> 2, wq = (Wo + DW_PX - 1) / DW_PX, bands = (Ho + band - 1) / band;
    const long long total = (long long)N * bands * wq * cgv;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int cg = (int)(idx % cgv);
        long long rest = idx / cgv;
        const int xq = (int)(rest % wq);
        rest /= wq;
        const int bi = (int)(rest % bands), n = (int)(rest / bands), ox0 = xq * DW_PX;
        const int oy0 = bi * band, oy1 = min(oy0 + band, Ho);
        int wm[9][4];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ww = *reinterpret_cast<const int *>(w9c + (size_t)t * C + 4 * cg);
#pragma unroll
            for (int j = 0; j < 4; ++j) wm[t][j] = ww & (0xff << (8 * j));
        }
        const v4i b4 = (bias && FASTQ == 0) ? *reinterpret_cast<const v4i *>(bias + 4 * cg) : v4i{0, 0, 0, 0};
        v4i m4 = {0, 0, 0, 0}, e4 = {0, 0, 0, 0};
        DyNt dq[4];
        if constexpr (REQ && FASTQ == 0) m4 = *reinterpret_cast<const v4i *>(mult + 4 * cg), e4 = *reinterpret_cast<const v4i *>(expo + 4 * cg);
        if constexpr (REQ && FASTQ != 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const v4i t4 = *reinterpret_cast<const v4i *>(mult + (size_t)(4 * cg + j) * 4);
                dq[j].m = t4.x, dq[j].s = t4.y & 31, dq[j].k = t4.y >> 8;
                dq[j].add = (long long)(((unsigned long long)(unsigned)t4.w << 32) | (unsigned)t4.z);
            }
        }
        // column offsets (clamped) and validity of the NC input columns
        int coff[NC];
        unsigned cmask = 0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int ix = ox0 * S - 1 + c;
            cmask |= ((unsigned)ix < (unsigned)W ? 1u : 0u) << c;
            coff[c] = min(max(ix, 0), W - 1) * C;
        }
        const int8_t *img = in + (size_t)n * H * W * C + 4 * cg;
        auto load_row = [&](int iy, int (&r)[NC]) {
            const bool rv = (unsigned)iy < (unsigned)H;
            const int8_t *rowp = img + (size_t)min(max(iy, 0), H - 1) * W * C;
#pragma unroll
            for (int c = 0; c < NC; ++c) r[c] = *reinterpret_cast<const int *>(rowp + coff[c]);
#pragma unroll
            for (int c = 0; c < NC; ++c) r[c] = (rv && ((cmask >> c) & 1)) ? r[c] : 0;
        };
        int r0[NC], r1[NC], r2[NC];
        load_row(oy0 * S - 1, r0);
        load_row(oy0 * S, r1);
        for (int oy = oy0; oy < oy1; ++oy) {
            load_row(oy * S + 1, r2);
            int acc[DW_PX][4];
#pragma unroll
            for (int p = 0; p < DW_PX; ++p)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[p][j] = b4[j];
            // sdot4 multiplies byte by byte: byte j of the masked weight meets byte j of the input, the other three products are 0
#pragma unroll
            for (int p = 0; p < DW_PX; ++p)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[p][j] = __builtin_amdgcn_sdot4(r0[p * S + kw], wm[kw][j], acc[p][j], false);
                        acc[p][j] = __builtin_amdgcn_sdot4(r1[p * S + kw], wm[3 + kw][j], acc[p][j], false);
                        acc[p][j] = __builtin_amdgcn_sdot4(r2[p * S + kw], wm[6 + kw][j], acc[p][j], false);
                    }
#pragma unroll
            for (int p = 0; p < DW_PX; ++p)
                if (ox0 + p < Wo) {
                    const size_t o4 = (((size_t)n * Ho + oy) * Wo + ox0 + p) * C + 4 * cg;
                    if (out) *reinterpret_cast<v4i *>(out + o4) = v4i{acc[p][0], acc[p][1], acc[p][2], acc[p][3]};
                    if constexpr (REQ) {
                        int qv[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if constexpr (FASTQ == 0) qv[j] = clampi(dyadic_rne(relu ? max(acc[p][j], 0) : acc[p][j], m4[j], e4[j]), q_lo, q_hi);
                            else qv[j] = med3i(FASTQ == 2 ? dyadic_tie(acc[p][j], dq[j]) : dyadic_nt(acc[p][j], dq[j]), q_lo, q_hi);   // ReLU is in q_lo >= 0
                        }
                        *reinterpret_cast<uint32_t *>(out_q + o4) = pack4_i8(qv[0], qv[1], qv[2], qv[3]);
                    }
                    for (int pc = cgv + cg; pc < cgs; pc += cgv) {   // padding groups: rne(0 * 0) = 0
                        const size_t z4 = o4 + 4 * (size_t)(pc - cg);
                        if (out) *reinterpret_cast<v4i *>(out + z4) = v4i{0, 0, 0, 0};
                        if constexpr (REQ) *reinterpret_cast<uint32_t *>(out_q + z4) = 0u;
                    }
                }
            if (oy + 1 < oy1) {
                if constexpr (S == 1) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) r0[c] = r1[c], r1[c] = r2[c];
                } else {
#pragma unroll
                    for (int c = 0; c < NC; ++c) r0[c] = r2[c];
                    load_row((oy + 1) * S, r1);
                }
            }
        }
    }
}

template <bool REQ, int FASTQ = 0>
int depthwise_launch(const int8_t *in, const int8_t *wgt9c, const int32_t *bias, const int32_t *m, const int32_t *e, int N, int H, int W, int C, int Cv, int stride,
                     int relu, int q_lo, int q_hi, int8_t *out_q, int32_t *out_acc, void *stream) {
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    HAWQ_REQUIRE(Ho > 0 && Wo > 0, "hawq_depthwise3x3: empty output");
    // rows per band: as tall as leaves ~4 waves per SIMD of the chip (2^18 threads), at most DW_MAX_BAND
    const long long per_row = (long long)N * ((Wo + DW_PX - 1) / DW_PX) * ((Cv + 3) / 4);
    long long band = per_row * Ho / (1 << 18);
    band = band < 1 ? 1 : (band > DW_MAX_BAND ? DW_MAX_BAND : band);
    if (band > Ho) band = Ho;
    const long long total = per_row * ((Ho + band - 1) / band);
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    auto kern = stride == 1 ? depthwise3x3_kernel<REQ, 1, FASTQ> : depthwise3x3_kernel<REQ, 2, FASTQ>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, wgt9c, bias, N, H, W, C, Cv, Ho, Wo, (int)band, out_acc, m, e, relu, q_lo, q_hi,
                       out_q);
    HAWQ_CHECK_HIP(hipGetLastError());
    return 0;
}
}  // namespace

extern "C" int hawq_depthwise3x3(const int8_t *in, const int8_t *wgt9c, const int32_t *bias, int32_t N, int32_t H, int32_t W, int32_t C,
                                 int32_t stride, int32_t *out_acc, void *stream) {
    HAWQ_REQUIRE(in && wgt9c && out_acc, "hawq_depthwise3x3: null pointer");
    HAWQ_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && (stride == 1 || stride == 2), "hawq_depthwise3x3: need C %% 4 == 0 and stride 1 or 2 (C=%d stride=%d)", C, stride);
    return depthwise_launch<false>(in, wgt9c, bias, nullptr, nullptr, N, H, W, C, C, stride, 0, 0, 0, nullptr, out_acc, stream);
}

extern "C" int hawq_depthwise3x3_requant(const int8_t *in, const int8_t *wgt9c, const int32_t *bias, const int32_t *m, const int32_t *e,
                                         int32_t N, int32_t H, int32_t W, int32_t C, int32_t C_valid, int32_t stride, int32_t relu, int32_t q_lo,
                                         int32_t q_hi, int8_t *out_q, int32_t *out_acc, void *stream) {
    HAWQ_REQUIRE(in && wgt9c && m && e && out_q, "hawq_depthwise3x3_requant: null pointer");
    HAWQ_REQUIRE(C_valid >= 0 && C_valid <= C, "hawq_depthwise3x3_requant: C_valid=%d outside [0, C=%d]", C_valid, C);
    HAWQ_REQUIRE(C_valid == 0 || C_valid == C || (q_lo <= 0 && q_hi >= 0), "hawq_depthwise3x3_requant: padding channels are written as 0, which the clamp must contain");
    HAWQ_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && (stride == 1 || stride == 2), "hawq_depthwise3x3_requant: need C %% 4 == 0 and stride 1 or 2 (C=%d stride=%d)", C, stride);
    HAWQ_REQUIRE(q_lo >= -128 && q_hi <= 127 && q_lo <= q_hi, "hawq_depthwise3x3_requant: the clamp must fit int8");
    return depthwise_launch<true>(in, wgt9c, bias, m, e, N, H, W, C, C_valid > 0 ? C_valid : C, stride, relu, q_lo, q_hi, out_q, out_acc, stream);
}

extern "C" int hawq_depthwise3x3_requant_fast(const int8_t *in, const int8_t *wgt9c, const int32_t *ctab, int32_t fast_tables, int32_t N, int32_t H, int32_t W,
                                              int32_t C, int32_t C_valid, int32_t stride, int32_t q_lo, int32_t q_hi, int8_t *out_q, void *stream) {
    HAWQ_REQUIRE(in && wgt9c && ctab && out_q, "hawq_depthwise3x3_requant_fast: null pointer");
    HAWQ_REQUIRE(fast_tables != 0, "hawq_depthwise3x3_requant_fast: fast_tables = 0 is hawq_depthwise3x3_requant");
    HAWQ_REQUIRE(C_valid >= 0 && C_valid <= C, "hawq_depthwise3x3_requant_fast: C_valid=%d outside [0, C=%d]", C_valid, C);
    HAWQ_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && (stride == 1 || stride == 2), "hawq_depthwise3x3_requant_fast: need C %% 4 == 0 and stride 1 or 2 (C=%d stride=%d)", C, stride);
    HAWQ_REQUIRE(q_lo >= 0 && q_hi <= 127 && q_lo <= q_hi, "hawq_depthwise3x3_requant_fast: the clamp carries the ReLU: 0 <= q_lo <= q_hi <= 127");
    HAWQ_REQUIRE(C_valid == 0 || C_valid == C || q_lo == 0, "hawq_depthwise3x3_requant_fast: padding channels are written as 0, which the clamp must contain");
    const int cv = C_valid > 0 ? C_valid : C;
    if (fast_tables & 4) return depthwise_launch<true, 2>(in, wgt9c, nullptr, ctab, nullptr, N, H, W, C, cv, stride, 1, q_lo, q_hi, out_q, nullptr, stream);
    return depthwise_launch<true, 1>(in, wgt9c, nullptr, ctab, nullptr, N, H, W, C, cv, stride, 1, q_lo, q_hi, out_q, nullptr, stream);
}


// ------------------------------------------------------------------------------------------------------------------
// Input quantiser + im2col for a 3x3 / stride 2 / pad 1 first convolution on 3 input channels (MobileNetV2's init block,
// q_mobilenetv2.py:182-186 after the input QuantAct, quant_modules.py:271-274): fp32 NCHW image in, one 64-byte int8 row per
// OUTPUT pixel out, holding the 27 quantised patch values in (kh, kw, c) order and 37 zeros.  The 3x3 convolution on a
// 3 -> 64 channel-padded tensor (9 x 64 bytes of K per output, 95 % of them padding) becomes a 1x1 convolution with K = 64
// on a tensor a quarter of the size.  q = clamp(rne(fl(1/S) * x)), one binary32 rounding as `1. / scale * input` has.
namespace {
__global__ __launch_bounds__(256) void quantize_im2col3x3s2_kernel(const float *__restrict__ x, int8_t *__restrict__ out, int N, int H, int W, int Ho, int Wo,
                                                                   float inv_scale, int lo, int hi) {
    const long long total = (long long)N * Ho * Wo;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ox = (int)(i % Wo);
        const long long r = i / Wo;
        const int oy = (int)(r % Ho), n = (int)(r / Ho);
        int q[28];
        q[27] = 0;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int iy = 2 * oy - 1 + kh;
            const bool rv = (unsigned)iy < (unsigned)H;
            const int iyc = min(max(iy, 0), H - 1);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ix = 2 * ox - 1 + kw;
                const bool ok = rv && (unsigned)ix < (unsigned)W;
                const int ixc = min(max(ix, 0), W - 1);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float v = x[(((long long)n * 3 + c) * H + iyc) * W + ixc];
                    float rr = rintf(__fmul_rn(inv_scale, v));
                    rr = fminf(fmaxf(rr, (float)lo), (float)hi);
                    q[(kh * 3 + kw) * 3 + c] = ok ? (int)rr : 0;
                }
            }
        }
        v4i *dst = reinterpret_cast<v4i *>(out + i * 64);
        v4i o0, o1;
#pragma unroll
        for (int k = 0; k < 4; ++k) o0[k] = (int)pack4_i8(q[4 * k], q[4 * k + 1], q[4 * k + 2], q[4 * k + 3]);
#pragma unroll
        for (int k = 0; k < 3; ++k) o1[k] = (int)pack4_i8(q[16 + 4 * k], q[17 + 4 * k], q[18 + 4 * k], q[19 + 4 * k]);
        o1[3] = 0;
        dst[0] = o0, dst[1] = o1, dst[2] = v4i{0, 0, 0, 0}, dst[3] = v4i{0, 0, 0, 0};
    }
}
}  // namespace

namespace {
// the same rows from uint8 NHWC pixels: ToTensor + Normalize + the input QuantAct are one table look-up per channel
// (lut[c][u], hawq_amd.quant_utils.input_quant_lut - the reference pipeline's own float operations, evaluated on the host)
__global__ __launch_bounds__(256) void quantize_im2col3x3s2_u8_kernel(const uint8_t *__restrict__ x, const int8_t *__restrict__ lut, int8_t *__restrict__ out,
                                                                      int N, int H, int W, int Ho, int Wo) {
    __shared__ int8_t lut_s[3 * 256];
    for (int i = threadIdx.x; i < 3 * 256 / 4; i += 256) reinterpret_cast<int *>(lut_s)[i] = reinterpret_cast<const int *>(lut)[i];
    __syncthreads();
    const long long total = (long long)N * Ho * Wo;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ox = (int)(i % Wo);
        const long long r = i / Wo;
        const int oy = (int)(r % Ho), n = (int)(r / Ho);
        int q[28];
        q[27] = 0;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int iy = 2 * oy - 1 + kh;
            const bool rv = (unsigned)iy < (unsigned)H;
            const int iyc = min(max(iy, 0), H - 1);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ix = 2 * ox - 1 + kw;
                const bool ok = rv && (unsigned)ix < (unsigned)W;
                const uint8_t *px = x + (((long long)n * H + iyc) * W + min(max(ix, 0), W - 1)) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) q[(kh * 3 + kw) * 3 + c] = ok ? (int)lut_s[c * 256 + px[c]] : 0;
            }
        }
        v4i *dst = reinterpret_cast<v4i *>(out + i * 64);
        v4i o0, o1;
#pragma unroll
        for (int k = 0; k < 4; ++k) o0[k] = (int)pack4_i8(q[4 * k], q[4 * k + 1], q[4 * k + 2], q[4 * k + 3]);
#pragma unroll
        for (int k = 0; k < 3; ++k) o1[k] = (int)pack4_i8(q[16 + 4 * k], q[17 + 4 * k], q[18 + 4 * k], q[19 + 4 * k]);
        o1[3] = 0;
        dst[0] = o0, dst[1] = o1, dst[2] = v4i{0, 0, 0, 0}, dst[3] = v4i{0, 0, 0, 0};
    }
}
}  // namespace

extern "C" int hawq_quantize_im2col3x3s2_u8(const uint8_t *x, const int8_t *lut, int8_t *out, int32_t N, int32_t C, int32_t H, int32_t W, void *stream) {
    HAWQ_REQUIRE(x && lut && out, "hawq_quantize_im2col3x3s2_u8: null pointer");
    HAWQ_REQUIRE(C == 3, "hawq_quantize_im2col3x3s2_u8: C=%d, the 64-byte patch row is laid out for 3 input channels", C);
    HAWQ_REQUIRE(N > 0 && H > 0 && W > 0, "hawq_quantize_im2col3x3s2_u8: bad geometry");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(quantize_im2col3x3s2_u8_kernel, dim3(grid_for((long long)N * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, x, lut, out, N, H, W, Ho, Wo);
    HAWQ_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int hawq_quantize_im2col3x3s2(const float *x, int8_t *out, int32_t N, int32_t C, int32_t H, int32_t W, float inv_scale, int32_t q_lo,
                                         int32_t q_hi, void *stream) {
    HAWQ_REQUIRE(x && out, "hawq_quantize_im2col3x3s2: null pointer");
    HAWQ_REQUIRE(C == 3, "hawq_quantize_im2col3x3s2: C=%d, the 64-byte patch row is laid out for 3 input channels", C);
    HAWQ_REQUIRE(N > 0 && H > 0 && W > 0 && q_lo >= -128 && q_hi <= 127 && q_lo <= q_hi, "hawq_quantize_im2col3x3s2: bad geometry or range");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(quantize_im2col3x3s2_kernel, dim3(grid_for((long long)N * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, x, out, N, H, W, Ho, Wo,
                       inv_scale, q_lo, q_hi);
    HAWQ_CHECK_HIP(hipGetLastError());
    return 0;
}


// ------------------------------------------------------------------------------------------------------------------
// Image pipeline in front of forward_uint8 (quant_train.py:428-440: transforms.Resize(256) + CenterCrop(224) on the
// decoded PIL image, then ToTensor + Normalize, which the stem's look-up table replays).  One separable pass of Pillow's
// 8-bit antialiased resampling (libImaging/Resample.c, ImagingResampleHorizontal_8bpc / ...Vertical_8bpc): fixed-point
// coefficients with 22 fractional bits, accumulator started at 2^21, result clipped to 0..255.  The coefficients are
// computed on the host in binary64 exactly as precompute_coeffs + normalize_coeffs_8bpc do (hawq_amd/image.py).
// uint8 HWC in and out; byte work, HBM-bound.
namespace {
__global__ void resample_u8_kernel(const uint8_t *__restrict__ in, int in_w, int C, const int32_t *__restrict__ bounds,
                                   const int32_t *__restrict__ coef, int ksize, int out_n, int horizontal, int lines, int line0,
                                   int out_w, uint8_t *__restrict__ out) {
    // horizontal: out[l][o][c] over lines l (input rows line0 + l) and output columns o;  in [.][in_w][C]
    // vertical:   out[o][x][c] over output rows o and columns x = l (line0 == 0);        in [.][in_w][C], in_w == out_w
    const long long total = (long long)lines * out_n * C;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        int l, o;
        if (horizontal) {
            o = (int)((idx / C) % out_n), l = (int)(idx / ((long long)C * out_n));
        } else {
            l = (int)((idx / C) % lines), o = (int)(idx / ((long long)C * lines));
        }
        const int k0 = bounds[2 * o], kn = bounds[2 * o + 1];
        const int32_t *k = coef + (size_t)o * ksize;
        int ss = 1 << 21;
        if (horizontal) {
            const uint8_t *row = in + ((size_t)(line0 + l) * in_w + k0) * C + c;
            for (int x = 0; x < kn; ++x) ss += (int)row[(size_t)x * C] * k[x];
            ss >>= 22;
            out[((size_t)l * out_n + o) * C + c] = (uint8_t)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
        } else {
            const uint8_t *col = in + ((size_t)k0 * in_w + l) * C + c;
            for (int y = 0; y < kn; ++y) ss += (int)col[(size_t)y * in_w * C] * k[y];
            ss >>= 22;
            out[((size_t)o * out_w + l) * C + c] = (uint8_t)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
        }
    }
}
}  // namespace

extern "C" int hawq_resample_u8(const uint8_t *in, int32_t in_w, int32_t C, const int32_t *bounds, const int32_t *coef, int32_t ksize,
                                int32_t out_n, int32_t horizontal, int32_t lines, int32_t line0, uint8_t *out, void *stream) {
    HAWQ_REQUIRE(in && bounds && coef && out, "hawq_resample_u8: null pointer");
    HAWQ_REQUIRE(in_w > 0 && C > 0 && ksize > 0 && out_n > 0 && lines > 0 && line0 >= 0, "hawq_resample_u8: bad geometry");
    const long long total = (long long)lines * out_n * C;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(resample_u8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, in_w, C, bounds, coef, ksize, out_n, horizontal,
                       lines, line0, horizontal ? out_n : lines, out);
    HAWQ_CHECK_HIP(hipGetLastError());
    return 0;
}
