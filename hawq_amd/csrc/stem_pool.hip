// Stem, pooling and small tail kernels of the integer ResNet pipeline (gfx950).
//   hawq_quantize_input      quant_modules.py:271-274 / quant_utils.py:73-97, 237-258
//   hawq_stem_conv7          quant_modules.py:489-494 + q_resnet.py:117-122 (conv, QuantAct16, ReLU)
//   hawq_maxpool3s2_requant  q_resnet.py:93,119 (+ first unit's QuantAct, q_resnet.py:234/239)
//   hawq_requant_residual    quant_modules.py:288-293 (S_w == 1)
//   hawq_avgpool_requant     quant_modules.py:596-600, quant_utils.py:334-337, q_resnet.py:129-131
#include <stdlib.h>

#include "common.h"

namespace {

// ------------------------------------------------------------------ input quantiser
// One thread per output pixel: reads 3 planes (coalesced along W), writes one dword (c0,c1,c2,0).
__global__ __launch_bounds__(256) void quantize_input_kernel(const float *__restrict__ x, int8_t *__restrict__ out,
                                                             int N, int C, int H, int W, int out_h, int out_w,
                                                             int pad_top, int pad_left, float inv_scale, int lo,
                                                             int hi) {
    const long long total = (long long)N * H * W;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int w = (int)(i % W);
        const long long r = i / W;
        const int hh = (int)(r % H);
        const int n = (int)(r / H);
        int q[4] = {0, 0, 0, 0};
        for (int c = 0; c < C; ++c) {
            const float v = x[((long long)(n * C + c) * H + hh) * W + w];
            const float t = __fmul_rn(inv_scale, v);  // `1. / scale * input` : one binary32 rounding
            float rr = rintf(t);                      // round-half-even == torch.round
            rr = fminf(fmaxf(rr, (float)lo), (float)hi);
            q[c] = (int)rr;
        }
        reinterpret_cast<uint32_t *>(out)[((long long)n * out_h + hh + pad_top) * out_w + w + pad_left] =
            pack4_i8(q[0], q[1], q[2], q[3]);
    }
}

// ------------------------------------------------------------------ stem 7x7/2 conv
__device__ __forceinline__ int cperm(int i) { return (((i >> 2) & 1) << 4) + (i & 3) + ((i >> 3) << 2); }

// Each wave keeps the whole 64x(7x32 B) weight matrix in registers (14 x v4i) and walks pixel
// tiles of 64 pixels; the B operand (7 rows x 32 contiguous bytes of the padded NHWC4 image per
// output pixel) is loaded straight from global/L1 - every input byte is reused ~12x by
// neighbouring pixels and rows, all of it cache hits.
__global__ __launch_bounds__(256) void stem_conv7_kernel(const int8_t *__restrict__ in, const int8_t *__restrict__ wgt,
                                                         const int32_t *__restrict__ bias,
                                                         const int32_t *__restrict__ m, const int32_t *__restrict__ e,
                                                         int N, int Hp, int Wp, int Ho, int Wo, int q_lo, int q_hi,
                                                         uint16_t *__restrict__ out16, int32_t *__restrict__ out_acc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    v4i wf[2][7];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int kh = 0; kh < 7; ++kh)
            wf[c][kh] = *reinterpret_cast<const v4i *>(wgt + ((c * 32 + cperm(l31)) * 7 + kh) * 32 + h * 16);
    const int M = N * Ho * Wo;
    const int ntiles = (M + 63) / 64;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        v16i acc[2][2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][p][r] = 0;
        const int8_t *src[2];
        int pix[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            pix[p] = tile * 64 + p * 32 + l31;
            const int mm = pix[p] < M ? pix[p] : M - 1;
            const int n = mm / (Ho * Wo);
            const int r = mm - n * (Ho * Wo);
            const int oy = r / Wo, ox = r - oy * Wo;
            src[p] = in + (((long long)n * Hp + 2 * oy) * Wp + 2 * ox) * 4 + h * 16;
        }
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
            v4i af[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int *s = reinterpret_cast<const int *>(src[p] + (long long)kh * Wp * 4);  // 8-B aligned
                v2i a = *reinterpret_cast<const v2i *>(s), b = *reinterpret_cast<const v2i *>(s + 2);
                af[p].x = a.x, af[p].y = a.y, af[p].z = b.x, af[p].w = b.y;
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    acc[c][p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[c][kh], af[p], acc[c][p], 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ch = c * 32 + h * 16;
            int bb[16], mm[16], ee[16];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v4i t0 = reinterpret_cast<const v4i *>(bias + ch)[i];
                v4i t1 = reinterpret_cast<const v4i *>(m + ch)[i];
                v4i t2 = reinterpret_cast<const v4i *>(e + ch)[i];
                bb[4 * i] = t0.x, bb[4 * i + 1] = t0.y, bb[4 * i + 2] = t0.z, bb[4 * i + 3] = t0.w;
                mm[4 * i] = t1.x, mm[4 * i + 1] = t1.y, mm[4 * i + 2] = t1.z, mm[4 * i + 3] = t1.w;
                ee[4 * i] = t2.x, ee[4 * i + 1] = t2.y, ee[4 * i + 2] = t2.z, ee[4 * i + 3] = t2.w;
            }
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                if (pix[p] >= M) continue;
                const size_t elem = (size_t)pix[p] * 64 + ch;
                int v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = acc[c][p][r] + bb[r];
                if (out_acc) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v4i w = {v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
                        reinterpret_cast<v4i *>(out_acc + elem)[i] = w;
                    }
                }
                if (out16) {
                    int w[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        // QuantAct 16-bit (clamp) then ReLU; both monotone, so they commute with max-pool
                        const int a = max(clampi(dyadic_rne(v[2 * i], mm[2 * i], ee[2 * i]), q_lo, q_hi), 0);
                        const int b = max(clampi(dyadic_rne(v[2 * i + 1], mm[2 * i + 1], ee[2 * i + 1]), q_lo, q_hi), 0);
                        w[i] = min(a, 65535) | (min(b, 65535) << 16);
                    }
                    v4i *dst = reinterpret_cast<v4i *>(out16 + elem);
                    v4i a = {w[0], w[1], w[2], w[3]}, b = {w[4], w[5], w[6], w[7]};
                    dst[0] = a;
                    dst[1] = b;
                }
            }
        }
    }
}


// ------------------------------------------------------------------ fused stem
// input quantiser + 7x7/2 conv + bias + 3x3/2 max-pool + QuantAct16 + ReLU + first unit's QuantAct in
// ONE kernel (q_resnet.py:115-122 + 234/239).  A workgroup owns an 8x8 block of POOLED pixels of two
// images (one 4x8 sub-block per wave).  The fp32 input patch it needs (39x39 pixels per image) is
// quantised straight into an int8 NHWC4 LDS patch; each wave then walks the 3x3 pooling window: for
// window position (dy,dx) one MFMA pixel tile holds conv pixel (2py+dy-1, 2px+dx-1) of the lane's pooled
// pixel, so the max over the window is a register max over 9 accumulator tiles - taken on the raw
// accumulators, because the per-channel requantisation is monotone (m > 0) and the bias is constant.
// The 16-bit conv output at 112x112 (205 MB per 128-image batch) never exists in memory.
constexpr int SF_PW = 40;                       // patch width in pixels (39 used + 1 zero column)
constexpr int SF_PATCH = 39 * SF_PW * 4;        // bytes per image patch

// U8: the image arrives as uint8 NHWC [N][H][W][Cin] (what an image decoder produces) together with a [3][256] int8
// table lut[c][u] = QuantAct(normalise_c(u / 255)) built on the host with the reference's own float operations
// (hawq_amd.engine.input_lut): the input quantiser becomes a table look-up - exact, and the input read shrinks 4x.
template <bool U8>
__global__ __launch_bounds__(256, 4) void stem_fused_kernel(
    const float *__restrict__ x, const uint8_t *__restrict__ xu8, const int8_t *__restrict__ lut,
    int N, int Cin, int H, int W, float inv_scale, int in_lo, int in_hi,
    const int8_t *__restrict__ wgt, const int32_t *__restrict__ bias, const int32_t *__restrict__ m,
    const int32_t *__restrict__ e, int a_lo, int a_hi, int Hc, int Wc, int Hp, int Wp,
    uint16_t *__restrict__ res_out, void *__restrict__ out_q, int out_bits, int mq, int eq, int q_lo, int q_hi,
    int fast, int dbg) {
    // Two copies of the int8 patch: `patch` and, 16 bytes into the second half, the SAME bytes shifted down by 8 (copy byte a =
    // patch byte a + 8).  A lane's B fragment is 16 consecutive bytes starting at a multiple of 8: window columns dx = 0 / 2
    // read it 16-byte aligned from `patch`, dx = 1 from the shifted copy - one conflict-free ds_read_b128 per MFMA instead of
    // two ds_read_b64 that reach only half of the LDS banks (lanes are 16 bytes apart: 8-byte reads leave every other bank
    // pair idle, and the 4 pooled rows of a wave fold onto each other two-way).  The 126 fragment reads per wave were the
    // kernel's largest single cost (4 MB of LDS reads per CU per 8 workgroups).
    __shared__ __attribute__((aligned(16))) char patch[4 * SF_PATCH + 32];
    // fast contract: per-channel fused requant constants {m, (e - 32) | k << 8, lo32(C), hi32(C)}, C = (bias << k) * m + 2^(e-1) - the form
    // of hawq_amd.packing.pack_ctab, built here from (bias, m, e) by the first 64 threads while the others already fetch the patch.  The
    // epilogue then reads ONE 16-byte LDS entry per output instead of three global table words, and a requant is shift + v_mad_i64_i32 +
    // shift (round 4: the epilogue was 11 of the launch's 54 us at batch 64, tools/stemprobe.py with HAWQ_DBG=64)
    __shared__ __attribute__((aligned(16))) int ctab_s[64 * 4];
    char *const patch8 = patch + 2 * SF_PATCH + 16;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (fast && t < 64) {
        const int mm = m[t], ek = e[t], ee = ek & 0xff, kk = ek >> 8;
        const long long cc = ((long long)bias[t] << kk) * (long long)mm + (1ll << (ee - 1));
        const v4i ent = {mm, (ee - 32) | (kk << 8), (int)(unsigned)cc, (int)(cc >> 32)};
        *reinterpret_cast<v4i *>(ctab_s + 4 * t) = ent;
    }
    const int l31 = lane & 31, h = lane >> 5;
    const int bw = (Wp + 7) >> 3, bh = (Hp + 7) >> 3;
    int b = blockIdx.x;
    const int bx = b % bw;
    b /= bw;
    const int by = b % bh;
    const int n0 = (b / bh) * 2;
    const int py0 = by * 8, px0 = bx * 8;

    if constexpr (U8) {
        // ---- uint8 input: table look-up straight into the LDS patch
        __shared__ int8_t lut_s[3 * 256];
        for (int i = t; i < 3 * 256 / 4; i += 256) reinterpret_cast<int *>(lut_s)[i] = reinterpret_cast<const int *>(lut)[i];
        __syncthreads();
        constexpr int NG = 2 * 39 * (SF_PW / 4);       // groups of 4 pixels (one 16-B LDS store each)
        constexpr int GPT = (NG + 255) / 256;          // groups per thread
        // all byte loads of the thread's groups are issued before any is consumed (the fill is latency-bound otherwise);
        // coordinates are clamped into the image so every load is legal, out-of-image pixels are zeroed by mask
        unsigned raw[GPT][3];   // the group's 12 bytes (4 pixels x 3 channels), little-endian
        unsigned gmask[GPT];
        int gaddr[GPT];
#pragma unroll
        for (int k = 0; k < GPT; ++k) {
            const int g = t + 256 * k;
            const int gg = g < NG ? g : NG - 1;
            const int img = gg / (39 * (SF_PW / 4)), rem = gg - img * (39 * (SF_PW / 4));
            const int r = rem / (SF_PW / 4), c = (rem - r * (SF_PW / 4)) * 4;
            const int iy = 4 * py0 - 5 + r, ix = 4 * px0 - 5 + c, n = n0 + img;
            const bool rowok = g < NG && n < N && (unsigned)iy < (unsigned)H && !HAWQ_DBG_BIT(dbg, 16);
            gaddr[k] = img * SF_PATCH + (r * SF_PW + c) * 4;
            gmask[k] = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (rowok && (unsigned)(ix + j) < (unsigned)W && c + j < 39) gmask[k] |= 1u << j;
            const int nn = n < N ? n : N - 1, yy = min(max(iy, 0), H - 1);
            const size_t rowoff = ((size_t)nn * H + yy) * W * Cin;
            if (Cin == 3 && ix >= 0 && ix + 5 < W) {
                // 4 aligned dwords cover the 12 bytes at any misalignment (and stay inside the row: ix + 5 < W)
                const size_t b0 = rowoff + (size_t)ix * 3;
                const unsigned *q4 = reinterpret_cast<const unsigned *>(xu8 + (b0 & ~(size_t)3));
                const unsigned d0 = q4[0], d1 = q4[1], d2 = q4[2], d3 = q4[3];
                const unsigned sh = (unsigned)(b0 & 3);
                raw[k][0] = __builtin_amdgcn_alignbyte(d1, d0, sh);
                raw[k][1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
                raw[k][2] = __builtin_amdgcn_alignbyte(d3, d2, sh);
            } else {  // image borders / fewer channels: byte loads with clamped coordinates
                unsigned char b[12];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint8_t *px = xu8 + rowoff + (size_t)min(max(ix + j, 0), W - 1) * Cin;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) b[3 * j + ch] = ch < Cin ? px[ch] : 0;
                }
#pragma unroll
                for (int d = 0; d < 3; ++d)
                    raw[k][d] = (unsigned)b[4 * d] | ((unsigned)b[4 * d + 1] << 8) | ((unsigned)b[4 * d + 2] << 16) | ((unsigned)b[4 * d + 3] << 24);
            }
        }
#pragma unroll
        for (int k = 0; k < GPT; ++k) {
            if (t + 256 * k >= NG) continue;
            int w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = (gmask[k] >> j) & 1;
                auto byte_at = [&](int i) { return (raw[k][i >> 2] >> (8 * (i & 3))) & 0xffu; };
                w[j] = ok ? (int)pack4_i8(lut_s[byte_at(3 * j)], Cin > 1 ? lut_s[256 + byte_at(3 * j + 1)] : 0,
                                          Cin > 2 ? lut_s[512 + byte_at(3 * j + 2)] : 0, 0)
                          : 0;
            }
            v4i o = {w[0], w[1], w[2], w[3]};
            *reinterpret_cast<v4i *>(patch + gaddr[k]) = o;
            *reinterpret_cast<v2i *>(patch8 + gaddr[k] - 8) = v2i{o.x, o.y};
            *reinterpret_cast<v2i *>(patch8 + gaddr[k]) = v2i{o.z, o.w};
        }
    } else {
    // ---- quantise the input patch(es) into LDS: q = clamp(rint(fl(1/S) * x))  (quant_utils.py:73-97)
    const size_t plane = (size_t)H * W;
    auto quant1 = [&](float v) {
        v = rintf(__fmul_rn(inv_scale, v));
        return (int)fminf(fmaxf(v, (float)in_lo), (float)in_hi);
    };
    // 4 pixels (one 16-B LDS store) per step.  All global loads of the thread's groups are issued before any
    // is consumed (coordinates are clamped into the image so every load is legal; out-of-image pixels are
    // zeroed by mask afterwards) - the fill is latency-bound otherwise.
    constexpr int NG = 2 * 39 * (SF_PW / 4);       // 780 groups of 4 pixels
    constexpr int GPT = (NG + 255) / 256;          // groups per thread
    float4 v[GPT][3];
    int gaddr[GPT];
    unsigned gmask[GPT];  // bit k: pixel k of the group is inside the image
#pragma unroll
    for (int k = 0; k < GPT; ++k) {
        const int g = t + 256 * k;
        const int gg = g < NG ? g : NG - 1;
        const int img = gg / (39 * (SF_PW / 4)), rem = gg - img * (39 * (SF_PW / 4));
        const int r = rem / (SF_PW / 4), c = (rem - r * (SF_PW / 4)) * 4;
        const int iy = 4 * py0 - 5 + r, ix = 4 * px0 - 5 + c, n = n0 + img;
        const bool rowok = g < NG && n < N && (unsigned)iy < (unsigned)H && !HAWQ_DBG_BIT(dbg, 16);
        gaddr[k] = img * SF_PATCH + (r * SF_PW + c) * 4;
        gmask[k] = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (rowok && (unsigned)(ix + j) < (unsigned)W && c + j < 39) gmask[k] |= 1u << j;
        const int nn = n < N ? n : N - 1, yy = min(max(iy, 0), H - 1);
        const float *src = x + ((size_t)nn * Cin) * plane + (size_t)yy * W;
        const bool interior = ix >= 0 && ix + 3 < W;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            if (ch < Cin) {
                if (interior) {
                    v[k][ch] = *reinterpret_cast<const float4 *>(src + ch * plane + ix);
                } else {
                    const float *s1 = src + ch * plane;
                    v[k][ch].x = s1[min(max(ix, 0), W - 1)], v[k][ch].y = s1[min(max(ix + 1, 0), W - 1)];
                    v[k][ch].z = s1[min(max(ix + 2, 0), W - 1)], v[k][ch].w = s1[min(max(ix + 3, 0), W - 1)];
                }
            } else {
                v[k][ch] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < GPT; ++k) {
        if (t + 256 * k >= NG) continue;
        const float px4[4][3] = {{v[k][0].x, v[k][1].x, v[k][2].x}, {v[k][0].y, v[k][1].y, v[k][2].y},
                                 {v[k][0].z, v[k][1].z, v[k][2].z}, {v[k][0].w, v[k][1].w, v[k][2].w}};
        int w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = (gmask[k] >> j) & 1;
            w[j] = ok ? (int)pack4_i8(quant1(px4[j][0]), Cin > 1 ? quant1(px4[j][1]) : 0, Cin > 2 ? quant1(px4[j][2]) : 0, 0) : 0;
        }
        v4i o = {w[0], w[1], w[2], w[3]};
        *reinterpret_cast<v4i *>(patch + gaddr[k]) = o;
        *reinterpret_cast<v2i *>(patch8 + gaddr[k] - 8) = v2i{o.x, o.y};
        *reinterpret_cast<v2i *>(patch8 + gaddr[k]) = v2i{o.z, o.w};
    }
    }
    __syncthreads();

    const int img = wave >> 1;
    const int pr = (wave & 1) * 4 + (l31 >> 3), pc = l31 & 7;  // pooled pixel of this lane inside the block
    const int py = py0 + pr, px = px0 + pc, n = n0 + img;
    const bool pvalid = n < N && py < Hp && px < Wp;
    const int poff = img * SF_PATCH + ((4 * pr) * SF_PW + 4 * pc + 4 * h) * 4;   // multiple of 16
    const char *pbase = patch + poff, *pbase8 = patch8 + poff;
    const size_t opix = ((size_t)n * Hp + py) * Wp + px;

    // The two 32-channel halves are walked one after the other: weights of ONE half (7 fragments, 28 VGPRs), one
    // accumulator tile and one running maximum live at a time - 80 instead of 138 VGPRs, 6 instead of 3 workgroups per
    // CU, whose fill (HBM), MFMA and max / requant (VALU) phases then overlap.  The price is reading every B fragment of
    // the LDS patch twice: 2 x 126 ds_read_b64 per wave, ~500 LDS cycles beside 4000 cycles of MFMA.
    // a block none of whose 17 x 17 conv pixels leaves the conv map (36 of ImageNet's 49 blocks) needs no select at all
    const bool interior = py0 > 0 && px0 > 0 && 2 * (py0 + 7) + 1 < Hc && 2 * (px0 + 7) + 1 < Wc;
#pragma unroll 1
    for (int c = 0; c < 2; ++c) {
    v4i wf[7];
#pragma unroll
    for (int kh = 0; kh < 7; ++kh)
        wf[kh] = *reinterpret_cast<const v4i *>(wgt + ((c * 32 + cperm(l31)) * 7 + kh) * 32 + h * 16);
    int best[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) best[r] = (int)0x80000000;
#pragma unroll 1
    for (int dy = 0; dy < (HAWQ_DBG_BIT(dbg, 32) ? 1 : 3); ++dy) {  // not unrolled: keeps the 9 window tiles from living at once
        const int cy = 2 * py + dy - 1;
#pragma unroll 1
        for (int dx = 0; dx < 3; ++dx) {
            const int cx = 2 * px + dx - 1;
            const bool cvalid = (unsigned)cy < (unsigned)Hc && (unsigned)cx < (unsigned)Wc;
            v16i acc0;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[r] = 0;
            const char *wbase = dx == 1 ? pbase8 : pbase + 8 * dx;   // patch bytes [8 dx, 8 dx + 16) of the row segment, 16-byte aligned
#pragma unroll
            for (int kh = 0; kh < 7; ++kh) {
                const v4i af = *reinterpret_cast<const v4i *>(wbase + ((2 * dy + kh) * SF_PW) * 4);
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[kh], af, acc0, 0, 0, 0);
            }
            // window positions that can fall outside the conv map (max-pool padding = -inf): the top row / left column, and - for
            // odd conv sizes only - the bottom row / right column of the last pooled pixel
            if (!interior && (dy == 0 || dx == 0 || (((Hc | Wc) & 1) && (dy == 2 || dx == 2)))) {
#pragma unroll
                for (int r = 0; r < 16; ++r) best[r] = max(best[r], cvalid ? acc0[r] : (int)0x80000000);
            } else {   // 4 of the 9 positions (ImageNet's even 112 x 112 map) are inside for every pooled pixel: no select
#pragma unroll
                for (int r = 0; r < 16; ++r) best[r] = max(best[r], acc0[r]);
            }
        }
    }
    {
        if (!pvalid || HAWQ_DBG_BIT(dbg, 64)) continue;
        const int ch = c * 32 + h * 16;
        int r16[16], qa[16];
        if (fast) {  // host-proved tie-free tables (hawq_amd.quant_utils.tables_are_fast): fused constants from LDS, 3-instruction requants
            const DyNt dq = dynt_prepare(mq, eq);
            const int lo0 = max(a_lo, 0);   // QuantAct16's clamp and the ReLU behind it in one v_med3 (a_hi >= 0)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const v4i t4 = *reinterpret_cast<const v4i *>(ctab_s + 4 * (ch + j));
                DyNt d;
                d.m = t4.x, d.s = t4.y & 31, d.k = t4.y >> 8;
                d.add = (long long)(((unsigned long long)(unsigned)t4.w << 32) | (unsigned)t4.z);
                const int v = med3i(dyadic_nt(best[j], d), lo0, a_hi);
                r16[j] = v;
                qa[j] = med3i(dyadic_nt(v, dq), q_lo, q_hi);
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4i b4 = *reinterpret_cast<const v4i *>(bias + ch + 4 * g), m4 = *reinterpret_cast<const v4i *>(m + ch + 4 * g),
                          e4 = *reinterpret_cast<const v4i *>(e + ch + 4 * g);
                const int bb[4] = {b4.x, b4.y, b4.z, b4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, ee[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int v = max(clampi(dyadic_rne(best[4 * g + j] + bb[j], mm[j], ee[j]), a_lo, a_hi), 0);
                    r16[4 * g + j] = v;
                    qa[4 * g + j] = clampi(dyadic_rne(v, mq, eq), q_lo, q_hi);
                }
            }
        }
        if (res_out) {
            v4i lo, hi;
            lo.x = r16[0] | (r16[1] << 16), lo.y = r16[2] | (r16[3] << 16), lo.z = r16[4] | (r16[5] << 16), lo.w = r16[6] | (r16[7] << 16);
            hi.x = r16[8] | (r16[9] << 16), hi.y = r16[10] | (r16[11] << 16), hi.z = r16[12] | (r16[13] << 16), hi.w = r16[14] | (r16[15] << 16);
            v4i *dst = reinterpret_cast<v4i *>(res_out + opix * 64 + ch);
            dst[0] = lo;
            dst[1] = hi;
        }
        if (out_q) {
            if (out_bits == 8) {
                v4i w = {(int)pack4_i8(qa[0], qa[1], qa[2], qa[3]), (int)pack4_i8(qa[4], qa[5], qa[6], qa[7]),
                         (int)pack4_i8(qa[8], qa[9], qa[10], qa[11]), (int)pack4_i8(qa[12], qa[13], qa[14], qa[15])};
                *reinterpret_cast<v4i *>((int8_t *)out_q + opix * 64 + ch) = w;
            } else {
                v2i w = {(int)pack8_u4(&qa[0]), (int)pack8_u4(&qa[8])};
                *reinterpret_cast<v2i *>((uint8_t *)out_q + ((opix * 64 + ch) >> 1)) = w;
            }
        }
    }
    }  // c
}

// ------------------------------------------------------------------ 3x3/2 max-pool (+ QuantAct)
// thread = 8 channels of one output pixel (16 B of uint16)
__global__ __launch_bounds__(256) void maxpool_requant_kernel(const uint16_t *__restrict__ in, int N, int H, int W,
                                                              int C, int Ho, int Wo, uint16_t *__restrict__ res_out,
                                                              void *__restrict__ out_q, int out_bits, int mq, int eq,
                                                              int q_lo, int q_hi) {
    const int cg = C >> 3;
    const long long total = (long long)N * Ho * Wo * cg;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int g = (int)(i % cg);
        long long r = i / cg;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho);
        const int n = (int)(r / Ho);
        int best[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) best[k] = 0;  // inputs are >= 0 (post-ReLU) and the window is never empty
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy - 1 + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * ox - 1 + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const v4i v = *reinterpret_cast<const v4i *>(in + (((long long)n * H + iy) * W + ix) * C + g * 8);
                const int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    best[2 * k] = max(best[2 * k], w[k] & 0xffff);
                    best[2 * k + 1] = max(best[2 * k + 1], (int)((unsigned)w[k] >> 16));
                }
            }
        }
        const long long elem = (((long long)n * Ho + oy) * Wo + ox) * C + g * 8;
        if (res_out) {
            v4i o;
            o.x = best[0] | (best[1] << 16);
            o.y = best[2] | (best[3] << 16);
            o.z = best[4] | (best[5] << 16);
            o.w = best[6] | (best[7] << 16);
            *reinterpret_cast<v4i *>(res_out + elem) = o;
        }
        if (out_q) {
            int q[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) q[k] = clampi(dyadic_rne(best[k], mq, eq), q_lo, q_hi);
            if (out_bits == 8) {
                v2i o;
                o.x = (int)pack4_i8(q[0], q[1], q[2], q[3]);
                o.y = (int)pack4_i8(q[4], q[5], q[6], q[7]);
                *reinterpret_cast<v2i *>((int8_t *)out_q + elem) = o;
            } else {
                *reinterpret_cast<uint32_t *>((uint8_t *)out_q + (elem >> 1)) = pack8_u4(q);
            }
        }
    }
}

// ------------------------------------------------------------------ stand-alone block-input QuantAct
__global__ __launch_bounds__(256) void requant_residual_kernel(const void *__restrict__ in, int in_bits, long long n8,
                                                               void *__restrict__ out_q, int out_bits, int mq, int eq,
                                                               int q_lo, int q_hi) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        int v[8];
        if (in_bits == 16) {
            const v4i t = reinterpret_cast<const v4i *>(in)[i];
            const int w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[2 * k] = w[k] & 0xffff;
                v[2 * k + 1] = (int)((unsigned)w[k] >> 16);
            }
        } else {
            const v4i a = reinterpret_cast<const v4i *>(in)[2 * i], b = reinterpret_cast<const v4i *>(in)[2 * i + 1];
            v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
        }
        int q[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) q[k] = clampi(dyadic_rne(v[k], mq, eq), q_lo, q_hi);
        if (out_bits == 8) {
            v2i o;
            o.x = (int)pack4_i8(q[0], q[1], q[2], q[3]);
            o.y = (int)pack4_i8(q[4], q[5], q[6], q[7]);
            reinterpret_cast<v2i *>(out_q)[i] = o;
        } else {
            reinterpret_cast<uint32_t *>(out_q)[i] = pack8_u4(q);
        }
    }
}

// ------------------------------------------------------------------ global average pool + QuantAct
// trunc(sum/HW + 0.01) in exact rationals (== floor(sum/HW) for sum >= 0), then the next QuantAct's table
__device__ __forceinline__ void avgpool_finish(long long s, int HW, int i, int8_t *out, int32_t *pooled, int mq, int eq,
                                               int q_lo, int q_hi) {
    const long long p = (100 * s + HW) / (100ll * HW);
    if (pooled) pooled[i] = (int32_t)p;
    out[i] = (int8_t)clampi(dyadic_rne((int32_t)p, mq, eq), q_lo, q_hi);
}

// general form: thread = one (n, c), strided over HW (int32 residuals, odd channel counts)
__global__ __launch_bounds__(256) void avgpool_requant_kernel(const void *__restrict__ in, int in_bits, int N, int HW,
                                                              int C, int8_t *__restrict__ out,
                                                              int32_t *__restrict__ pooled, int mq, int eq, int q_lo,
                                                              int q_hi) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    long long s = 0;
    for (int k = 0; k < HW; ++k) {
        const size_t idx = ((size_t)n * HW + k) * C + c;
        s += in_bits == 16 ? (long long)((const uint16_t *)in)[idx] : (long long)((const int32_t *)in)[idx];
    }
    avgpool_finish(s, HW, i, out, pooled, mq, eq, q_lo, q_hi);
}

// uint16 residuals, C % 256 == 0: workgroup = (image, 256 channels); lane group g = t / 32 takes pixels g, g+8, ...
// with 16-byte loads (8 channels per lane: a wave instruction reads two full 512-byte pixel rows), partial sums
// meet in LDS.  HW * 65535 < 2^31 is checked by the launcher.
__global__ __launch_bounds__(256) void avgpool_requant_u16_kernel(const uint16_t *__restrict__ in, int HW, int C,
                                                                  int8_t *__restrict__ out, int32_t *__restrict__ pooled,
                                                                  int mq, int eq, int q_lo, int q_hi) {
    __shared__ int part[8][256];
    const int t = threadIdx.x, cg = t & 31, g = t >> 5;
    const int n = blockIdx.x / (C >> 8), c0 = (blockIdx.x % (C >> 8)) << 8;
    int s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint16_t *base = in + (size_t)n * HW * C + c0 + cg * 8;
    for (int k = g; k < HW; k += 8) {
        const uint4 v = *reinterpret_cast<const uint4 *>(base + (size_t)k * C);
        s[0] += v.x & 0xffff, s[1] += v.x >> 16, s[2] += v.y & 0xffff, s[3] += v.y >> 16;
        s[4] += v.z & 0xffff, s[5] += v.z >> 16, s[6] += v.w & 0xffff, s[7] += v.w >> 16;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) part[g][cg * 8 + j] = s[j];
    __syncthreads();
    long long tot = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) tot += part[j][t];
    avgpool_finish(tot, HW, n * C + c0 + t, out, pooled, mq, eq, q_lo, q_hi);
}

inline int grid_for(long long work_items) {
    long long g = (work_items + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" int hawq_quantize_input(const float *x, int8_t *out, int32_t N, int32_t C, int32_t H, int32_t W,
                                   int32_t out_h, int32_t out_w, int32_t pad_top, int32_t pad_left, float inv_scale,
                                   int32_t lo, int32_t hi, void *stream) {
    HAWQ_REQUIRE(x && out, "hawq_quantize_input: null pointer");
    HAWQ_REQUIRE(C >= 1 && C <= 4, "hawq_quantize_input: C=%d must be 1..4", C);
    HAWQ_REQUIRE(N > 0 && H > 0 && W > 0 && out_h >= H + pad_top && out_w >= W + pad_left,
                 "hawq_quantize_input: bad geometry");
    hipLaunchKernelGGL(quantize_input_kernel, dim3(grid_for((long long)N * H * W)), dim3(256), 0, (hipStream_t)stream,
                       x, out, N, C, H, W, out_h, out_w, pad_top, pad_left, inv_scale, lo, hi);
    HAWQ_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int hawq_stem_conv7(const int8_t *in, const int8_t *wgt, const int32_t *bias, const int32_t *m,
                               const int32_t *e, int32_t N, int32_t Hp, int32_t Wp, int32_t Ho, int32_t Wo,
                               int32_t q_lo, int32_t q_hi, uint16_t *out16, int32_t *out_acc, void *stream) {
    HAWQ_REQUIRE(in && wgt && bias, "hawq_stem_conv7: null pointer");
    HAWQ_REQUIRE(out16 || out_acc, "hawq_stem_conv7: no output requested");
    HAWQ_REQUIRE(!out16 || (m && e), "hawq_stem_conv7: out16 needs m/e tables");
    HAWQ_REQUIRE(Hp >= 2 * (Ho - 1) + 7 && Wp >= 2 * (Wo - 1) + 8 && (Wp % 2) == 0,
                 "hawq_stem_conv7: padded input %dx%d too small for output %dx%d", Hp, Wp, Ho, Wo);
    const long long M = (long long)N * Ho * Wo;
    HAWQ_REQUIRE(M > 0 && M < (1ll << 30), "hawq_stem_conv7: bad size");
    const long long tiles = (M + 63) / 64;
    long long grid = (tiles + 3) / 4;
    if (grid > 2048) grid = 2048;
    if (!m) m = bias, e = bias;  // never dereferenced for out16 == NULL, keeps loads in-bounds
    hipLaunchKernelGGL(stem_conv7_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, in, wgt, bias, m, e, N,
                       Hp, Wp, Ho, Wo, q_lo, q_hi, out16, out_acc);
    HAWQ_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int hawq_maxpool3s2_requant(const uint16_t *in, int32_t N, int32_t H, int32_t W, int32_t C,
                                       uint16_t *res_out, void *out_q, int32_t out_bits, int32_t mq, int32_t eq,
                                       int32_t q_lo, int32_t q_hi, void *stream) {
    HAWQ_REQUIRE(in && (res_out || out_q), "hawq_maxpool3s2_requant: null pointer");
    HAWQ_REQUIRE(C > 0 && C % 8 == 0, "hawq_maxpool3s2_requant: C must be a multiple of 8");
    HAWQ_REQUIRE(!out_q || out_bits == 8 || out_bits == 4, "hawq_maxpool3s2_requant: out_bits 4/8");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool_requant_kernel, dim3(grid_for((long long)N * Ho * Wo * (C / 8))), dim3(256), 0,
                       (hipStream_t)stream, in, N, H, W, C, Ho, Wo, res_out, out_q, out_bits, mq, eq, q_lo, q_hi);
    HAWQ_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int hawq_requant_residual(const void *in, int32_t in_bits, int64_t n, void *out_q, int32_t out_bits,
                                     int32_t mq, int32_t eq, int32_t q_lo, int32_t q_hi, void *stream) {
    HAWQ_REQUIRE(in && out_q, "hawq_requant_residual: null pointer");
    HAWQ_REQUIRE(n > 0 && n % 8 == 0, "hawq_requant_residual: n must be a positive multiple of 8");
    HAWQ_REQUIRE((in_bits == 16 || in_bits == 32) && (out_bits == 8 || out_bits == 4),
                 "hawq_requant_residual: in_bits 16/32, out_bits 4/8");
    hipLaunchKernelGGL(requant_residual_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, in, in_bits,
                       (long long)(n / 8), out_q, out_bits, mq, eq, q_lo, q_hi);
    HAWQ_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int hawq_avgpool_requant(const void *in, int32_t in_bits, int32_t N, int32_t HW, int32_t C, int8_t *out,
                                    int32_t *pooled_out, int32_t mq, int32_t eq, int32_t q_lo, int32_t q_hi,
                                    void *stream) {
    HAWQ_REQUIRE(in && out, "hawq_avgpool_requant: null pointer");
    HAWQ_REQUIRE(in_bits == 16 || in_bits == 32, "hawq_avgpool_requant: in_bits 16/32");
    HAWQ_REQUIRE(N > 0 && HW > 0 && C > 0, "hawq_avgpool_requant: bad geometry");
    if (in_bits == 16 && C % 256 == 0 && HW <= 32768 && (reinterpret_cast<size_t>(in) & 15) == 0)
        hipLaunchKernelGGL(avgpool_requant_u16_kernel, dim3(N * (C / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t *)in, HW, C, out, pooled_out, mq, eq, q_lo, q_hi);
    else
        hipLaunchKernelGGL(avgpool_requant_kernel, dim3((N * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, in,
                           in_bits, N, HW, C, out, pooled_out, mq, eq, q_lo, q_hi);
    HAWQ_CHECK_HIP(hipGetLastError());
    return 0;
}

namespace {
int stem_fused_launch(const float *x, const uint8_t *xu8, const int8_t *lut, int32_t N, int32_t C, int32_t H, int32_t W,
                      float inv_scale, int32_t in_lo, int32_t in_hi, const int8_t *wgt, const int32_t *bias, const int32_t *m,
                      const int32_t *e, int32_t a_lo, int32_t a_hi, uint16_t *res_out, void *out_q, int32_t out_bits,
                      int32_t mq, int32_t eq, int32_t q_lo, int32_t q_hi, int32_t fast_tables, void *stream) {
    HAWQ_REQUIRE((x || (xu8 && lut)) && wgt && bias && m && e, "hawq_stem_fused: null pointer");
    HAWQ_REQUIRE(!fast_tables || ((eq & 0xff) >= 33 && (eq & 0xff) <= 62), "hawq_stem_fused: fast_tables needs eq in [33,62]");
    HAWQ_REQUIRE(res_out || out_q, "hawq_stem_fused: no output requested");
    HAWQ_REQUIRE(C >= 1 && C <= 3 && N > 0 && H >= 7 && W >= 7, "hawq_stem_fused: bad geometry (C <= 3)");
    HAWQ_REQUIRE(!out_q || out_bits == 8 || out_bits == 4, "hawq_stem_fused: out_bits 4/8");
    HAWQ_REQUIRE(a_lo >= -32768 && a_hi <= 65535 && a_hi >= 0, "hawq_stem_fused: 16-bit activation range expected");
    HAWQ_REQUIRE(!lut || (reinterpret_cast<size_t>(lut) & 3) == 0, "hawq_stem_fused_u8: lut must be 4-byte aligned");
    const int Hc = (H + 6 - 7) / 2 + 1, Wc = (W + 6 - 7) / 2 + 1;    // conv 7x7 / 2, pad 3
    const int Hp = (Hc + 2 - 3) / 2 + 1, Wp = (Wc + 2 - 3) / 2 + 1;  // max-pool 3x3 / 2, pad 1
    const long long blocks = (long long)((Wp + 7) / 8) * ((Hp + 7) / 8) * ((N + 1) / 2);
    HAWQ_REQUIRE(blocks < (1ll << 31), "hawq_stem_fused: problem too large");
    const int dbg = HAWQ_DBG_ENV();
    if (xu8)
        hipLaunchKernelGGL(stem_fused_kernel<true>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, xu8, lut, N, C, H,
                           W, inv_scale, in_lo, in_hi, wgt, bias, m, e, a_lo, a_hi, Hc, Wc, Hp, Wp, res_out, out_q, out_bits,
                           mq, eq, q_lo, q_hi, fast_tables, dbg);
    else
        hipLaunchKernelGGL(stem_fused_kernel<false>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, xu8, lut, N, C, H,
                           W, inv_scale, in_lo, in_hi, wgt, bias, m, e, a_lo, a_hi, Hc, Wc, Hp, Wp, res_out, out_q, out_bits,
                           mq, eq, q_lo, q_hi, fast_tables, dbg);
    HAWQ_CHECK_HIP(hipGetLastError());
    return 0;
}
}  // namespace

extern "C" int hawq_stem_fused(const float *x, int32_t N, int32_t C, int32_t H, int32_t W, float inv_scale, int32_t in_lo,
                               int32_t in_hi, const int8_t *wgt, const int32_t *bias, const int32_t *m, const int32_t *e,
                               int32_t a_lo, int32_t a_hi, uint16_t *res_out, void *out_q, int32_t out_bits, int32_t mq,
                               int32_t eq, int32_t q_lo, int32_t q_hi, int32_t fast_tables, void *stream) {
    HAWQ_REQUIRE(x, "hawq_stem_fused: null pointer");
    return stem_fused_launch(x, nullptr, nullptr, N, C, H, W, inv_scale, in_lo, in_hi, wgt, bias, m, e, a_lo, a_hi, res_out,
                             out_q, out_bits, mq, eq, q_lo, q_hi, fast_tables, stream);
}

extern "C" int hawq_stem_fused_u8(const uint8_t *x, const int8_t *lut, int32_t N, int32_t C, int32_t H, int32_t W,
                                  const int8_t *wgt, const int32_t *bias, const int32_t *m, const int32_t *e, int32_t a_lo,
                                  int32_t a_hi, uint16_t *res_out, void *out_q, int32_t out_bits, int32_t mq, int32_t eq,
                                  int32_t q_lo, int32_t q_hi, int32_t fast_tables, void *stream) {
    HAWQ_REQUIRE(x && lut, "hawq_stem_fused_u8: null pointer");
    return stem_fused_launch(nullptr, x, lut, N, C, H, W, 0.f, -128, 127, wgt, bias, m, e, a_lo, a_hi, res_out, out_q, out_bits,
                             mq, eq, q_lo, q_hi, fast_tables, stream);
}
