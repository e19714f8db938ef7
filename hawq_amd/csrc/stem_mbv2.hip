// MobileNetV2's init block as ONE launch (round 4): input QuantAct + 3x3 / stride 2 / pad 1 convolution on 3 channels + ReLU6 (as ReLU +
// clamp) + quant_act_int32 + the first unit's block-input QuantAct.
//
// Reference path: Q_MobileNetV2.forward, q_mobilenetv2.py:176-186 (quant_input -> init_block -> quant_act_int32 -> the first
// Q_LinearBottleneck's quant_act, :60-65) with the input case of QuantAct (quant_modules.py:271-274).
//
// Round 3 ran it as two launches: hawq_quantize_im2col3x3s2 wrote one 64-byte patch row per output pixel (103 MB at batch 128, 27 of
// the 64 bytes real) and a 1x1 hawq_conv2d read them back: 43 + 87 us.  Here a workgroup owns an 8 x 16 tile of output pixels:
//   1. the tile's 3 x 17 x 33 input window is quantised ONCE per element into LDS (fp32 NCHW through `fl(1/S) * x`, or uint8 NHWC through
//      the ToTensor + Normalize + QuantAct look-up table), zeros outside the image;
//   2. every lane gathers its pixel's half of the (kh, kw, c) patch (16 of the 27 + 5 bytes) - the B operand of ONE v_mfma_i32_32x32x32_i8
//      against the same K = 64 weight rows the im2col path uses (lane = pixel, 16 registers = 16 consecutive channels);
//   3. the closing arithmetic of hawq_conv2d's direct RESIDUAL epilogue (3-instruction requants against the fused constants).
// HBM bytes: the image once (77 MB fp32 / 19 MB uint8 at batch 128) + the int8 block input of unit 1 (51 MB).
#include "common.h"

namespace {

struct StemP {
    const float *x;
    const uint8_t *xu;
    const int8_t *lut;
    int N, H, W, Ho, Wo;
    float inv_scale;
    int in_lo, in_hi;
    const int8_t *wgt;   // [64][64] int8, row = output channel, bytes = the 27 taps in (kh, kw, c) order, then zeros
    const int32_t *ctab;
    int relu, clamp16, mq, eq, q_lo, q_hi, out_pitch;
    int8_t *out_q;
    int32_t *res_out;
    int tiles_x, tiles_y;
};

constexpr int ST_TH = 8, ST_TW = 16, ST_WH = 2 * ST_TH + 1, ST_WW = 2 * ST_TW + 1, ST_RP = 36, ST_PL = ST_WH * ST_RP;

// LDS byte offset of patch tap t (kh, kw, c order) relative to the pixel's window origin; K bytes 27 .. 31 meet zero weights, any
// readable byte will do for them
__host__ __device__ constexpr int tap_off(int t) { return t < 27 ? (t % 3) * ST_PL + (t / 9) * ST_RP + (t % 9) / 3 : 0; }

template <bool U8, bool TIE>
__global__ __launch_bounds__(256) void stem3x3s2_kernel(const StemP p) {
    __shared__ __attribute__((aligned(16))) int8_t qs[3 * ST_PL];
    __shared__ int8_t lut_s[U8 ? 768 : 4];
    __shared__ v4i cts[32];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, h = lane >> 5;
    int bid = blockIdx.x;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y, n = bid / p.tiles_y;
    const int oy0 = ty * ST_TH, ox0 = tx * ST_TW, iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;
    // weights of this lane's MFMA row (channel cperm(l31), K bytes 16 h ..) and the table rows: issued first, used last
    const v4i wf = *reinterpret_cast<const v4i *>(p.wgt + (size_t)cperm(l31) * 64 + h * 16);
    if (t < 32) cts[t] = *reinterpret_cast<const v4i *>(p.ctab + t * 4);
    if constexpr (U8) {
        if (t < 192) reinterpret_cast<int *>(lut_s)[t] = reinterpret_cast<const int *>(p.lut)[t];
        __syncthreads();
    }
    // 1. quantised window -> LDS.  A thread keeps its window column (fp32: wx = t % 36, seven groups of threads walk the 3 x 17 (channel,
    //    row) pairs; uint8: byte t % 100 of the 99-byte NHWC window row, two groups walk the 17 rows): no per-element index arithmetic
    //    beyond an add, and consecutive lanes read consecutive addresses.
    if constexpr (U8) {
        const int bx = t % 100, rg = t / 100, wx = bx / 3, c = bx - wx * 3, ix = ix0 + wx;
        const bool mine = bx < 3 * ST_WW && rg < 2, cok = (unsigned)ix < (unsigned)p.W;
        const int ixc = min(max(ix, 0), p.W - 1);
        // (24-bit multiplies: full-rate VALU; image rows / planes are far below 2^24 elements)
        const uint8_t *xb = p.xu + (size_t)n * p.H * p.W * 3 + ixc * 3 + c;
        const unsigned w3 = (unsigned)p.W * 3;
        unsigned char u[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int iyc = min(max(iy0 + min(rg + 2 * i, ST_WH - 1), 0), p.H - 1);
            u[i] = xb[__umul24((unsigned)iyc, w3)];
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int wy = rg + 2 * i;
            const int q = (cok && (unsigned)(iy0 + wy) < (unsigned)p.H) ? (int)lut_s[c * 256 + u[i]] : 0;
            if (mine && wy < ST_WH) qs[c * ST_PL + wy * ST_RP + wx] = (int8_t)q;
        }
    } else {
        // all eight loads of a thread are issued before the first is used (clamped addresses, no branches around them)
        const int wx = t % 36, rg = t / 36, ix = ix0 + wx;
        const bool mine = wx < ST_WW && rg < 7, cok = (unsigned)ix < (unsigned)p.W;
        const int ixc = min(max(ix, 0), p.W - 1);
        const float flo = (float)p.in_lo, fhi = (float)p.in_hi;
        // (no divisions, 24-bit multiplies: the integer arithmetic of these addresses was a third of the kernel's VALU time)
        const unsigned hw = (unsigned)p.H * (unsigned)p.W;
        const float *xb = p.x + (size_t)n * 3 * hw + ixc;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pr = min(rg + 7 * i, 3 * ST_WH - 1), c = (pr >= ST_WH) + (pr >= 2 * ST_WH), wy = pr - c * ST_WH;   // (channel, window row) pair rg + 7 i
            const int iyc = min(max(iy0 + wy, 0), p.H - 1);
            v[i] = xb[__umul24((unsigned)c, hw) + __umul24((unsigned)iyc, (unsigned)p.W)];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pr = rg + 7 * i, c = (pr >= ST_WH) + (pr >= 2 * ST_WH), wy = pr - c * ST_WH;
            float rr = rintf(__fmul_rn(p.inv_scale, v[i]));   // one binary32 rounding, as `1. / scale * input` has
            rr = fminf(fmaxf(rr, flo), fhi);
            const int q = (cok && (unsigned)(iy0 + wy) < (unsigned)p.H) ? (int)rr : 0;
            if (mine && pr < 3 * ST_WH) qs[c * ST_PL + wy * ST_RP + wx] = (int8_t)q;
        }
    }
    __syncthreads();
    // 2. this lane's half of its pixel's patch: taps 16 h .. 16 h + 15
    const int pl = wave * 32 + l31, py = pl >> 4, px = pl & 15;
    const int8_t *org = qs + (2 * py) * ST_RP + 2 * px;
    int b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) b[i] = org[h ? tap_off(16 + i) : tap_off(i)];
    v4i af;
#pragma unroll
    for (int g = 0; g < 4; ++g) af[g] = (int)pack4_i8(b[4 * g], b[4 * g + 1], b[4 * g + 2], b[4 * g + 3]);
    v16i acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0;
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, af, acc, 0, 0, 0);
    // 3. closing
    const int gy = oy0 + py, gx = ox0 + px;
    if (gy >= p.Ho || gx >= p.Wo) return;
    const int ch = h * 16;
    if (ch >= p.out_pitch) return;
    const size_t elem = (((size_t)n * p.Ho + gy) * p.Wo + gx) * p.out_pitch + ch;
    DyNt dq = dynt_prepare(p.mq, p.eq);
    asm volatile("" : "+v"(dq.add));
    const int rlo = p.relu ? 0 : (int)0x80000000;
    const int clo = p.clamp16 ? -32768 : (int)0x80000000, chi = p.clamp16 ? 32767 : 0x7fffffff;
    int qw[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        int o[4], qv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const v4i t4 = cts[ch + 4 * g + k];
            DyNt d;
            d.m = t4.x, d.s = t4.y & 31, d.k = t4.y >> 8;
            d.add = (long long)(((unsigned long long)(unsigned)t4.w << 32) | (unsigned)t4.z);
            asm volatile("" : "+v"(d.add));
            int ov = TIE ? dyadic_tie(acc[4 * g + k], d) : dyadic_nt(acc[4 * g + k], d);
            ov = med3i(max(ov, rlo), clo, chi);
            o[k] = ov;
            qv[k] = med3i(TIE ? dyadic_tie(ov, dq) : dyadic_nt(ov, dq), p.q_lo, p.q_hi);
        }
        if (p.res_out) *reinterpret_cast<v4i *>(p.res_out + elem + 4 * g) = v4i{o[0], o[1], o[2], o[3]};
        qw[g] = pack4_fast(qv[0], qv[1], qv[2], qv[3]);
    }
    if (p.out_q) *reinterpret_cast<v4i *>(p.out_q + elem) = v4i{qw[0], qw[1], qw[2], qw[3]};
}

const char *stem_refusal(const float *x, const uint8_t *xu, const int8_t *lut, int H, int W, const hawq_conv_args *c) {
    if (!c) return "null conv description";
    if ((x != nullptr) == (xu != nullptr)) return "exactly one of x (fp32 NCHW) and x_u8 (uint8 NHWC) must be given";
    if (xu && !lut) return "uint8 images need the look-up table";
    if (H <= 0 || W <= 0 || c->N <= 0) return "empty input";
    if (c->H != (H - 1) / 2 + 1 || c->W != (W - 1) / 2 + 1) return "conv->H / W must be the 3x3 / stride 2 / pad 1 output grid";
    if (c->KH != 1 || c->KW != 1 || c->stride != 1 || c->pad != 0 || c->Cin != 64 || c->Cout != 64 || c->in_bits != 8 || c->w_bits != 8 || c->in2)
        return "conv: the 1x1 / K = 64 form of the im2col path (27 taps in (kh, kw, c) order), 64 packed output rows";
    if (c->epilogue != HAWQ_EPI_RESIDUAL || !c->fast_tables || !c->ctab || c->res_in || !c->wgt) return "conv: RESIDUAL epilogue without identity, fast_tables with ctab";
    if (c->out_pitch != 16 && c->out_pitch != 32) return "out_pitch must be 16 or 32 (at most 32 output channels)";
    if (c->res_out && c->res_out_bits != 32) return "res_out must be the int32 carrier";
    if (!c->out_q && !c->res_out) return "nothing to write";
    if (c->out_q && (c->out_bits != 8 || c->q_lo < -128 || c->q_hi > 127 || c->q_lo > c->q_hi || c->mq < 0 || (c->eq & 0xff) < 33 || (c->eq & 0xff) > 62))
        return "int8 out_q with a fast (mq, eq)";
    return nullptr;
}

}  // namespace

extern "C" int hawq_stem3x3s2_ok(const float *x, const uint8_t *x_u8, const int8_t *lut, int32_t H, int32_t W, const hawq_conv_args *conv) {
    return stem_refusal(x, x_u8, lut, H, W, conv) == nullptr ? 1 : 0;
}

extern "C" int hawq_stem3x3s2(const float *x, const uint8_t *x_u8, const int8_t *lut, int32_t H, int32_t W, float inv_scale, int32_t in_lo,
                              int32_t in_hi, const hawq_conv_args *c, void *stream) {
    const char *why = stem_refusal(x, x_u8, lut, H, W, c);
    HAWQ_REQUIRE(!why, "hawq_stem3x3s2: %s", why);
    HAWQ_REQUIRE(x_u8 || (in_lo >= -128 && in_hi <= 127 && in_lo <= in_hi), "hawq_stem3x3s2: the input clamp must fit int8");
    StemP p;
    p.x = x, p.xu = x_u8, p.lut = lut;
    p.N = c->N, p.H = H, p.W = W, p.Ho = c->H, p.Wo = c->W;
    p.inv_scale = inv_scale, p.in_lo = in_lo, p.in_hi = in_hi;
    p.wgt = (const int8_t *)c->wgt, p.ctab = c->ctab;
    p.relu = !c->res_no_relu, p.clamp16 = c->res_clamp16;
    p.mq = c->out_q ? c->mq : 0, p.eq = c->out_q ? c->eq : 33, p.q_lo = c->q_lo, p.q_hi = c->q_hi, p.out_pitch = c->out_pitch;
    p.out_q = (int8_t *)c->out_q, p.res_out = (int32_t *)c->res_out;
    p.tiles_x = (p.Wo + ST_TW - 1) / ST_TW, p.tiles_y = (p.Ho + ST_TH - 1) / ST_TH;
    const long long grid = (long long)p.N * p.tiles_x * p.tiles_y;
    HAWQ_REQUIRE(grid <= 0x7fffffffll, "hawq_stem3x3s2: grid too large");
    const bool tie = (c->fast_tables & 4) != 0;
    auto fn = x_u8 ? (tie ? stem3x3s2_kernel<true, true> : stem3x3s2_kernel<true, false>) : (tie ? stem3x3s2_kernel<false, true> : stem3x3s2_kernel<false, false>);
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p);
    HAWQ_CHECK_HIP(hipGetLastError());
    return 0;
}
