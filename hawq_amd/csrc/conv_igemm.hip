// Implicit-GEMM integer convolution for gfx950 (MI355X) with fused HAWQ epilogues.
//
// GEMM view: D[Cout][M] = W[Cout][K] * X[K][M],  M = N*Ho*Wo output pixels,
// K = KH*KW*Cin.  Weights are the MFMA "A" operand (rows = output channels), pixels the "B"
// operand (cols), so that after v_mfma_i32_32x32x32_i8 every lane owns ONE pixel and, thanks
// to a row permutation applied when the weight fragment is read from LDS, 16 CONSECUTIVE
// output channels of it: the epilogue then stores 16 int8 (one dwordx4) / 16 uint16 (two
// dwordx4) per lane straight into the NHWC output.
//
// K is walked in chunks of 64 input channels of one filter tap.  Each chunk of both operands is
// fetched global->registers (16 B per thread per row, im2col addressing with zero fill for
// padding), unpacked to int8 if the tensor is 4-bit (hawq4 nibble format), and written to a
// double-buffered, XOR-swizzled LDS tile [rows][64 B]; fragments are read back with
// conflict-free ds_read_b128.  All MACs are exact int32.
//
// Reference arithmetic replaced: quant_modules.py:489-494 (conv), q_resnet.py:242-258 (ReLU,
// residual add), quant_utils.py:390-456 (fixedpoint_fn case 0 / case 1).
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace {

struct ConvP {
    const uint8_t *in, *wgt;
    const int32_t *bias;
    int N, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo, M;
    int in_bits, w_bits;
    const uint8_t *in2, *wgt2;
    const int32_t *bias2;
    int H2, W2, Cin2, stride2, in2_bits, w2_bits;
    int relu;
    const int32_t *m, *e, *m_id, *e_id;
    int m_id_s, e_id_s;
    const void *res_in;
    int res_in_bits;
    void *res_out;
    int res_out_bits;
    void *out_q;
    int out_bits, q_lo, q_hi, mq, eq;
    int32_t *out_acc;
    float *out_f32;
    const float *fscale;
    int ldo, n_valid;
    int32_t *flags;
    const int32_t *ctab, *ctab_id;
    int k0;   // fast path requant mode (dyadic_mode): 0 tie-free, 1 tie-free + every pre-shift 0, 2 exact tie handling
    int ck0;  // every PER-CHANNEL pre-shift (ctab, ctab_id) is zero (fast_tables bit 3): epilogue_fast<..., CK0 = true>
    int res_no_relu, res_clamp16;   // exact general RESIDUAL epilogue: no ReLU after the sum / clamp to the int16 range (hawq_conv_args)
    int ring_bytes;  // LDS bytes of the operand ring actually allocated (fewer stages when the K loop is shorter than the ring)
    int in_planar, out_planar;  // activation layout of in / out_q: 0 = NHWC rows, 1 = channel-group planes (hawq_mi355.h)
    int gfast;                  // general (direct) RESIDUAL epilogue with the fast contract's arithmetic: 32-bit / signed residuals whose tables the host has proved
    int in_pitch, out_pitch;    // bytes per pixel row of `in` / channels per pixel row of out_q, res_out, res_in (hawq_conv_args, ABI 4; always > 0 here)
    int dbg;  // HAWQ_DBG ablation bits (timing experiments only): 1 = skip operand loads, 2 = skip MFMAs
    long long *dbgbuf;  // HAWQ_DBG & 128: per-phase cycle sums of workgroup 0 / wave 0 (band kernel)
};

template <int BM_, int BN_, int WM_, int WN_, int NS_, int KSUB_ = 1, int MINB_ = 2, int KG_ = 1>
struct Cfg {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
    static constexpr int MINB = MINB_;       // waves per SIMD the register allocation must allow (launch_bounds)
    static constexpr int NS = NS_;           // LDS ring stages of the asynchronous (global_load_lds) pipeline
    static constexpr int KSUB = KSUB_;       // 64-channel sub-chunks per ring stage (one barrier per stage)
    // KG > 1: intra-workgroup split-K.  KG groups of WM x WN waves each own the sub-chunks sub % KG == group of every ring
    // stage - they issue their LDS-DMA and run their MFMAs - and the partial accumulators are summed through LDS before
    // the epilogue.  For the long-K, few-pixel layers (reduce convs of stages 3-4, FC): the K loop of a tile is bound by
    // the per-wave LDS-DMA rate, so KG x the issuing waves per tile shortens it almost KG-fold.
    static constexpr int KG = KG_;
    static constexpr int NWG = WM_ * WN_;    // waves per K group
    static constexpr int NW = NWG * KG;      // waves per workgroup
    static constexpr int NT = NW * 64;       // threads per workgroup
    static constexpr int NTG = NWG * 64;     // threads per K group
    static constexpr int RPP = NTG / 4;      // operand rows staged per pass of a K group (4 lanes x 16 B per 64-B row)
    static constexpr int PT = BM / WM / 32;  // pixel MFMA tiles per wave
    static constexpr int CT = BN / WN / 32;  // channel MFMA tiles per wave
    static constexpr int AL = BM / RPP;      // 16-B A loads per thread per sub-chunk
    static constexpr int WL = BN / RPP;
    static constexpr int STAGE_BYTES = KSUB * (BM + BN) * 64;
    static constexpr int LDS_BYTES = NS * STAGE_BYTES;  // the register-staged path uses the first two stages
    static_assert(NW == 4 || NW == 8 || NW == 16, "4, 8 or 16 waves per workgroup");
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must fill whole staging passes");
    static_assert(KSUB % KG == 0 && (KG == 1 || BM * BN * 4 <= LDS_BYTES), "split-K: sub-chunks per group, partial sums staged on the ring");
};

// 8 bytes of hawq4 (16 channels) -> 16 int8.  Unsigned: zero-extended.  Signed weights come
// out as value*16 (nibble moved to the top of its byte); the accumulator is shifted back by 4
// after the K loop, which is exact.
template <bool SIGNED16>
__device__ __forceinline__ v4i unpack16(unsigned x0, unsigned x1) {
    v4i r;
    if (SIGNED16) {
        r.x = (int)((x0 << 4) & 0xF0F0F0F0u);
        r.y = (int)(x0 & 0xF0F0F0F0u);
        r.z = (int)((x1 << 4) & 0xF0F0F0F0u);
        r.w = (int)(x1 & 0xF0F0F0F0u);
    } else {
        r.x = (int)(x0 & 0x0F0F0F0Fu);
        r.y = (int)((x0 >> 4) & 0x0F0F0F0Fu);
        r.z = (int)(x1 & 0x0F0F0F0Fu);
        r.w = (int)((x1 >> 4) & 0x0F0F0F0Fu);
    }
    return r;
}

template <int BITS>
__device__ __forceinline__ v4i load_chunk16(const uint8_t *p, bool valid) {
    v4i z = {0, 0, 0, 0};
    if (!valid) return z;
    if (BITS == 8) return *reinterpret_cast<const v4i *>(p);
    v2i t = *reinterpret_cast<const v2i *>(p);
    z.x = t.x;
    z.y = t.y;
    return z;
}

// One GEMM segment: acc[ct][pt] += W_tile * X_tile over all taps and input channels.
template <class C, int A_BITS, int W_BITS>
__device__ __forceinline__ void gemm_segment(v16i (&acc)[C::CT][C::PT], const uint8_t *__restrict__ in,
                                             const uint8_t *__restrict__ wgt, int H, int W, int Cin, int KH,
                                             int KW, int stride, int pad, int Ho, int Wo, int M, int Cout,
                                             int m0, int c0, char *smem, int apitch) {
    // apitch: bytes from one pixel row of `in` to the next (Cin * A_BITS / 8 when dense)
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wave_m = wave % C::WM, wave_c = wave / C::WM;
    const int lrow = t >> 2, lslot = t & 3;
    constexpr int ABYTES = A_BITS == 8 ? 16 : 8;  // bytes per 16 channels in global memory
    constexpr int WBYTES = W_BITS == 8 ? 16 : 8;

    // per-thread im2col row bookkeeping
    int pix_base[C::AL], iy0[C::AL], ix0[C::AL];
    bool mval[C::AL];
#pragma unroll
    for (int i = 0; i < C::AL; ++i) {
        const int m = m0 + lrow + C::RPP * i;
        mval[i] = m < M;
        const int mm = mval[i] ? m : 0;
        const int n = mm / (Ho * Wo);
        const int r = mm - n * (Ho * Wo);
        const int oy = r / Wo, ox = r - oy * Wo;
        iy0[i] = oy * stride - pad;
        ix0[i] = ox * stride - pad;
        pix_base[i] = (n * H + iy0[i]) * W + ix0[i];
    }
    const int taps = KH * KW;
    const int cchunks = Cin >> 6;
    const int nk = taps * cchunks;
    const size_t wrow_bytes = (size_t)taps * Cin * W_BITS / 8;

    char *ldsA = smem;                 // [2][BM][64]
    char *ldsW = smem + 2 * C::BM * 64;  // [2][BN][64]

    v4i ra[C::AL], rw[C::WL];
    int kh = 0, kw = 0, cc = 0;  // coordinates of the chunk being LOADED

    auto load_regs = [&]() {
#pragma unroll
        for (int i = 0; i < C::AL; ++i) {
            const int iy = iy0[i] + kh, ix = ix0[i] + kw;
            const bool v = mval[i] && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const size_t off = (size_t)(pix_base[i] + kh * W + kw) * apitch + (size_t)(cc << 6) * A_BITS / 8 + lslot * ABYTES;
            ra[i] = load_chunk16<A_BITS>(in + (v ? off : 0), v);
        }
#pragma unroll
        for (int j = 0; j < C::WL; ++j) {
            const int c = c0 + lrow + C::RPP * j;
            const bool v = c < Cout;
            const size_t off = (size_t)c * wrow_bytes + ((size_t)((kh * KW + kw) * Cin + (cc << 6))) * W_BITS / 8 +
                               lslot * WBYTES;
            rw[j] = load_chunk16<W_BITS>(wgt + (v ? off : 0), v);
        }
        if (++cc == cchunks) {
            cc = 0;
            if (++kw == KW) {
                kw = 0;
                ++kh;
            }
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < C::AL; ++i) {
            v4i v = ra[i];
            if (A_BITS == 4) v = unpack16<false>((unsigned)v.x, (unsigned)v.y);
            *reinterpret_cast<v4i *>(ldsA + buf * C::BM * 64 + lds_off(lrow + C::RPP * i, lslot)) = v;
        }
#pragma unroll
        for (int j = 0; j < C::WL; ++j) {
            v4i v = rw[j];
            if (W_BITS == 4) v = unpack16<true>((unsigned)v.x, (unsigned)v.y);
            *reinterpret_cast<v4i *>(ldsW + buf * C::BN * 64 + lds_off(lrow + C::RPP * j, lslot)) = v;
        }
    };

    const int l31 = lane & 31, h = lane >> 5;
    int arow[C::PT], wrow[C::CT];
#pragma unroll
    for (int p = 0; p < C::PT; ++p) arow[p] = wave_m * (C::PT * 32) + p * 32 + l31;
#pragma unroll
    for (int c = 0; c < C::CT; ++c) wrow[c] = wave_c * (C::CT * 32) + c * 32 + cperm(l31);

    load_regs();
    store_lds(0);
    __syncthreads();
    for (int k = 0; k < nk; ++k) {
        const int buf = k & 1;
        if (k + 1 < nk) load_regs();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int slot = 2 * ks + h;
            v4i wf[C::CT], af[C::PT];
#pragma unroll
            for (int c = 0; c < C::CT; ++c)
                wf[c] = *reinterpret_cast<const v4i *>(ldsW + buf * C::BN * 64 + lds_off(wrow[c], slot));
#pragma unroll
            for (int p = 0; p < C::PT; ++p)
                af[p] = *reinterpret_cast<const v4i *>(ldsA + buf * C::BM * 64 + lds_off(arow[p], slot));
#pragma unroll
            for (int c = 0; c < C::CT; ++c)
#pragma unroll
                for (int p = 0; p < C::PT; ++p)
                    acc[c][p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[c], af[p], acc[c][p], 0, 0, 0);
        }
        if (k + 1 < nk) store_lds(buf ^ 1);
        __syncthreads();
    }
    if (W_BITS == 4) {
#pragma unroll
        for (int c = 0; c < C::CT; ++c)
#pragma unroll
            for (int p = 0; p < C::PT; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][p][r] >>= 4;
    }
}

// ------------------------------------------------------------------------------------------
// Asynchronous int8 x int8 pipeline: both operand tiles of every K chunk are streamed
// global -> LDS with global_load_lds (no VGPR staging, no ds_write) into an NS-stage ring, NS-1
// chunks ahead of the MFMAs.  One s_barrier per chunk: [wait chunk k landed] [barrier: everyone is
// also done reading the stage chunk k+NS-1 will overwrite] [issue chunk k+NS-1] [MFMA chunk k].
// The LDS image is lane-linear (16 rows x 64 B per wave instruction), so the XOR swizzle that keeps
// the fragment reads conflict-free is applied to the per-lane SOURCE address; padding taps and
// rows beyond M read a 16-byte zero page.  With a second branch (identity conv) its chunks simply
// continue the same chunk sequence and accumulate into acc2, so its loads overlap the first
// branch's MFMAs.
__device__ __attribute__((aligned(16))) const int g_zero16[4] = {0, 0, 0, 0};


// NIB: both operands are hawq4 nibble-packed.  A ring-stage row still holds 64 BYTES (= 128 channels), the
// packed bytes travel through LDS untouched and every fragment (8 B = 16 channels per lane) is unpacked to
// int8 in registers right before its MFMA (activations zero-extended, weights as value*16 with the
// accumulators shifted back by 4 at the end - exact).  Needs Cin % 128 == 0 (launcher-checked).
template <class C, bool DUAL, bool NIB>
__device__ __forceinline__ void gemm_pipeline(v16i (&acc)[C::CT][C::PT], v16i (&acc2)[DUAL ? C::CT : 1][DUAL ? C::PT : 1],
                                              const ConvP &p, int m0, int c0, char *smem) {
    constexpr int CSH = NIB ? 7 : 6;  // log2(channels per 64-byte chunk)
    constexpr int NS = C::NS, L = C::AL + C::WL, STAGE = C::STAGE_BYTES;
    static_assert((NS - 2) * L * C::KSUB / C::KG <= 60, "vmcnt range");
    const int t = threadIdx.x;
    const int lane = t & 63, wave = (t >> 6) % C::NWG, kg = (t >> 6) / C::NWG;   // wave inside its K group, K group
    const int wave_m = wave % C::WM, wave_c = wave / C::WM;
    const int tg = t % C::NTG;                                                    // thread inside its K group
    const int lrow = tg >> 2, lslot = tg & 3;
    const char *zero = reinterpret_cast<const char *>(g_zero16);

    // im2col bookkeeping of this thread's rows (first branch) and of the second branch's 1x1/stride-s2 rows
    int pix_base[C::AL], iy0[C::AL], ix0[C::AL], pix2[DUAL ? C::AL : 1];
    bool mval[C::AL];
    int asw[C::AL];  // source slot (16-B unit) after the swizzle
#pragma unroll
    for (int i = 0; i < C::AL; ++i) {
        const int row = lrow + C::RPP * i;
        const int m = m0 + row;
        mval[i] = m < p.M;
        const int mm = mval[i] ? m : 0;
        const int n = mm / (p.Ho * p.Wo);
        const int r = mm - n * (p.Ho * p.Wo);
        const int oy = r / p.Wo, ox = r - oy * p.Wo;
        iy0[i] = oy * p.stride - p.pad;
        ix0[i] = ox * p.stride - p.pad;
        pix_base[i] = (n * p.H + iy0[i]) * p.W + ix0[i];
        if (DUAL) pix2[DUAL ? i : 0] = (n * p.H2 + oy * p.stride2) * p.W2 + ox * p.stride2;
        asw[i] = (lslot ^ ((row >> 2) & 3)) << 4;
    }
    int wsw[C::WL];
#pragma unroll
    for (int j = 0; j < C::WL; ++j) wsw[j] = (lslot ^ (((lrow + C::RPP * j) >> 2) & 3)) << 4;

    const int taps = p.KH * p.KW, cch1 = p.Cin >> CSH;
    const int nk1 = taps * cch1, nk2 = DUAL ? (p.Cin2 >> CSH) : 0, nk = nk1 + nk2;
    const int rowb1 = NIB ? p.Cin >> 1 : p.Cin, rowb2 = DUAL ? (NIB ? p.Cin2 >> 1 : p.Cin2) : 0;  // bytes per pixel / tap row
    const size_t wrow1 = (size_t)taps * rowb1, wrow2 = (size_t)rowb2;

    constexpr int KSUB = C::KSUB, SUBB = (C::BM + C::BN) * 64;  // bytes of one 64-channel sub-chunk (A rows, then W rows)
    int kh = 0, kw = 0, cc = 0, jissue = 0, istage = 0;  // coordinates of the next sub-chunk to ISSUE
    auto issue_sub = [&](int sub) {
        char *sa = smem + istage * STAGE + sub * SUBB + wave * 1024;  // + i * RPP * 64: RPP rows x 64 B per pass
        char *sw = sa + C::BM * 64;
        const bool mine = C::KG == 1 || (sub % C::KG) == kg;           // split-K: the sub-chunk's own group loads it
        if (!mine) {
            if (!DUAL || jissue < nk1) {
                if (++cc == cch1) {
                    cc = 0;
                    if (++kw == p.KW) {
                        kw = 0;
                        ++kh;
                    }
                }
            }
            ++jissue;
            return;
        }
        if (!DUAL || jissue < nk1) {
            const int tap_off = kh * p.W + kw;
#pragma unroll
            for (int i = 0; i < C::AL; ++i) {
                const int iy = iy0[i] + kh, ix = ix0[i] + kw;
                const bool v = mval[i] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const char *src = v ? (const char *)p.in + (size_t)(pix_base[i] + tap_off) * p.in_pitch + (cc << 6) + asw[i] : zero;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(sa + i * (C::RPP * 64)), 16, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < C::WL; ++j) {
                const char *src = (const char *)p.wgt + (size_t)(c0 + lrow + C::RPP * j) * wrow1 +
                                  (size_t)((kh * p.KW + kw) * rowb1 + (cc << 6)) + wsw[j];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(sw + j * (C::RPP * 64)), 16, 0, 0);
            }
            if (++cc == cch1) {
                cc = 0;
                if (++kw == p.KW) {
                    kw = 0;
                    ++kh;
                }
            }
        } else {
            const int c2 = jissue - nk1;
#pragma unroll
            for (int i = 0; i < C::AL; ++i) {
                const char *src = mval[i] ? (const char *)p.in2 + (size_t)pix2[DUAL ? i : 0] * rowb2 + (c2 << 6) + asw[i] : zero;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(sa + i * (C::RPP * 64)), 16, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < C::WL; ++j) {
                const char *src = (const char *)p.wgt2 + (size_t)(c0 + lrow + C::RPP * j) * wrow2 + (size_t)(c2 << 6) + wsw[j];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(sw + j * (C::RPP * 64)), 16, 0, 0);
            }
        }
        ++jissue;
    };
    auto issue = [&]() {  // one ring stage = KSUB consecutive sub-chunks (nk1 and nk2 are multiples of KSUB)
#pragma unroll
        for (int sub = 0; sub < KSUB; ++sub) issue_sub(sub);
        if (++istage == NS) istage = 0;
    };

    const int l31 = lane & 31, h = lane >> 5;
    int arow[C::PT], wrow[C::CT];
#pragma unroll
    for (int q = 0; q < C::PT; ++q) arow[q] = wave_m * (C::PT * 32) + q * 32 + l31;
#pragma unroll
    for (int c = 0; c < C::CT; ++c) wrow[c] = wave_c * (C::CT * 32) + c * 32 + cperm(l31);

    auto compute = [&](auto &a, int stage) {
        // fragments of K-step s+1 are fetched from LDS while the MFMAs of K-step s run
        v4i wf[2][C::CT], af[2][C::PT];
        constexpr int SPS = NIB ? 4 : 2;  // MFMA K-steps per 64-byte sub-chunk
        auto fetch = [&](int s, int buf) {
            const char *ldsA = smem + stage * STAGE + (s / SPS) * SUBB, *ldsW = ldsA + C::BM * 64;
            if constexpr (NIB) {
                const int slot = s % SPS;  // 16 packed bytes = 32 channels; lane-half h takes 8 of them
#pragma unroll
                for (int c = 0; c < C::CT; ++c) {
                    const v2i t2 = *reinterpret_cast<const v2i *>(ldsW + lds_off(wrow[c], slot) + h * 8);
                    wf[buf][c] = unpack16<true>((unsigned)t2.x, (unsigned)t2.y);
                }
#pragma unroll
                for (int q = 0; q < C::PT; ++q) {
                    const v2i t2 = *reinterpret_cast<const v2i *>(ldsA + lds_off(arow[q], slot) + h * 8);
                    af[buf][q] = unpack16<false>((unsigned)t2.x, (unsigned)t2.y);
                }
                return;
            }
            const int slot = 2 * (s & 1) + h;
#pragma unroll
            for (int c = 0; c < C::CT; ++c) wf[buf][c] = *reinterpret_cast<const v4i *>(ldsW + lds_off(wrow[c], slot));
#pragma unroll
            for (int q = 0; q < C::PT; ++q) af[buf][q] = *reinterpret_cast<const v4i *>(ldsA + lds_off(arow[q], slot));
        };
        // the K-steps this wave owns: all of the stage, or (split-K) those of the sub-chunks sub % KG == kg
        constexpr int NOWN = SPS * KSUB / C::KG;
        auto step_of = [&](int o) { return C::KG == 1 ? o : ((o / SPS) * C::KG + kg) * SPS + (o % SPS); };
        fetch(step_of(0), 0);
#pragma unroll
        for (int o = 0; o < NOWN; ++o) {
            if (o + 1 < NOWN) fetch(step_of(o + 1), (o + 1) & 1);
#pragma unroll
            for (int c = 0; c < C::CT; ++c)
#pragma unroll
                for (int q = 0; q < C::PT; ++q)
                    a[c][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[o & 1][c], af[o & 1][q], a[c][q], 0, 0, 0);
        }
    };

    const int ns1 = nk1 / KSUB, nst = nk / KSUB;  // ring stages of the first branch / in total
#pragma unroll
    for (int j = 0; j < NS - 1; ++j)
        if (j < nst) issue();
    int cstage = 0;
    constexpr int LS = L * KSUB / C::KG;  // loads per thread per stage
    auto step = [&](auto &a, int k) {
        // stage k must have landed: at most min(NS-2, stages issued after k) stages may stay in flight
        const int after = min(NS - 2, nst - 1 - k);
        if (after >= NS - 2) {
            wait_vmcnt<(NS - 2) * LS>();  // steady state
        } else {  // tail: fewer stages behind stage k, wait for exactly those to remain
            switch (after) {
                case 5: wait_vmcnt<(NS > 6 ? 5 : 0) * LS>(); break;
                case 4: wait_vmcnt<(NS > 5 ? 4 : 0) * LS>(); break;
                case 3: wait_vmcnt<(NS > 4 ? 3 : 0) * LS>(); break;
                case 2: wait_vmcnt<(NS > 3 ? 2 : 0) * LS>(); break;
                case 1: wait_vmcnt<(NS > 2 ? 1 : 0) * LS>(); break;
                default: wait_vmcnt<0>(); break;
            }
        }
        __builtin_amdgcn_s_barrier();
        if (jissue < nk && !HAWQ_DBG_BIT(p.dbg, 1)) issue();
        if (!HAWQ_DBG_BIT(p.dbg, 2)) compute(a, cstage);
        if (++cstage == NS) cstage = 0;
    };
    for (int k = 0; k < ns1; ++k) step(acc, k);
    if constexpr (DUAL)
        for (int k = ns1; k < nst; ++k) step(acc2, k);
    if constexpr (NIB) {  // weights were unpacked as value*16
#pragma unroll
        for (int c = 0; c < C::CT; ++c)
#pragma unroll
            for (int q = 0; q < C::PT; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[c][q][r] >>= 4;
                    if constexpr (DUAL) acc2[DUAL ? c : 0][DUAL ? q : 0][r] >>= 4;
                }
    }
    __syncthreads();  // all MFMA fragment reads done: the ring may be reused by the epilogue
}

// BITS: 0 = decide at run time (generic kernels), else (a_bits << 4) | w_bits.
template <class C, int BITS>
__device__ __forceinline__ void run_segment(v16i (&acc)[C::CT][C::PT], const uint8_t *in, const uint8_t *wgt,
                                            int a_bits, int w_bits, int H, int W, int Cin, int KH, int KW,
                                            int stride, int pad, int Ho, int Wo, int M, int Cout, int m0, int c0,
                                            char *smem, int apitch) {
    if (BITS == 0x88 || (BITS == 0 && a_bits == 8 && w_bits == 8))
        gemm_segment<C, 8, 8>(acc, in, wgt, H, W, Cin, KH, KW, stride, pad, Ho, Wo, M, Cout, m0, c0, smem, apitch);
    else if (BITS == 0x44 || (BITS == 0 && a_bits == 4 && w_bits == 4))
        gemm_segment<C, 4, 4>(acc, in, wgt, H, W, Cin, KH, KW, stride, pad, Ho, Wo, M, Cout, m0, c0, smem, apitch);
    else if (BITS == 0x84 || (BITS == 0 && a_bits == 8 && w_bits == 4))
        gemm_segment<C, 8, 4>(acc, in, wgt, H, W, Cin, KH, KW, stride, pad, Ho, Wo, M, Cout, m0, c0, smem, apitch);
    else
        gemm_segment<C, 4, 8>(acc, in, wgt, H, W, Cin, KH, KW, stride, pad, Ho, Wo, M, Cout, m0, c0, smem, apitch);
}

__device__ __forceinline__ v4i ld4(const int32_t *p) { return *reinterpret_cast<const v4i *>(p); }


// =============================================================== exact general epilogue (BITS == 0)
// Direct per-lane global accesses, dyadic_rne everywhere: any e in [1,62], any pre-shift, ties
// handled, 16- or 32-bit residuals, run-time operand widths.  Slow but always right.
template <class C, int EPI, bool DUAL>
__device__ __forceinline__ void epilogue_generic(const ConvP &p, v16i (&acc)[C::CT][C::PT],
                                                 v16i (&acc2)[DUAL ? C::CT : 1][DUAL ? C::PT : 1], int m0, int c0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wave_m = wave % C::WM, wave_c = wave / C::WM;
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int c = 0; c < C::CT; ++c) {
        const int ch = c0 + wave_c * (C::CT * 32) + c * 32 + h * 16;
#pragma unroll
        for (int q = 0; q < C::PT; ++q) {
            const int pix = m0 + wave_m * (C::PT * 32) + q * 32 + l31;
            if (pix >= p.M) continue;
            // RAW / DEQUANT address the dense [M][Cout] / [M][ldo] tensors; REQUANT / RESIDUAL rows are out_pitch channels apart (ABI 4)
            constexpr bool PITCHED = EPI == HAWQ_EPI_REQUANT || EPI == HAWQ_EPI_RESIDUAL;
            const int opitch = PITCHED ? p.out_pitch : p.Cout;
            const size_t elem = (size_t)pix * opitch + ch;
            bool ovf = false;
            int qw[4] = {0, 0, 0, 0};   // int8 output: the lane's 16 channels leave as ONE 16-byte store after the loop
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (PITCHED && ch + 4 * g >= opitch) continue;   // beyond the stored row: nothing to compute, nothing to write
                const v4i b4 = ld4(p.bias + ch + 4 * g);
                const int bb[4] = {b4.x, b4.y, b4.z, b4.w};
                int v[4], qv[4] = {0, 0, 0, 0}, o[4] = {0, 0, 0, 0};
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[c][q][4 * g + j] + bb[j];
                if constexpr (EPI == HAWQ_EPI_RAW) {
                    v4i w = {v[0], v[1], v[2], v[3]};
                    reinterpret_cast<v4i *>(p.out_acc + elem)[g] = w;
                } else if constexpr (EPI == HAWQ_EPI_DEQUANT) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (ch + 4 * g + j < p.n_valid)
                            p.out_f32[(size_t)pix * p.ldo + ch + 4 * g + j] = (float)v[j] * p.fscale[ch + 4 * g + j];
                } else if (p.n_valid > 0 && ch + 4 * g >= p.n_valid) {
                    // padding channels (caller's promise: zero weights / bias / tables, zero identity): zeros without the arithmetic.
                    // Narrow layers padded to the 64-channel tile (MobileNetV2's 16 / 24 / 32-channel projections) spend most of
                    // this exact epilogue's 64-bit requants on them otherwise.
                    if (EPI == HAWQ_EPI_RESIDUAL && p.res_out) {
                        if (p.res_out_bits == 16) reinterpret_cast<v2i *>((uint16_t *)p.res_out + elem)[g] = v2i{0, 0};
                        else reinterpret_cast<v4i *>((int32_t *)p.res_out + elem)[g] = v4i{0, 0, 0, 0};
                    }
                    if ((EPI == HAWQ_EPI_REQUANT || p.out_q) && p.out_bits == 4 && !(g & 1)) reinterpret_cast<uint32_t *>((uint8_t *)p.out_q + (elem >> 1) + (g >> 1) * 4)[0] = 0u;
                } else if (EPI == HAWQ_EPI_RESIDUAL && !DUAL && p.gfast) {
                    // Fast-contract arithmetic on the direct epilogue (round 4): 32-bit / signed residual tensors (MobileNetV2's carriers)
                    // keep this epilogue's per-lane accesses, but every table has been lifted and bounded by the host, so a requant is
                    // shift + v_mad_i64_i32 + shift against the fused constants of `ctab` (bias folded in) instead of dyadic_rne's
                    // ~15 instructions; p.k0 == 2 keeps the exact round-half-even correction where a tie could not be excluded.
                    const bool tie = p.k0 == 2;
                    const DyNt dids = dynt_prepare(p.m_id_s, p.e_id_s), dq = dynt_prepare(p.mq, p.eq);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const v4i t4 = ld4(p.ctab + (size_t)(ch + 4 * g + j) * 4);
                        DyNt dm;
                        dm.m = t4.x, dm.s = t4.y & 31, dm.k = t4.y >> 8;
                        dm.add = (long long)(((unsigned long long)(unsigned)t4.w << 32) | (unsigned)t4.z);
                        const int av = acc[c][q][4 * g + j];
                        int ov = tie ? dyadic_tie(av, dm) : dyadic_nt(av, dm);
                        if (p.res_in) {
                            const int r = p.res_in_bits == 16 ? (int)((const uint16_t *)p.res_in)[elem + 4 * g + j]
                                                              : ((const int32_t *)p.res_in)[elem + 4 * g + j];
                            ov += tie ? dyadic_tie(r, dids) : dyadic_nt(r, dids);
                        }
                        if (!p.res_no_relu) ov = max(ov, 0);
                        if (p.res_clamp16) ov = clampi(ov, -32768, 32767);
                        o[j] = ov;
                        qv[j] = clampi(tie ? dyadic_tie(ov, dq) : dyadic_nt(ov, dq), p.q_lo, p.q_hi);
                        ovf |= ov > 65535;
                    }
                    if (p.res_out) {
                        if (p.res_out_bits == 16) {
                            v2i w = {min(o[0], 65535) | (min(o[1], 65535) << 16), min(o[2], 65535) | (min(o[3], 65535) << 16)};
                            reinterpret_cast<v2i *>((uint16_t *)p.res_out + elem)[g] = w;
                        } else {
                            v4i w = {o[0], o[1], o[2], o[3]};
                            reinterpret_cast<v4i *>((int32_t *)p.res_out + elem)[g] = w;
                        }
                    }
                    if (p.out_q) {
                        if (p.out_bits == 8) {
                            qw[g] = (int)pack4_i8(qv[0], qv[1], qv[2], qv[3]);
                        } else {
                            uint8_t *dst = (uint8_t *)p.out_q + (elem >> 1) + (g >> 1) * 4;
#pragma unroll
                            for (int j = 0; j < 4; ++j) dst[j] = (g & 1) ? (uint8_t)((dst[j] & 0x0f) | (qv[j] << 4)) : (uint8_t)(qv[j] & 0x0f);
                        }
                    }
                } else {
                    const v4i m4 = ld4(p.m + ch + 4 * g), e4 = ld4(p.e + ch + 4 * g);
                    const int mm[4] = {m4.x, m4.y, m4.z, m4.w}, ee[4] = {e4.x, e4.y, e4.z, e4.w};
                    if constexpr (EPI == HAWQ_EPI_REQUANT) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            qv[j] = clampi(dyadic_rne(p.relu ? max(v[j], 0) : v[j], mm[j], ee[j]), p.q_lo, p.q_hi);
                    } else {
                        int idv[4];
                        if constexpr (DUAL) {
                            const v4i b2 = ld4(p.bias2 + ch + 4 * g), mi = ld4(p.m_id + ch + 4 * g),
                                      ei = ld4(p.e_id + ch + 4 * g);
                            const int bb2[4] = {b2.x, b2.y, b2.z, b2.w}, m1[4] = {mi.x, mi.y, mi.z, mi.w},
                                      e1[4] = {ei.x, ei.y, ei.z, ei.w};
#pragma unroll
                            for (int j = 0; j < 4; ++j) idv[j] = dyadic_rne(acc2[DUAL ? c : 0][DUAL ? q : 0][4 * g + j] + bb2[j], m1[j], e1[j]);
                        } else if (p.res_in) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int r = p.res_in_bits == 16 ? (int)((const uint16_t *)p.res_in)[elem + 4 * g + j]
                                                                  : ((const int32_t *)p.res_in)[elem + 4 * g + j];
                                idv[j] = dyadic_rne(r, p.m_id_s, p.e_id_s);
                            }
                        } else {   // no identity branch (MobileNetV2 units that change shape): fixedpoint_fn case 0
#pragma unroll
                            for (int j = 0; j < 4; ++j) idv[j] = 0;
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            o[j] = dyadic_rne(v[j], mm[j], ee[j]) + idv[j];          // no clamp with an identity: quant_utils.py:456
                            if (!p.res_no_relu) o[j] = max(o[j], 0);
                            if (p.res_clamp16) o[j] = clampi(o[j], -32768, 32767);   // case 0 clamps to its 16-bit range (quant_utils.py:409-413)
                            qv[j] = clampi(dyadic_rne(o[j], p.mq, p.eq), p.q_lo, p.q_hi);
                            ovf |= o[j] > 65535;
                        }
                        if (p.res_out) {
                            if (p.res_out_bits == 16) {
                                v2i w = {min(o[0], 65535) | (min(o[1], 65535) << 16), min(o[2], 65535) | (min(o[3], 65535) << 16)};
                                reinterpret_cast<v2i *>((uint16_t *)p.res_out + elem)[g] = w;
                            } else {
                                v4i w = {o[0], o[1], o[2], o[3]};
                                reinterpret_cast<v4i *>((int32_t *)p.res_out + elem)[g] = w;
                            }
                        }
                    }
                    if (EPI == HAWQ_EPI_REQUANT || p.out_q) {
                        if (p.out_bits == 8) {
                            qw[g] = (int)pack4_i8(qv[0], qv[1], qv[2], qv[3]);
                        } else {  // hawq4: byte k of an 8-channel group = c_k | c_{k+4} << 4
                            uint8_t *dst = (uint8_t *)p.out_q + (elem >> 1) + (g >> 1) * 4;
                            if (g & 1) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) dst[j] = (uint8_t)((dst[j] & 0x0f) | (qv[j] << 4));
                            } else {
#pragma unroll
                                for (int j = 0; j < 4; ++j) dst[j] = (uint8_t)(qv[j] & 0x0f);
                            }
                        }
                    }
                }
            }
            if constexpr (PITCHED) {
                // (rows are multiples of 16 channels and ch is one: the lane's 16 bytes are either entirely inside the row or not at all)
                if ((EPI == HAWQ_EPI_REQUANT || p.out_q) && p.out_bits == 8 && ch < opitch)
                    *reinterpret_cast<v4i *>((char *)p.out_q + elem) = v4i{qw[0], qw[1], qw[2], qw[3]};
            }
            if (EPI == HAWQ_EPI_RESIDUAL && ovf && p.res_out && p.res_out_bits == 16) atomicOr(p.flags, 1);
        }
    }
}

// =============================================================== fast epilogue (BITS != 0)
// Host-proved tables (see dyadic_nt), uint16 residuals, everything staged through LDS so that
// each global access of the residual / output tensors is a full-line coalesced 16 B-per-lane access:
//   res tile [BM][BN] uint16 behind the operand staging buffers (filled asynchronously with
//   global_load_lds before the K loop, overwritten in place with the new residual),
//   q tile   [BM][BN] int8 positions, aliased onto the staging buffers (free after the K loop).
// 16-B chunks are XOR-swizzled by the pixel row so that both the per-lane (one pixel, 16 channels)
// and the row-contiguous access patterns are bank-conflict free.
template <class C>
struct Stage {
    static constexpr int RCPR = C::BN / 8;   // 16-B chunks per residual-tile row (uint16)
    static constexpr int QCPR = C::BN / 16;  // 16-B chunks per q-tile row
    __device__ static __forceinline__ int rsw(int row) { return row & (RCPR - 1); }
    __device__ static __forceinline__ int qsw(int row) {
        return QCPR >= 8 ? (row & (QCPR - 1)) : ((row >> 1) & (QCPR - 1));
    }
};

template <class C>
__device__ __forceinline__ void prefetch_residual(const ConvP &p, int m0, int c0, char *res_tile) {
    using S = Stage<C>;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < (C::BM * S::RCPR + C::NT - 1) / C::NT; ++i) {
        const int base = (i * C::NW + wave) * 64;
        if (base >= C::BM * S::RCPR) break;   // (16-wave workgroups on a 64-pixel tile: half of the waves have no piece)
        const int idx = base + lane;
        const int row = idx / S::RCPR, j = idx % S::RCPR;
        const int grow = (m0 + row < p.M) ? m0 + row : m0;
        const char *src = (const char *)p.res_in + ((size_t)grow * p.Cout + c0) * 2 + ((j ^ S::rsw(row)) << 4);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(res_tile + base * 16), 16, 0, 0);
    }
}

// MODE: see dyadic_mode (0 = tie-free tables, 2 = exact tie handling for every table; 1 = tie-free and shift-free is
// implemented but not instantiated: as a run-time variant it cost registers (spills in the 64x64 dual kernels) for
// a gain inside the measurement noise)
// CK0: the per-channel tables carry no pre-shift: the table word is the shift amount itself (3 VALU instructions per requantised
// accumulator fewer).  NOT instantiated: a wave-uniform branch between the two forms in every fast kernel measured neutral in the
// forward (same-plan A/B 93.08 vs 93.04 k img/s, profiles/r03_valu_trims.md) - these epilogues are not VALU-bound; the fused
// expand -> reduce kernels (fused_er.hip / fused_wp.hip), which are, have their own all-k-zero instantiations
template <class C, int EPI, bool DUAL, int MODE = 0, bool CK0 = false>
__device__ __forceinline__ void epilogue_fast(const ConvP &p, v16i (&acc)[C::CT][C::PT],
                                              v16i (&acc2)[DUAL ? C::CT : 1][DUAL ? C::PT : 1], int m0, int c0,
                                              char *q_tile, char *res_tile, const char *ctab_lds,
                                              const bool active = true) {
    // active == false: a wave that owns no accumulators (band-kernel producer); it only helps with the stores
    using S = Stage<C>;
    constexpr bool RES = EPI == HAWQ_EPI_RESIDUAL;
    constexpr bool K0 = MODE == 1 || CK0;                      // per-channel tables
    constexpr int MODE_C = (CK0 && MODE == 0) ? 1 : MODE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wave_m = wave % C::WM, wave_c = (wave / C::WM) % C::WN;
    const int l31 = lane & 31, h = lane >> 5;
    const int lrow0 = wave_m * (C::PT * 32) + l31;  // tile-local pixel row of pixel tile 0
    DyNt dids = dynt_prepare(p.m_id_s, p.e_id_s), dq = dynt_prepare(p.mq, p.eq);
    asm volatile("" : "+s"(dids.add), "+s"(dq.add));   // opaque rounding constants: one v_mad_i64_i32 instead of v_mul_hi_i32 + v_add (see fused_er.hip)
    const int qhi2 = (p.q_hi & 0xffff) | (p.q_hi << 16);   // RESIDUAL: q >= 0, clamped from above as packed int16 pairs
    unsigned oor = 0;               // OR of all residual outputs: bits >= 16 set <=> uint16 overflow
    unsigned rowmask[C::PT];
#pragma unroll
    for (int q = 0; q < C::PT; ++q) rowmask[q] = (m0 + lrow0 + q * 32 < p.M) ? 0xffffffffu : 0u;
    if (active)
#pragma unroll
    for (int c = 0; c < C::CT; ++c) {
        const int lch = wave_c * (C::CT * 32) + c * 32 + h * 16;  // tile-local first channel of this lane
        // Two accumulator sets (DUAL) leave no room to keep every pixel tile's packed outputs live: walk the pixel tiles one at a
        // time there (the channel constants are re-read from LDS).  The same holds for the RESIDUAL epilogue of a kernel that must fit
        // 4 waves per SIMD (<= 128 registers: the 16-wave band kernels, the 3x3 + residual launches of ResNet18/34): with both pixel
        // tiles live it held 2 x (8 residual + 8 packed residual + 4 packed q) registers beside 64 accumulators and spilled 88-160
        // of them (round 4: found by the scratch check of tests/test_host_logic.py).
        constexpr int WPS = (C::NT / 256 > C::MINB) ? C::NT / 256 : C::MINB;   // waves per SIMD the kernel is built for
        constexpr int QPASS = (DUAL || (RES && WPS >= 4)) ? C::PT : 1, QPER = C::PT / QPASS;
#pragma unroll
        for (int qp = 0; qp < QPASS; ++qp) {
        v4i rin[C::PT][2];             // old residual values of this pass's pixel tiles (32 bytes per lane and tile)
        if constexpr (RES && !DUAL) {
#pragma unroll
            for (int qq = 0; qq < QPER; ++qq) {
                const int q = qp * QPER + qq;
                const int row = lrow0 + q * 32;
                const char *base = res_tile + row * (S::RCPR * 16);
                rin[q][0] = *reinterpret_cast<const v4i *>(base + (((lch >> 3) ^ S::rsw(row)) << 4));
                rin[q][1] = *reinterpret_cast<const v4i *>(base + ((((lch >> 3) + 1) ^ S::rsw(row)) << 4));
            }
        }
        int qpack[QPER][4];            // 16 x int8, or 2 dwords of hawq4 in [0..1]
        int rpack[QPER][RES ? 8 : 1];  // 16 x uint16
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // fused constants {m, s, C} with C = bias*m + 2^(e-1): (acc+bias)*m + 2^(e-1) == acc*m + C
            DyNt dm[4], di[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const v4i t = *reinterpret_cast<const v4i *>(ctab_lds + (lch + 4 * g + j) * 16);
                dm[j].m = t.x, dm[j].s = K0 ? t.y : (t.y & 31), dm[j].k = K0 ? 0 : (t.y >> 8);
                dm[j].add = (long long)(((unsigned long long)(unsigned)t.w << 32) | (unsigned)t.z);
                if constexpr (DUAL) {
                    const v4i u = *reinterpret_cast<const v4i *>(ctab_lds + C::BN * 16 + (lch + 4 * g + j) * 16);
                    di[j].m = u.x, di[j].s = K0 ? u.y : (u.y & 31), di[j].k = K0 ? 0 : (u.y >> 8);
                    di[j].add = (long long)(((unsigned long long)(unsigned)u.w << 32) | (unsigned)u.z);
                } else {
                    di[j] = dids;
                }
            }
#pragma unroll
            for (int qq = 0; qq < QPER; ++qq) {
                const int q = qp * QPER + qq;
                int qv[4];
                if constexpr (!RES) {
                    // ReLU commutes with the (monotone, 0 -> 0) requantisation: it is folded into q_lo
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        qv[j] = med3i(dyadic_mode<MODE_C>(acc[c][q][4 * g + j], dm[j]), p.q_lo, p.q_hi);
                } else {
                    int idin[4], o[4];
                    if constexpr (DUAL) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) idin[j] = acc2[DUAL ? c : 0][DUAL ? q : 0][4 * g + j];
                    } else {
                        const unsigned w0 = (unsigned)rin[q][g >> 1][(g & 1) * 2], w1 = (unsigned)rin[q][g >> 1][(g & 1) * 2 + 1];
                        idin[0] = (int)(w0 & 0xffffu), idin[1] = (int)(w0 >> 16);
                        idin[2] = (int)(w1 & 0xffffu), idin[3] = (int)(w1 >> 16);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int a = dyadic_mode<MODE_C>(acc[c][q][4 * g + j], dm[j]);
                        const int b = DUAL ? dyadic_mode<MODE_C>(idin[j], di[j]) : dyadic_mode<MODE == 2 ? 2 : 0>(idin[j], di[j]);
                        o[j] = max(a + b, 0);  // no clamp: quant_utils.py:456
                        qv[j] = dyadic_mode<MODE>(o[j], dq);  // o >= 0 and m >= 0: q >= 0 >= q_lo; the upper clamp runs on the packed pairs
                    }
                    // rows beyond M hold bias-only garbage: they never reach memory and must not raise the flag
                    oor |= ((unsigned)(o[0] | o[1]) | (unsigned)(o[2] | o[3])) & rowmask[q];
                    rpack[qq][RES ? 2 * g : 0] = pack2_u16_sat(o[0], o[1]);
                    rpack[qq][RES ? 2 * g + 1 : 0] = pack2_u16_sat(o[2], o[3]);
                }
                const int w = RES ? pack4_min(qv[0], qv[1], qv[2], qv[3], qhi2) : pack4_fast(qv[0], qv[1], qv[2], qv[3]);
                if (p.out_bits == 8) {
                    qpack[qq][g] = w;
                } else if (g & 1) {  // hawq4: low nibbles = channels 0-3 of the 8-group, high = 4-7
                    qpack[qq][g >> 1] |= w << 4;
                } else {
                    qpack[qq][g >> 1] = w;
                }
            }
        }
#pragma unroll
        for (int qq = 0; qq < QPER; ++qq) {
            const int q = qp * QPER + qq;
            const int row = lrow0 + q * 32;
            if constexpr (RES) {
                char *base = res_tile + row * (S::RCPR * 16);
                v4i a = {rpack[qq][0], rpack[qq][1], rpack[qq][2], rpack[qq][3]};
                v4i b = {rpack[qq][4], rpack[qq][5], rpack[qq][6], rpack[qq][7]};
                *reinterpret_cast<v4i *>(base + (((lch >> 3) ^ S::rsw(row)) << 4)) = a;
                *reinterpret_cast<v4i *>(base + ((((lch >> 3) + 1) ^ S::rsw(row)) << 4)) = b;
            }
            char *qb = q_tile + row * (S::QCPR * 16) + (((lch >> 4) ^ S::qsw(row)) << 4);
            if (p.out_bits == 8) {
                v4i w = {qpack[qq][0], qpack[qq][1], qpack[qq][2], qpack[qq][3]};
                *reinterpret_cast<v4i *>(qb) = w;
            } else {
                v2i w = {qpack[qq][0], qpack[qq][1]};
                *reinterpret_cast<v2i *>(qb) = w;
            }
        }
        }  // qp
    }
    if (RES && (oor >> 16) != 0 && p.res_out && !HAWQ_DBG_BIT(p.dbg, ~0)) atomicOr(p.flags, 1);
    if (HAWQ_DBG_BIT(p.dbg, 8)) return;  // rows beyond M never reach memory but may flag: harmless
    __syncthreads();
    const int t = threadIdx.x;
    if constexpr (RES) {
        if (p.res_out) {
#pragma unroll
            for (int i = 0; i < (C::BM * S::RCPR + C::NT - 1) / C::NT; ++i) {
                const int idx = t + C::NT * i;
                const int row = idx / S::RCPR, j = idx % S::RCPR;
                if (idx < C::BM * S::RCPR && m0 + row < p.M) {
                    char *dst = (char *)p.res_out + ((size_t)(m0 + row) * p.Cout + c0) * 2 + ((j ^ S::rsw(row)) << 4);
                    *reinterpret_cast<v4i *>(dst) = *reinterpret_cast<const v4i *>(res_tile + idx * 16);
                }
            }
        }
    }
    if ((!RES || p.out_q) && p.out_planar) {
        // channel-group planes [C / G][M][16 B] (G = 16 int8 / 32 hawq4 channels): what the 3x3 band kernel's LDS-DMA
        // fill wants - 64 consecutive pixels of one plane are one contiguous KiB (tools/ubench/ingest_shape.hip:
        // 16-byte segments a pixel row apart reach 16 GB/s per CU, contiguous ones 57-87).  Pixel index fastest
        // across lanes: 16-B stores of consecutive pixels are contiguous in a plane.
        if (p.out_bits == 8) {
#pragma unroll
            for (int i = 0; i < (C::BM * S::QCPR + C::NT - 1) / C::NT; ++i) {
                const int idx = t + C::NT * i;
                const int ch = idx / C::BM, row = idx % C::BM;
                if (idx < C::BM * S::QCPR && m0 + row < p.M) {
                    char *dst = (char *)p.out_q + ((size_t)((c0 >> 4) + ch) * p.M + (m0 + row)) * 16;
                    *reinterpret_cast<v4i *>(dst) = *reinterpret_cast<const v4i *>(q_tile + (row * S::QCPR + (ch ^ S::qsw(row))) * 16);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < (C::BM * (S::QCPR / 2) + C::NT - 1) / C::NT; ++i) {
                const int idx = t + C::NT * i;
                const int u = idx / C::BM, row = idx % C::BM;
                if (idx < C::BM * (S::QCPR / 2) && m0 + row < p.M) {
                    const v2i a = *reinterpret_cast<const v2i *>(q_tile + (row * S::QCPR + ((2 * u) ^ S::qsw(row))) * 16);
                    const v2i b = *reinterpret_cast<const v2i *>(q_tile + (row * S::QCPR + ((2 * u + 1) ^ S::qsw(row))) * 16);
                    const v4i w = {a.x, a.y, b.x, b.y};
                    char *dst = (char *)p.out_q + ((size_t)((c0 >> 5) + u) * p.M + (m0 + row)) * 16;
                    *reinterpret_cast<v4i *>(dst) = w;
                }
            }
        }
    } else if (!RES || p.out_q) {
#pragma unroll
        for (int i = 0; i < (C::BM * S::QCPR + C::NT - 1) / C::NT; ++i) {
            const int idx = t + C::NT * i;
            const int row = idx / S::QCPR, j = idx % S::QCPR;
            const int cofs = c0 + ((j ^ S::qsw(row)) << 4);   // first channel of this 16-channel piece
            if (idx < C::BM * S::QCPR && m0 + row < p.M && cofs < p.out_pitch) {   // (out_pitch < Cout: the row ends before the tile does, ABI 4)
                const size_t e0 = (size_t)(m0 + row) * p.out_pitch + cofs;
                if (p.out_bits == 8)
                    *reinterpret_cast<v4i *>((char *)p.out_q + e0) = *reinterpret_cast<const v4i *>(q_tile + idx * 16);
                else
                    *reinterpret_cast<v2i *>((char *)p.out_q + (e0 >> 1)) = *reinterpret_cast<const v2i *>(q_tile + idx * 16);
            }
        }
    }
}

// TIE: the launch's tables are not provably tie-free -> every requant of the fast epilogue applies the exact
// round-half-even correction (dyadic_tie).  A separate instantiation, not a run-time branch: the correction costs
// registers, and register allocation is per kernel (as a branch it pushed the 64x64 dual kernels into 44 spills).
template <class C, int EPI, bool DUAL, int BITS, int BITS2, bool TIE = false>
__global__ __launch_bounds__(C::NT, C::MINB) void conv_kernel(const ConvP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const bool prof = HAWQ_DBG_BIT(p.dbg, 128) && p.dbgbuf;  // HAWQ_DBG=128: cycle stamps of one wave (tools/convprobe.py)
    const long long t_entry = prof ? (long long)__builtin_readcyclecounter() : 0;
    // XCD-aware tile order: consecutive workgroup ids land on different XCDs (id % 8), so give
    // each XCD a contiguous run of pixel tiles that share the same weight tile in its L2.
    const int tiles_m = (p.M + C::BM - 1) / C::BM;
    const int tiles_c = p.Cout / C::BN;
    const int nwg = tiles_m * tiles_c;
    int wg = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tc = wg % tiles_c, tm = wg / tiles_c;  // channel tiles of one pixel tile are adjacent
    const int m0 = tm * C::BM, c0 = tc * C::BN;
    constexpr bool FAST = BITS != 0 && (EPI == HAWQ_EPI_REQUANT || EPI == HAWQ_EPI_RESIDUAL);
    char *res_tile = smem + p.ring_bytes;
    char *ctab_lds = res_tile + (EPI == HAWQ_EPI_RESIDUAL ? C::BM * C::BN * 2 : 0);  // [BN][16 B] (+ second branch)
    if constexpr (FAST) {
        // the per-channel requant constants of this tile go to LDS asynchronously, off the epilogue's critical path
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (wave < C::BN / 64)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p.ctab + (size_t)(c0 + wave * 64 + lane) * 4),
                                             (__attribute__((address_space(3))) void *)(ctab_lds + wave * 1024), 16, 0, 0);
        if constexpr (DUAL) {
            if (wave >= C::NW / 2 && wave - C::NW / 2 < C::BN / 64)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(p.ctab_id + (size_t)(c0 + (wave - C::NW / 2) * 64 + lane) * 4),
                    (__attribute__((address_space(3))) void *)(ctab_lds + C::BN * 16 + (wave - C::NW / 2) * 1024), 16, 0, 0);
        }
    }
    if constexpr (FAST && EPI == HAWQ_EPI_RESIDUAL && !DUAL) prefetch_residual<C>(p, m0, c0, res_tile);

    v16i acc[C::CT][C::PT];
#pragma unroll
    for (int c = 0; c < C::CT; ++c)
#pragma unroll
        for (int q = 0; q < C::PT; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][q][r] = 0;
    v16i acc2[DUAL ? C::CT : 1][DUAL ? C::PT : 1];
    if constexpr (DUAL) {
#pragma unroll
        for (int c = 0; c < C::CT; ++c)
#pragma unroll
            for (int q = 0; q < C::PT; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[c][q][r] = 0;
    }
    const long long t_begin = prof ? (long long)__builtin_readcyclecounter() : 0;
    constexpr bool ASYNC8 = BITS == 0x88 && (!DUAL || BITS2 == 0x88);
    constexpr bool ASYNC4 = BITS == 0x44 && (!DUAL || BITS2 == 0x44);
    if constexpr (ASYNC8) {
        gemm_pipeline<C, DUAL, false>(acc, acc2, p, m0, c0, smem);
    } else if (ASYNC4 && (p.Cin & 127) == 0 && (!DUAL || (p.Cin2 & 127) == 0) &&
               ((p.KH * p.KW * (p.Cin >> 7)) % C::KSUB) == 0 && (!DUAL || ((p.Cin2 >> 7) % C::KSUB) == 0)) {
        if constexpr (ASYNC4) gemm_pipeline<C, DUAL, true>(acc, acc2, p, m0, c0, smem);
    } else {
        run_segment<C, BITS>(acc, p.in, p.wgt, p.in_bits, p.w_bits, p.H, p.W, p.Cin, p.KH, p.KW, p.stride, p.pad,
                             p.Ho, p.Wo, p.M, p.Cout, m0, c0, smem, p.in_pitch);
        if constexpr (DUAL)
            run_segment<C, BITS2>(acc2, p.in2, p.wgt2, p.in2_bits, p.w2_bits, p.H2, p.W2, p.Cin2, 1, 1, p.stride2,
                                  0, p.Ho, p.Wo, p.M, p.Cout, m0, c0, smem, p.Cin2 * p.in2_bits / 8);
    }
    if (HAWQ_DBG_BIT(p.dbg, 4)) return;
    const long long t_loop_end = prof ? (long long)__builtin_readcyclecounter() : 0;
    const int kgrp = (threadIdx.x >> 6) / C::NWG;
    if constexpr (C::KG > 1) {
        // split-K: groups 1 .. KG-1 hand their partial accumulators to group 0 through the (now free) ring, one group per
        // round; [wave][tile][register][lane] dwords: conflict-free, every lane finds its own registers
        int *dump = reinterpret_cast<int *>(smem);
        const int slot0 = (((threadIdx.x >> 6) % C::NWG) * (C::CT * C::PT)) * 16 * 64 + (threadIdx.x & 63);
        for (int g = 1; g < C::KG; ++g) {
            if (kgrp == g) {
#pragma unroll
                for (int c = 0; c < C::CT; ++c)
#pragma unroll
                    for (int q = 0; q < C::PT; ++q)
#pragma unroll
                        for (int r = 0; r < 16; ++r) dump[slot0 + ((c * C::PT + q) * 16 + r) * 64] = acc[c][q][r];
            }
            __syncthreads();
            if (kgrp == 0) {
#pragma unroll
                for (int c = 0; c < C::CT; ++c)
#pragma unroll
                    for (int q = 0; q < C::PT; ++q)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[c][q][r] += dump[slot0 + ((c * C::PT + q) * 16 + r) * 64];
            }
            __syncthreads();
            if constexpr (DUAL) {
                if (kgrp == g) {
#pragma unroll
                    for (int c = 0; c < C::CT; ++c)
#pragma unroll
                        for (int q = 0; q < C::PT; ++q)
#pragma unroll
                            for (int r = 0; r < 16; ++r) dump[slot0 + ((c * C::PT + q) * 16 + r) * 64] = acc2[DUAL ? c : 0][DUAL ? q : 0][r];
                }
                __syncthreads();
                if (kgrp == 0) {
#pragma unroll
                    for (int c = 0; c < C::CT; ++c)
#pragma unroll
                        for (int q = 0; q < C::PT; ++q)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc2[DUAL ? c : 0][DUAL ? q : 0][r] += dump[slot0 + ((c * C::PT + q) * 16 + r) * 64];
                }
                __syncthreads();
            }
        }
    }
    if constexpr (FAST) {
        epilogue_fast<C, EPI, DUAL, TIE ? 2 : 0>(p, acc, acc2, m0, c0, smem, res_tile, ctab_lds, kgrp == 0);
    } else {
        if (C::KG > 1 && kgrp != 0) return;   // the direct epilogue has no barriers: the other groups are done
        epilogue_generic<C, EPI, DUAL>(p, acc, acc2, m0, c0);
    }
    if (prof && blockIdx.x == 8 && threadIdx.x == 0) {
        p.dbgbuf[0] = t_begin - t_entry, p.dbgbuf[1] = t_loop_end - t_begin;
        p.dbgbuf[2] = (long long)__builtin_readcyclecounter() - t_loop_end, p.dbgbuf[3] = 0;
    }
}

// =============================================================== 3x3 / stride 1 / pad 1 "band" kernel
// The generic pipeline re-fetches the activation tile once per filter tap (9x) and synchronises the workgroup
// once per 64-deep K chunk.  Here a workgroup that owns BM consecutive output pixels keeps, per 64-channel
// slice, the whole input BAND those pixels touch (their image rows plus one halo row above and below, each
// row with a zero column left and right) in LDS and reads the nine taps' B fragments from it at shifted
// addresses: activation traffic into LDS drops ~7x, only the weight tiles of the slice are streamed.  One
// pipeline step covers a whole filter ROW (3 taps, K = 192): one barrier per 24 MFMAs per wave.  Band rows are
// GLOBAL rows G = n*Ho + y, so the fill needs no image logic.
//   band ring : BSTAGES x 4 planes x BAND_PX band pixels x 16 B: plane j holds channels 16j..16j+15 of every band
//               pixel.  Consecutive pixels are 16 B apart, so a B-fragment read (16 lanes = 16 mostly consecutive
//               pixels per LDS cycle) is conflict-free WITHOUT a swizzle and every (tap, k-half) address is
//               base + immediate: the inner loop carries no address arithmetic.  The last 4 entries of each
//               plane are zeros: taps whose row lies outside the pixel's image read those.
//   W ring    : WS (3 or 4) stages x (3 taps x BN x 64 B), one stage per (slice, filter row) step, issued WS - 1
//               steps ahead
// Wave specialisation (NPROD > 0).  Measured on gfx950 (tools/ubench/mfma_rate.hip, HAWQ_DBG=128 stamps):
// the matrix pipe serves the OLDEST ready wave of a SIMD first, an LDS-DMA issue blocks its wave for 60-150
// cycles, and hipcc cannot overlap LDS reads with MFMAs once LDS-DMA is in the loop (see lds_read16).  With all
// waves symmetric, a step cost ~3300 cycles for 1536 cycles of MFMA work per SIMD.  So NPROD extra "producer"
// waves own every LDS-DMA issue and every vmcnt wait; the NW "consumer" waves only read fragments (hand
// software-pipelined) and issue MFMAs, and the two kinds meet at the per-step barrier.
template <int BM_, int BN_, int WM_, int WN_, int BAND_PX_, int BSTAGES_, int MINB_, int NPROD_, int WS_ = 3>
struct BandCfg {
    static constexpr int WS = WS_;                            // W ring stages; W(s + WS - 1) is issued during step s
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, MINB = MINB_, NPROD = NPROD_;
    static constexpr int NW = WM * WN;                        // consumer (MFMA) waves
    static constexpr int NT = (NW + NPROD) * 64;
    static constexpr int ND = NPROD > 0 ? NPROD : NW;         // waves that issue LDS-DMA
    static constexpr int PT = BM / WM / 32, CT = BN / WN / 32;
    static constexpr int BAND_PX = BAND_PX_, BSTAGES = BSTAGES_;  // band pixels per stage (launcher-checked bound)
    static constexpr int PLANE = BAND_PX * 16, BAND_BYTES = 4 * PLANE;
    static constexpr int BPI = BAND_PX / 16 / ND;             // band pieces (1 KiB LDS-DMA instructions) per issuing wave per stage
    static constexpr int RG = BN / 16;                        // 16-row groups of a W tap tile
    static constexpr int RPI = RG / ND, WPI = 3 * RPI;        // W row groups / pieces per issuing wave per step
    static constexpr int WTAP = BN * 64, WSTAGE = 3 * WTAP;
    static constexpr int LDS_BYTES = BSTAGES * BAND_BYTES + WS * WSTAGE;
    static_assert(RG % ND == 0 && (BAND_PX / 16) % ND == 0 && PT >= 1 && CT >= 1 && WS >= 3 && WS <= 5, "tile shape");
    static_assert(BM * BN * 2 <= LDS_BYTES, "epilogue staging aliases the rings");
};

// NIB: both operands hawq4 nibble-packed (Cin % 128 == 0).  A 64-byte slice is then 128 channels, the packed bytes
// travel through LDS untouched and every 16-byte fragment read (32 channels) feeds TWO MFMA K-steps after
// unpacking in registers (activations zero-extended, weights as value*16, accumulators shifted back by 4 - exact);
// both operands pair the same channels with the same K-step, so the channel order inside a slice is irrelevant.
// EPI: REQUANT, or RESIDUAL (single branch, uint16 residuals: the 3x3 second conv of a ResNet18/34 basic block).  The
// residual tile is fetched into the W ring once the K loop has released it (no extra LDS).
template <class C, bool NIB = false, bool TIE = false, int EPI = HAWQ_EPI_REQUANT>
__global__ __launch_bounds__(C::NT, C::MINB) void conv3x3_band_kernel(const ConvP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const bool prof = HAWQ_DBG_BIT(p.dbg, 128) && p.dbgbuf;
    const long long t_entry = prof ? (long long)__builtin_readcyclecounter() : 0;
    const int tiles_c = p.Cout / C::BN;
    const int nwg = ((p.M + C::BM - 1) / C::BM) * tiles_c;
    int wg = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tc = wg % tiles_c, tm = wg / tiles_c;
    const int m0 = tm * C::BM, c0 = tc * C::BN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool consumer = wave < C::NW;
    const bool issuer = C::NPROD > 0 ? !consumer : true;
    const int dw = C::NPROD > 0 ? wave - C::NW : wave;       // index among the issuing waves
    char *band = smem, *wring = smem + C::BSTAGES * C::BAND_BYTES, *ctab_lds = smem + C::LDS_BYTES;
    const char *zero = reinterpret_cast<const char *>(g_zero16);

    const int Wo = p.Wo, Wb = Wo + 2;
    const int G0 = m0 / Wo - 1;                 // global row held by band row 0
    const int rowb = NIB ? p.Cin >> 1 : p.Cin;  // bytes per pixel / per filter tap of one output channel
    const int cchunks = rowb >> 6;
    const int nsteps = 3 * cchunks;             // step s = cc * 3 + kh
    const size_t slice_stride = p.in_planar ? (size_t)p.M * 64 : 64;  // bytes from one 64-byte channel slice to the next

    // ------------------------------------------------------------------ LDS-DMA side (issuing waves)
    const char *bsrc[C::BPI];   // band piece j = i * ND + dw fills 64 pixels of plane (j & 3)
    const char *wsrc[C::RPI];   // W row group r * ND + dw (16 rows), source-side swizzled like every operand tile
    if (issuer) {
        const int rows_total = p.N * p.Ho;
#pragma unroll
        for (int i = 0; i < C::BPI; ++i) {
            const int j = i * C::ND + dw;
            const int bpx = (j >> 2) * 64 + lane;
            const int br = bpx / Wb, bc = bpx - br * Wb;
            const int G = G0 + br, x = bc - 1;
            const bool v = (unsigned)G < (unsigned)rows_total && (unsigned)x < (unsigned)Wo && bpx < C::BAND_PX - 4;
            // NHWC: 16 bytes of a pixel row (64 lanes = 64 segments a row apart: slow, see ingest_shape.hip); planar:
            // plane (cc * 4 + (j & 3)) is [M][16 B], the lanes of a band row read consecutive 16-byte units
            bsrc[i] = !v ? nullptr
                      : p.in_planar ? (const char *)p.in + ((size_t)(j & 3) * p.M + (size_t)G * Wo + x) * 16
                                    : (const char *)p.in + ((size_t)G * Wo + x) * rowb + ((j & 3) << 4);
        }
#pragma unroll
        for (int r = 0; r < C::RPI; ++r) {
            const int row = (r * C::ND + dw) * 16 + (lane >> 2);
            wsrc[r] = (const char *)p.wgt + (size_t)(c0 + row) * ((size_t)9 * rowb) + (((lane & 3) ^ ((row >> 2) & 3)) << 4);
        }
    }
    auto issue_band = [&](int cc) {
        char *dst = band + (C::BSTAGES > 1 ? (cc & 1) * C::BAND_BYTES : 0);
#pragma unroll
        for (int i = 0; i < C::BPI; ++i) {
            const int j = i * C::ND + dw;
            const char *src = bsrc[i] ? bsrc[i] + (size_t)cc * slice_stride : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(dst + (j & 3) * C::PLANE + (j >> 2) * 1024), 16, 0, 0);
        }
    };
    auto issue_w = [&](int s, int cc, int kh) {  // the 3 taps of filter row kh, channel slice cc -> ring stage s % WS
        char *dst = wring + (s % C::WS) * C::WSTAGE + dw * 1024;
        const size_t off = (size_t)(kh * 3) * rowb + (cc << 6);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int r = 0; r < C::RPI; ++r)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wsrc[r] + off + (size_t)kw * rowb),
                                                 (__attribute__((address_space(3))) void *)(dst + kw * C::WTAP + r * (C::ND * 1024)), 16, 0, 0);
    };
    // after barrier #s (which closed step s-1): ring stage (s-1) % WS and band stage (cc+1)&1 are free
    auto issue_step = [&](int s, int cc, int kh) {
        if (C::BSTAGES > 1 && kh == 0 && cc + 1 < cchunks) issue_band(cc + 1);
        const int s2 = s + C::WS - 1;
        if (s2 < nsteps) issue_w(s2, s2 / 3, s2 % 3);
    };
    // Before barrier #(s+1) the issuing waves must have seen W(s+2) land.  With WS == 3 that is the W issued in this
    // very step (everything must have landed).  With WS >= 4 it was issued WS - 3 steps earlier: everything issued in
    // this step and in the WS - 4 steps before it - younger in program order, and LDS-DMA completes in order - may stay
    // in flight (a band prefetch happens in steps with kh == 0 only, so at most one of two consecutive steps has one).
    auto wait_step = [&](int s, int cc, int kh) {
        if (C::WS <= 3) {
            wait_vmcnt<0>();
            return;
        }
        const bool band_now = C::BSTAGES > 1 && kh == 0 && cc + 1 < cchunks;
        const bool w_now = s + C::WS - 1 < nsteps;
        if (C::WS == 4) {
            if (band_now) {
                if (w_now) wait_vmcnt<C::BPI + C::WPI>(); else wait_vmcnt<C::BPI>();
            } else {
                if (w_now) wait_vmcnt<C::WPI>(); else wait_vmcnt<0>();
            }
            return;
        }
        // WS == 5: this step's and the previous step's issues
        const int khp = kh == 0 ? 2 : kh - 1, ccp = kh == 0 ? cc - 1 : cc;
        const bool band_prev = C::BSTAGES > 1 && s >= 1 && khp == 0 && ccp + 1 < cchunks;
        const bool w_prev = s >= 1 && s - 1 + C::WS - 1 < nsteps;
        const int nw = (w_now ? 1 : 0) + (w_prev ? 1 : 0);
        if (band_now || band_prev) {
            if (nw == 2) wait_vmcnt<C::BPI + 2 * C::WPI>(); else if (nw == 1) wait_vmcnt<C::BPI + C::WPI>(); else wait_vmcnt<C::BPI>();
        } else {
            if (nw == 2) wait_vmcnt<2 * C::WPI>(); else if (nw == 1) wait_vmcnt<C::WPI>(); else wait_vmcnt<0>();
        }
    };

    if (issuer) {
        if (dw < C::BN / 64)  // requant constants of this channel tile -> LDS (oldest load: landed before anything else)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p.ctab + (size_t)(c0 + dw * 64 + lane) * 4),
                                             (__attribute__((address_space(3))) void *)(ctab_lds + dw * 1024), 16, 0, 0);
        issue_band(0);
#pragma unroll
        for (int s = 0; s < C::WS - 1; ++s)
            if (s < nsteps) issue_w(s, s / 3, s % 3);
    }

    // ------------------------------------------------------------------ MFMA side (consumer waves)
    const int wave_m = wave % C::WM, wave_c = (wave / C::WM) % C::WN;
    const int l31 = lane & 31, h = lane >> 5;
    int bp0[C::PT], yy[C::PT];
#pragma unroll
    for (int q = 0; q < C::PT; ++q) {
        const int m = m0 + wave_m * (C::PT * 32) + q * 32 + l31;
        const int G = m / Wo, x = m - G * Wo;
        bp0[q] = (G - G0 - 1) * Wb + x;   // band pixel of tap (kh=0, kw=0)
        yy[q] = G % p.Ho;
    }
    int wofs[C::CT][2];  // A-fragment byte offsets inside a W tap tile (swizzled rows), k-halves 0/1
#pragma unroll
    for (int c = 0; c < C::CT; ++c)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wofs[c][ks] = lds_off(wave_c * (C::CT * 32) + c * 32 + cperm(l31), 2 * ks + h);

    v16i acc[C::CT][C::PT];
#pragma unroll
    for (int c = 0; c < C::CT; ++c)
#pragma unroll
        for (int q = 0; q < C::PT; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][q][r] = 0;

    const long long t_begin = prof ? (long long)__builtin_readcyclecounter() : 0;
    // Barrier #0: band(0), W(0), W(1) visible.  Barrier #(s+1) closes step s; BEFORE it the issuing waves have
    // seen W(s+2) (and a band prefetch) land, so that the MFMA waves may request the first fragments of step
    // s+1 already at the end of step s and cross the barrier with their LDS reads in flight (the exposed
    // barrier + first-fetch latency was ~500 of ~2000 cycles per step).  Ring stage (s+2)%3 was last read in
    // step s-1, i.e. before barrier #s, after which W(s+2) is issued.
    if (C::NPROD > 0 && !consumer) {
        int cc = 0, kh = 0;
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        for (int s = 0; s < nsteps; ++s) {
            if (!HAWQ_DBG_BIT(p.dbg, 1)) issue_step(s, cc, kh);
            wait_step(s, cc, kh);
            __builtin_amdgcn_s_barrier();
            if (++kh == 3) kh = 0, ++cc;
        }
    } else {
        if (C::NPROD > 0) __builtin_amdgcn_s_setprio(2);  // MFMA waves win issue arbitration against the producers
        const bool early = wave < C::NW / 2;  // waves w and w + NW/2 share a SIMD (only used when NPROD == 0)
        unsigned ap[C::PT], wp[C::CT][2];
        // operand bases of step (s, cc, kh); everything inside the batches is base + compile-time immediate
        auto bases = [&](int s, int cc, int kh) {
            const char *bst = band + (C::BSTAGES > 1 ? (cc & 1) * C::BAND_BYTES : 0) + h * C::PLANE, *wst = wring + (s % C::WS) * C::WSTAGE;
#pragma unroll
            for (int q = 0; q < C::PT; ++q) {
                const bool rowok = (unsigned)(yy[q] + kh - 1) < (unsigned)p.Ho;  // else: another image's row / padding -> zeros
                ap[q] = lds_addr(bst) + (rowok ? bp0[q] + kh * Wb : C::BAND_PX - 4) * 16;
            }
#pragma unroll
            for (int c = 0; c < C::CT; ++c)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) wp[c][ks] = lds_addr(wst) + wofs[c][ks];
        };
        // 6 batches per step (tap kw = b / 2, k-half ks = b % 2) of CT x PT MFMAs, software-pipelined by hand: the
        // fragments of the next batch are requested from LDS before the MFMAs of the current one are issued.
        v4i wf[2][C::CT], af[2][C::PT];
#define BAND_FETCH(B, BUF)                                                                                        \
    if (!HAWQ_DBG_BIT(p.dbg, 4)) {                                                                                           \
        _Pragma("unroll") for (int c = 0; c < C::CT; ++c)                                                         \
            wf[BUF][c] = lds_read16<((B) >> 1) * C::WTAP>(wp[c][(B) & 1]);                                        \
        _Pragma("unroll") for (int q = 0; q < C::PT; ++q)                                                         \
            af[BUF][q] = lds_read16<((B) >> 1) * 16 + ((B) & 1) * (2 * C::PLANE)>(ap[q]);                         \
    }
#define BAND_MMA(B)                                                                                               \
    {                                                                                                             \
        _Pragma("unroll") for (int c = 0; c < C::CT; ++c) pin(wf[(B) & 1][c]);                                    \
        _Pragma("unroll") for (int q = 0; q < C::PT; ++q) pin(af[(B) & 1][q]);                                    \
        if constexpr (NIB) {                                                                                      \
            _Pragma("unroll") for (int hf = 0; hf < 2; ++hf) {                                                    \
                v4i w8[C::CT], a8[C::PT];                                                                         \
                _Pragma("unroll") for (int c = 0; c < C::CT; ++c)                                                 \
                    w8[c] = unpack16<true>((unsigned)wf[(B) & 1][c][2 * hf], (unsigned)wf[(B) & 1][c][2 * hf + 1]); \
                _Pragma("unroll") for (int q = 0; q < C::PT; ++q)                                                 \
                    a8[q] = unpack16<false>((unsigned)af[(B) & 1][q][2 * hf], (unsigned)af[(B) & 1][q][2 * hf + 1]); \
                _Pragma("unroll") for (int c = 0; c < C::CT; ++c)                                                 \
                    _Pragma("unroll") for (int q = 0; q < C::PT; ++q)                                             \
                        acc[c][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w8[c], a8[q], acc[c][q], 0, 0, 0);      \
            }                                                                                                     \
        } else {                                                                                                  \
            _Pragma("unroll") for (int c = 0; c < C::CT; ++c)                                                     \
                _Pragma("unroll") for (int q = 0; q < C::PT; ++q)                                                 \
                    acc[c][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[(B) & 1][c], af[(B) & 1][q], acc[c][q], 0, 0, 0); \
        }                                                                                                         \
    }
#define BAND_BATCH(B)                                                                                             \
    {                                                                                                             \
        BAND_FETCH((B) + 1, ((B) + 1) & 1)                                                                        \
        wait_lgkm<C::CT + C::PT>();                                                                               \
        BAND_MMA(B)                                                                                               \
    }
        int cc = 0, kh = 0;
        if (C::NPROD == 0) wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        bases(0, 0, 0);
        BAND_FETCH(0, 0)
        for (int s = 0; s < nsteps; ++s) {
            BAND_BATCH(0)
            if (C::NPROD == 0 && early && !HAWQ_DBG_BIT(p.dbg, 1)) issue_step(s, cc, kh);
            BAND_BATCH(1)
            BAND_BATCH(2)
            BAND_BATCH(3)
            if (C::NPROD == 0 && !early && !HAWQ_DBG_BIT(p.dbg, 1)) issue_step(s, cc, kh);
            BAND_BATCH(4)
            if (++kh == 3) kh = 0, ++cc;
            if (s + 1 < nsteps) {  // first fragments of the next step (its operands are visible since the previous barrier)
                bases(s + 1, cc, kh);
                BAND_FETCH(0, 0)
                wait_lgkm<C::CT + C::PT>();
            } else {
                wait_lgkm<0>();
            }
            BAND_MMA(5)
            if (C::NPROD == 0) wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
        }
#undef BAND_FETCH
#undef BAND_MMA
#undef BAND_BATCH
        if (C::NPROD > 0) __builtin_amdgcn_s_setprio(0);
    }
    const long long t_loop_end = prof ? (long long)__builtin_readcyclecounter() : 0;
    if constexpr (NIB) {  // weights were unpacked as value*16
#pragma unroll
        for (int c = 0; c < C::CT; ++c)
#pragma unroll
            for (int q = 0; q < C::PT; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][q][r] >>= 4;
    }
    __syncthreads();
    if constexpr (EPI == HAWQ_EPI_RESIDUAL) {
        using S = Stage<C>;
        static_assert(C::BM * C::BN * 2 <= C::WS * C::WSTAGE && C::BM * C::BN <= C::BSTAGES * C::BAND_BYTES, "epilogue tiles alias the rings");
        constexpr int NWALL = C::NT / 64, PPW = C::BM * S::RCPR / 64 / NWALL;
        static_assert(PPW * NWALL * 64 == C::BM * S::RCPR, "residual tile pieces");
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int base = (i * NWALL + wave) * 64, idx = base + lane;
            const int row = idx / S::RCPR, j = idx % S::RCPR;
            const int grow = (m0 + row < p.M) ? m0 + row : m0;
            const char *src = (const char *)p.res_in + ((size_t)grow * p.Cout + c0) * 2 + ((j ^ S::rsw(row)) << 4);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(wring + base * 16), 16, 0, 0);
        }
        wait_vmcnt<0>();
        __syncthreads();
    }
    v16i dummy[1][1];
    epilogue_fast<C, EPI, false, TIE ? 2 : 0>(p, acc, dummy, m0, c0, smem, wring, ctab_lds, consumer);
    if (prof && blockIdx.x == 8 && t == 0) {
        p.dbgbuf[0] = t_begin - t_entry, p.dbgbuf[1] = t_loop_end - t_begin;
        p.dbgbuf[2] = (long long)__builtin_readcyclecounter() - t_loop_end, p.dbgbuf[3] = nsteps;
    }
}

using B0 = BandCfg<256, 64, 4, 1, 512, 1, 2, 0>;    // Cin == Cout == 64 (stage 1): 4 waves x (64 px x 64 ch), 2 workgroups per CU
using B1 = BandCfg<256, 128, 4, 2, 512, 2, 1, 8>;   // 8 MFMA waves x (64 px x 64 ch) + 8 producer waves (LDS-DMA ingest scales with the issuing waves)
using B2 = BandCfg<128, 128, 2, 4, 256, 2, 1, 8, 4>;   // 8 MFMA waves x (64 px x 32 ch) + 8 producers: twice the workgroups for 14x14 / 7x7
using B3 = BandCfg<256, 128, 4, 2, 384, 2, 1, 8, 4>;   // B1 with a 384-pixel band (14x14 / 7x7 maps): room for a 4-stage W ring
using B4 = BandCfg<128, 128, 2, 4, 256, 2, 1, 8, 5>;   // B2 with a 5-stage W ring (W issued 4 filter rows ahead)
// round 3: in stages 3-4 a launch has fewer workgroups than the chip has CUs and lasts as long as ONE workgroup (profiles/
// r03_mallprobe.md), so halve the work per workgroup and let two of them share a CU: 128 px x 64 ch, 4 MFMA waves (64 px x 32 ch)
// + 4 producers, 3-stage W ring of 12 KiB stages: 68 KiB.  4x the workgroups of B1 / B3 (392 for the 7x7 layers at batch 128)
using B5 = BandCfg<128, 64, 2, 2, 256, 2, 4, 4, 3>;
constexpr int NUM_BAND_TILES = 6;

typedef void (*KernelFn)(const ConvP);
struct BandInfo { KernelFn fn[2][2][2]; int bm, bn, band_px, bstages, lds, nt; };  // fn[hawq4][exact-tie][residual]
#define BAND_FN(B, N, T) {conv3x3_band_kernel<B, N, T, HAWQ_EPI_REQUANT>, conv3x3_band_kernel<B, N, T, HAWQ_EPI_RESIDUAL>}
#define BAND_ENTRY(B) {{{BAND_FN(B, false, false), BAND_FN(B, false, true)}, {BAND_FN(B, true, false), BAND_FN(B, true, true)}}, B::BM, B::BN, B::BAND_PX, B::BSTAGES, B::LDS_BYTES + B::BN * 16, B::NT}
const BandInfo kBand[NUM_BAND_TILES] = {BAND_ENTRY(B0), BAND_ENTRY(B1), BAND_ENTRY(B2), BAND_ENTRY(B3), BAND_ENTRY(B4), BAND_ENTRY(B5)};
// tile ids tried, in this order, when the caller leaves the choice open for a layer that needs a band kernel
const int kBandPreference[NUM_BAND_TILES] = {0, 3, 1, 2, 4, 5};

// Does band tile `bi` take this layer?  (3x3 / stride 1 / pad 1, single branch, fast-contract tables, int8 or hawq4
// operands, REQUANT or 16-bit RESIDUAL epilogue, the band of a pixel tile fits the LDS stage)
bool band_applies(const BandInfo &bi, const hawq_conv_args *a) {
    const int wo = a->W, band_rows = (bi.bm + wo - 1) / wo + 1 + 2;
    const bool nib = a->in_bits == 4 && a->w_bits == 4;
    const bool res = a->epilogue == HAWQ_EPI_RESIDUAL;
    const bool epi_ok = (a->epilogue == HAWQ_EPI_REQUANT && a->out_q) ||
                        (res && a->res_in && a->res_in_bits == 16 && (!a->res_out || a->res_out_bits == 16));
    const bool dense = (a->in_pitch == 0 || a->in_pitch == a->Cin * a->in_bits / 8) && (a->out_pitch == 0 || a->out_pitch == a->Cout);
    return a->KH == 3 && a->KW == 3 && a->stride == 1 && a->pad == 1 && a->in2 == nullptr && a->fast_tables != 0 && epi_ok && dense &&
           ((a->in_bits == 8 && a->w_bits == 8) || (nib && a->Cin % 128 == 0)) && a->Cout % bi.bn == 0 &&
           band_rows * (wo + 2) <= bi.band_px - 4 && (bi.bstages > 1 || (a->Cin >> (nib ? 7 : 6)) == 1);
}

using T0 = Cfg<128, 128, 2, 2, 3>;
using T1 = Cfg<256, 64, 4, 1, 3>;
using T2 = Cfg<64, 64, 2, 2, 4>;
using T3 = Cfg<128, 64, 2, 2, 4>;
// two sub-chunks (K = 128) per ring stage and barrier for the long-K / few-workgroup layers (stages 3-4);
// need an even chunk count, otherwise the launcher falls back to the single-sub-chunk twin
using T4 = Cfg<128, 128, 2, 2, 3, 2>;  // K = 128 per barrier
using T5 = Cfg<64, 64, 2, 2, 4, 2>;
using T6 = Cfg<128, 64, 2, 2, 3, 2>;
// 8-wave workgroups: twice the waves issuing LDS-DMA for the same tile (the DMA rate per wave, not the ring
// depth, bounds the long-K layers when there is at most ~1 workgroup per CU - profiles/r01_ubench_dma_rate.txt)
using T7 = Cfg<128, 128, 2, 4, 3>;
using T8 = Cfg<128, 128, 2, 4, 3, 2>;
using T9 = Cfg<256, 128, 4, 2, 3>;
// shallow rings for the short-K, epilogue-heavy layers (1x1 expand + residual): less LDS per workgroup = more
// workgroups per CU (4 / 5 instead of 3), whose load, epilogue-VALU and store phases then overlap
using T10 = Cfg<64, 64, 2, 2, 3, 1, 6>;
using T11 = Cfg<64, 64, 2, 2, 2, 1, 6>;
using T12 = Cfg<128, 64, 2, 2, 2, 1, 3>;
using T13 = Cfg<128, 128, 2, 4, 2, 1, 2>;
// intra-workgroup split-K (KG groups of waves on alternating 64-channel sub-chunks) for the long-K, few-pixel layers
using T14 = Cfg<64, 64, 2, 2, 3, 2, 4, 2>;     // 8 waves: 2 groups x (2 x 2 waves of 32 x 32), 48 KiB
// (measured, not kept: 16 waves in 4 K groups, and 5-6-stage rings for the same layers - 96-120 KiB of LDS leave one workgroup
// per CU, and these launches live on the overlap BETWEEN workgroups: 15-60 % slower.  Split-K itself ties with the best
// plain tile; it stays as an autotuner choice.)
constexpr int NUM_TILES = 15;

// single-branch kernels: epilogue {RAW, REQUANT, RESIDUAL, DEQUANT} x bit variant {run-time, 8/8, 4/4};
// dual-branch (RESIDUAL + identity conv): {run-time, 88/88, 44/44, 88/44, 44/88}
struct TileInfo {
    int BM, BN, lds, ksub, twin, nt, ns, kg;  // twin: tile id to fall back to when KSUB does not divide the chunk count / the layer is not on the asynchronous pipeline
    KernelFn single[4][3];
    KernelFn dual[5];
    KernelFn tie_single[2][2];  // exact-tie instantiations: {REQUANT, RESIDUAL} x {8/8, 4/4}
    KernelFn tie_dual[2];       // {88/88, 44/44}
};
#define SINGLE_ROW(T, E) \
    { conv_kernel<T, E, false, 0, 0>, conv_kernel<T, E, false, 0x88, 0>, conv_kernel<T, E, false, 0x44, 0> }
#define TILE_ENTRY(T, TWIN)                                                                                          \
    {                                                                                                          \
        T::BM, T::BN, T::LDS_BYTES, T::KSUB, TWIN, T::NT, T::NS, T::KG,                                                   \
            {SINGLE_ROW(T, HAWQ_EPI_RAW), SINGLE_ROW(T, HAWQ_EPI_REQUANT), SINGLE_ROW(T, HAWQ_EPI_RESIDUAL),   \
             SINGLE_ROW(T, HAWQ_EPI_DEQUANT)},                                                                 \
        {                                                                                                      \
            conv_kernel<T, HAWQ_EPI_RESIDUAL, true, 0, 0>, conv_kernel<T, HAWQ_EPI_RESIDUAL, true, 0x88, 0x88>, \
                conv_kernel<T, HAWQ_EPI_RESIDUAL, true, 0x44, 0x44>,                                           \
                conv_kernel<T, HAWQ_EPI_RESIDUAL, true, 0x88, 0x44>,                                           \
                conv_kernel<T, HAWQ_EPI_RESIDUAL, true, 0x44, 0x88>                                            \
        },                                                                                                     \
        {{conv_kernel<T, HAWQ_EPI_REQUANT, false, 0x88, 0, true>, conv_kernel<T, HAWQ_EPI_REQUANT, false, 0x44, 0, true>},   \
         {conv_kernel<T, HAWQ_EPI_RESIDUAL, false, 0x88, 0, true>, conv_kernel<T, HAWQ_EPI_RESIDUAL, false, 0x44, 0, true>}}, \
        {conv_kernel<T, HAWQ_EPI_RESIDUAL, true, 0x88, 0x88, true>, conv_kernel<T, HAWQ_EPI_RESIDUAL, true, 0x44, 0x44, true>} \
    }
const TileInfo kTiles[NUM_TILES] = {TILE_ENTRY(T0, 0), TILE_ENTRY(T1, 1), TILE_ENTRY(T2, 2), TILE_ENTRY(T3, 3),
                                    TILE_ENTRY(T4, 0), TILE_ENTRY(T5, 2), TILE_ENTRY(T6, 3),
                                    TILE_ENTRY(T7, 7), TILE_ENTRY(T8, 7), TILE_ENTRY(T9, 9),
                                    TILE_ENTRY(T10, 10), TILE_ENTRY(T11, 11), TILE_ENTRY(T12, 12), TILE_ENTRY(T13, 13),
                                    TILE_ENTRY(T14, 2)};

// kernels whose staged epilogue needs more than the default 64 KiB of dynamic LDS
bool raise_lds_limits() {
    bool ok = true;
    for (const TileInfo &ti : kTiles) {
        const int lds = ti.lds + ti.BM * ti.BN * 2 + ti.BN * 32;
        if (lds <= 64 * 1024) continue;
        for (int v = 0; v < 2; ++v)
            ok &= hipFuncSetAttribute((const void *)ti.tie_single[1][v], hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess &&
                  hipFuncSetAttribute((const void *)ti.tie_dual[v], hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
        for (int v = 1; v < 3; ++v)
            ok &= hipFuncSetAttribute((const void *)ti.single[2][v], hipFuncAttributeMaxDynamicSharedMemorySize, lds) ==
                  hipSuccess;
        for (int v = 1; v < 5; ++v)
            ok &= hipFuncSetAttribute((const void *)ti.dual[v], hipFuncAttributeMaxDynamicSharedMemorySize, lds) ==
                  hipSuccess;
    }
    return ok;
}

int pick_tile(int M, int Cout, bool dual) {
    // Enough workgroups to fill 256 CUs a few times over, the largest tile that allows it.
    auto nwg = [&](int t) { return ((M + kTiles[t].BM - 1) / kTiles[t].BM) * (Cout / kTiles[t].BN); };
    // two accumulator sets: only the 32-channel-per-wave tiles stay within 256 VGPRs without spilling
    if (dual) return nwg(3) >= 512 ? 3 : 2;
    if (Cout % 128 == 0 && nwg(0) >= 1024) return 0;
    if (nwg(1) >= 1024 && !dual) return 1;
    if (Cout % 128 == 0 && nwg(0) >= 512) return 0;
    if (nwg(3) >= 512) return 3;
    return 2;
}

}  // namespace

// band_persist.hip: weight-stationary persistent 3x3 kernel for Cin == Cout == 64 (the id after the band tiles)
bool band_persist_applies(const hawq_conv_args *a);
int band_persist_launch(const hawq_conv_args *a, int exact_tie, int dbg, int wgs_per_cu, void *stream);

// band_v2.hip: the round-5 3x3 kernels (ids after the weight-stationary kernel's two); gemm_v2.hip: the round-5 streaming 1x1 kernels (last ids)
int gemm_v2_count(void);
bool gemm_v2_applies(const hawq_conv_args *a, int v);
int gemm_v2_launch(const hawq_conv_args *a, int v, int exact_tie, int dbg, void *stream);
int band_v2_count(void);
bool band_v2_applies(const hawq_conv_args *a, int v);
int band_v2_launch(const hawq_conv_args *a, int v, int exact_tie, int dbg, void *stream);

extern "C" int hawq_conv2d_num_tiles(void) { return NUM_TILES + NUM_BAND_TILES + 2 + band_v2_count() + gemm_v2_count(); }  // the special-purpose kernels are the last ids
extern "C" int hawq_conv2d_num_gemm2_tiles(void) { return gemm_v2_count(); }
extern "C" int hawq_conv2d_gemm2_first(void) { return NUM_TILES + NUM_BAND_TILES + 2 + band_v2_count() + 1; }
// + the weight-stationary kernel of band_persist.hip with 1 / 2 workgroups per CU + the round-5 kernels of band_v2.hip
extern "C" int hawq_conv2d_num_band_tiles(void) { return NUM_BAND_TILES + 2 + band_v2_count() + gemm_v2_count(); }

extern "C" int hawq_conv2d_num_band2_tiles(void) { return band_v2_count(); }
extern "C" int hawq_conv2d_band2_tile(const hawq_conv_args *a) {
    if (!a || a->W <= 0 || a->H <= 0 || a->Cin <= 0 || a->Cout <= 0) return 0;
    for (int v = 0; v < band_v2_count(); ++v)
        if (band_v2_applies(a, v)) return NUM_TILES + NUM_BAND_TILES + 2 + v + 1;
    return 0;
}

extern "C" int hawq_conv2d_band_tile(const hawq_conv_args *a) {
    if (!a || a->W <= 0 || a->H <= 0 || a->Cin <= 0 || a->Cout <= 0) return 0;
    for (int k = 0; k < NUM_BAND_TILES; ++k)
        if (band_applies(kBand[kBandPreference[k]], a)) return NUM_TILES + kBandPreference[k] + 1;
    return 0;
}

extern "C" int hawq_conv2d(const hawq_conv_args *a, void *stream) {
    HAWQ_REQUIRE(a != nullptr, "hawq_conv2d: null args");
    HAWQ_REQUIRE(a->in && a->wgt && a->bias, "hawq_conv2d: in/wgt/bias must be non-null");
    HAWQ_REQUIRE(a->Cin > 0 && a->Cin % 64 == 0, "hawq_conv2d: Cin=%d must be a positive multiple of 64", a->Cin);
    HAWQ_REQUIRE(a->Cout > 0 && a->Cout % 64 == 0, "hawq_conv2d: Cout=%d must be a positive multiple of 64", a->Cout);
    HAWQ_REQUIRE((a->in_bits == 8 || a->in_bits == 4) && (a->w_bits == 8 || a->w_bits == 4),
                 "hawq_conv2d: in_bits/w_bits must be 4 or 8 (got %d/%d)", a->in_bits, a->w_bits);
    HAWQ_REQUIRE(a->KH > 0 && a->KW > 0 && a->stride > 0 && a->pad >= 0 && a->N > 0 && a->H > 0 && a->W > 0,
                 "hawq_conv2d: bad geometry");
    ConvP p;
    p.in = (const uint8_t *)a->in;
    p.wgt = (const uint8_t *)a->wgt;
    p.bias = a->bias;
    p.N = a->N, p.H = a->H, p.W = a->W, p.Cin = a->Cin, p.Cout = a->Cout;
    p.KH = a->KH, p.KW = a->KW, p.stride = a->stride, p.pad = a->pad;
    p.Ho = (a->H + 2 * a->pad - a->KH) / a->stride + 1;
    p.Wo = (a->W + 2 * a->pad - a->KW) / a->stride + 1;
    HAWQ_REQUIRE(p.Ho > 0 && p.Wo > 0, "hawq_conv2d: empty output");
    const long long M = (long long)a->N * p.Ho * p.Wo;
    HAWQ_REQUIRE(M * (long long)a->Cout < (1ll << 40) && M < (1ll << 30), "hawq_conv2d: problem too large");
    p.M = (int)M;
    p.in_bits = a->in_bits, p.w_bits = a->w_bits;
    const bool dual = a->in2 != nullptr;
    p.in2 = (const uint8_t *)a->in2, p.wgt2 = (const uint8_t *)a->wgt2, p.bias2 = a->bias2;
    p.H2 = a->H2, p.W2 = a->W2, p.Cin2 = a->Cin2, p.stride2 = a->stride2;
    p.in2_bits = a->in2_bits, p.w2_bits = a->w2_bits;
    if (dual) {
        HAWQ_REQUIRE(a->epilogue == HAWQ_EPI_RESIDUAL, "hawq_conv2d: second branch needs HAWQ_EPI_RESIDUAL");
        HAWQ_REQUIRE(a->wgt2 && a->bias2 && a->m_id && a->e_id, "hawq_conv2d: second branch tables missing");
        HAWQ_REQUIRE(a->Cin2 > 0 && a->Cin2 % 64 == 0, "hawq_conv2d: Cin2 must be a multiple of 64");
        HAWQ_REQUIRE((a->in2_bits == 8 || a->in2_bits == 4) && (a->w2_bits == 8 || a->w2_bits == 4),
                     "hawq_conv2d: in2_bits/w2_bits must be 4 or 8");
        HAWQ_REQUIRE((a->H2 - 1) / a->stride2 + 1 == p.Ho && (a->W2 - 1) / a->stride2 + 1 == p.Wo,
                     "hawq_conv2d: second branch output grid differs");
    }
    p.relu = a->relu;
    p.m = a->m, p.e = a->e, p.m_id = a->m_id, p.e_id = a->e_id;
    p.m_id_s = a->m_id_scalar, p.e_id_s = a->e_id_scalar;
    p.res_in = a->res_in, p.res_in_bits = a->res_in_bits;
    p.res_no_relu = a->res_no_relu, p.res_clamp16 = a->res_clamp16;
    p.res_out = a->res_out, p.res_out_bits = a->res_out_bits;
    p.out_q = a->out_q, p.out_bits = a->out_bits, p.q_lo = a->q_lo, p.q_hi = a->q_hi, p.mq = a->mq, p.eq = a->eq;
    p.out_acc = a->out_acc, p.out_f32 = a->out_f32, p.fscale = a->fscale, p.ldo = a->ldo, p.n_valid = a->n_valid;
    {   // ABI 4: pitches of narrow tensors (0 = dense)
        const int dense_in = a->Cin * a->in_bits / 8;
        p.in_pitch = a->in_pitch ? a->in_pitch : dense_in, p.out_pitch = a->out_pitch ? a->out_pitch : a->Cout;
        if (p.in_pitch != dense_in)
            HAWQ_REQUIRE(a->in_pitch > 0 && a->in_pitch % 16 == 0 && a->in_pitch < dense_in && a->in_bits == 8 && !a->in_planar,
                         "hawq_conv2d: in_pitch=%d needs a multiple of 16 below the dense row (%d bytes), int8 activations, NHWC rows", a->in_pitch, dense_in);
        if (p.out_pitch != a->Cout) {
            const bool fast_ = a->fast_tables != 0;
            HAWQ_REQUIRE(a->out_pitch > 0 && a->out_pitch % 16 == 0 && a->out_pitch < a->Cout, "hawq_conv2d: out_pitch=%d needs a multiple of 16 below Cout=%d", a->out_pitch, a->Cout);
            HAWQ_REQUIRE((a->epilogue == HAWQ_EPI_REQUANT || a->epilogue == HAWQ_EPI_RESIDUAL) && !a->in2 && !a->out_planar && (!a->out_q || a->out_bits == 8),
                         "hawq_conv2d: out_pitch exists for single-branch REQUANT / RESIDUAL launches with int8 NHWC outputs");
            const bool direct_ = a->epilogue == HAWQ_EPI_RESIDUAL &&   // the direct epilogue addresses pitched rows (ConvP.gfast), the fast RESIDUAL tiles are dense
                                 (a->res_no_relu || a->res_clamp16 || !a->res_in || a->res_in_bits == 32 || (a->res_out && a->res_out_bits == 32));
            HAWQ_REQUIRE(!fast_ || a->epilogue == HAWQ_EPI_REQUANT || direct_,
                         "hawq_conv2d: out_pitch with fast_tables needs the REQUANT epilogue or a signed / 32-bit RESIDUAL (the unsigned 16-bit fast RESIDUAL tiles are dense)");
            HAWQ_REQUIRE(a->n_valid <= a->out_pitch, "hawq_conv2d: n_valid=%d exceeds out_pitch=%d", a->n_valid, a->out_pitch);
        }
    }
    p.flags = a->flags;
    p.ctab = a->ctab, p.ctab_id = a->ctab_id;
    static const int dbg_env = HAWQ_DBG_ENV();
    p.dbg = a->fast_tables ? dbg_env : 0;  // ablations only touch the fused-plan launches
    p.dbgbuf = nullptr;
    static long long *dbg_dev = nullptr;
    if (HAWQ_DBG_BIT(p.dbg, 128) && !dbg_dev) (void)hipMalloc(&dbg_dev, 64 * sizeof(long long));
    if (HAWQ_DBG_BIT(p.dbg, 128)) p.dbgbuf = dbg_dev;
    const bool fast = a->fast_tables != 0;
    p.k0 = (a->fast_tables & 4) ? 2 : ((a->fast_tables & 2) ? 1 : 0);
    p.ck0 = (a->fast_tables & 8) != 0;
    if (fast && (a->epilogue == HAWQ_EPI_REQUANT || a->epilogue == HAWQ_EPI_RESIDUAL)) {
        HAWQ_REQUIRE(a->ctab && (!dual || a->ctab_id), "hawq_conv2d: fast_tables needs ctab (and ctab_id)");
        if (a->epilogue == HAWQ_EPI_REQUANT && a->relu && p.q_lo < 0) p.q_lo = 0;  // ReLU folded into the clamp
        const bool direct_form = a->epilogue == HAWQ_EPI_RESIDUAL && !dual &&   // runs the direct epilogue (clamps q on both sides), see ConvP.gfast
                                 (a->res_no_relu || a->res_clamp16 || !a->res_in || a->res_in_bits == 32 || (a->res_out && a->res_out_bits == 32));
        HAWQ_REQUIRE(a->epilogue != HAWQ_EPI_RESIDUAL || !a->out_q || a->q_lo <= 0 || direct_form,
                     "hawq_conv2d: fast RESIDUAL needs q_lo <= 0");
    }
    auto e_fast = [](int ek) { return (ek & 0xff) >= 33 && (ek & 0xff) <= 62; };
    auto e_any = [](int ek) { return (ek & 0xff) >= 1 && (ek & 0xff) <= 62 && (ek >> 8) >= 0 && (ek >> 8) < 31; };
    if (a->epilogue == HAWQ_EPI_RESIDUAL) {
        if (a->out_q) {
            HAWQ_REQUIRE(a->mq >= 0 && e_any(a->eq), "hawq_conv2d: bad (mq, eq)");
            HAWQ_REQUIRE(!fast || e_fast(a->eq), "hawq_conv2d: fast_tables needs eq in [33,62]");
            HAWQ_REQUIRE(!fast || (a->q_hi >= 0 && a->q_hi <= 32767), "hawq_conv2d: fast_tables needs 0 <= q_hi <= 32767 for the next QuantAct");
            HAWQ_REQUIRE(p.k0 != 1 || (a->eq >> 8) == 0, "hawq_conv2d: fast_tables bit 1 (no pre-shifts) but eq carries one");
        } else {
            p.mq = 0, p.eq = 33;
        }
        if (!dual && !a->res_in) {   // no identity branch at all (ABI 3, general path)
            p.m_id_s = 0, p.e_id_s = 33;
        } else if (!dual) {
            HAWQ_REQUIRE(a->m_id_scalar >= 0 && e_any(a->e_id_scalar), "hawq_conv2d: bad (m_id_scalar, e_id_scalar)");
            HAWQ_REQUIRE(!fast || e_fast(a->e_id_scalar), "hawq_conv2d: fast_tables needs e_id_scalar in [33,62]");
        } else {
            p.m_id_s = 0, p.e_id_s = 33;
        }
    } else {
        p.mq = 0, p.eq = 33, p.m_id_s = 0, p.e_id_s = 33;
    }
    int slot = -1;
    switch (a->epilogue) {
        case HAWQ_EPI_RAW:
            HAWQ_REQUIRE(a->out_acc, "hawq_conv2d: RAW needs out_acc");
            slot = 0;
            break;
        case HAWQ_EPI_REQUANT:
            HAWQ_REQUIRE(a->out_q && a->m && a->e, "hawq_conv2d: REQUANT needs out_q, m, e");
            HAWQ_REQUIRE(a->out_bits == 8 || a->out_bits == 4, "hawq_conv2d: out_bits must be 4 or 8");
            slot = 1;
            break;
        case HAWQ_EPI_RESIDUAL:
            HAWQ_REQUIRE(a->m && a->e, "hawq_conv2d: RESIDUAL needs m, e");
            HAWQ_REQUIRE(dual || !a->res_in || a->res_in_bits == 16 || a->res_in_bits == 32, "hawq_conv2d: res_in_bits 16/32");
            HAWQ_REQUIRE(!(dual && (a->res_no_relu || a->res_clamp16)), "hawq_conv2d: res_no_relu / res_clamp16 exist for single-branch launches");
            HAWQ_REQUIRE(!fast || !(a->res_no_relu || a->res_clamp16 || (!dual && !a->res_in)) || (a->fast_tables & 2) == 0,
                         "hawq_conv2d: fast_tables bit 1 is not defined for the signed / identity-free RESIDUAL forms");
            HAWQ_REQUIRE(!a->res_no_relu || !a->res_out || a->res_out_bits == 32, "hawq_conv2d: a residual stored without ReLU is signed: res_out_bits must be 32");
            HAWQ_REQUIRE(!a->res_out || a->res_out_bits == 32 || (a->res_out_bits == 16 && a->flags),
                         "hawq_conv2d: res_out_bits 16 (with flags) or 32");
            HAWQ_REQUIRE(!a->out_q || a->out_bits == 8 || a->out_bits == 4, "hawq_conv2d: out_bits must be 4 or 8");
            HAWQ_REQUIRE(a->res_out || a->out_q, "hawq_conv2d: RESIDUAL needs res_out and/or out_q");
            slot = 2;
            break;
        case HAWQ_EPI_DEQUANT:
            HAWQ_REQUIRE(a->out_f32 && a->fscale && a->ldo > 0, "hawq_conv2d: DEQUANT needs out_f32, fscale, ldo");
            slot = 3;
            break;
        default:
            HAWQ_REQUIRE(false, "hawq_conv2d: unknown epilogue %d", a->epilogue);
    }
    p.in_planar = a->in_planar, p.out_planar = a->out_planar;
    HAWQ_REQUIRE((a->in_planar | a->out_planar | 1) == 1, "hawq_conv2d: in_planar / out_planar must be 0 or 1");
    HAWQ_REQUIRE(!a->out_planar || (fast && a->out_q && (a->epilogue == HAWQ_EPI_REQUANT || a->epilogue == HAWQ_EPI_RESIDUAL) &&
                                    !(a->epilogue == HAWQ_EPI_RESIDUAL && a->res_out && a->res_out_bits == 32)),
                 "hawq_conv2d: out_planar needs the fast-contract REQUANT / RESIDUAL epilogue");
    int tile = a->tile > 0 ? a->tile - 1 : pick_tile(p.M, p.Cout, dual);
    if (a->tile == 0 && a->in_planar) {  // only the band kernels read planar activations: first one that takes the layer
        tile = -1;
        for (int k = 0; k < NUM_BAND_TILES && tile < 0; ++k)
            if (band_applies(kBand[kBandPreference[k]], a)) tile = NUM_TILES + kBandPreference[k];
        for (int v = 0; v < band_v2_count() && tile < 0; ++v)   // a layer only the round-5 kernels take
            if (band_v2_applies(a, v)) tile = NUM_TILES + NUM_BAND_TILES + 2 + v;
        HAWQ_REQUIRE(tile >= 0, "hawq_conv2d: in_planar input but no 3x3 band kernel takes this layer");
    }
    if (tile >= NUM_TILES + NUM_BAND_TILES + 2 + band_v2_count() && tile < NUM_TILES + NUM_BAND_TILES + 2 + band_v2_count() + gemm_v2_count()) {
        const int v = tile - (NUM_TILES + NUM_BAND_TILES + 2 + band_v2_count());
        HAWQ_REQUIRE(gemm_v2_applies(a, v), "hawq_conv2d: tile %d (round-5 1x1 kernel) does not apply to this layer", a->tile);
        return gemm_v2_launch(a, v, p.k0 == 2, p.dbg, stream);
    }
    if (tile >= NUM_TILES + NUM_BAND_TILES + 2 && tile < NUM_TILES + NUM_BAND_TILES + 2 + band_v2_count()) {
        const int v = tile - (NUM_TILES + NUM_BAND_TILES + 2);
        HAWQ_REQUIRE(band_v2_applies(a, v), "hawq_conv2d: tile %d (round-5 3x3 kernel) does not apply to this layer", a->tile);
        return band_v2_launch(a, v, p.k0 == 2, p.dbg, stream);
    }
    if (tile == NUM_TILES + NUM_BAND_TILES || tile == NUM_TILES + NUM_BAND_TILES + 1) {
        HAWQ_REQUIRE(band_persist_applies(a), "hawq_conv2d: tile %d (weight-stationary 3x3 kernel) does not apply to this layer", a->tile);
        return band_persist_launch(a, p.k0 == 2, p.dbg, tile - (NUM_TILES + NUM_BAND_TILES) + 1, stream);
    }
    if (tile >= NUM_TILES && tile < NUM_TILES + NUM_BAND_TILES) {
        // 3x3 band kernels (LDS-resident input band shared by the 9 taps): fast-contract int8 / hawq4 layers only
        const BandInfo &bi = kBand[tile - NUM_TILES];
        const bool nib = a->in_bits == 4 && a->w_bits == 4;
        const bool res = a->epilogue == HAWQ_EPI_RESIDUAL;
        HAWQ_REQUIRE(band_applies(bi, a), "hawq_conv2d: tile %d (3x3 band kernel) does not apply to this layer", a->tile);
        static const bool band_attrs = [] {
            bool good = true;
            for (const BandInfo &b : kBand)
                for (int i = 0; i < 8; ++i)
                    good &= hipFuncSetAttribute((const void *)b.fn[i >> 2][(i >> 1) & 1][i & 1], hipFuncAttributeMaxDynamicSharedMemorySize, b.lds) == hipSuccess;
            return good;
        }();
        HAWQ_REQUIRE(band_attrs, "hawq_conv2d: hipFuncSetAttribute failed for the band kernels");
        const int grid_b = ((p.M + bi.bm - 1) / bi.bm) * (p.Cout / bi.bn);
        hipLaunchKernelGGL(bi.fn[nib ? 1 : 0][p.k0 == 2 ? 1 : 0][res ? 1 : 0], dim3(grid_b), dim3(bi.nt), bi.lds, (hipStream_t)stream, p);
        HAWQ_CHECK_HIP(hipGetLastError());
        if (HAWQ_DBG_BIT(p.dbg, 128) && p.dbgbuf) {  // experiment hook: per-phase cycles of one wave (synchronises!)
            long long hbuf[4];
            (void)hipStreamSynchronize((hipStream_t)stream);
            (void)hipMemcpy(hbuf, p.dbgbuf, sizeof(hbuf), hipMemcpyDeviceToHost);
            fprintf(stderr, "[band bm=%d bn=%d M=%d Cin=%d Cout=%d] steps %lld: prologue %lld | K loop %lld | epilogue %lld cycles (wave 0 of workgroup 8, s_memtime)\n",
                    bi.bm, bi.bn, p.M, p.Cin, p.Cout, hbuf[3], hbuf[0], hbuf[1], hbuf[2]);
        }
        return 0;
    }
    HAWQ_REQUIRE(!a->in_planar, "hawq_conv2d: in_planar activations are read by the 3x3 band kernels only (tile %d is not one)", a->tile);
    HAWQ_REQUIRE(tile >= 0 && tile < NUM_TILES, "hawq_conv2d: bad tile id %d", a->tile);
    if (p.Cout % kTiles[tile].BN != 0) tile = 2;
    if (kTiles[tile].ksub > 1) {  // K = 128 (or more) per barrier: async pipeline with chunk counts the stage divides
        const int nk1 = a->KH * a->KW * (a->Cin >> 6), nk2 = dual ? (a->Cin2 >> 6) : 0;
        const bool all88 = a->in_bits == 8 && a->w_bits == 8 && (!dual || (a->in2_bits == 8 && a->w2_bits == 8));
        // (4/4 layers check the divisibility of their 128-channel chunk count inside the kernel and otherwise
        //  run the register-staged loop, which works for any tile without split-K)
        if (all88 && ((nk1 % kTiles[tile].ksub) || (nk2 % kTiles[tile].ksub))) tile = kTiles[tile].twin;
        if (kTiles[tile].kg > 1) {   // split-K tiles exist on the asynchronous pipelines only
            const bool all44 = a->in_bits == 4 && a->w_bits == 4 && (!dual || (a->in2_bits == 4 && a->w2_bits == 4));
            const int ks = kTiles[tile].ksub;
            const bool nib_ok = all44 && (a->Cin & 127) == 0 && (!dual || (a->Cin2 & 127) == 0) &&
                                ((a->KH * a->KW * (a->Cin >> 7)) % ks) == 0 && (!dual || ((a->Cin2 >> 7) % ks) == 0);
            const bool wide_res_ = a->epilogue == HAWQ_EPI_RESIDUAL && ((!dual && a->res_in_bits == 32) || (a->res_out && a->res_out_bits == 32) ||
                                                                        (!dual && (a->res_no_relu || a->res_clamp16 || !a->res_in)));
            const bool tables = a->epilogue == HAWQ_EPI_REQUANT || a->epilogue == HAWQ_EPI_RESIDUAL;
            if (!(all88 || nib_ok) || wide_res_ || (tables && !fast)) tile = kTiles[tile].twin;
        }
    }
    const TileInfo &ti = kTiles[tile];
    const int grid = ((p.M + ti.BM - 1) / ti.BM) * (p.Cout / ti.BN);
    // 32-bit residual tensors use the generic (run-time bit-width, direct epilogue) kernels
    const bool wide_res = a->epilogue == HAWQ_EPI_RESIDUAL &&
                          ((!dual && a->res_in_bits == 32) || (a->res_out && a->res_out_bits == 32));
    const bool needs_tables = slot == 1 || slot == 2;
    // the signed / identity-free RESIDUAL forms (MobileNetV2, ABI 3) live on the direct epilogue like the 32-bit residual tensors do;
    // with fast_tables they run it with the fast contract's arithmetic (ConvP.gfast)
    const bool signed_res = a->epilogue == HAWQ_EPI_RESIDUAL && !dual && (a->res_no_relu || a->res_clamp16 || !a->res_in);
    p.gfast = fast && a->epilogue == HAWQ_EPI_RESIDUAL && !dual && (wide_res || signed_res);
    auto variant = [&](int ab, int wb) {
        if (wide_res || signed_res || (needs_tables && !fast)) return 0;
        return ab == 8 && wb == 8 ? 1 : (ab == 4 && wb == 4 ? 2 : 0);
    };
    KernelFn fn;
    int lds = ti.lds;
    {   // a K loop shorter than the ring needs fewer stages: less LDS = more resident workgroups (the K = 64 expand
        // convs of stage 1 run a single stage).  Only the all-int8 asynchronous pipeline; the register-staged
        // and 4-bit paths keep the full ring.
        const int stages = (a->KH * a->KW * (a->Cin >> 6) + (dual ? (a->Cin2 >> 6) : 0)) / ti.ksub;
        const bool all88 = a->in_bits == 8 && a->w_bits == 8 && (!dual || (a->in2_bits == 8 && a->w2_bits == 8));
        const int stage_bytes = ti.lds / ti.ns;
        static const bool full_ring = getenv("HAWQ_FULL_RING") != nullptr;  // A/B switch for measurements
        if (!full_ring && all88 && fast && needs_tables && !wide_res && !signed_res && stages >= 1 && stages < ti.ns &&
            stages * stage_bytes >= ti.BM * ti.BN * (ti.kg > 1 ? 4 : 1))  // (the int8 output tile - split-K: the partial sums - is staged on top of the ring)
            lds = stages * stage_bytes;
    }
    p.ring_bytes = lds;
    if (dual) {
        const int v1 = variant(p.in_bits, p.w_bits), v2 = variant(p.in2_bits, p.w2_bits);
        fn = (v1 == 0 || v2 == 0) ? ti.dual[0] : ti.dual[v1 == v2 ? v1 : (v1 == 1 ? 3 : 4)];
        if (p.k0 == 2 && v1 != 0 && v2 != 0) {
            HAWQ_REQUIRE(v1 == v2, "hawq_conv2d: exact-tie mode (fast_tables bit 2) needs equal operand widths in both branches");
            fn = ti.tie_dual[v1 - 1];
        }
        if (v1 != 0 && v2 != 0) lds += ti.BM * ti.BN * 2 + ti.BN * 32;
    } else {
        const int v = variant(p.in_bits, p.w_bits);
        fn = ti.single[slot][v];
        if (p.k0 == 2 && v != 0 && needs_tables) fn = ti.tie_single[slot - 1][v - 1];
        if (v != 0 && slot == 2) lds += ti.BM * ti.BN * 2;
        if (v != 0 && needs_tables) lds += ti.BN * 16;
    }
    static const bool attrs_ok = raise_lds_limits();  // once per process; never inside a capture
    HAWQ_REQUIRE(attrs_ok, "hawq_conv2d: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
    hipLaunchKernelGGL(fn, dim3(grid), dim3(ti.nt), lds, (hipStream_t)stream, p);
    HAWQ_CHECK_HIP(hipGetLastError());
    if (HAWQ_DBG_BIT(p.dbg, 128) && p.dbgbuf) {  // experiment hook: per-phase cycles of one wave (synchronises!)
        long long hbuf[4];
        (void)hipStreamSynchronize((hipStream_t)stream);
        (void)hipMemcpy(hbuf, p.dbgbuf, sizeof(hbuf), hipMemcpyDeviceToHost);
        fprintf(stderr, "[conv tile %d (%dx%d, %d thr) M=%d K=%dx%dx%d(+%d) Cout=%d epi=%d] grid %d lds %d: prologue %lld | K loop %lld | epilogue %lld cycles\n",
                tile + 1, ti.BM, ti.BN, ti.nt, p.M, a->KH, a->KW, a->Cin, dual ? a->Cin2 : 0, p.Cout, a->epilogue, grid, lds,
                hbuf[0], hbuf[1], hbuf[2]);
    }
    return 0;
}
