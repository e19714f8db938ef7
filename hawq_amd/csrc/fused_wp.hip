// "Wave-private" expand (+ reduce) kernel for gfx950 (MI355X): the 1x1 expand conv of a bottleneck unit with its residual
// epilogue and - optionally - the 1x1 reduce conv of the next unit, like fused_er.hip, but organised so that a wave never
// waits for another wave's arithmetic.
//
// Reference graph (q_resnet.py:231-260): conv3(i) -> x + identity -> quant_act_int32 -> ReLU |unit i+1| quant_act -> conv1
// -> ReLU -> quant_act1; rounding points as in SURVEY.md App. A (quant_utils.py:390-456): both residual branches are
// requantised separately, summed un-clamped, ReLU; q is the block-input QuantAct of that sum; the reduce conv accumulates
// exact int32 over all C3 channels before its own requant.
//
// Why another kernel (profiles/r02_fused_er_steps.md, r03_mallprobe.md): in fused_er.hip a workgroup walks the C3 output
// channels in slices of 64 with two workgroup barriers per slice, q and the residual go through LDS staging tiles, and the
// phases of a slice (LDS-DMA issue, GEMM1, 277 VALU instructions of epilogue per lane, GEMM2, stores) run in lock step:
// 6.5 k cycles per slice for 1.0 k of matrix pipe and 1.8 k of VALU.  What a launch costs the forward is the CU-time it
// holds, so the phases have to overlap INSIDE the CU.  Here:
//   * a wave owns 32 pixels and ALL 64 channels of the slice (two 32x32 MFMA tiles): with tile t <-> channels 32t + 16h + r
//     (h = lane half, r = accumulator register) the 16 requantised outputs of a lane per tile are exactly the 16 K-bytes that
//     lane must supply as B operand of GEMM2's K-step t - q goes from the epilogue's registers straight into the MFMA,
//     no LDS, no barrier between epilogue and GEMM2;
//   * the resident input pixels are MFMA B fragments in registers (loaded once); the residual slice goes through a
//     WAVE-PRIVATE 4 KiB staging tile [32 px][128 B] (LDS-DMA in, in-place update, whole 128-byte rows out), so every global
//     access is a full line (lane-per-pixel accesses straight from registers - 32 lines per instruction - were 25-50 % slower:
//     the L1 tag rate, not the bytes, bounds them) and no other wave ever touches the tile: no barrier, only counted waits;
//   * what waves share is a two-stage LDS ring of {W3 slice, W1 slice} and the ctab slice, filled by LDS-DMA that every wave
//     issues a share of; ONE workgroup barrier per slice (ring hand-over).  Between barriers a wave runs GEMM1 -> epilogue ->
//     GEMM2 at its own pace, so the waves of a SIMD drift apart and one wave's epilogue VALU runs beside another's MFMAs.
// int8 x int8, fast-contract tables (exact-tie mode / all-k-zero mode as separate instantiations), uint16 residuals.
#include <stdio.h>
#include <utility>

#include "common.h"

namespace {

struct WPP {
    const uint8_t *x2, *w3, *w1;
    const int32_t *ctab3, *ctab1;
    const uint16_t *res_in;
    uint16_t *res_out;   // may be null (the next unit is a resize unit: it reads only q)
    uint8_t *y;          // REDUCE: the reduce conv's output [M][C] int8 (or channel-group planes)
    uint8_t *q_out;      // !REDUCE: block input of the next unit [M][C3] int8 - or (q_nib, round 6) hawq4 [M][C3 / 2] -, or null
    int q_nib;
    int M, C3;
    int m_id_s, e_id_s, mq, eq, q_hi;
    int y_lo, y_hi, y_planar, y_nib;   // y_nib: the reduce conv's output is stored hawq4 (two 4-bit channels per byte)
    int spb;             // slices per workgroup along gridDim.y (REDUCE: all of them)
    int32_t *flags;
    long long *dbgbuf;   // probe builds (HAWQ_ABLATE, HAWQ_DBG=128): per-phase cycle sums of wave 0 of workgroup 8
};

template <int... Is, class Fn>
__device__ __forceinline__ void sfor_impl(std::integer_sequence<int, Is...>, Fn &&f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class Fn>
__device__ __forceinline__ void sfor(Fn &&f) {
    sfor_impl(std::make_integer_sequence<int, N>{}, f);
}

__device__ __forceinline__ void wp_dma16(const char *src, char *dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
}
__device__ __forceinline__ void wp_dma4(const char *src, char *dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)dst, 4, 0, 0);
}
// ctab entry {m, s | k << 8, lo32(add), hi32(add)}; K0: the host guarantees k == 0 for every channel of this table
template <bool K0>
__device__ __forceinline__ DyNt wp_ctab(const char *ctab, int ch) {
    const v4i t = *reinterpret_cast<const v4i *>(ctab + ch * 16);
    DyNt d;
    d.m = t.x, d.s = K0 ? t.y : (t.y & 31), d.k = K0 ? 0 : (t.y >> 8);
    d.add = (long long)(((unsigned long long)(unsigned)t.w << 32) | (unsigned)t.z);
    return d;
}

// C: channels of the expand conv's input (= of the reduce conv's output).  NW: waves per workgroup = 32-pixel tiles.
// REDUCE: the next unit's reduce conv runs in the same launch.  MINW: waves per SIMD the register allocation must allow.
template <int C_, int NW_, bool REDUCE_, int MINW_>
struct WPCfg {
    static constexpr int C = C_, NW = NW_, MINW = MINW_;
    static constexpr bool REDUCE = REDUCE_;
    static constexpr int NT = 64 * NW, BM = 32 * NW;
    static constexpr int KC = C / 64;            // 64-byte chunks of GEMM1's K
    static constexpr int CT2 = C / 32;           // 32-channel MFMA tiles of GEMM2's output
    static constexpr int RPP = NT / 4;           // operand rows per LDS-DMA pass (4 lanes x 16 B per 64-byte row)
    static constexpr int W3_BYTES = 64 * C;      // [KC][64 rows][64 B]
    static constexpr int W1_BYTES = REDUCE ? 64 * C : 0;   // [C rows][64 B]
    static constexpr int STAGE = W3_BYTES + W1_BYTES;
    static constexpr int W3PASS = (64 * KC + RPP - 1) / RPP, W1PASS = REDUCE ? (C + RPP - 1) / RPP : 0;
    static constexpr int CTPASS = NW >= 4 ? 1 : 4 / NW;
    static constexpr int NDMA = W3PASS + W1PASS + CTPASS;   // LDS-DMA instructions per wave per slice
    static constexpr int OFF_CT = 2 * STAGE;
    static constexpr int OFF_STG = OFF_CT + 2 * 1024;        // [NW] wave-private staging tiles [32 px][128 B]
    static constexpr int LDS_BYTES = OFF_STG + NW * 4096;
    static_assert(RPP % 16 == 0 && (NW == 1 || NW == 2 || NW % 4 == 0), "producer passes must cover whole swizzle groups");
};

template <class F, bool TIE, bool CK0, bool QK0 = false>
__global__ __launch_bounds__(F::NT, F::MINW) void expand_wp_kernel(const WPP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MODE_S = TIE ? 2 : 0;                  // scalar identity table: uniform pre-shift
    constexpr int MODE_Q = TIE ? 2 : (QK0 ? 1 : 0);      // scalar table of the next QuantAct (QK0: no pre-shift)
    constexpr int MODE_C = TIE ? 2 : (CK0 ? 1 : 0);      // per-channel tables
    constexpr bool K0 = CK0 && !TIE;
    constexpr int C = F::C;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, h = lane >> 5;
    const int pix = blockIdx.x * F::BM + wave * 32 + l31;
    const bool valid = pix < p.M;
    const size_t row = valid ? pix : p.M - 1;            // rows beyond M read a valid row and are never stored
    const int nsl = p.C3 >> 6;
    const int j0 = blockIdx.y * p.spb, j1 = (j0 + p.spb < nsl) ? j0 + p.spb : nsl;
    char *ct = smem + F::OFF_CT;
    // ---------------------------------------------------------------- LDS-DMA: every wave issues its 16 rows of each pass
    const int prow = t >> 2, pslot = t & 3;
    const int sw = (pslot ^ ((prow >> 2) & 3)) << 4;     // source-side swizzle (same for rows prow + k * RPP)
    auto issue = [&](int j, int stage) {   // exactly NDMA instructions per wave (the counted waits rely on it): a pass wider than
        char *dst = smem + stage * F::STAGE;   // its operand wraps around and rewrites identical bytes
#pragma unroll
        for (int i = 0; i < F::W3PASS; ++i) {            // rows j*64 .. j*64+63 of W3 [C3][C] as KC chunks of [64 rows][64 B]
            const int idx = (i * F::RPP + prow) & (64 * F::KC - 1), first = (i * F::RPP + wave * 16) & (64 * F::KC - 1);
            wp_dma16((const char *)p.w3 + (size_t)(j * 64 + (idx & 63)) * C + (idx >> 6) * 64 + sw, dst + first * 64);
        }
#pragma unroll
        for (int i = 0; i < F::W1PASS; ++i) {            // columns j*64 .. j*64+63 of W1 [C][C3] as [C rows][64 B]
            const int r = (i * F::RPP + prow) & (C - 1), first = (i * F::RPP + wave * 16) & (C - 1);
            wp_dma16((const char *)p.w1 + (size_t)r * p.C3 + j * 64 + sw, dst + F::W3_BYTES + first * 64);
        }
#pragma unroll
        for (int i = 0; i < F::CTPASS; ++i) {            // ctab3 slice: 4 x 256 B
            const int q4 = (wave + i * F::NW) & 3;
            wp_dma4((const char *)p.ctab3 + (size_t)j * 1024 + q4 * 256 + lane * 4, ct + (j & 1) * 1024 + q4 * 256);
        }
    };
    // ---------------------------------------------------------------- wave-private staging tile [32 px][8 slots of 16 B]
    // physical slot = logical slot ^ (row & 7).  Coalesced side (LDS-DMA in, row stores out): instruction k covers rows
    // 8k .. 8k+7, lane -> (row 8k + lane / 8, physical slot lane % 8).  Lane-per-pixel side: row l31, logical slots 4t + 2h + {0, 1}
    char *stg = smem + F::OFF_STG + wave * 4096;
    const int pix0 = blockIdx.x * F::BM + wave * 32;
    const int crow = lane >> 3, cslot = lane & 7;
    // byte offset of (row 8k + crow, physical slot cslot) in a [M][C3] uint16 tensor, slice 0: shared by the residual DMA and the
    // row stores; 32-bit (the launcher refuses tensors of 4 GiB and more), rows beyond M clamp to the last row (never stored)
    unsigned coff[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = 8 * k + crow;
        coff[k] = (unsigned)((pix0 + r < p.M) ? pix0 + r : p.M - 1) * (unsigned)(p.C3 * 2) + ((cslot ^ (r & 7)) << 4);
    }
    auto dma_res = [&](int j) {
#pragma unroll
        for (int k = 0; k < 4; ++k) wp_dma16((const char *)p.res_in + (coff[k] + (unsigned)(j * 128)), stg + k * 1024);
    };
    const unsigned sl = lds_addr(stg) + l31 * 128;   // this lane's row; slot s at ((s ^ (l31 & 7)) << 4)
    const int x7 = l31 & 7;
    // ---------------------------------------------------------------- prologue: resident B fragments through the staging tile
    v4i xf[F::KC][2];                                    // K bytes [32 ks + 16 h, +16) of chunk kc of this lane's pixel
    {
        const int r2 = lane >> 2, s2 = lane & 3;         // [32 px][64 B] chunks: instruction k covers rows 16k .. 16k+15
#pragma unroll
        for (int kc0 = 0; kc0 < F::KC; kc0 += 2) {
#pragma unroll
            for (int kc = kc0; kc < kc0 + 2 && kc < F::KC; ++kc)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int r = 16 * k + r2;
                    const size_t gr = (pix0 + r < p.M) ? pix0 + r : p.M - 1;
                    wp_dma16((const char *)p.x2 + gr * C + kc * 64 + ((s2 ^ ((r >> 2) & 3)) << 4), stg + (kc - kc0) * 2048 + k * 1024);
                }
            wait_vmcnt<0>();
#pragma unroll
            for (int kc = kc0; kc < kc0 + 2 && kc < F::KC; ++kc)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    xf[kc][ks] = *reinterpret_cast<const v4i *>(stg + (kc - kc0) * 2048 + lds_off(l31, 2 * ks + h));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the next round / the residual DMA overwrites the tile
#pragma unroll
            for (int kc = kc0; kc < kc0 + 2 && kc < F::KC; ++kc) pin(xf[kc][0]), pin(xf[kc][1]);
        }
    }
    issue(j0, 0);
    dma_res(j0);
    v16i acc2[F::REDUCE ? F::CT2 : 1];
    if constexpr (F::REDUCE) {
#pragma unroll
        for (int c = 0; c < F::CT2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[c][r] = 0;
    }
    DyNt dids = dynt_prepare(p.m_id_s, p.e_id_s), dq = dynt_prepare(p.mq, p.eq);
    constexpr bool SC0 = QK0 && !TIE;   // round 6: the QK0 instantiations also take the identity table in its shift-free form (launcher: ids0_form)
    DyS0 dis0 = dys0_prepare(p.m_id_s, p.e_id_s);
    asm volatile("" : "+s"(dids.add), "+s"(dq.add), "+s"(dis0.add));   // opaque rounding constants: one v_mad_i64_i32 instead of v_mul_hi_i32 + v_add (see fused_er.hip)
    const unsigned rowmask = valid ? 0xffffffffu : 0u;
    const int qhi2 = (p.q_hi & 0xffff) | (p.q_hi << 16);
    unsigned oor = 0;
    const int cp = cperm(l31);
    const unsigned a0 = lds_addr(smem) + lds_off(cp, h), a1 = lds_addr(smem) + lds_off(cp, 2 + h);   // K-steps 0 / 1 of a 64-byte row
#ifdef HAWQ_ABLATE
    const bool prof = p.dbgbuf != nullptr;
    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = prof ? (long long)__builtin_readcyclecounter() : 0;
#define WP_STAMP(K)                                                    \
    if (prof) {                                                        \
        const long long now = (long long)__builtin_readcyclecounter(); \
        ph[K] += now - tprev;                                          \
        tprev = now;                                                   \
    }
#else
#define WP_STAMP(K)
#endif
    for (int j = j0; j < j1; ++j) {
        const int st = (j - j0) & 1;
        // In flight from this wave: {ring DMA(j)} < {stores(j-1)} < {residual DMA(j): 4} (issue order).  Loads return in order, so
        // with <= 4 operations left the ring group has landed (stores may retire at any time; they only make the wait stricter)
        wait_vmcnt<4>();
        __builtin_amdgcn_s_barrier();   // every wave's share of stage st has landed; every wave is done with slice j-1
        WP_STAMP(0)
        const bool more = j + 1 < j1;
        if (more) issue(j + 1, st ^ 1);
        WP_STAMP(1)
        // ------------------------------------------------------------ GEMM1: acc1[t] = W3[slice rows 32t..32t+31] . x2
        v16i acc1[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[0][r] = 0, acc1[1][r] = 0;
        {
            const unsigned b0 = a0 + st * F::STAGE, b1 = a1 + st * F::STAGE;
            v4i wf[2][2];   // {tile 0, tile 1} of one K-step, double-buffered
            wf[0][0] = lds_read16<0>(b0), wf[0][1] = lds_read16<2048>(b0);
            sfor<2 * F::KC>([&](auto SI) {
                constexpr int s = decltype(SI)::value, kc = s >> 1, ks = s & 1, cur = s & 1, nxt = cur ^ 1;
                if constexpr (s + 1 < 2 * F::KC) {
                    constexpr int kn = (s + 1) >> 1;
                    wf[nxt][0] = lds_read16<kn * 4096>(ks ? b0 : b1), wf[nxt][1] = lds_read16<kn * 4096 + 2048>(ks ? b0 : b1);
                    wait_lgkm<2>();
                } else {
                    wait_lgkm<0>();
                }
                pin(wf[cur][0]), pin(wf[cur][1]);
                acc1[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[cur][0], xf[kc][ks], acc1[0], 0, 0, 0);
                acc1[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[cur][1], xf[kc][ks], acc1[1], 0, 0, 0);
            });
        }
        WP_STAMP(2)
        // ------------------------------------------------------------ epilogue 1: residual add, ReLU, next QuantAct
        // younger than the residual DMA of this slice: only the ring DMA(j+1)
        if (more) wait_vmcnt<F::NDMA>(); else wait_vmcnt<0>();
        v4i rin[4];
        rin[0] = lds_read16<0>(sl + ((4 * 0 + 2 * h) ^ x7) * 16), rin[1] = lds_read16<0>(sl + ((4 * 0 + 2 * h + 1) ^ x7) * 16);
        rin[2] = lds_read16<0>(sl + ((4 * 1 + 2 * h) ^ x7) * 16), rin[3] = lds_read16<0>(sl + ((4 * 1 + 2 * h + 1) ^ x7) * 16);
        wait_lgkm<0>();
        pin(rin[0]), pin(rin[1]), pin(rin[2]), pin(rin[3]);
        WP_STAMP(3)
        v4i qf[2];
        v4i rout[4];
        {
            const char *ctb = ct + (j & 1) * 1024;
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                int qp[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned w0 = (unsigned)rin[2 * tl + (g >> 1)][(g & 1) * 2], w1 = (unsigned)rin[2 * tl + (g >> 1)][(g & 1) * 2 + 1];
                    const int idin[4] = {(int)(w0 & 0xffffu), (int)(w0 >> 16), (int)(w1 & 0xffffu), (int)(w1 >> 16)};
                    int o[4], qv[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const DyNt dm = wp_ctab<K0>(ctb, 32 * tl + 16 * h + 4 * g + k);
                        const int a = dyadic_mode<MODE_C>(acc1[tl][4 * g + k], dm);
                        const int b = SC0 ? dyadic_s0(idin[k], dis0) : dyadic_mode<MODE_S>(idin[k], dids);
                        o[k] = max(a + b, 0);                                  // no clamp: quant_utils.py:456
                        qv[k] = dyadic_mode<MODE_Q>(o[k], dq);                 // o >= 0, m >= 0: q >= 0; clamped from above in the pack
                    }
                    oor |= ((unsigned)(o[0] | o[1]) | (unsigned)(o[2] | o[3])) & rowmask;
                    rout[2 * tl + (g >> 1)][(g & 1) * 2] = pack2_u16_sat(o[0], o[1]);
                    rout[2 * tl + (g >> 1)][(g & 1) * 2 + 1] = pack2_u16_sat(o[2], o[3]);
                    qp[g] = pack4_min(qv[0], qv[1], qv[2], qv[3], qhi2);
                }
                qf[tl] = v4i{qp[0], qp[1], qp[2], qp[3]};
            }
        }
        WP_STAMP(4)
        // new residual slice: in place into the staging tile, then whole 128-byte rows to memory
        if (p.res_out) {
            *reinterpret_cast<v4i *>(stg + l31 * 128 + ((4 * 0 + 2 * h) ^ x7) * 16) = rout[0];
            *reinterpret_cast<v4i *>(stg + l31 * 128 + ((4 * 0 + 2 * h + 1) ^ x7) * 16) = rout[1];
            *reinterpret_cast<v4i *>(stg + l31 * 128 + ((4 * 1 + 2 * h) ^ x7) * 16) = rout[2];
            *reinterpret_cast<v4i *>(stg + l31 * 128 + ((4 * 1 + 2 * h + 1) ^ x7) * 16) = rout[3];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = 8 * k + crow;
                const v4i v = *reinterpret_cast<const v4i *>(stg + k * 1024 + lane * 16);
                if (pix0 + r < p.M) *reinterpret_cast<v4i *>((char *)p.res_out + (coff[k] + (unsigned)(j * 128))) = v;
            }
        }
        if constexpr (!F::REDUCE) {   // the next unit's block input [32 px][64 B] the same way (slots 2t + h, lds_off swizzle)
            if (p.q_out && p.q_nib) {
                // hawq4 block input (the next unit is a 4-bit layer whose input is stored as nibbles): a lane's 16 channels of tile tl are 8
                // bytes - dword = (c0..c3 bytes) | (c4..c7 bytes) << 4, include/hawq_mi355.h - at byte 16 tl + 8 h of the pixel's 32-byte slice
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the row reads above are done with the tile
#pragma unroll
                for (int tl = 0; tl < 2; ++tl)
                    *reinterpret_cast<v2i *>(stg + l31 * 32 + 16 * tl + 8 * h) = v2i{qf[tl].x | (qf[tl].y << 4), qf[tl].z | (qf[tl].w << 4)};
                const int r = lane >> 1, s2 = lane & 1;
                const v4i v = *reinterpret_cast<const v4i *>(stg + lane * 16);   // (wave-private tile: LDS serves a wave's DS instructions in order)
                if (pix0 + r < p.M)
                    *reinterpret_cast<v4i *>((char *)p.q_out + (size_t)(pix0 + r) * (p.C3 >> 1) + j * 32 + s2 * 16) = v;
            } else if (p.q_out) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the row reads above are done with the tile
                *reinterpret_cast<v4i *>(stg + lds_off(l31, h)) = qf[0];
                *reinterpret_cast<v4i *>(stg + lds_off(l31, 2 + h)) = qf[1];
                const int r2 = lane >> 2, s2 = lane & 3;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int r = 16 * k + r2;
                    const v4i v = *reinterpret_cast<const v4i *>(stg + k * 1024 + lane * 16);
                    if (pix0 + r < p.M)
                        *reinterpret_cast<v4i *>((char *)p.q_out + (size_t)(pix0 + r) * p.C3 + j * 64 + ((s2 ^ ((r >> 2) & 3)) << 4)) = v;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every read of the tile has returned: the DMA may overwrite it
        if (more) dma_res(j + 1);
        WP_STAMP(5)
        // ------------------------------------------------------------ GEMM2: acc2[c] += W1[rows 32c..32c+31][slice] . q
        if constexpr (F::REDUCE) {
            const unsigned b0 = a0 + st * F::STAGE + F::W3_BYTES, b1 = a1 + st * F::STAGE + F::W3_BYTES;
            v4i wf[2][2];   // {K-step 0, K-step 1} of one channel tile, double-buffered
            wf[0][0] = lds_read16<0>(b0), wf[0][1] = lds_read16<0>(b1);
            sfor<F::CT2>([&](auto CI) {
                constexpr int c = decltype(CI)::value, cur = c & 1, nxt = cur ^ 1;
                if constexpr (c + 1 < F::CT2) {
                    wf[nxt][0] = lds_read16<(c + 1) * 2048>(b0), wf[nxt][1] = lds_read16<(c + 1) * 2048>(b1);
                    wait_lgkm<2>();
                } else {
                    wait_lgkm<0>();
                }
                pin(wf[cur][0]), pin(wf[cur][1]);
                acc2[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[cur][0], qf[0], acc2[c], 0, 0, 0);
                acc2[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[cur][1], qf[1], acc2[c], 0, 0, 0);
            });
        }
        WP_STAMP(6)
    }
#ifdef HAWQ_ABLATE
    if (prof && blockIdx.x == 8 && blockIdx.y == 0 && t == 0)
        for (int k = 0; k < 8; ++k) p.dbgbuf[k] = ph[k];
#endif
#undef WP_STAMP
    if ((oor >> 16) != 0) atomicOr(p.flags, 1);
    // ---------------------------------------------------------------- epilogue 2: the reduce conv's QuantAct
    if constexpr (F::REDUCE) {
        wait_vmcnt<0>();
#pragma unroll
        for (int c = 0; c < F::CT2; ++c) {
            const int ch0 = 32 * c + 16 * h;
            int w[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int qv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    qv[k] = med3i(dyadic_mode<MODE_C>(acc2[c][4 * g + k], wp_ctab<K0>((const char *)p.ctab1, ch0 + 4 * g + k)), p.y_lo, p.y_hi);
                w[g] = pack4_fast(qv[0], qv[1], qv[2], qv[3]);
            }
            const v4i ww = {w[0], w[1], w[2], w[3]};
            if (valid && p.y_nib) {
                // hawq4: byte k of an 8-channel group = c_k | c_{k+4} << 4; the int8 dwords hold 4 channels each, values 0..15
                const v2i nn = {w[0] | (w[1] << 4), w[2] | (w[3] << 4)};
                if (p.y_planar)   // planes [C / 32][M][16 B]
                    *reinterpret_cast<v2i *>((char *)p.y + ((size_t)(ch0 >> 5) * p.M + row) * 16 + ((ch0 >> 4) & 1) * 8) = nn;
                else
                    *reinterpret_cast<v2i *>((char *)p.y + row * (C / 2) + (ch0 >> 1)) = nn;
            } else if (valid) {
                if (p.y_planar)   // channel-group planes [C / 16][M][16 B] (hawq_conv_args.out_planar)
                    *reinterpret_cast<v4i *>((char *)p.y + ((size_t)(ch0 >> 4) * p.M + row) * 16) = ww;
                else
                    *reinterpret_cast<v4i *>((char *)p.y + row * C + ch0) = ww;
            }
        }
    }
}

// ---- variant table ------------------------------------------------------------------------------------------------
typedef void (*WPFn)(const WPP);
struct WPInfo { WPFn fn[5]; int c, bm, nt, lds; bool reduce; int ysplit; };   // fn: {general, per-channel k zero, exact-tie, ck0 + next-QuantAct k zero, only the latter}
#define WP_ENTRY(F, YS) {{expand_wp_kernel<F, false, false>, expand_wp_kernel<F, false, true>, expand_wp_kernel<F, true, false>, expand_wp_kernel<F, false, true, true>, \
                          expand_wp_kernel<F, false, false, true>}, F::C, F::BM, F::NT, F::LDS_BYTES, F::REDUCE, YS}
using W64 = WPCfg<64, 4, true, 3>;       // stage 1: 128 pixels, 18 KiB of LDS
using W64B = WPCfg<64, 8, true, 3>;      //          256 pixels: half the weight traffic per pixel
using W128 = WPCfg<128, 4, true, 2>;     // stage 2: 34 KiB
using W128B = WPCfg<128, 8, true, 2>;
using W256 = WPCfg<256, 4, true, 1>;     // stage 3: 82 KiB; one wave per SIMD (> 256 registers): the 2-wave allocation spilled, and a spill
                                         // reload is a vmcnt(0) wait in the middle of the LDS-DMA issue sequence
// (a 256-pixel, two-waves-per-SIMD form of the pair, WPCfg<256, 8, true, 2>, reloaded three spilled registers inside its slice loop - a
// scratch_load with its s_waitcnt vmcnt(0) in the middle of the LDS-DMA issue sequence - and was the slowest variant of every pair (74-81 us
// against 26-43): removed in round 4, tests/test_host_logic.py::test_shipped_kernels_do_not_spill keeps such a kernel from shipping again)
using N64 = WPCfg<64, 4, false, 4>;      // expand conv alone (last unit of a stage): 10 KiB
using N128 = WPCfg<128, 4, false, 3>;
using N256 = WPCfg<256, 4, false, 3>;
using N512 = WPCfg<512, 4, false, 2>;    // stage 4: 66 KiB; gridDim.y splits the C3 / 64 slices
const WPInfo kWP[] = {WP_ENTRY(W64, 1), WP_ENTRY(W64B, 1), WP_ENTRY(W128, 1), WP_ENTRY(W128B, 1), WP_ENTRY(W256, 1),
                      WP_ENTRY(N64, 1), WP_ENTRY(N128, 1), WP_ENTRY(N256, 1), WP_ENTRY(N256, 2), WP_ENTRY(N256, 4),
                      WP_ENTRY(N512, 1), WP_ENTRY(N512, 2), WP_ENTRY(N512, 4), WP_ENTRY(N512, 8)};
constexpr int NUM_WP = sizeof(kWP) / sizeof(kWP[0]);

bool wp_expand_ok(const hawq_conv_args &e) {
    return (e.in_pitch == 0 || e.in_pitch == e.Cin) && (e.out_pitch == 0 || e.out_pitch == e.Cout) && e.KH == 1 && e.KW == 1 && e.stride == 1 && e.pad == 0 && e.in_bits == 8 && e.w_bits == 8 && e.fast_tables != 0 && !e.in2 &&
           !e.in_planar && e.epilogue == HAWQ_EPI_RESIDUAL && e.res_in && e.res_in_bits == 16 && (!e.res_out || e.res_out_bits == 16) &&
           e.flags && e.ctab && e.Cout % 64 == 0 && (e.out_bits == 8 || (e.out_bits == 4 && e.q_lo >= 0 && e.q_hi <= 15));   // (4: the expand conv alone writing a hawq4 block input)
}

// index into kWP of the nth (1-based) variant that takes this launch, or -1
int wp_variant(const hawq_expand_reduce_args *a, int nth) {
    const hawq_conv_args &e = a->expand, &r = a->reduce;
    if (!wp_expand_ok(e)) return -1;
    const bool reduce = r.wgt != nullptr;
    if (reduce && e.out_bits != 8) return -1;   // (a fused pair keeps the block input on chip: its storage width is moot, the caller says 8)
    if (reduce) {
        if (!(r.KH == 1 && r.KW == 1 && r.stride == 1 && r.pad == 0 && r.in_bits == 8 && r.w_bits == 8 && r.fast_tables != 0 && !r.in2)) return -1;
        if (r.epilogue != HAWQ_EPI_REQUANT || !r.ctab || !r.out_q || (r.out_bits != 8 && r.out_bits != 4) || !e.res_out) return -1;
        if (r.out_bits == 4 && (r.q_lo < 0 || r.q_hi > 15)) return -1;   // hawq4 stores unsigned nibbles
        if (r.Cin != e.Cout || r.Cout != e.Cin || r.N != e.N || r.H != e.H || r.W != e.W) return -1;
    }
    int n = 0;
    for (int i = 0; i < NUM_WP; ++i)
        if (kWP[i].c == e.Cin && kWP[i].reduce == reduce && (e.Cout / 64) % kWP[i].ysplit == 0 && ++n == nth) return i;
    return -1;
}

}  // namespace

// number of wave-private variants that take this launch (reduce.wgt == NULL: the expand conv alone)
int wp_num_variants(const hawq_expand_reduce_args *a) {
    int n = 0;
    while (wp_variant(a, n + 1) >= 0) ++n;
    return n;
}

int wp_launch(const hawq_expand_reduce_args *a, int nth, void *stream) {
    const int v = wp_variant(a, nth);
    HAWQ_REQUIRE(v >= 0, "hawq_conv_expand_reduce: no wave-private variant %d for this launch", nth);
    const hawq_conv_args &e = a->expand, &r = a->reduce;
    auto e_fast = [](int ek) { return (ek & 0xff) >= 33 && (ek & 0xff) <= 62; };
    HAWQ_REQUIRE(e.mq >= 0 && e_fast(e.eq) && e.m_id_scalar >= 0 && e_fast(e.e_id_scalar), "hawq_conv_expand_reduce: scalar tables outside the fast contract");
    HAWQ_REQUIRE(e.q_lo <= 0, "hawq_conv_expand_reduce: the block-input QuantAct clamp must admit 0");
    const WPInfo &wi = kWP[v];
    WPP p;
    p.x2 = (const uint8_t *)e.in, p.w3 = (const uint8_t *)e.wgt, p.w1 = (const uint8_t *)r.wgt;
    p.ctab3 = e.ctab, p.ctab1 = wi.reduce ? r.ctab : nullptr;
    p.res_in = (const uint16_t *)e.res_in, p.res_out = (uint16_t *)e.res_out;
    p.y = wi.reduce ? (uint8_t *)r.out_q : nullptr;
    p.q_out = wi.reduce ? nullptr : (uint8_t *)e.out_q;
    p.q_nib = !wi.reduce && e.out_bits == 4;
    const long long M = (long long)e.N * e.H * e.W;
    HAWQ_REQUIRE(M > 0 && M * e.Cout * 2 < (1ll << 32), "hawq_conv_expand_reduce: bad problem size");
    p.M = (int)M, p.C3 = e.Cout;
    p.m_id_s = e.m_id_scalar, p.e_id_s = e.e_id_scalar, p.mq = e.mq, p.eq = e.eq, p.q_hi = e.q_hi;
    p.y_lo = wi.reduce ? (r.relu && r.q_lo < 0 ? 0 : r.q_lo) : 0, p.y_hi = wi.reduce ? r.q_hi : 0;
    p.y_planar = wi.reduce ? r.out_planar : 0;
    p.y_nib = wi.reduce && r.out_bits == 4;
    p.spb = (e.Cout / 64) / wi.ysplit;
    p.flags = e.flags;
    p.dbgbuf = nullptr;
#ifdef HAWQ_ABLATE
    static const int dbg_env = HAWQ_DBG_ENV();
    static long long *dbg_dev = nullptr;
    if ((dbg_env & 128) && !dbg_dev) (void)hipMalloc(&dbg_dev, 8 * sizeof(long long));
    if (dbg_env & 128) p.dbgbuf = dbg_dev;
#endif
    static const bool attrs = [] {
        bool good = true;
        for (const WPInfo &k : kWP)
            for (int i = 0; i < 5; ++i)
                good &= hipFuncSetAttribute((const void *)k.fn[i], hipFuncAttributeMaxDynamicSharedMemorySize, k.lds) == hipSuccess;
        return good;
    }();
    HAWQ_REQUIRE(attrs, "hawq_conv_expand_reduce: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
    const int ft = e.fast_tables | (wi.reduce ? r.fast_tables : 0);
    // bit 2: some table is not provably tie-free; bit 3 (on BOTH convs): every per-channel pre-shift is zero
    const bool tie = (ft & 4) != 0, ck0 = (e.fast_tables & 8) && (!wi.reduce || (r.fast_tables & 8));
    const bool qk0 = (e.eq >> 8) == 0 && ids0_form(e.e_id_scalar);   // the QK0 instantiations: next-QuantAct table without pre-shift AND identity table in the shift-free form
    HAWQ_REQUIRE(e.q_hi >= 0 && e.q_hi <= 32767, "hawq_conv_expand_reduce: q_hi outside [0, 32767]");
    hipLaunchKernelGGL(wi.fn[tie ? 2 : (ck0 ? (qk0 ? 3 : 1) : (qk0 ? 4 : 0))], dim3((p.M + wi.bm - 1) / wi.bm, wi.ysplit), dim3(wi.nt), wi.lds, (hipStream_t)stream, p);
    HAWQ_CHECK_HIP(hipGetLastError());
    if (p.dbgbuf) {   // probe builds only (synchronises!)
        long long hb[8];
        (void)hipStreamSynchronize((hipStream_t)stream);
        (void)hipMemcpy(hb, p.dbgbuf, sizeof(hb), hipMemcpyDeviceToHost);
        fprintf(stderr, "[wave-private C=%d bm=%d reduce=%d ysplit=%d M=%d C3=%d] cycles of wave 0 / workgroup 8 over %d slices: wait+barrier %lld | issue ring DMA %lld | "
                        "GEMM1 %lld | wait + read residual %lld | epilogue %lld | stage + store + residual DMA %lld | GEMM2 %lld\n",
                wi.c, wi.bm, (int)wi.reduce, wi.ysplit, p.M, p.C3, p.spb, hb[0], hb[1], hb[2], hb[3], hb[4], hb[5], hb[6]);
    }
    return 0;
}
