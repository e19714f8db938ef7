// Weight-stationary, persistent 3x3 / stride 1 / pad 1 kernel for Cin == Cout == 64 (ResNet50 stage 1 conv2, the
// first 3x3 of a ResNet18/34 stage-1 block; QuantBnConv2d + ReLU + QuantAct, quant_modules.py:527-545, 233-260).
//
// Why a second 3x3 kernel.  For these layers K = 576 only: a 256-pixel tile is 2304 matrix-pipe cycles, while the
// band kernel of conv_igemm.hip spends 4.9 k cycles per tile waiting for its operands (36 KiB of weights - the SAME
// 36 KiB for every tile - plus a 32 KiB band whose plane layout makes every LDS-DMA piece a gather of 64 sixteen-byte
// segments, the one slow shape of tools/ubench/ingest_shape.hip) and 4.1 k in the epilogue, with two workgroups per
// CU to overlap them (profiles/r02_band_persist.md).  Here
//   - the weights are loaded ONCE per workgroup and stay in LDS (36 KiB) while it walks its tiles;
//   - the band is a dense, source-side swizzled [band pixel][64 B] tile: an LDS-DMA piece is 16 pixels x 64 contiguous
//     bytes (full-rate shape; from channel-group planes: four runs of 256 contiguous bytes); the tap addresses cost 6 VALU per (tap, pixel tile), hidden between the MFMAs;
//   - 4 producer waves refill the band for the next tile while the 4 MFMA waves run the epilogue of the current one;
//   - the epilogue needs no workgroup barrier: each wave transposes its own 64 px x 64 ch through a private 2 KiB
//     staging area and writes whole 64-byte pixel rows (1 KiB contiguous per store instruction).
// 77 KiB of LDS, 8 waves, <= 128 VGPRs: two workgroups per CU, whose MFMA and epilogue phases interleave.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

struct BPP {
    const uint8_t *in, *wgt;
    const int32_t *ctab;
    uint8_t *out;              // int8: NHWC [M][64] or (round 5) planes [4][M][16 B]; null = none (RESIDUAL)
    const uint16_t *res_in;    // RESIDUAL (round 5): [M][64] uint16
    uint16_t *res_out;         // [M][64] uint16 or null
    int *flags;
    int out_planar, mq, eq, m_id, e_id;
    int M, rows_total, Ho, Wo;
    int in_planar;   // activations as channel-group planes [4][M][16 B] (hawq_conv_args.in_planar)
    float rcp_wo, rcp_ho;
    int q_lo, q_hi;
    int ntiles;
    int ck0;         // every per-channel pre-shift of ctab is zero (hawq_conv_args.fast_tables bit 3): the shorter requant
    long long *dbgbuf;
};

__device__ __attribute__((aligned(16))) const int g_bp_zero16[4] = {0, 0, 0, 0};

constexpr int BM = 256, BAND_PX = 512, ZP = BAND_PX - 4;   // band pixels ZP.. are zeros (taps in another image / the padding rows)
constexpr int NCW = 4, NPW = 4, NT = (NCW + NPW) * 64;
constexpr int OFF_W = 0, W_BYTES = 9 * 4096;               // [tap][64 rows][64 B], rows swizzled like every operand tile
constexpr int OFF_BAND = OFF_W + W_BYTES;                  // [BAND_PX][64 B]
constexpr int OFF_STAGE = OFF_BAND + BAND_PX * 64;         // [NCW][32 px][64 B]
constexpr int OFF_CTAB = OFF_STAGE + NCW * 2048;           // [64][16 B]
constexpr int LDS_BYTES = OFF_CTAB + 1024;
constexpr int BPIECES = BAND_PX / 16 / NPW;                // band pieces (16 pixels = 1 KiB) per producer wave per tile
static_assert((OFF_BAND & 63) == 0, "band addresses are XOR-ed below bit 6");

__device__ __forceinline__ void dma16(const char *src, char *dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
}
// floor(m / d) for 0 <= m < 2^23 with r = 1.0f / d: the float estimate is off by at most one
__device__ __forceinline__ int fdiv(int m, int d, float r) {
    int g = (int)((float)m * r);
    g -= (g * d > m);
    g += ((g + 1) * d <= m);
    return g;
}

// RES (round 5): RESIDUAL epilogue - the second 3x3 conv of a ResNet18/34 stage-1 basic block (q_resnet.py:300-316): uint16 residual in and
// out (quant_utils.py:415-456), the next block's QuantAct as int8 output.
template <bool TIE, bool RES = false>
__global__ __launch_bounds__(NT, RES ? 3 : 4) void band_persist_kernel(const BPP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MODE = TIE ? 2 : 0;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int G = gridDim.x;
    int wg = blockIdx.x;
    {   // contiguous tile ranges per XCD (neighbouring tiles share their halo rows in that XCD's L2)
        const int q = G >> 3, r = G & 7, xcd = wg & 7, idx = wg >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int Wo = p.Wo, Wb = Wo + 2;
    const char *zero = reinterpret_cast<const char *>(g_bp_zero16);
    const bool prof = HAWQ_DBG_BIT(~0, 128) && p.dbgbuf != nullptr;
    long long ph[5] = {0, 0, 0, 0, 0};
    long long tprev = prof ? (long long)__builtin_readcyclecounter() : 0;
#define BP_STAMP(K)                                                    \
    if (prof) {                                                        \
        const long long now = (long long)__builtin_readcyclecounter(); \
        ph[K] += now - tprev;                                          \
        tprev = now;                                                   \
    }
    if (wave >= NCW) {
        // ------------------------------------------------------------------ producer waves: every LDS-DMA of the kernel
        const int dw = wave - NCW;
        int boff[BPIECES], brow[BPIECES];   // per piece: byte offset of this lane's segment relative to global row G0; band row (or -2^20: never valid)
#pragma unroll
        for (int i = 0; i < BPIECES; ++i) {
            const int bpx = (i * NPW + dw) * 16 + (lane >> 2);
            const int br = bpx / Wb, bc = bpx - br * Wb, x = bc - 1;
            const bool v = (unsigned)x < (unsigned)Wo && bpx < ZP;
            const int slot = (lane & 3) ^ ((bpx >> 2) & 3);   // the 16 channels this lane fetches (source-side swizzle)
            // NHWC: 4 lanes cover the 64 bytes of a pixel; planes: the lanes of one slot read consecutive 16-byte units of plane `slot`
            boff[i] = p.in_planar ? (slot * p.M + br * Wo + x) * 16 : (br * Wo + x) * 64 + (slot << 4);
            brow[i] = v ? br : -(1 << 20);
        }
        const int px_bytes = p.in_planar ? 16 : 64;
        auto issue_band = [&](int tile) {
            const int G0 = (tile * BM) / Wo - 1;
            const char *base = (const char *)p.in + (long long)G0 * Wo * px_bytes;
#pragma unroll
            for (int i = 0; i < BPIECES; ++i) {
                const bool v = (unsigned)(G0 + brow[i]) < (unsigned)p.rows_total;
                dma16(v ? base + boff[i] : zero, smem + OFF_BAND + (i * NPW + dw) * 1024);
            }
        };
        {   // weights: 36 pieces of 16 rows x 64 B (row = output channel, 9 taps), ctab, first band
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const int row = dw * 16 + (lane >> 2);
                dma16((const char *)p.wgt + (size_t)row * 576 + i * 64 + (((lane & 3) ^ ((row >> 2) & 3)) << 4), smem + OFF_W + i * 4096 + dw * 1024);
            }
            if (dw == 0) dma16((const char *)p.ctab + lane * 16, smem + OFF_CTAB);
            if (wg < p.ntiles) issue_band(wg);
            wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();   // A(first tile)
        for (int tile = wg; tile < p.ntiles; tile += G) {
            __builtin_amdgcn_s_barrier();   // B(tile): the MFMA waves are done with the band
            if (tile + G < p.ntiles) issue_band(tile + G);
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();   // A(next tile)
        }
        return;
    }
    // ---------------------------------------------------------------------- MFMA waves: 64 pixels x 64 channels each
    const int l31 = lane & 31, h = lane >> 5;
    unsigned wofs[2][2];   // A (weight) fragment addresses inside a tap tile
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wofs[c][ks] = lds_addr(smem + OFF_W) + lds_off(c * 32 + cperm(l31), 2 * ks + h);
    const unsigned band0 = lds_addr(smem + OFF_BAND);
    const char *ctb = smem + OFF_CTAB;
    BP_STAMP(0)
    __builtin_amdgcn_s_barrier();   // A(first tile)
    for (int tile = wg; tile < p.ntiles; tile += G) {
        const int m0 = tile * BM;
        const int G0 = m0 / Wo - 1;
        int rowbase[2][3];   // band pixel of tap (kh, kw = 0) per pixel tile, or the zero area
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int m = m0 + wave * 64 + q * 32 + l31;
            const int Gr = fdiv(m, Wo, p.rcp_wo), x = m - Gr * Wo;
            const int yy = Gr - fdiv(Gr, p.Ho, p.rcp_ho) * p.Ho;
            const int bp0 = (Gr - G0 - 1) * Wb + x;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) rowbase[q][kh] = (unsigned)(yy + kh - 1) < (unsigned)p.Ho ? bp0 + kh * Wb : ZP;
        }
        v16i acc[2][2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][q][r] = 0;
        BP_STAMP(1)
        __builtin_amdgcn_s_setprio(2);
        // 18 batches (tap, k-half) of 4 MFMAs; the fragments of batch b + 1 are requested before the MFMAs of batch b
        v4i wf[2][2], af[2][2];
        unsigned aaddr[2];   // address of k-half 0 of the current tap (k-half 1 = ^ 32)
#define BP_ADDR(TAP)                                                                      \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                       \
        const int tp = rowbase[q][(TAP) / 3] + (TAP) % 3;                                 \
        aaddr[q] = band0 + (unsigned)(tp << 6) + (unsigned)((h ^ ((tp >> 2) & 3)) << 4);  \
    }
#define BP_FETCH(B, BUF)                                                                  \
    {                                                                                     \
        if (((B) & 1) == 0) { BP_ADDR((B) >> 1) }                                         \
        _Pragma("unroll") for (int c = 0; c < 2; ++c) wf[BUF][c] = lds_read16<((B) >> 1) * 4096>(wofs[c][(B) & 1]); \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) af[BUF][q] = lds_read16<0>(((B) & 1) ? aaddr[q] ^ 32u : aaddr[q]); \
    }
#define BP_MMA(B)                                                                         \
    {                                                                                     \
        _Pragma("unroll") for (int c = 0; c < 2; ++c) pin(wf[(B) & 1][c]);                \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) pin(af[(B) & 1][q]);                \
        _Pragma("unroll") for (int c = 0; c < 2; ++c)                                     \
            _Pragma("unroll") for (int q = 0; q < 2; ++q)                                 \
                acc[c][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[(B) & 1][c], af[(B) & 1][q], acc[c][q], 0, 0, 0); \
    }
#define BP_BATCH(B)                   \
    {                                 \
        BP_FETCH((B) + 1, ((B) + 1) & 1) \
        wait_lgkm<4>();               \
        BP_MMA(B)                     \
    }
        BP_FETCH(0, 0)
        BP_BATCH(0) BP_BATCH(1) BP_BATCH(2) BP_BATCH(3) BP_BATCH(4) BP_BATCH(5) BP_BATCH(6) BP_BATCH(7) BP_BATCH(8)
        BP_BATCH(9) BP_BATCH(10) BP_BATCH(11) BP_BATCH(12) BP_BATCH(13) BP_BATCH(14) BP_BATCH(15) BP_BATCH(16)
        wait_lgkm<0>();
        BP_MMA(17)
#undef BP_ADDR
#undef BP_FETCH
#undef BP_MMA
#undef BP_BATCH
        __builtin_amdgcn_s_setprio(0);
        BP_STAMP(2)
        __builtin_amdgcn_s_barrier();   // B(tile): the band may be refilled
        // ------------------------------------------------------------------ epilogue: requant, private transposition, 64-byte rows
        int w[2][2][4];   // [pixel tile][channel tile][4 channels each]
        if constexpr (!RES) {
        // the kernel is bound by the ISSUE of this requantisation (profiles/r02_band_persist.md): with all pre-shifts zero (the
        // usual case, a wave-uniform flag) the table word IS the shift amount - no field extraction, no pre-shift: 3.75 instead
        // of 5.75 VALU instructions per output.  Uniform branch between two instantiations of the same code
        auto requant_all = [&](auto K0c) {
            constexpr bool K0 = decltype(K0c)::value;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    DyNt dm[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const v4i e = *reinterpret_cast<const v4i *>(ctb + (c * 32 + h * 16 + 4 * g + j) * 16);
                        dm[j].m = e.x, dm[j].s = K0 ? e.y : (e.y & 31), dm[j].k = K0 ? 0 : (e.y >> 8);
                        dm[j].add = (long long)(((unsigned long long)(unsigned)e.w << 32) | (unsigned)e.z);
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        int qv[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) qv[j] = med3i(dyadic_mode<(K0 && MODE == 0) ? 1 : MODE>(acc[c][q][4 * g + j], dm[j]), p.q_lo, p.q_hi);
                        w[q][c][g] = pack4_fast(qv[0], qv[1], qv[2], qv[3]);
                    }
                }
        };
        requant_all(std::false_type{});   // (the all-k-zero form under a wave-uniform branch measured neutral in the forward: not kept)
        } else {
            // RESIDUAL: o = ReLU(requant(acc + bias) + requant(identity)) un-clamped -> uint16 (sticky overflow flag); q = next QuantAct of o.
            // A lane owns 16 consecutive channels of a pixel = 32 contiguous bytes of the residual rows: loaded and stored from registers
            const DyNt dids = dynt_prepare(p.m_id, p.e_id), dq = dynt_prepare(p.mq, p.eq);   // (scalars: this kernel has no registers to spare)
            const int qhi2 = (p.q_hi & 0xffff) | (p.q_hi << 16);
            unsigned oor = 0;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int m = m0 + wave * 64 + q * 32 + l31, mr = m < p.M ? m : p.M - 1;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    v4i rin[2];
                    {
                        const v4i *rp = reinterpret_cast<const v4i *>(p.res_in + (size_t)mr * 64 + c * 32 + h * 16);
                        rin[0] = rp[0], rin[1] = rp[1];
                    }
                    int rpk[8], wq[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const unsigned w0 = (unsigned)rin[g >> 1][(g & 1) * 2], w1 = (unsigned)rin[g >> 1][(g & 1) * 2 + 1];
                        const int idin[4] = {(int)(w0 & 0xffffu), (int)(w0 >> 16), (int)(w1 & 0xffffu), (int)(w1 >> 16)};
                        int o[4], qv[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const v4i e = *reinterpret_cast<const v4i *>(ctb + (c * 32 + h * 16 + 4 * g + j) * 16);
                            DyNt dm;
                            dm.m = e.x, dm.s = e.y & 31, dm.k = e.y >> 8;
                            dm.add = (long long)(((unsigned long long)(unsigned)e.w << 32) | (unsigned)e.z);
                            const int a = dyadic_mode<MODE>(acc[c][q][4 * g + j], dm);
                            const int b = dyadic_mode<MODE == 2 ? 2 : 0>(idin[j], dids);
                            o[j] = max(a + b, 0);                  // no clamp: quant_utils.py:456
                            qv[j] = dyadic_mode<MODE>(o[j], dq);   // o >= 0, m >= 0: q >= 0; clamped from above in the pack
                        }
                        if (m < p.M) oor |= (unsigned)(o[0] | o[1]) | (unsigned)(o[2] | o[3]);
                        rpk[2 * g] = pack2_u16_sat(o[0], o[1]);
                        rpk[2 * g + 1] = pack2_u16_sat(o[2], o[3]);
                        wq[g] = pack4_min(qv[0], qv[1], qv[2], qv[3], qhi2);
                    }
                    if (m < p.M && p.res_out) {
                        v4i *dst = reinterpret_cast<v4i *>(p.res_out + (size_t)m * 64 + c * 32 + h * 16);
                        const v4i ra = {rpk[0], rpk[1], rpk[2], rpk[3]}, rb = {rpk[4], rpk[5], rpk[6], rpk[7]};
                        dst[0] = ra, dst[1] = rb;
                    }
                    const v4i ww = {wq[0], wq[1], wq[2], wq[3]};
                    if (p.out != nullptr && p.out_planar) {
                        if (m < p.M) *reinterpret_cast<v4i *>((char *)p.out + ((size_t)(2 * c + h) * p.M + m) * 16) = ww;
                    } else if (p.out != nullptr) {
                        *reinterpret_cast<v4i *>(smem + OFF_STAGE + wave * 2048 + lds_off(l31, 2 * c + h)) = ww;
                    }
                }
                if (p.out != nullptr && !p.out_planar) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {   // 16 pixel rows x 64 B per instruction: 1 KiB contiguous
                        const int row = i * 16 + (lane >> 2), mm = m0 + wave * 64 + q * 32 + row;
                        const v4i d = *reinterpret_cast<const v4i *>(smem + OFF_STAGE + wave * 2048 + row * 64 + ((lane & 3) << 4));
                        if (mm < p.M) *reinterpret_cast<v4i *>((char *)p.out + (size_t)mm * 64 + (((lane & 3) ^ ((row >> 2) & 3)) << 4)) = d;
                    }
                }
            }
            if ((oor >> 16) != 0 && p.res_out) atomicOr(p.flags, 1);
        }
        if (RES) {
        } else if (p.out != nullptr && p.out_planar) {   // planes [4][M][16 B]: a lane's 16 channels are one unit; 32 lanes = 512 contiguous bytes
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int m = m0 + wave * 64 + q * 32 + l31;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const v4i ww = {w[q][c][0], w[q][c][1], w[q][c][2], w[q][c][3]};
                    if (m < p.M) *reinterpret_cast<v4i *>((char *)p.out + ((size_t)(2 * c + h) * p.M + m) * 16) = ww;
                }
            }
        } else if (p.out != nullptr) {
        char *stage = smem + OFF_STAGE + wave * 2048;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const v4i ww = {w[q][c][0], w[q][c][1], w[q][c][2], w[q][c][3]};
                *reinterpret_cast<v4i *>(stage + lds_off(l31, 2 * c + h)) = ww;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {   // 16 pixel rows x 64 B per instruction: 1 KiB contiguous
                const int row = i * 16 + (lane >> 2), m = m0 + wave * 64 + q * 32 + row;
                const v4i d = *reinterpret_cast<const v4i *>(stage + row * 64 + ((lane & 3) << 4));
                if (m < p.M) *reinterpret_cast<v4i *>((char *)p.out + (size_t)m * 64 + (((lane & 3) ^ ((row >> 2) & 3)) << 4)) = d;
            }
        }
        }
        BP_STAMP(3)
        __builtin_amdgcn_s_barrier();   // A(next tile): its band has landed
        BP_STAMP(4)
    }
    if (prof && blockIdx.x == 8 && t == 0)
        for (int k = 0; k < 5; ++k) p.dbgbuf[k] = ph[k];
#undef BP_STAMP
}

}  // namespace

bool band_persist_applies(const hawq_conv_args *a) {
    const int wo = a->W, band_rows = (BM + wo - 1) / wo + 1 + 2;
    const bool epi_ok = (a->epilogue == HAWQ_EPI_REQUANT && a->out_q && a->out_bits == 8) ||
                        (a->epilogue == HAWQ_EPI_RESIDUAL && a->res_in && a->res_in_bits == 16 && (!a->res_out || (a->res_out_bits == 16 && a->flags)) &&
                         !a->res_no_relu && !a->res_clamp16 && (a->res_out || a->out_q) && (!a->out_q || a->out_bits == 8));   // (round 5)
    return a->KH == 3 && a->KW == 3 && a->stride == 1 && a->pad == 1 && a->in2 == nullptr && a->fast_tables != 0 &&
           epi_ok && a->in_bits == 8 && a->w_bits == 8 && a->Cin == 64 &&
           a->Cout == 64 && a->ctab && band_rows * (wo + 2) <= ZP && (a->in_pitch == 0 || a->in_pitch == 64) &&
           (a->out_pitch == 0 || a->out_pitch == 64) &&
           (long long)a->N * a->H * a->W < (1ll << 23);
}

// wgs_per_cu: 1 leaves half of every CU (LDS, wave slots) to whatever runs beside this launch - the other sub-batch chain
// of the engine -, 2 is the faster one when the launch has the chip to itself
int band_persist_launch(const hawq_conv_args *a, int exact_tie, int dbg, int wgs_per_cu, void *stream) {
    BPP p;
    p.in = (const uint8_t *)a->in, p.wgt = (const uint8_t *)a->wgt, p.ctab = a->ctab, p.out = (uint8_t *)a->out_q;
    const bool res = a->epilogue == HAWQ_EPI_RESIDUAL;
    p.res_in = (const uint16_t *)a->res_in, p.res_out = (uint16_t *)a->res_out, p.flags = a->flags, p.out_planar = a->out_planar;
    p.mq = a->out_q && res ? a->mq : 0, p.eq = a->out_q && res ? a->eq : 33, p.m_id = a->m_id_scalar, p.e_id = res ? a->e_id_scalar : 33;
    p.M = a->N * a->H * a->W, p.rows_total = a->N * a->H, p.Ho = a->H, p.Wo = a->W;
    p.in_planar = a->in_planar;
    p.rcp_wo = 1.0f / (float)a->W, p.rcp_ho = 1.0f / (float)a->H;
    p.q_lo = a->relu && a->q_lo < 0 ? 0 : a->q_lo, p.q_hi = a->q_hi;
    p.ntiles = (p.M + BM - 1) / BM;
    p.ck0 = (a->fast_tables & 8) != 0;
    static long long *dbg_dev = nullptr;
    if (HAWQ_DBG_BIT(dbg, 128) && !dbg_dev) (void)hipMalloc(&dbg_dev, 8 * sizeof(long long));
    p.dbgbuf = HAWQ_DBG_BIT(dbg, 128) ? dbg_dev : nullptr;
    static const int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        return n;
    }();
    HAWQ_REQUIRE(n_cu > 0, "hawq_conv2d: cannot read the CU count of the device");
    static const bool attrs = hipFuncSetAttribute((const void *)band_persist_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) == hipSuccess &&
                              hipFuncSetAttribute((const void *)band_persist_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) == hipSuccess &&
                              hipFuncSetAttribute((const void *)band_persist_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) == hipSuccess &&
                              hipFuncSetAttribute((const void *)band_persist_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) == hipSuccess;
    HAWQ_REQUIRE(attrs, "hawq_conv2d: hipFuncSetAttribute failed for the weight-stationary 3x3 kernel");
    const int grid = p.ntiles < wgs_per_cu * n_cu ? p.ntiles : wgs_per_cu * n_cu;
    if (res && exact_tie)
        hipLaunchKernelGGL((band_persist_kernel<true, true>), dim3(grid), dim3(NT), LDS_BYTES, (hipStream_t)stream, p);
    else if (res)
        hipLaunchKernelGGL((band_persist_kernel<false, true>), dim3(grid), dim3(NT), LDS_BYTES, (hipStream_t)stream, p);
    else if (exact_tie)
        hipLaunchKernelGGL(band_persist_kernel<true>, dim3(grid), dim3(NT), LDS_BYTES, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(band_persist_kernel<false>, dim3(grid), dim3(NT), LDS_BYTES, (hipStream_t)stream, p);
    HAWQ_CHECK_HIP(hipGetLastError());
    if (p.dbgbuf) {   // experiment hook (synchronises!)
        long long hb[5];
        (void)hipStreamSynchronize((hipStream_t)stream);
        (void)hipMemcpy(hb, p.dbgbuf, sizeof(hb), hipMemcpyDeviceToHost);
        fprintf(stderr, "[band-persist M=%d grid=%d] cycles of wave 0 / workgroup 8: prologue %lld | tile setup %lld | MFMA phase %lld | barrier B + epilogue %lld | wait for the next band %lld\n",
                p.M, grid, hb[0], hb[1], hb[2], hb[3], hb[4]);
    }
    return 0;
}
