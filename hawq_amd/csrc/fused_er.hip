// Fused "expand -> reduce" launch for gfx950 (MI355X): the 1x1 expand conv of bottleneck unit i with its residual
// epilogue AND the 1x1 reduce conv of unit i+1 in one kernel.
//
// Reference graph (q_resnet.py:231-260): ... conv3(i) -> x + identity -> quant_act_int32 -> ReLU |unit i+1| quant_act
// -> conv1 -> ReLU -> quant_act1.  Launched separately, conv3's kernel is bound by memory (uint16 residual in and out)
// and epilogue VALU while the matrix pipe idles (2-6 % busy), writes the 8-bit block input `q` of unit i+1, and conv1's
// kernel reads it straight back - a short latency-bound GEMM.  Here a workgroup owns BM pixels and walks the expand
// conv's C3 output channels in slices of 64:
//     GEMM1  acc1[BM x 64]  = x2[BM x C] . W3[slice][C]^T              (K = C, operands: resident x2 tile, W3 slice)
//     epi 1  o = ReLU(requant(acc1) + requant(residual)); residual slice out (uint16); q = QuantAct_{i+1}(o) -> LDS
//     GEMM2  acc2[BM x C'] += q[BM x 64] . W1[:, slice]^T              (K = 64 of the reduce conv's K = C3)
// and after the last slice  y = QuantAct1(ReLU(acc2 + bias1)) -> int8 [M][C'].  The block input q never reaches memory
// (one write + one read of M x C3 bytes per unit) and the reduce conv's MACs run under the expand epilogue's
// memory / VALU time.  Every rounding point of SURVEY.md App. A is kept: each residual branch is requantised separately,
// summed un-clamped (quant_utils.py:416-456), q is the block-input QuantAct of the stored 16-bit value
// (quant_modules.py:288-293), the reduce conv accumulates exact int32 over all C3 channels before its own requant.
//
// int8 x int8, fast-contract tables (see hawq_conv_args.fast_tables; exact-tie mode as a separate instantiation),
// uint16 residuals, C = C' in {64, 128, 256} (ResNet50 stages 1-3).
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace {

struct ERP {
    const uint8_t *x2, *w3, *w1;
    const uint8_t *xid, *wid;          // DUAL: input and weights of the unit's 1x1 identity conv (same pixels: stride 1, C channels)
    const int32_t *ctab3, *ctab1, *ctab_id;
    const uint16_t *res_in;
    uint16_t *res_out;
    uint8_t *y;
    int M, C3;
    int m_id_s, e_id_s, mq, eq, q_hi;  // scalar tables: identity pass-through, QuantAct of unit i+1 (q >= 0: post-ReLU)
    int y_lo, y_hi;                    // clamp of the reduce conv's QuantAct (ReLU folded into y_lo)
    int y_planar;
    int y_nib;                         // the reduce conv's output is stored hawq4 (4-bit values, two channels per byte) for a nibble 3x3 conv
    int32_t *flags;
    int dbg;             // HAWQ_DBG ablations (timing experiments only): 256 = no residual stores, 512 = no residual loads
    long long *dbgbuf;   // HAWQ_DBG=128: per-phase cycle sums of wave 0 of workgroup 8 (timing experiments only)
};

__device__ __attribute__((aligned(16))) const int g_er_zero16[4] = {0, 0, 0, 0};

// C: channels of the expand conv's input = of the reduce conv's output.  WM: pixel MFMA tiles (32 pixels) per
// workgroup = compute waves along the pixel axis (x 2 along the channel axis).  NP: producer waves that own every
// LDS-DMA instruction (0: the compute waves issue them themselves).  An LDS-DMA instruction costs its wave 180-330
// cycles (HAWQ_DBG=128 stamps, profiles/r02_fused_er_steps.md) - the per-wave ingest cap of tools/ubench/ingest_shape.hip
// seen from inside - so the ingest rate of a CU is set by HOW MANY waves issue, and producers only pay when they add
// issuing waves without costing compute occupancy (they get the same register allocation as the compute waves).
// RESDMA: the old residual slice is prefetched one slice ahead into LDS by the producers (memory-bound stage 1: the
// HBM latency needs a whole slice of cover); otherwise each compute lane loads its 32 bytes straight into registers at
// the top of the slice (L2 / Infinity-Cache resident in the later stages; 16 KiB less LDS, more workgroups per CU).
// DUAL: the first unit of a stage whose identity branch is a 1x1 / stride-1 conv over the same pixels (ResNet50 stage 1):
// the identity conv is a second GEMM1 (its own resident input tile, its weight slice in the same ring stage as W3's) and its
// requantised accumulators take the place of the stored residual (q_resnet.py:236-251).
template <int C_, int WM_, int NP_, bool RESDMA_, int MINB_, bool DUAL_ = false>
struct ERCfg {
    static constexpr int C = C_, WM = WM_, NP = NP_, MINB = MINB_;
    static constexpr bool RESDMA = RESDMA_, DUAL = DUAL_;
    static constexpr int BM = 32 * WM, NW = 2 * WM, NTC = 64 * NW, NT = NTC + 64 * NP;
    static constexpr int NI = NP > 0 ? NP : NW, NPT = 64 * NI;   // waves / threads that issue LDS-DMA (NP == 0: the compute waves themselves)
    static constexpr int KC = C / 64;          // 64-byte chunks of GEMM1's K
    static constexpr int CT2 = C / 64;         // 32-channel MFMA tiles per wave in GEMM2 (2 waves across the C channels)
    static constexpr int RPP = NPT / 4;        // operand rows per producer LDS-DMA pass (4 lanes x 16 B per 64-byte row)
    static constexpr int NSW = 3;
    static constexpr int WSTAGE = (DUAL ? 2 : 1) * 64 * C;   // W3 slice [KC][64 rows][64 B] (+ identity slice)  >=  W1 slice [C rows][64 B]
    static constexpr int W1PASS = 64 * C / (RPP * 64);        // LDS-DMA instructions per thread of a W1 slice
    static constexpr int WPASS = WSTAGE / (RPP * 64);   // LDS-DMA instructions per producer thread per ring stage
    static constexpr int XPASS = BM / RPP;     // per 64-byte chunk of the x2 tile
    static constexpr int RPASS = BM * 8 / NPT; // residual slice: BM rows x 8 chunks of 16 B
    static constexpr int X2_BYTES = (DUAL ? 2 : 1) * BM * C;   // resident input tile(s): x2 (+ the identity conv's input)
    static constexpr int Q_BYTES = BM * 64, RES_BYTES = BM * 128;
    // LDS map.  q is single-buffered: written between B1(j) and B2(j), read between B2(j) and B1(j+1); so is the residual
    // staging tile unless it doubles as the prefetch target (RESDMA: in place, two buffers)
    static constexpr int OFF_X2 = 0;
    static constexpr int OFF_RING = OFF_X2 + X2_BYTES;
    static constexpr int OFF_Q = OFF_RING + NSW * WSTAGE;       // [BM][64 B]
    static constexpr int OFF_RES = OFF_Q + Q_BYTES;             // [1 or 2][BM][64] uint16
    static constexpr int OFF_CT3 = OFF_RES + (RESDMA ? 2 : 1) * RES_BYTES;   // [2][64][16 B] (DUAL: + the identity conv's)
    static constexpr int LDS_BYTES = OFF_CT3 + (DUAL ? 4 : 2) * 1024;
    static_assert(!(DUAL && RESDMA), "a dual-branch unit has no stored residual to prefetch");
    static_assert(WSTAGE % (RPP * 64) == 0 && BM % RPP == 0 && (BM * 8) % NPT == 0 && C % RPP == 0 && (RPP & 15) == 0,
                  "tiles must fill whole producer passes");
    static_assert(BM * C <= NSW * WSTAGE, "the output tile is staged on the weight ring");
};

__device__ __forceinline__ void dma16(const char *src, char *dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
}
__device__ __forceinline__ void dma4(const char *src, char *dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)dst, 4, 0, 0);
}
// K0: the host guarantees k == 0 for every channel of this table (hawq_conv_args.fast_tables bit 3): the entry's second word IS the
// shift, two extraction instructions and the pre-shift disappear (3 of ~17 VALU instructions per residual output; the stage-1/2
// launches of this kernel are 40-60 % VALU-busy, profiles/r02_e_pmc_MFMA.md)
template <bool K0 = false>
__device__ __forceinline__ DyNt ctab_entry(const char *ctab, int ch) {
    const v4i t = *reinterpret_cast<const v4i *>(ctab + ch * 16);
    DyNt d;
    d.m = t.x, d.s = K0 ? t.y : (t.y & 31), d.k = K0 ? 0 : (t.y >> 8);
    d.add = (long long)(((unsigned long long)(unsigned)t.w << 32) | (unsigned)t.z);
    return d;
}

// QK0: the next unit's QuantAct table (mq, eq) carries no pre-shift either (the usual case: a 16-bit -> 8-bit ratio is ~2^-7)
template <class F, bool TIE, bool CK0 = false, bool QK0 = false>
__global__ __launch_bounds__(F::NT, F::MINB) void expand_reduce_kernel(const ERP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MODE = TIE ? 2 : 0;                    // scalar identity table (uniform pre-shift)
    constexpr int MODE_Q = TIE ? 2 : (QK0 ? 1 : 0);      // scalar table of the next QuantAct
    constexpr int MODE_C = TIE ? 2 : (CK0 ? 1 : 0);      // per-channel tables
    constexpr bool K0 = CK0 && !TIE;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool producer = F::NP > 0 && wave >= F::NW;
    const int m0 = blockIdx.x * F::BM;
    const int nslices = p.C3 >> 6;
    char *x2t = smem + F::OFF_X2, *ring = smem + F::OFF_RING, *qt = smem + F::OFF_Q, *rest = smem + F::OFF_RES;
    char *ct3 = smem + F::OFF_CT3;
    const bool prof = HAWQ_DBG_BIT(~0, 128) && p.dbgbuf != nullptr;
    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = prof ? (long long)__builtin_readcyclecounter() : 0;
#define ER_STAMP(K)                                                    \
    if (prof) {                                                        \
        const long long now = (long long)__builtin_readcyclecounter(); \
        ph[K] += now - tprev;                                          \
        tprev = now;                                                   \
    }
    // Ring stage of W3(j) is (2j) % 3, of W1(j) is (2j + 1) % 3.  Two barriers per slice:
    //   B1(j): W3(j), ctab3(j) (and residual(j)) have landed; every wave is done with slice j-1
    //   B2(j): W1(j) has landed; q(j) and the new residual slice are written; GEMM1(j) has released W3(j)'s stage
    // ---------------------------------------------------------------- LDS-DMA issue (producer waves, or every compute wave)
    const int pt = F::NP > 0 ? t - F::NTC : t, pw = F::NP > 0 ? wave - F::NW : wave;   // index among the issuing threads / waves
    const int prow = pt >> 2, pslot = pt & 3;
    const int sw = (pslot ^ ((prow >> 2) & 3)) << 4;   // source-side swizzle of this thread's 16-byte slot (same for rows prow + k * RPP)
    const char *zero = reinterpret_cast<const char *>(g_er_zero16);
    auto issue_w3 = [&](int j, int stage) {   // rows j*64 .. j*64+63 of W3 [C3][C], as KC chunks of [64 rows][64 B] (+ identity weights)
        char *dst = ring + stage * F::WSTAGE + pw * 1024;
#pragma unroll
        for (int i = 0; i < F::W1PASS; ++i) {
            const int idx = i * F::RPP + prow, chunk = idx >> 6, row = idx & 63;
            dma16((const char *)p.w3 + (size_t)(j * 64 + row) * F::C + chunk * 64 + sw, dst + i * (F::RPP * 64));
            if constexpr (F::DUAL)
                dma16((const char *)p.wid + (size_t)(j * 64 + row) * F::C + chunk * 64 + sw, dst + 64 * F::C + i * (F::RPP * 64));
        }
    };
    auto issue_w1 = [&](int j, int stage) {   // columns j*64 .. j*64+63 of W1 [C][C3], as [C rows][64 B]
        char *dst = ring + stage * F::WSTAGE + pw * 1024;
#pragma unroll
        for (int i = 0; i < F::W1PASS; ++i) {
            const int row = i * F::RPP + prow;
            dma16((const char *)p.w1 + (size_t)row * p.C3 + j * 64 + sw, dst + i * (F::RPP * 64));
        }
    };
    auto issue_grp = [&](int j) {   // ctab3 slice: 4 x 256 B (issuing waves beyond the 4th repeat); residual slice if RESDMA
        dma4((const char *)p.ctab3 + (size_t)j * 1024 + (pw & 3) * 256 + lane * 4, ct3 + (j & 1) * 1024 + (pw & 3) * 256);
        if constexpr (F::DUAL)
            dma4((const char *)p.ctab_id + (size_t)j * 1024 + (pw & 3) * 256 + lane * 4, ct3 + 2048 + (j & 1) * 1024 + (pw & 3) * 256);
        if constexpr (F::RESDMA) {
            char *dst = rest + (j & 1) * F::RES_BYTES;
#pragma unroll
            for (int i = 0; i < F::RPASS; ++i) {
                const int base = (i * F::NI + pw) * 64, idx = base + lane;
                const int row = idx >> 3, jj = idx & 7;
                const int grow = (m0 + row < p.M) ? m0 + row : m0;
                dma16(HAWQ_DBG_BIT(p.dbg, 512) ? zero : (const char *)p.res_in + ((size_t)grow * p.C3 + j * 64) * 2 + ((jj ^ (row & 7)) << 4), dst + base * 16);
            }
        }
    };
    auto issue_prologue = [&]() {
#pragma unroll
        for (int kc = 0; kc < F::KC; ++kc)   // resident x2 tile: KC chunks of [BM pixel rows][64 B]
#pragma unroll
            for (int i = 0; i < F::XPASS; ++i) {
                const int row = i * F::RPP + prow;
                dma16(m0 + row < p.M ? (const char *)p.x2 + (size_t)(m0 + row) * F::C + kc * 64 + sw : zero,
                      x2t + kc * (F::BM * 64) + i * (F::RPP * 64) + pw * 1024);
                if constexpr (F::DUAL)
                    dma16(m0 + row < p.M ? (const char *)p.xid + (size_t)(m0 + row) * F::C + kc * 64 + sw : zero,
                          x2t + F::BM * F::C + kc * (F::BM * 64) + i * (F::RPP * 64) + pw * 1024);
            }
        issue_w3(0, 0);
        issue_grp(0);
        issue_w1(0, 1);
        wait_vmcnt<F::W1PASS>();   // everything but W1(0) has landed
    };
    constexpr int N_A = F::WPASS + (F::DUAL ? 2 : 1) + (F::RESDMA ? F::RPASS : 0);   // instructions per thread of one {W3 (+ identity), ctab3, residual} group
    if (F::NP > 0 && producer) {
        // ------------------------------------------------------------ producer waves: all LDS-DMA, nothing else
        issue_prologue();
        for (int j = 0; j < nslices; ++j) {
            __builtin_amdgcn_s_barrier();   // B1(j)
            if (j + 1 < nslices) {
                issue_w3(j + 1, (2 * j + 2) % 3);   // the stage GEMM2(j-1) read
                issue_grp(j + 1);
                wait_vmcnt<N_A>();                  // W1(j) is older than this group
            } else {
                wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();   // B2(j)
            if (j + 1 < nslices) {
                issue_w1(j + 1, (2 * j + 3) % 3);   // the stage GEMM1(j) read
                wait_vmcnt<F::W1PASS>();            // the {W3, ctab3, residual}(j+1) group has landed (no stores here: counts are exact)
            }
        }
        __syncthreads();
        __syncthreads();
        return;
    }
    // ---------------------------------------------------------------- compute waves
    const int wave_m = wave % F::WM, wave_c = wave / F::WM;   // GEMM1: 32 px x 32 ch per wave; GEMM2: 32 px x C/2 ch
    const int l31 = lane & 31, h = lane >> 5;
    v16i acc2[F::CT2];
#pragma unroll
    for (int c = 0; c < F::CT2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[c][r] = 0;
    const int arow = wave_m * 32 + l31;                           // this lane's pixel row (both GEMMs, both epilogues)
    const int wrow1 = wave_c * 32 + cperm(l31);                   // GEMM1: W3 slice row
    const int lch = wave_c * 32 + h * 16;                         // slice-local first channel of this lane's 16 outputs
    DyNt dids = dynt_prepare(p.m_id_s, p.e_id_s), dq = dynt_prepare(p.mq, p.eq);
    // hide the (zero) low words of the scalar tables' rounding constants from the optimiser: knowing them it splits the 64-bit
    // multiply-add into v_mul_hi_i32 + v_add; opaque, it emits ONE v_mad_i64_i32 with the constant as an SGPR-pair addend
    // (2 of ~17 VALU instructions per residual output)
    asm volatile("" : "+s"(dids.add), "+s"(dq.add));
    const unsigned rowmask = (m0 + arow < p.M) ? 0xffffffffu : 0u;
    const int qhi2 = (p.q_hi & 0xffff) | (p.q_hi << 16);
    const int res_row = (m0 + arow < p.M) ? m0 + arow : m0;       // rows beyond M read a valid row and are never stored
    unsigned oor = 0;
    if constexpr (F::NP == 0) issue_prologue();
    ER_STAMP(0)
    for (int j = 0; j < nslices; ++j) {
        const int st3 = (2 * j) % 3, st1 = (2 * j + 1) % 3;
        __builtin_amdgcn_s_barrier();   // B1(j)
        ER_STAMP(1)
        // !RESDMA: this lane's 16 residual values of slice j (32 contiguous bytes) straight into registers.  Hand-issued
        // and waited for (the only younger memory operations of a compute wave are none; its stores are older)
        v4i rin[2];
        if constexpr (!F::RESDMA && !F::DUAL) {
            if (HAWQ_DBG_BIT(p.dbg, 512)) {
                rin[0] = rin[1] = v4i{0, 0, 0, 0};
            } else {
            const char *rp = (const char *)p.res_in + ((size_t)res_row * p.C3 + j * 64 + lch) * 2;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rin[0]) : "v"(rp) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(rin[1]) : "v"(rp) : "memory");
            }
        }
        if constexpr (F::NP == 0)   // after the residual loads: the counted wait in front of the epilogue leaves this group in flight
            if (j + 1 < nslices) {
                issue_w3(j + 1, (2 * j + 2) % 3);
                issue_grp(j + 1);
            }
        ER_STAMP(2)
        // ------------------------------------------------------------ GEMM1
        v16i acc1, acc_id;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = 0, acc_id[r] = 0;
        {
            const char *w3s = ring + st3 * F::WSTAGE;
#pragma unroll
            for (int kc = 0; kc < F::KC; ++kc)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const v4i wf = *reinterpret_cast<const v4i *>(w3s + kc * 4096 + lds_off(wrow1, 2 * ks + h));
                    const v4i af = *reinterpret_cast<const v4i *>(x2t + kc * (F::BM * 64) + lds_off(arow, 2 * ks + h));
                    acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, af, acc1, 0, 0, 0);
                    if constexpr (F::DUAL) {
                        const v4i wi = *reinterpret_cast<const v4i *>(w3s + 64 * F::C + kc * 4096 + lds_off(wrow1, 2 * ks + h));
                        const v4i ai = *reinterpret_cast<const v4i *>(x2t + F::BM * F::C + kc * (F::BM * 64) + lds_off(arow, 2 * ks + h));
                        acc_id = __builtin_amdgcn_mfma_i32_32x32x32_i8(wi, ai, acc_id, 0, 0, 0);
                    }
                }
        }
        ER_STAMP(3)
        // ------------------------------------------------------------ epilogue 1: residual add, ReLU, next QuantAct
        {
            char *rb = rest + (F::RESDMA ? (j & 1) * F::RES_BYTES : 0) + arow * 128;
            const char *ctb = ct3 + (j & 1) * 1024;
            if constexpr (F::DUAL) {
                if constexpr (F::NP == 0) {   // W1(j) must have landed before B2; nothing else to wait for here
                    if (j + 1 < nslices) wait_vmcnt<N_A>(); else wait_vmcnt<0>();
                }
            } else if constexpr (F::RESDMA) {
                rin[0] = *reinterpret_cast<const v4i *>(rb + (((lch >> 3) ^ (arow & 7)) << 4));
                rin[1] = *reinterpret_cast<const v4i *>(rb + ((((lch >> 3) + 1) ^ (arow & 7)) << 4));
            } else {
                // older than the residual loads: W1(j), the stores of slice j-1; younger: only the DMA group (NP == 0)
                if (F::NP == 0 && j + 1 < nslices) wait_vmcnt<N_A>(); else wait_vmcnt<0>();
                asm volatile("" : "+v"(rin[0]), "+v"(rin[1]));
            }
            int rpack[8], qpack[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int idin[4];
                if constexpr (!F::DUAL) {
                    const unsigned w0 = (unsigned)rin[g >> 1][(g & 1) * 2], w1 = (unsigned)rin[g >> 1][(g & 1) * 2 + 1];
                    idin[0] = (int)(w0 & 0xffffu), idin[1] = (int)(w0 >> 16), idin[2] = (int)(w1 & 0xffffu), idin[3] = (int)(w1 >> 16);
                }
                int o[4], qv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const DyNt dm = ctab_entry<K0>(ctb, lch + 4 * g + k);
                    const int a = dyadic_mode<MODE_C>(acc1[4 * g + k], dm);
                    const int b = F::DUAL ? dyadic_mode<MODE_C>(acc_id[4 * g + k], ctab_entry<K0>(ctb + 2048, lch + 4 * g + k))
                                          : dyadic_mode<MODE>(idin[k], dids);
                    o[k] = max(a + b, 0);                                  // no clamp: quant_utils.py:456
                    qv[k] = dyadic_mode<MODE_Q>(o[k], dq);                 // o >= 0, m >= 0: q >= 0; clamped from above in the pack
                }
                oor |= ((unsigned)(o[0] | o[1]) | (unsigned)(o[2] | o[3])) & rowmask;
                rpack[2 * g] = pack2_u16_sat(o[0], o[1]);
                rpack[2 * g + 1] = pack2_u16_sat(o[2], o[3]);
                qpack[g] = pack4_min(qv[0], qv[1], qv[2], qv[3], qhi2);
            }
            const v4i ra = {rpack[0], rpack[1], rpack[2], rpack[3]}, rc = {rpack[4], rpack[5], rpack[6], rpack[7]};
            *reinterpret_cast<v4i *>(rb + (((lch >> 3) ^ (arow & 7)) << 4)) = ra;
            *reinterpret_cast<v4i *>(rb + ((((lch >> 3) + 1) ^ (arow & 7)) << 4)) = rc;
            const v4i qw = {qpack[0], qpack[1], qpack[2], qpack[3]};
            *reinterpret_cast<v4i *>(qt + lds_off(arow, lch >> 4)) = qw;
        }
        ER_STAMP(4)
        if constexpr (F::NP == 0 && F::RESDMA) {   // W1(j) is older than this slice's group
            if (j + 1 < nslices) wait_vmcnt<N_A>(); else wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();   // B2(j)
        ER_STAMP(5)
        // ------------------------------------------------------------ GEMM2 partial sum over this slice's 64 channels
        {
            const char *w1s = ring + st1 * F::WSTAGE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const v4i af = *reinterpret_cast<const v4i *>(qt + lds_off(arow, 2 * ks + h));
#pragma unroll
                for (int c = 0; c < F::CT2; ++c) {
                    const v4i wf = *reinterpret_cast<const v4i *>(w1s + lds_off(wave_c * (F::CT2 * 32) + c * 32 + cperm(l31), 2 * ks + h));
                    acc2[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, af, acc2[c], 0, 0, 0);
                }
            }
        }
        ER_STAMP(6)
        // Stores count on vmcnt like loads and may retire before OLDER loads: every load that a later counted wait
        // targets must have landed before a store is issued ({W3, ctab3, residual}(j+1) had the whole slice)
        if constexpr (F::NP == 0) wait_vmcnt<0>();
        {   // new residual slice -> memory, whole 128-byte rows
            const char *src = rest + (F::RESDMA ? (j & 1) * F::RES_BYTES : 0);
#pragma unroll
            for (int i = 0; i < F::BM * 8 / F::NTC; ++i) {
                const int idx = t + F::NTC * i, row = idx >> 3, jj = idx & 7;
                if (m0 + row < p.M && !HAWQ_DBG_BIT(p.dbg, 256))
                    *reinterpret_cast<v4i *>((char *)p.res_out + ((size_t)(m0 + row) * p.C3 + j * 64) * 2 + ((jj ^ (row & 7)) << 4)) =
                        *reinterpret_cast<const v4i *>(src + idx * 16);
            }
        }
        if constexpr (F::NP == 0)
            if (j + 1 < nslices) issue_w1(j + 1, (2 * j + 3) % 3);   // into the stage GEMM1(j) read (all waves are past B2(j))
        ER_STAMP(7)
    }
    if (prof && blockIdx.x == 8 && t == 0)
        for (int k = 0; k < 8; ++k) p.dbgbuf[k] = ph[k];
#undef ER_STAMP
    if ((oor >> 16) != 0) atomicOr(p.flags, 1);
    // ---------------------------------------------------------------- epilogue 2: the reduce conv's QuantAct
    __syncthreads();   // all GEMM2 fragment reads done: the ring becomes the output staging tile [BM][C B]
    {
        constexpr int CPR = F::C / 16;   // 16-byte chunks per output row
        char *yt = ring;
        if (p.y_nib) {
            // hawq4 output (the next 3x3 conv runs its nibble pipeline): 16 channels of a lane = 8 bytes, byte k of an 8-channel
            // group = c_k | c_{k+4} << 4 (include/hawq_mi355.h); rows are C / 2 bytes, 16-byte chunks hold 32 channels
            constexpr int CPN = F::C / 32;
#pragma unroll
            for (int c = 0; c < F::CT2; ++c) {
                const int ch0 = wave_c * (F::CT2 * 32) + c * 32 + h * 16;
                int qv[16];
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    qv[k] = med3i(dyadic_mode<MODE_C>(acc2[c][k], ctab_entry<K0>((const char *)p.ctab1, ch0 + k)), p.y_lo, p.y_hi);
                const v2i ww = {(int)pack8_u4(&qv[0]), (int)pack8_u4(&qv[8])};
                *reinterpret_cast<v2i *>(yt + (arow * CPN + ((ch0 >> 5) ^ (arow & (CPN - 1)))) * 16 + ((ch0 >> 4) & 1) * 8) = ww;
            }
            __syncthreads();
            if (p.y_planar) {   // planes [C / 32][M][16 B]
                for (int idx = t; idx < F::BM * CPN; idx += F::NTC) {
                    const int ch = idx / F::BM, row = idx % F::BM;
                    if (m0 + row < p.M)
                        *reinterpret_cast<v4i *>((char *)p.y + ((size_t)ch * p.M + (m0 + row)) * 16) =
                            *reinterpret_cast<const v4i *>(yt + (row * CPN + (ch ^ (row & (CPN - 1)))) * 16);
                }
            } else {
                for (int idx = t; idx < F::BM * CPN; idx += F::NTC) {
                    const int row = idx / CPN, jj = idx % CPN;
                    if (m0 + row < p.M)
                        *reinterpret_cast<v4i *>((char *)p.y + (size_t)(m0 + row) * (F::C / 2) + ((jj ^ (row & (CPN - 1))) << 4)) =
                            *reinterpret_cast<const v4i *>(yt + idx * 16);
                }
            }
            return;
        }
#pragma unroll
        for (int c = 0; c < F::CT2; ++c) {
            const int ch0 = wave_c * (F::CT2 * 32) + c * 32 + h * 16;
            int w[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int qv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    qv[k] = med3i(dyadic_mode<MODE_C>(acc2[c][4 * g + k], ctab_entry<K0>((const char *)p.ctab1, ch0 + 4 * g + k)), p.y_lo, p.y_hi);
                w[g] = pack4_fast(qv[0], qv[1], qv[2], qv[3]);
            }
            const v4i ww = {w[0], w[1], w[2], w[3]};
            *reinterpret_cast<v4i *>(yt + (arow * CPR + ((ch0 >> 4) ^ (arow & (CPR - 1)))) * 16) = ww;
        }
        __syncthreads();
        if (p.y_planar) {   // channel-group planes [C / 16][M][16 B] (hawq_conv_args.out_planar)
#pragma unroll
            for (int i = 0; i < F::BM * CPR / F::NTC; ++i) {
                const int idx = t + F::NTC * i, ch = idx / F::BM, row = idx % F::BM;
                if (m0 + row < p.M)
                    *reinterpret_cast<v4i *>((char *)p.y + ((size_t)ch * p.M + (m0 + row)) * 16) =
                        *reinterpret_cast<const v4i *>(yt + (row * CPR + (ch ^ (row & (CPR - 1)))) * 16);
            }
        } else {
#pragma unroll
            for (int i = 0; i < F::BM * CPR / F::NTC; ++i) {
                const int idx = t + F::NTC * i, row = idx / CPR, jj = idx % CPR;
                if (m0 + row < p.M)
                    *reinterpret_cast<v4i *>((char *)p.y + (size_t)(m0 + row) * F::C + ((jj ^ (row & (CPR - 1))) << 4)) =
                        *reinterpret_cast<const v4i *>(yt + idx * 16);
            }
        }
    }
}

// MINB = waves per SIMD the register allocation must allow (launch_bounds)
using E64 = ERCfg<64, 2, 0, true, 4>;       // stage 1: 64 pixels, 4 waves, 38 KiB of LDS -> 4 workgroups per CU; residual prefetched into LDS
using E64R = ERCfg<64, 2, 0, false, 4>;     //          residual through registers, 30 KiB
using E128 = ERCfg<128, 2, 0, false, 3>;    // stage 2: 64 pixels, 4 waves, 46 KiB -> 3 workgroups per CU
using E128D = ERCfg<128, 2, 0, true, 3>;    //          residual prefetched into LDS, 54 KiB
using E128P = ERCfg<128, 2, 4, true, 4>;    //          4 + 4 waves (producers own the LDS-DMA), two workgroups per CU
using E256 = ERCfg<256, 4, 0, false, 2>;    // stage 3: 128 pixels, 8 waves, 106 KiB, one workgroup per CU
using E256P = ERCfg<256, 4, 4, true, 3>;    //          8 + 4 waves, residual prefetched into LDS, 122 KiB
using E256S = ERCfg<256, 2, 0, false, 2>;   //          64 pixels, 4 waves, residual through registers, 78 KiB: two workgroups per CU, twice the workgroups
using E256SP = ERCfg<256, 2, 4, true, 2>;    //          64 pixels, 4 + 4 waves, 86 KiB
using E64D = ERCfg<64, 2, 0, false, 4, true>;   // stage 1, first unit: identity conv as a second GEMM1, 46 KiB
constexpr int NUM_ER = 10;

typedef void (*ERFn)(const ERP);
struct ERInfo { ERFn fn[5]; int c, bm, nt, lds; bool dual; };   // fn: {general, exact-tie, per-channel k all zero, + next-QuantAct k zero, only the latter}
#define ER_ENTRY(F) {{expand_reduce_kernel<F, false>, expand_reduce_kernel<F, true>, expand_reduce_kernel<F, false, true>, expand_reduce_kernel<F, false, true, true>, \
                      expand_reduce_kernel<F, false, false, true>}, F::C, F::BM, F::NT, F::LDS_BYTES, F::DUAL}
const ERInfo kER[NUM_ER] = {ER_ENTRY(E64), ER_ENTRY(E64R), ER_ENTRY(E128), ER_ENTRY(E128D), ER_ENTRY(E128P), ER_ENTRY(E256), ER_ENTRY(E256P), ER_ENTRY(E256S), ER_ENTRY(E256SP), ER_ENTRY(E64D)};

bool conv_is_1x1_int8_fast(const hawq_conv_args &a, bool dual_ok = false) {
    return (a.in_pitch == 0 || a.in_pitch == a.Cin) && (a.out_pitch == 0 || a.out_pitch == a.Cout) && a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && a.in_bits == 8 && a.w_bits == 8 && a.fast_tables != 0 &&
           (dual_ok || !a.in2) && !a.in_planar;
}

// the expand conv carries a second (identity) branch the dual variant can take: 1x1 / stride 1 over the same pixels, same channel count
bool dual_branch_fits(const hawq_conv_args &e) {
    return e.in2 && e.wgt2 && e.ctab_id && e.Cin2 == e.Cin && e.stride2 == 1 && e.H2 == e.H && e.W2 == e.W && e.in2_bits == 8 && e.w2_bits == 8;
}

// variant index (into kER) for this pair, or -1.  `tile` 0 = default, 1.. = the variants that take channel count C in table order
int er_variant(const hawq_expand_reduce_args *a) {
    const hawq_conv_args &e = a->expand, &r = a->reduce;
    if (!conv_is_1x1_int8_fast(e, true) || !conv_is_1x1_int8_fast(r)) return -1;
    const bool dual = e.in2 != nullptr;
    if (dual && !dual_branch_fits(e)) return -1;
    if (e.epilogue != HAWQ_EPI_RESIDUAL || r.epilogue != HAWQ_EPI_REQUANT) return -1;
    if (!dual && (!e.res_in || e.res_in_bits != 16)) return -1;
    if (!e.res_out || e.res_out_bits != 16 || !e.flags || !e.ctab || !r.ctab || !r.out_q) return -1;
    if ((r.out_bits != 8 && r.out_bits != 4) || e.out_bits != 8) return -1;
    if (r.out_bits == 4 && (r.q_lo < 0 || r.q_hi > 15)) return -1;   // hawq4 stores unsigned nibbles
    if (r.Cin != e.Cout || r.Cout != e.Cin || r.N != e.N || r.H != e.H || r.W != e.W || e.Cout % 64) return -1;
    int nth = 0, first = -1;
    for (int i = 0; i < NUM_ER; ++i)
        if (kER[i].c == e.Cin && kER[i].dual == dual) {
            if (first < 0) first = i;
            if (++nth == a->tile) return i;
        }
    return a->tile == 0 ? first : -1;
}

}  // namespace

// wave-private variants (fused_wp.hip): numbered after the variants of this file; they also take reduce.wgt == NULL
int wp_num_variants(const hawq_expand_reduce_args *a);
int wp_launch(const hawq_expand_reduce_args *a, int nth, void *stream);
// software-pipelined pair variants (fused_er2.hip): numbered after the wave-private ones
int er2_num_variants(const hawq_expand_reduce_args *a);
int er2_launch(const hawq_expand_reduce_args *a, int nth, void *stream);

namespace {
int er_count(const hawq_expand_reduce_args *a) {   // variants of THIS file that take the pair
    if (!a->reduce.wgt) return 0;
    hawq_expand_reduce_args q = *a;
    q.tile = 0;
    if (er_variant(&q) < 0) return 0;
    int n = 0;
    for (int i = 0; i < NUM_ER; ++i) n += kER[i].c == a->expand.Cin && kER[i].dual == (a->expand.in2 != nullptr);
    return n;
}
}  // namespace

extern "C" int hawq_conv_expand_reduce_variants(const hawq_expand_reduce_args *a) {
    if (!a) return 0;
    return er_count(a) + wp_num_variants(a) + er2_num_variants(a);
}

extern "C" int hawq_conv_expand_reduce(const hawq_expand_reduce_args *a, void *stream) {
    HAWQ_REQUIRE(a != nullptr, "hawq_conv_expand_reduce: null args");
    const int n_er = er_count(a), n_wp = wp_num_variants(a);
    if (a->tile > n_er + n_wp) return er2_launch(a, a->tile - n_er - n_wp, stream);
    if (a->tile > n_er || (a->tile == 0 && n_er == 0)) return wp_launch(a, a->tile == 0 ? 1 : a->tile - n_er, stream);
    const int v = er_variant(a);
    HAWQ_REQUIRE(v >= 0, "hawq_conv_expand_reduce: this pair of layers cannot be fused (need 1x1/stride-1 int8 fast-contract convs, "
                         "uint16 residual in and out - or a same-shape 1x1 identity branch with Cin 64 -, Cin in {64,128,256}, reduce.Cin == expand.Cout, reduce.Cout == expand.Cin; tile %d)", a->tile);
    const hawq_conv_args &e = a->expand, &r = a->reduce;
    auto e_fast = [](int ek) { return (ek & 0xff) >= 33 && (ek & 0xff) <= 62; };
    HAWQ_REQUIRE(e.mq >= 0 && e_fast(e.eq) && (e.in2 || (e.m_id_scalar >= 0 && e_fast(e.e_id_scalar))), "hawq_conv_expand_reduce: scalar tables outside the fast contract");
    HAWQ_REQUIRE(e.q_lo <= 0, "hawq_conv_expand_reduce: the block-input QuantAct clamp must admit 0");
    ERP p;
    p.x2 = (const uint8_t *)e.in, p.w3 = (const uint8_t *)e.wgt, p.w1 = (const uint8_t *)r.wgt;
    p.ctab3 = e.ctab, p.ctab1 = r.ctab;
    p.xid = (const uint8_t *)e.in2, p.wid = (const uint8_t *)e.wgt2, p.ctab_id = e.ctab_id;
    p.res_in = (const uint16_t *)e.res_in, p.res_out = (uint16_t *)e.res_out;
    p.y = (uint8_t *)r.out_q;
    const long long M = (long long)e.N * e.H * e.W;
    HAWQ_REQUIRE(M > 0 && M < (1ll << 30), "hawq_conv_expand_reduce: bad problem size");
    p.M = (int)M, p.C3 = e.Cout;
    p.m_id_s = e.in2 ? 0 : e.m_id_scalar, p.e_id_s = e.in2 ? 33 : e.e_id_scalar, p.mq = e.mq, p.eq = e.eq, p.q_hi = e.q_hi;
    p.y_lo = r.relu && r.q_lo < 0 ? 0 : r.q_lo, p.y_hi = r.q_hi;
    p.y_planar = r.out_planar;
    p.y_nib = r.out_bits == 4;
    p.flags = e.flags;
    static const int dbg_env = HAWQ_DBG_ENV();
    static long long *dbg_dev = nullptr;
    if (HAWQ_DBG_BIT(dbg_env, 128) && !dbg_dev) (void)hipMalloc(&dbg_dev, 64 * sizeof(long long));
    p.dbgbuf = HAWQ_DBG_BIT(dbg_env, 128) ? dbg_dev : nullptr;
    p.dbg = dbg_env;
    const ERInfo &ei = kER[v];
    static const bool attrs = [] {
        bool good = true;
        for (const ERInfo &k : kER)
            for (int i = 0; i < 5; ++i)
                good &= hipFuncSetAttribute((const void *)k.fn[i], hipFuncAttributeMaxDynamicSharedMemorySize, k.lds) == hipSuccess;
        return good;
    }();
    HAWQ_REQUIRE(attrs, "hawq_conv_expand_reduce: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
    const bool tie = ((e.fast_tables | r.fast_tables) & 4) != 0, ck0 = (e.fast_tables & 8) && (r.fast_tables & 8);
    const bool qk0 = (e.eq >> 8) == 0;
    HAWQ_REQUIRE(e.q_hi >= 0 && e.q_hi <= 32767, "hawq_conv_expand_reduce: q_hi outside [0, 32767]");
    hipLaunchKernelGGL(ei.fn[tie ? 1 : (ck0 ? (qk0 ? 3 : 2) : (qk0 ? 4 : 0))], dim3((p.M + ei.bm - 1) / ei.bm), dim3(ei.nt), ei.lds, (hipStream_t)stream, p);
    HAWQ_CHECK_HIP(hipGetLastError());
    if (p.dbgbuf) {  // experiment hook (synchronises!)
        long long hb[8];
        (void)hipStreamSynchronize((hipStream_t)stream);
        (void)hipMemcpy(hb, p.dbgbuf, sizeof(hb), hipMemcpyDeviceToHost);
        fprintf(stderr, "[expand-reduce C=%d bm=%d M=%d C3=%d] cycles of wave 0 / workgroup 8: prologue %lld | B1 wait %lld | issue W3+res %lld | GEMM1 %lld | "
                        "epilogue1 %lld | W1 wait + B2 %lld | GEMM2 %lld | vmcnt0 + stores + issue W1 %lld\n",
                ei.c, ei.bm, p.M, p.C3, hb[0], hb[1], hb[2], hb[3], hb[4], hb[5], hb[6], hb[7]);
    }
    return 0;
}
